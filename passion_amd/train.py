"""Training step of the PASSION hot path on MI355X.

Counterpart of the reference's code/train.py hot loop (:198-289 PASSION branch, :390-448 plain
branch): model forward (losses are computed inside it), the per-iteration loss assembly with the
relative-preference mask, backward, AdamW(amsgrad) step -- and of the epoch-level imbalance update
(:325-335).  Differences that matter on a ~tens-of-ms GPU step:
  * everything stays on the device: the reference's 36 `.item()/.cpu()` syncs per iteration
    (train.py:291-321) are replaced by device-side accumulators read once per epoch;
  * multi-GPU is one process per GPU with RCCL gradient all-reduce (passion_amd.parallel) instead of
    single-process DataParallel (train.py:90).  Losses are SUMS over samples (train.py:229-231), so
    gradients are summed -- not averaged -- across ranks, which equals DataParallel on the global batch.
"""
import os
import time

import numpy as np
import torch

from .utils import criterions


class LossState:
    """imb_beta / modal_weight and the per-epoch accumulators (reference train.py:163-175,185-195)"""

    def __init__(self, device, modal_weight=None, eta=0.01, eta_ext=1.5):
        self.imb_beta = torch.ones(4, device=device)
        self.modal_weight = torch.ones(4, device=device) if modal_weight is None else modal_weight.to(device)
        self.eta, self.eta_ext = eta, eta_ext
        self.epoch_dist_m = torch.zeros(4, device=device)

    def state_dict(self):
        """plain tensors / floats only: the checkpoint stays loadable with torch.load(weights_only=True)"""
        return {"imb_beta": self.imb_beta.detach().cpu(), "modal_weight": self.modal_weight.detach().cpu(), "eta": float(self.eta),
                "eta_ext": float(self.eta_ext)}

    def load_state_dict(self, sd):
        # in place: a recorded step (graph.GraphedStep) reads these tensors by ADDRESS
        self.imb_beta.copy_(sd["imb_beta"].to(self.imb_beta.device).float())
        self.modal_weight.copy_(sd["modal_weight"].to(self.modal_weight.device).float())
        self.eta, self.eta_ext = float(sd["eta"]), float(sd.get("eta_ext", self.eta_ext))

    def end_epoch(self, epoch, warmup=False):
        """train.py:325-335: beta <- clamp(beta - eta*rp_epoch, .1, 4), renormalised to |beta|_2 = 2"""
        from .parallel import all_reduce_sum_
        all_reduce_sum_(self.epoch_dist_m)           # every rank saw its own samples only; the reference sums over all of them
        avg = self.epoch_dist_m.sum() / 4.0
        rp_epoch = (avg - self.epoch_dist_m) / avg
        if not warmup:
            if epoch % 100 == 0:
                self.eta *= self.eta_ext
            beta = torch.clamp(self.imb_beta - self.eta * rp_epoch, 0.1, 4.0)
            self.imb_beta.copy_(2 * beta / beta.pow(2).sum().sqrt())       # in place: a recorded step reads this tensor by address
        self.epoch_dist_m.zero_()
        return rp_epoch


class _SumLossFn(torch.autograd.Function):
    """fuse_loss + (sep * mask).sum() + prm_loss of the plain step (train.py:228-240 of the reference) as one concatenation + one sum, with the
    constant gradients handed back directly: between the end of the forward and the first backward kernel these few-element
    reductions, adds and their autograd twins were sixteen launches of ~5 us each in a row on the main stream (a recorded step
    replays them one after the other: nothing else can run there); four are left."""

    @staticmethod
    def forward(ctx, pair, prm, sep, mk):
        ctx.mk, ctx.shapes = mk, (pair.shape, prm.shape, sep.shape)
        sm = sep * mk if mk is not None else sep
        return torch.cat([pair.reshape(-1), prm.reshape(-1), sm.reshape(-1)]).sum()

    @staticmethod
    def backward(ctx, g):
        ps, rs, ss = ctx.shapes
        return g.expand(ps), g.expand(rs), (g * ctx.mk if ctx.mk is not None else g.expand(ss)), None


SUM_LOSS_FUSED = os.environ.get("PASSION_SUM_LOSS_FUSED", "1") != "0"


def assemble_loss(outs, target, mask, mask_type="idt", use_passion=True, state=None, warmup=False, modal_num=None,
                  iter_per_epoch=None, rp_override=None):
    """Per-iteration loss (train.py:228-280 / :413-440), no host synchronisation.  `modal_num` / `iter_per_epoch`: the
    normaliser of the epoch accumulators (train.py:295-304: / modal_num for mask_type 'idt', / iter_per_epoch otherwise)."""
    pred = outs[0]
    B = pred.shape[0]
    dev = pred.device
    if SUM_LOSS_FUSED and not use_passion and not warmup and pred.is_cuda:
        mk = mask.to(device=dev, dtype=torch.float32) if mask_type != "pdt" else None
        return _SumLossFn.apply(criterions._seg_pair(pred, target, None), outs[1], outs[2], mk)
    # softmax_weighted_loss_bs + dice_loss_bs (train.py:228): both columns of ONE pass over the prediction
    fuse_loss = criterions.seg_pair_bs(pred, target).sum()
    prm_loss = outs[1].sum()
    idt = mask_type != "pdt"
    mk = mask.to(device=dev, dtype=torch.float32)
    red = (lambda t: (t * mk).sum(0)) if idt else (lambda t: t.sum(0))
    sep_m = red(outs[2])
    if not use_passion:
        sep = sep_m.sum()
        return (sep + 0.0 * (fuse_loss + prm_loss)) if warmup else (fuse_loss + sep + prm_loss)
    beta = state.imb_beta if state is not None else torch.ones(4, device=dev)
    w = (state.modal_weight if (state is not None and idt) else torch.ones(4, device=dev))
    kl_m, proto_m, dist_bs = red(outs[3]), red(outs[4]), outs[5]
    # rp_override: a given relative-preference gate instead of this step's own (tests: replaying a reference trajectory whose
    # discrete gates are known, so that a near-tie flipped by rounding does not end the comparison)
    rp_mask = relative_preference(dist_bs, mk, idt) if rp_override is None else rp_override.to(device=dev, dtype=torch.float32)
    kl = (beta * w * kl_m).sum()
    proto = (rp_mask * w * proto_m).sum()
    if state is not None:
        dm = red(dist_bs).detach()
        if mask_type == "idt" and modal_num is not None:
            dm = dm / modal_num
        elif mask_type != "idt" and iter_per_epoch:
            dm = dm / iter_per_epoch
        state.epoch_dist_m += dm
    if warmup:
        return (beta * w * sep_m).sum() + 0.0 * (fuse_loss + prm_loss + kl + proto)
    return fuse_loss + (rp_mask * beta * w * sep_m).sum() + prm_loss + 0.5 * kl + 0.1 * proto


def relative_preference(dist_bs, mk, idt=True):
    """rp_mask of train.py:266-270: per sample, dist / (mean over that sample's present modalities) - 1, summed over the
    samples of the GLOBAL batch (the reference forms it on the outputs DataParallel gathered from all devices: with one
    process per GPU that is a 4-float all-reduce), thresholded at 0.  No gradient flows through it."""
    from .parallel import all_reduce_sum_
    d = dist_bs.detach()
    if idt:
        rp = (mk * (d / (d.sum(1, keepdim=True) / mk.sum(1, keepdim=True)) - 1)).sum(0)
    else:
        rp = (d / d.mean(1, keepdim=True) - 1).sum(0)
    return (all_reduce_sum_(rp.contiguous()) > 0).float()


def make_optimizer(model, lr=2e-4, weight_decay=1e-4, capturable=False):
    """AdamW(betas=(0.9,0.999), eps=1e-8, amsgrad=True) as train.py:95-96; capturable=True keeps the step counter on
    the device so that the step can be recorded in a HIP graph (passion_amd.graph)"""
    params = [{"params": list(model.parameters()), "lr": lr, "weight_decay": weight_decay}]
    on_gpu = all(p.is_cuda and p.dtype == torch.float32 for p in params[0]["params"])
    if on_gpu and not capturable:      # same math and state layout, all parameters updated in a handful of launches
        from .optim import FusedAdamW
        return FusedAdamW(params, betas=(0.9, 0.999), eps=1e-08, amsgrad=True)
    return torch.optim.AdamW(params, betas=(0.9, 0.999), eps=1e-08, amsgrad=True, capturable=capturable)


def poly_lr(base_lr, epoch, num_epochs, power=0.9):
    """LR_Scheduler (utils/lr_scheduler.py:17,43)"""
    return round(base_lr * np.power(1 - np.float32(epoch) / np.float32(num_epochs), power), 8)


EARLY_STEP = os.environ.get("PASSION_EARLY_STEP", "1") != "0"    # AdamW of the non-encoder parameters under the encoder backward
PACK_BEHIND_STEP = os.environ.get("PASSION_PACK_BEHIND_STEP", "1") != "0"   # conv weight images re-packed right behind the optimizer passes, not at the top of the next forward
DIRECT_GRADS = os.environ.get("PASSION_DIRECT_GRADS", "1") != "0"  # single process: the conv nodes fill .grad themselves (no AccumulateGrad nodes)


def _early_step_callback(model, optimizer):
    """-> callback for ops.BackwardMarkerFn (single process): when the backward reaches the encoders, every other parameter's
    gradient has been enqueued (on the main stream or on one of the model's side streams): AdamW for those 33 of 35.6 M
    parameters goes to a stream of its own, behind events of all of them, and runs UNDER the encoder backward (which touches
    none of those weights) instead of after it.  The closing optimizer.step() updates the encoders and joins."""
    dev = next(model.parameters()).device
    st = model.__dict__.setdefault("_opt_stream", {})
    if dev not in st:
        # HIP maps streams onto GPU_MAX_HW_QUEUES = 4 hardware queues (more cost 40 %) and streams that share a queue
        # serialise: the model's fourth side stream carries nothing while the encoders run grouped on the main stream -- the
        # optimizer takes it instead of a stream of its own (PASSION_OPT_OWN_STREAM=1: a new stream, for A/B)
        side = model.__dict__.get("_streams_dev", {}).get(dev, [])
        own = os.environ.get("PASSION_OPT_OWN_STREAM") == "1" or len(side) < 4
        st[dev] = (torch.cuda.Stream(device=dev) if own else side[3], torch.cuda.Event())
    opt_stream, done = st[dev]

    def cb():
        cur = torch.cuda.current_stream(dev)
        side = model.__dict__.get("_streams_dev", {}).get(dev, [])
        for s_ in {id(s_): s_ for s_ in [cur] + list(side)}.values():
            ev = torch.cuda.Event()
            ev.record(s_)
            opt_stream.wait_event(ev)
        from . import ops as _ops
        with torch.cuda.stream(opt_stream):
            optimizer.step_ready()
            if PACK_BEHIND_STEP:                 # ... and their MFMA operand images / bf16 GEMM operands, still under the encoder backward
                model.__dict__["_early_packed"] = set(optimizer._stepped)
                _ops.repack_stale(only=model.__dict__["_early_packed"])
                _ops.recast_stale(only=model.__dict__["_early_packed"])
            done.record(opt_stream)
        model.__dict__["_opt_pending"] = done
    return cb


def train_step(model, optimizer, x, mask, target, temp=4.0, state=None, reducer=None, warmup=False, modal_num=None,
               iter_per_epoch=None, phase="all"):
    """one iteration: forward, loss assembly, backward, (gradient all-reduce,) optimizer step.
    phase: "all"; "fb" = forward + loss + backward only (what graph.GraphedStep records for N > 1), "tail" = all-reduce + optimizer step"""
    from . import ops
    if phase == "tail":
        if reducer is not None:
            reducer.all_reduce_grads()
        optimizer.step()
        return None
    net = model
    net.is_training = True
    ops._TCL_STEP[0] += 1                # ops.target_to_cl: one conversion of the label volume per step, recordings included
    early = EARLY_STEP and hasattr(optimizer, "step_ready") and (reducer is None or not reducer.active) and x.is_cuda and phase == "all"
    if hasattr(optimizer, "_stepped"):
        optimizer._stepped.clear()       # a backward that raised after the early part left its marks: this iteration starts clean
    ops._MARKER_CB[0] = _early_step_callback(net, optimizer) if early else None
    from . import _lib as L
    host = L.host() if x.is_cuda else None
    # ~210 AccumulateGrad nodes less per step for the autograd engine (csrc/host_ext.cpp); with more than one rank the
    # GradReducer's overlap hangs hooks on exactly those nodes, so only here
    direct = DIRECT_GRADS and host is not None and (reducer is None or not reducer.active)
    if direct:
        host.set_direct_grads(True)
        ops._DIRECT[0] = True
    managed = PACK_BEHIND_STEP and x.is_cuda and phase == "all"
    ops._PACK_MANAGED[0] = managed
    net.__dict__.pop("_early_packed", None)
    try:
        outs = net(x, mask, target=target, temp=temp)
        loss = assemble_loss(outs, target, mask, net.mask_type, net.use_passion, state, warmup, modal_num, iter_per_epoch)
        optimizer.zero_grad()            # set_to_none: no 288 fill launches, backward assigns the fresh gradients
        loss.backward()
    finally:
        ops._MARKER_CB[0] = None
        ops._PACK_MANAGED[0] = False
        if direct:
            host.set_direct_grads(False)
            ops._DIRECT[0] = False
    if phase == "fb":
        return loss.detach()
    if reducer is not None:
        reducer.all_reduce_grads()
    optimizer.step()
    if managed:                          # the images of what this closing step updated (the encoders; everything without the early pass)
        early_set = net.__dict__.pop("_early_packed", set())
        ops.repack_stale(exclude=early_set)
        ops.recast_stale(exclude=early_set)
    pending = net.__dict__.pop("_opt_pending", None)
    if pending is not None:              # the early part ran on its own stream: the next forward waits for it
        torch.cuda.current_stream(x.device).wait_event(pending)
    return loss.detach()


def synthetic_batch(B, S, device, seed=7, mask=None):
    """SURVEY 8d synthetic inputs: randn volumes, uniform random one-hot float64 targets"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, S, S, S, generator=g)
    lab = torch.randint(0, 4, (B, S, S, S), generator=g)
    target = torch.nn.functional.one_hot(lab, 4).permute(0, 4, 1, 2, 3).double()
    if mask is None:
        mask = torch.ones(B, 4, dtype=torch.bool)
    return x.to(device), target.to(device), mask.to(device)


# mask_id -> present modalities [flair, t1ce, t1, t2] (reference data/datasets_nii.py:27-30) and the empirical mask_id histogram
# of datasets/BraTS/brats_split/Brats2020_imb_split_mr2468.csv (the --imbmrpath default; 219 subjects, modalities present
# [90, 135, 184, 43]): what BASELINE.json configs[2] ("imbalanced missing-rate masks (brats_split)") samples from
MASK_TABLE = [[False, False, False, True], [False, True, False, False], [False, False, True, False], [True, False, False, False],
              [False, True, False, True], [False, True, True, False], [True, False, True, False], [False, False, True, True],
              [True, False, False, True], [True, True, False, False], [True, True, True, False], [True, False, True, True],
              [True, True, False, True], [False, True, True, True], [True, True, True, True]]
SPLIT_MR2468_HIST = [2, 12, 38, 6, 3, 53, 23, 9, 1, 9, 35, 5, 2, 12, 9]


def split_modal_weight(batch_size=1, world=1):
    """modal_weight = iter_per_epoch / modal_num of the mr2468 split (train.py:163-171)"""
    hist = np.asarray(SPLIT_MR2468_HIST, dtype=np.float64)
    modal_num = (hist[:, None] * np.asarray(MASK_TABLE, dtype=np.float64)).sum(0)
    return torch.tensor(-(-int(hist.sum()) // (batch_size * world)) / modal_num, dtype=torch.float32)   # len(DataLoader): ceil


def sample_split_masks(rng, n):
    """n masks drawn i.i.d. from the split's empirical mask_id distribution (rng: numpy RandomState)"""
    hist = np.asarray(SPLIT_MR2468_HIST, dtype=np.float64)
    ids = rng.choice(15, size=n, p=hist / hist.sum())
    return torch.tensor([MASK_TABLE[i] for i in ids], dtype=torch.bool)


def epoch_order(n, epoch, seed, world, rank, batch_size, iters):
    """(iters, batch_size) subject indices of this rank for one epoch: shuffle=True of the reference's DataLoader (train.py:124)
    as a fresh permutation per epoch, identical on every rank, cut to iters * batch_size * world subjects and dealt so that
    global batch `it` = the ranks' rows `it` side by side.  Every rank gets exactly `iters` rows: each step issues collectives."""
    perm = np.random.RandomState(seed + epoch).permutation(n)
    gb = batch_size * world
    if n < iters * gb:              # the last global batch of ceil(n / gb) iterations is short: wrap around
        perm = np.resize(perm, gb * max(1, iters))
    return perm[:iters * gb].reshape(iters, world, batch_size)[:, rank]


def main(argv=None):
    """python -m passion_amd.train [--synthetic] --use_passion --patch 128 --dtype bf16 ...  (under torchrun for several
    GPUs).  Without --synthetic the npy dataset of the reference is read through passion_amd.data (GPU input pipeline)."""
    from . import parallel
    from .models import mmformer
    from .options import args_parser
    args = args_parser(argv)
    # --gpu (reference train.py:33: CUDA_VISIBLE_DEVICES = args.gpu): honoured for a single process when the box has those devices
    # and nothing else selected one; under torchrun LOCAL_RANK picks the device.  Said aloud when it is not applied.
    if "WORLD_SIZE" not in os.environ and not any(v in os.environ for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES",
                                                                          "PASSION_DEVICE_INDEX")):
        try:
            ids = [int(t) for t in str(args.gpu).split(",") if t.strip() != ""]
        except ValueError:
            ids = []
        if ids and max(ids) < torch.cuda.device_count():       # device_count() does not initialise the GPU
            os.environ["HIP_VISIBLE_DEVICES"] = ",".join(str(i) for i in ids)
        elif ids:
            print(f"[train] --gpu {args.gpu}: this box has {torch.cuda.device_count()} device(s); using device 0", flush=True)
    ctx = parallel.init()
    torch.manual_seed(args.seed)
    if args.model == "mmformer":
        mmformer.patch_size = args.patch // 16
        model = mmformer.Model(num_cls=4).to(ctx.device)
    elif args.model == "rfnet":
        from .models import rfnet
        model = rfnet.Model(num_cls=4).to(ctx.device)
    elif args.model == "m2ftrans":
        from .models import m2ftrans
        m2ftrans.patch_size = args.patch // 16
        model = m2ftrans.Model(num_cls=4).to(ctx.device)
    else:
        raise SystemExit(f"unknown --model {args.model!r} (mmformer, rfnet, m2ftrans)")
    model.mask_type, model.use_passion = args.mask_type, args.use_passion
    model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model.train()
    start_epoch = 0
    ckpt = None
    if args.resume is not None:         # train.py:143-152 (--use_pretrain: weights only) / checkpoint layout of :358-364
        ckpt = torch.load(args.resume, map_location="cpu", weights_only=True)
        sd = {k[7:] if k.startswith("module.") else k: v for k, v in ckpt["state_dict"].items()}
        own = model.state_dict()
        own.update({k: v for k, v in sd.items() if k in own})
        model.load_state_dict(own)
    reducer = parallel.GradReducer(model, ctx)
    optimizer = make_optimizer(model, args.lr, args.weight_decay)
    if ckpt is not None and not args.use_pretrain:      # continue the run: optimizer moments and the epoch counter too
        optimizer.load_state_dict(ckpt["optim_dict"])
        start_epoch = int(ckpt["epoch"]) + 1
    state = LossState(ctx.device)
    resumed_state = False
    if ckpt is not None and not args.use_pretrain:
        # every checkpoint is written after end_epoch() has moved imb_beta (and, every 100 epochs, eta): a continued run must
        # pick those up, or it silently follows another loss weighting than the uninterrupted one
        if "loss_state" in ckpt:
            state.load_state_dict(ckpt["loss_state"])
            resumed_state = True
        elif args.use_passion and ctx.rank == 0:
            print("[train] the checkpoint carries no PASSION loss state (a reference checkpoint?): imb_beta = 1, eta = 0.01 again",
                  flush=True)
    data = None
    gb = args.batch_size * ctx.world                    # global batch: one DataParallel batch of the reference = all ranks' samples
    if not args.synthetic:      # the reference's npy dataset through the GPU input pipeline (train.py:112-131)
        from .data.datasets_nii import Brats_loadall_train_nii_idt
        csv = os.path.join(args.datarootPath, args.imbmrpath)
        data = Brats_loadall_train_nii_idt(transforms=args.train_transforms, root=args.datasetPath, num_cls=4,
                                           mask_type=args.mask_type, train_file=csv, device=ctx.device)
    # iterations per epoch from a RANK-INDEPENDENT quantity: every step issues collectives (rp_mask, gradient buckets), so all
    # ranks must run the same number of steps.  The reference's DataLoader has no drop_last (train.py:122-128): an epoch is
    # ceil(N / batch) iterations and the last batch is short; here the last global batch is wrap-padded with the first subjects of
    # the epoch's permutation instead (every rank keeps a full batch).  `iter_ref` = the reference's iter_per_epoch, which feeds
    # modal_weight and the pdt normaliser (train.py:160,171) whatever --iters_per_epoch says.
    iter_ref = 10 if data is None else -(-len(data) // gb)
    iters = args.iters_per_epoch or iter_ref
    modal_num = None
    if data is not None:        # initial imbalance weights (train.py:163-171): iterations per epoch / subjects that have the modality
        modal_num = torch.zeros(4, device=ctx.device)
        for sample_mask in data.samples_masks:
            modal_num += torch.tensor(eval(sample_mask), dtype=torch.float32, device=ctx.device)
        if not resumed_state:
            state.modal_weight.copy_(iter_ref / modal_num)
    loss = torch.zeros((), device=ctx.device)
    for epoch in range(start_epoch, args.num_epochs):
        # per-epoch random streams (synthetic masks, dropout): epoch e of a resumed run draws what epoch e of the uninterrupted run drew
        rng = np.random.RandomState(args.seed + ctx.rank + 7919 * epoch)
        torch.manual_seed(args.seed + 7919 * epoch)
        lr = float(poly_lr(args.lr, epoch, args.num_epochs))     # python float: keeps the checkpoint weights_only-loadable
        for g_ in optimizer.param_groups:
            g_["lr"] = lr
        if data is not None:
            order = epoch_order(len(data), epoch, args.seed, ctx.world, ctx.rank, args.batch_size, iters)
        t0 = time.time()
        for it in range(iters):
            if data is None:
                mk = sample_split_masks(rng, args.batch_size)
                x, target, mk = synthetic_batch(args.batch_size, args.patch, ctx.device, seed=epoch * iters + it, mask=mk)
            else:
                samples = [data[int(j)] for j in order[it]]
                x = torch.stack([s_[0] for s_ in samples])
                target = torch.stack([s_[1] for s_ in samples])
                mk = torch.stack([s_[2] for s_ in samples])
            loss = train_step(model, optimizer, x, mk, target, args.temp, state, reducer,
                              warmup=epoch < args.region_fusion_start_epoch, modal_num=modal_num, iter_per_epoch=iter_ref)
        torch.cuda.synchronize()
        if args.use_passion:
            state.end_epoch(epoch, epoch < args.region_fusion_start_epoch)
        if ctx.rank == 0:
            print(f"epoch {epoch + 1}/{args.num_epochs} lr {lr} loss {float(loss):.4f} "
                  f"{iters * gb / (time.time() - t0):.3f} vol/s", flush=True)
            os.makedirs(args.savepath, exist_ok=True)
            torch.save({"epoch": epoch, "state_dict": {"module." + k: v for k, v in model.state_dict().items()},
                        "optim_dict": optimizer.state_dict(), "loss_state": state.state_dict()},
                       os.path.join(args.savepath, "model_last.pth"))
        if epoch == args.stop_after_epoch:
            break
    return model, state


if __name__ == "__main__":
    main()
