#!/usr/bin/env python
"""bench.py -- headline metric of BASELINE.json on MI355X: train volumes/sec of the PASSION hot path
(mmformer, 4-modality synthetic 128^3 volumes, batch 1 per GPU, bf16 storage / f32 accumulate).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One "step" = one full training iteration on one synthetic batch already resident in HBM:
Model.forward (all losses are computed inside it) + train.py's loss assembly + backward +
gradient all-reduce (N>1) + AdamW(amsgrad) step.  Ranks own independent volumes (weak scaling, no
data-path collective).  Rank 0 prints ONE JSON line; besides the contract's fields it carries
  roofline     : the conv kernel family with the largest total time in the step (forward 3x3x3 / forward 1x1x1 /
                 dgrad / wgrad): algorithmic bytes (activations in + out + weights) / average launch time,
                 taken live with hipEvents on the launch stream inside the timed region; peak = 8 TB/s HBM3E.
                 `avg_launch_ms_single_stream` / `frac_single_stream`: the same kernel in three extra steps with all work
                 on one stream (no other stream's launches sharing the CUs) -- the figure a rocprofv3 summary shows.
                 (--graph replays the step as one HIP graph; hipEventElapsedTime cannot time event nodes of a graph, so
                 there the event pairs come from eager iterations of the same step right after the timed region.)
  cpu_baseline : the oracle (oracle/, a CPU restatement proven equal to the reference) timed on the
                 host cores, rank 0, N=1 only, on one 64^3 training step (1/8 of a 128^3 volume).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prime", type=int, default=6,
                    help="untimed set-up iterations before the W warm-up steps: the caching allocator keeps one pool per HIP "
                         "stream (the four encoder branches run on their own streams) and needs a few iterations to stop "
                         "calling hipMalloc; part of building the step, like compiling a graph would be")
    ap.add_argument("--workload", default="mmformer128", choices=["mmformer128", "mmformer128_passion", "rfnet128_passion"],
                    help="BASELINE.json configs[1] (default) / configs[2] (+ PASSION self-distillation) / "
                         "configs[3] (RFNet + PASSION, batch 2 per GPU)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="EXPERIMENTAL: replay the whole step as one HIP graph (N=1; slower than eager on ROCm 7.2 (30 vs 19.5 ms), and on ROCm 7.2 / torch 2.10 "
                         "replays were observed to corrupt gradients intermittently -- DESIGN.md, so it is not the default)")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def cpu_baseline(passion):
    """oracle training step (forward + loss assembly + backward + AdamW) on the host cores"""
    from oracle import train_step_ref
    from passion_amd.models import mmformer
    S = 64
    cores = min(os.cpu_count() or 1, 32)   # more threads than that only add contention on these op sizes
    torch.set_num_threads(cores)
    mmformer.patch_size = S // 16
    torch.manual_seed(1037)
    m = mmformer.Model(num_cls=4)       # parameter holder only (CPU); the forward below is the oracle's
    mmformer.patch_size = 5
    P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    opt = torch.optim.AdamW(list(P.values()), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, amsgrad=True)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 4, S, S, S, generator=g)
    lab = torch.randint(0, 4, (1, S, S, S), generator=g)
    target = torch.nn.functional.one_hot(lab, 4).permute(0, 4, 1, 2, 3).double()
    mask = torch.ones(1, 4, dtype=torch.bool)
    times = []
    for it in range(2):                 # 1 warm-up + 1 timed
        t0 = time.time()
        loss, _, _ = train_step_ref.train_step(P, x, mask, target, 4.0, "idt", passion, drop_p=0.1)
        opt.zero_grad()
        loss.backward()
        opt.step()
        times.append(time.time() - t0)
    t = times[-1]
    return {"value": round((S / 128.0) ** 3 / t, 6), "unit": "volumes/s (128^3-equivalent)", "cores": cores,
            "kind": "port", "sample": f"1 training step (fwd+loss+bwd+AdamW, dropout on) of the same model on one "
            f"4x{S}^3 synthetic volume = 1/8 of a 128^3 volume by voxel count; {t:.2f} s on {cores} threads"}


def measured_copy_gbs(device):
    """device-to-device copy of 1 GiB (read + write counted), the 'measured peak' SURVEY 8(d) asks for beside the nominal one"""
    a = torch.empty(1 << 29, dtype=torch.bfloat16, device=device)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 5 * 2 * a.numel() * 2 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def main():
    args = parse()
    from passion_amd import ops, parallel
    from passion_amd import train as ptrain
    from passion_amd.models import mmformer

    ctx = parallel.init()
    assert torch.cuda.is_available(), "bench.py measures the HIP path: a GPU is required"
    if ctx.world != args.gpus and ctx.rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={ctx.world}", file=sys.stderr)
    S = args.size
    passion = args.workload.endswith("passion")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    is_rfnet = args.workload.startswith("rfnet")
    batch = 2 if is_rfnet else 1
    torch.manual_seed(1037)
    if is_rfnet:
        from passion_amd.models import rfnet
        model = rfnet.Model(num_cls=4).to(ctx.device)
    else:
        mmformer.patch_size = S // 16
        model = mmformer.Model(num_cls=4).to(ctx.device)
    model.mask_type, model.use_passion, model.compute_dtype = "idt", passion, dtype
    model.train()                                   # dropout live, as in the reference's training loop
    reducer = parallel.GradReducer(model, ctx)
    use_graph = ctx.world == 1 and args.graph and not args.no_graph
    opt = ptrain.make_optimizer(model, capturable=use_graph)
    state = ptrain.LossState(ctx.device)
    x, target, mask = ptrain.synthetic_batch(batch, S, ctx.device, seed=7 + ctx.rank)   # resident in HBM before timing

    def eager_step():
        return ptrain.train_step(model, opt, x, mask, target, 4.0, state, reducer)

    step = eager_step
    if use_graph:           # the whole iteration (weight packing, forward, losses, backward, AdamW) as one HIP graph
        from passion_amd import graph as pgraph
        try:
            gstep = pgraph.GraphedTrainStep(model, opt, x, mask, target, temp=4.0, state=state)
            step = lambda: gstep(x, mask, target)   # the batch is copied into the graph's static buffers every step
        except Exception as e:  # noqa: BLE001
            print(f"[bench] HIP-graph capture failed ({type(e).__name__}: {e}); eager steps", file=sys.stderr)
            use_graph = False

    # warm-up; with the roofline leg on, its event pairs bracket EVERY conv launch so that the kernel family that
    # dominates the step is known before the timed region starts -- there only that family keeps its event pairs
    # (326 event pairs per step cost ~5 % of throughput, one family ~1.5 %)
    for _ in range(args.prime):
        step()
    want_events = not args.no_roofline and not use_graph
    if want_events:
        ops.prof_read()
        ops.prof_enable(0xF)
    for _ in range(args.warmup):
        wl = step()
        if os.environ.get("BENCH_DEBUG") == "2":
            print(f"[bench] warmup loss {float(wl):.5f}", file=sys.stderr)
    parallel.barrier(ctx)
    torch.cuda.synchronize()
    live_events = want_events
    fam_warm = {}
    if want_events:
        ops.prof_enable(0)
        dom_kind, tot_w = 2, {}
        for r in ops.prof_read():
            if r[1] >= 0:
                tot_w[(r[0], r[4])] = tot_w.get((r[0], r[4]), 0.0) + r[1]
                fam_warm[r[0]] = fam_warm.get(r[0], 0.0) + r[1]
        if fam_warm:          # the family (fwd3 / fwd1 / dgrad / wgrad) with the largest total; its top kernel is reported
            dom_kind = max(fam_warm, key=fam_warm.get)
        ops.prof_enable(1 << dom_kind)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
        if os.environ.get("BENCH_DEBUG"):
            print(f"[bench] loss {float(loss):.5f}", file=sys.stderr)
    t_host = time.perf_counter() - t0                # all K steps enqueued; the GPU may still be working
    torch.cuda.synchronize()
    parallel.barrier(ctx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if os.environ.get("BENCH_HOST") and ctx.rank == 0:
        print(f"[bench] host enqueue {1e3 * t_host / args.steps:.2f} ms/step, wall {1e3 * dt / args.steps:.2f} ms/step", file=sys.stderr)
    ops.prof_enable(0)
    dt = parallel.all_reduce_max(ctx, dt)
    final_loss = float(loss)
    prof_steps = args.steps
    if not args.no_roofline and use_graph and ctx.rank == 0:
        prof_steps = min(args.steps, 3)             # same step, eager, with the event pairs (see the module docstring)
        ops.prof_read()
        ops.prof_enable(0xF)
        for _ in range(prof_steps):
            eager_step()
        torch.cuda.synchronize()
        ops.prof_enable(0)

    # The same kernel without stream overlap: three more steps (every rank, the gradient all-reduce is collective) with the
    # modality encoders / decoder_sep on the main stream.  Inside the timed region the launches of five streams share the CUs and
    # stretch each other; rocprofv3 (whose interception makes the step host-bound) sees nearly stand-alone launches.  Reporting
    # both lets the live number be checked against profiles/*kernel_stats.csv.
    timed_recs = ops.prof_read() if not args.no_roofline else []
    alone_recs = []
    if want_events and live_events:
        was = mmformer.ENC_STREAMS
        mmformer.ENC_STREAMS = False
        ops.prof_enable(1 << dom_kind)
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize()
        ops.prof_enable(0)
        mmformer.ENC_STREAMS = was
        alone_recs = [r for r in ops.prof_read() if r[1] >= 0]

    if ctx.rank != 0:
        return
    out = {
        "metric": "train volumes/sec (4-modality 128^3, bs=%d/GPU)" % batch, "value": round(batch * ctx.world * args.steps / dt, 4),
        "unit": "volumes/s", "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if dtype == torch.bfloat16 else "f32", "data": "synthetic",
        "config": {"workload": f"{'rfnet' if is_rfnet else 'mmformer'}{'+PASSION' if passion else ''} train step, 4x{S}^3 volume, "
                               f"batch {batch}/GPU, mask_type idt, all modalities present, AdamW(amsgrad)"
                               f"{'' if is_rfnet else ', dropout 0.1'} "
                               f"(BASELINE.json configs[{3 if is_rfnet else (2 if passion else 1)}])",
                   "global_batch": batch * ctx.world, "parallelism": f"dp{ctx.world} (RCCL grad all-reduce, sum)",
                   "final_loss": round(final_loss, 5),
                   "launch": "one HIP graph per step" if use_graph else "eager"},
    }
    if not args.no_roofline:
        recs = [r for r in timed_recs if r[1] >= 0]
        fam = {0: "fwd3", 1: "fwd1", 2: "dgrad", 3: "wgrad"}
        what = {0: "fused InstanceNorm+LeakyReLU+Conv3d 3x3x3 forward", 1: "fused InstanceNorm+LeakyReLU+Conv3d 1x1x1 forward",
                2: "conv input gradient + LeakyReLU'/mask + InstanceNorm-backward sums",
                3: "conv weight + bias gradient (InstanceNorm+LeakyReLU recomputed in the loader)"}
        tot = {}
        for r in recs:                              # per kernel instantiation, as rocprofv3 names it
            tot[r[4]] = tot.get(r[4], 0.0) + r[1]
        if tot:
            dom = max(tot, key=tot.get)             # the kernel with the largest total time in the timed region
            rk = [r for r in recs if r[4] == dom]
            ms, by, fl = sum(r[1] for r in rk), sum(r[2] for r in rk), sum(r[3] for r in rk)
            ach = by / (ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get("bytes_per_launch", {}).get(dom)
            fam_ms = {}
            for r in recs:
                fam_ms[fam[r[0]]] = fam_ms.get(fam[r[0]], 0.0) + r[1]
            copy_gbs = measured_copy_gbs(ctx.device)
            ak = [r for r in alone_recs if r[4] == dom]
            alone_ms = sum(r[1] for r in ak) / len(ak) if ak else None
            alone_ach = (sum(r[2] for r in ak) / (sum(r[1] for r in ak) * 1e-3) / 1e9) if ak else None
            out["roofline"] = {"kernel": f"{dom} ({what[rk[0][0]]})", "bound": "hbm", "achieved": round(ach, 2),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "measured_copy_peak": round(copy_gbs, 1), "frac_of_measured_copy": round(ach / copy_gbs, 5),
                               "avg_launch_ms_single_stream": round(alone_ms, 5) if alone_ms else None,
                               "frac_single_stream": round(alone_ach / HBM_PEAK_GBS, 5) if alone_ach else None,
                               "launches_per_step": len(rk) // prof_steps, "avg_launch_ms": round(ms / len(rk), 5),
                               "algorithmic_bytes_per_launch": round(by / len(rk), 1),
                               "tflops": round(fl / (ms * 1e-3) / 1e12, 3),
                               "events": "live, inside the timed region" if live_events else
                                         f"{prof_steps} eager iterations of the same step after the timed graph replays",
                               "ms_per_step_by_family": ({fam[k]: round(v / max(1, args.warmup), 3) for k, v in sorted(fam_warm.items())}
                                                         if fam_warm else
                                                         {k: round(v / prof_steps, 3) for k, v in sorted(fam_ms.items())})}
    if not args.no_cpu_baseline and ctx.world == 1 and not is_rfnet:
        out["cpu_baseline"] = cpu_baseline(passion)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
