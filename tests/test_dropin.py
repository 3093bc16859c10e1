"""The reference's import surface (code/train.py:15-24) resolved by the alias packages under dropin/: `from models import mmformer`,
`from utils import criterions`, `from options import args_parser` give the PRODUCT's modules (the same objects as passion_amd.*),
and the product Model answers the attribute / call sequence of code/train.py:91-92,212,222,228 unchanged."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r'''
import sys
sys.path.insert(0, sys.argv[1])          # the one line INTEGRATION.md 1 asks for
from models import rfnet, mmformer, m2ftrans
from options import args_parser
from utils import criterions
from utils.predict import softmax_output_dice_class4
import passion_amd.models.mmformer as pm, passion_amd.models.rfnet as pr, passion_amd.models.m2ftrans as p2
import passion_amd.utils.criterions as pc
assert mmformer is pm and rfnet is pr and m2ftrans is p2 and criterions is pc
mmformer.patch_size = 2                  # module constants set through the alias reach the product (same module object)
assert pm.patch_size == 2
args = args_parser(["--use_passion", "--mask_type", "idt"])
for mod in (mmformer, rfnet, m2ftrans):
    assert callable(mod.Model)
for f in ("softmax_weighted_loss_bs", "dice_loss_bs", "temp_kl_loss_bs", "prototype_passion_loss_bs"):
    assert callable(getattr(criterions, f))
print("ok", args.use_passion, args.mask_type)
'''


def test_alias_packages_resolve_to_the_product():
    out = subprocess.run([sys.executable, "-c", PROBE, os.path.join(ROOT, "dropin")], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().endswith("ok True idt")


def test_submodules_the_product_does_not_replace_come_from_the_reference_package(tmp_path):
    """utils.lr_scheduler / utils.parser (logging, schedules: outside the hot path) still resolve when the reference's code/ directory
    follows dropin/ on sys.path"""
    ref = tmp_path / "code"
    (ref / "utils").mkdir(parents=True)
    (ref / "utils" / "__init__.py").write_text("")
    (ref / "utils" / "lr_scheduler.py").write_text("MARK = 'reference file'\n")
    probe = ("import sys; sys.path.insert(0, sys.argv[2]); sys.path.insert(0, sys.argv[1])\n"
             "from utils import criterions\nfrom utils.lr_scheduler import MARK\n"
             "import passion_amd.utils.criterions as pc\nassert criterions is pc\nprint(MARK)\n")
    out = subprocess.run([sys.executable, "-c", probe, os.path.join(ROOT, "dropin"), str(ref)], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip() == "reference file"


def test_model_module_property_is_the_model():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _util import build_product_model
    m = build_product_model(32, small=True)
    assert m.module is m
    m.module.mask_type = "pdt"           # code/train.py:91-92
    m.module.use_passion = True
    m.module.is_training = True          # code/train.py:212
    assert (m.mask_type, m.use_passion, m.is_training) == ("pdt", True, True)
    assert "module" not in dict(m.named_modules())          # a property, not a registered child: state-dict keys are unchanged
    assert not any(k.startswith("module.") for k in m.state_dict())


@pytest.mark.gpu
def test_reference_train_loop_lines_drive_the_product():
    """code/train.py:83-96, 207-289 as written there (only `.cuda()` instead of DataParallel), against passion_amd.train's assembly"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.") or k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    from models import mmformer                      # train.py:19
    from options import args_parser                  # train.py:22
    from utils import criterions                     # train.py:23
    from _util import small_weights, synth, SMALL_DIMS, FULL_DIMS
    from passion_amd import train as ptrain
    args = args_parser(["--use_passion", "--mask_type", "idt", "--temp", "4.0"])
    num_cls = 4
    mmformer.patch_size = 2
    for k, v in SMALL_DIMS.items():
        setattr(mmformer, k, v)
    try:
        model = mmformer.Model(num_cls=num_cls)      # train.py:83
    finally:
        for k, v in FULL_DIMS.items():
            setattr(mmformer, k, v)
        mmformer.patch_size = 5
    model.load_state_dict(small_weights())
    model = model.cuda()                             # train.py:90 without the DataParallel wrapper
    model.module.mask_type = args.mask_type          # train.py:91
    model.module.use_passion = args.use_passion      # train.py:92
    train_params = [{'params': model.parameters(), 'lr': args.lr, 'weight_decay': args.weight_decay}]
    optimizer = torch.optim.AdamW(train_params, betas=(0.9, 0.999), eps=1e-08, amsgrad=True)     # train.py:95-96
    temp = args.temp
    x, target = synth(2, 32, blobs=True)
    mask = torch.tensor([[True, False, True, True], [True, True, False, True]])
    x, target, mask = x.cuda(non_blocking=True), target.cuda(non_blocking=True), mask.cuda(non_blocking=True)      # train.py:207-211
    model.eval()                                     # dropout off: the two evaluations below must agree
    model.module.is_training = True                  # train.py:212
    fuse_pred, prm_loss_bs, sep_loss_m_bs, kl_loss_m_bs, proto_loss_m_bs, dist_m_bs = model(x, mask, target=target, temp=temp)   # :222
    fuse_loss_bs = criterions.softmax_weighted_loss_bs(fuse_pred, target, num_cls=num_cls) + \
        criterions.dice_loss_bs(fuse_pred, target, num_cls=num_cls)                               # train.py:228
    fuse_loss = torch.sum(fuse_loss_bs)
    prm_loss = torch.sum(prm_loss_bs)
    imb_beta = torch.ones(4).cuda()
    modal_weight = torch.ones(4).cuda()
    kl_loss_m = torch.sum(kl_loss_m_bs * mask, dim=0)                                             # train.py:233-247 (idt)
    sep_loss_m = torch.sum(sep_loss_m_bs * mask, dim=0)
    proto_loss_m = torch.sum(proto_loss_m_bs * mask, dim=0)
    rp_iter = torch.zeros(4).cuda()
    for bs_n in range(x.size(0)):                                                                 # train.py:262-270
        dist_avg = sum(dist_m_bs[bs_n] * mask[bs_n]) / sum(mask[bs_n])
        rp_iter = rp_iter + (dist_m_bs[bs_n] / dist_avg - 1) * mask[bs_n]
    rp_mask = (rp_iter > 0).float()
    kl_loss = (imb_beta * modal_weight * kl_loss_m).sum()
    sep_loss = (rp_mask * imb_beta * modal_weight * sep_loss_m).sum()
    proto_loss = (rp_mask * modal_weight * proto_loss_m).sum()
    loss = fuse_loss + sep_loss + prm_loss + kl_loss * 0.5 + proto_loss * 0.1                     # train.py:279-280
    optimizer.zero_grad()                                                                         # train.py:287-289
    loss.backward()
    optimizer.step()
    outs = (fuse_pred, prm_loss_bs, sep_loss_m_bs, kl_loss_m_bs, proto_loss_m_bs, dist_m_bs)
    mine = ptrain.assemble_loss([o.detach() for o in outs], target, mask, "idt", True, ptrain.LossState(x.device))
    assert abs(float(loss) - float(mine)) < 1e-5 * abs(float(mine)), (float(loss), float(mine))
    assert all(p.grad is not None for p in model.parameters())
