"""--resume continues a run: 2 epochs in one go == 1 epoch, checkpoint, resume, 1 epoch -- weights, optimizer
moments and the PASSION loss state (imb_beta, eta, modal_weight), bit for bit in bf16 (the bf16 step is bit-reproducible and
every epoch draws from its own random streams)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_epochs_equal_one_epoch_plus_resume_plus_one(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from passion_amd import train as ptrain
    base = ["--synthetic", "--use_passion", "--patch", "32", "--dtype", "bf16", "--num_epochs", "2", "--iters_per_epoch", "3", "--lr", "1e-3"]
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    m_a, st_a = ptrain.main(base + ["--savepath", a])
    ptrain.main(base + ["--savepath", b, "--stop_after_epoch", "0"])
    ck = torch.load(os.path.join(b, "model_last.pth"), map_location="cpu", weights_only=True)
    assert ck["epoch"] == 0 and "loss_state" in ck and not torch.allclose(ck["loss_state"]["imb_beta"], torch.ones(4))
    m_b, st_b = ptrain.main(base + ["--savepath", b, "--resume", os.path.join(b, "model_last.pth")])
    assert torch.equal(st_a.imb_beta, st_b.imb_beta) and st_a.eta == st_b.eta
    sa, sb = m_a.state_dict(), m_b.state_dict()
    worst = max(float((sa[k].float() - sb[k].float()).abs().max()) for k in sa)
    assert worst == 0.0, worst
    fa = torch.load(os.path.join(a, "model_last.pth"), map_location="cpu", weights_only=True)
    fb = torch.load(os.path.join(b, "model_last.pth"), map_location="cpu", weights_only=True)
    for i in fa["optim_dict"]["state"]:
        for key in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            assert torch.equal(fa["optim_dict"]["state"][i][key], fb["optim_dict"]["state"][i][key]), (i, key)
