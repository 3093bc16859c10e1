"""Are column sums (the bias gradient of a token GEMM) bit-reproducible on this stack, alone and while another stream keeps the
GPU busy?  torch's `sum((0,1))` against the library's passion_colsum, and the library's loss reductions (same last-block fold).
(rows x 4096 and rows x 512: the FFN's two linears; 1280 rows = 5 passes x 256 tokens)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from passion_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
side = torch.cuda.Stream()
busy_a = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
big = torch.randn(64 << 20, device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000


def trial(name, fn, conc):
    ref, bad, worst = None, 0, 0.0
    for it in range(N):
        if conc and it % 4 == 0:
            with torch.cuda.stream(side):
                for _ in range(3):
                    busy_a @ busy_a
                big.mul_(1.0001)
        s = fn()
        s = torch.cat([t.flatten().float() for t in s]) if isinstance(s, (tuple, list)) else s
        if ref is None:
            ref = s.clone()
        elif not torch.equal(ref, s):
            bad += 1
            worst = max(worst, float((ref.float() - s.float()).abs().max()))
    torch.cuda.synchronize()
    print(f"{name:44s} concurrent={conc!s:5s}: {bad} of {N - 1} results differ from the first (max diff {worst:.2e})", flush=True)


for conc in (False, True):
    for rows, cols in ((1280, 4096), (1280, 512), (320, 4096)):
        gy = (torch.randn(5, rows // 5, cols, device=dev) * 1e-3).to(torch.bfloat16)
        g2 = gy.view(-1, cols)
        want = g2.double().sum(0)
        err = float((ops.colsum(g2).double() - want).abs().max() / want.abs().max())
        assert err < 1e-5, err
        trial(f"torch sum        [{rows} x {cols}] bf16", lambda: gy.sum((0, 1)), conc)
        trial(f"passion_colsum   [{rows} x {cols}] bf16", lambda: ops.colsum(g2), conc)
    V = 64 ** 3
    p = torch.softmax(torch.randn(5, V, 4, device=dev), -1).view(5, 64, 64, 64, 4)
    q = torch.softmax(torch.randn(5, V, 4, device=dev), -1).view(5, 64, 64, 64, 4)
    t = torch.nn.functional.one_hot(torch.randint(0, 4, (5, V), device=dev), 4).float().view(5, 64, 64, 64, 4)
    f1 = torch.randn(5, 64, 64, 64, 8, device=dev).to(torch.bfloat16)
    f2 = torch.randn(5, 64, 64, 64, 8, device=dev).to(torch.bfloat16)
    trial("passion wce_dice [5 x 64^3]", lambda: ops.wce_dice(p, t, "cl"), conc)
    trial("passion kl       [5 x 64^3]", lambda: ops.kl_probs(p, q, 4.0), conc)
    trial("passion proto    [5 x 64^3 x 8]", lambda: ops.proto_loss(f1, f2, t), conc)
