"""PASSION_WGRAD_FORK under stream capture: where does it break?  Each case in its own process (a segfault is an answer).
    python tools/fork_capture_probe.py            -> runs all cases
    python tools/fork_capture_probe.py CASE       -> one case
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["conv1", "conv3", "conv_deep3", "conv_side3", "conv_side3_x2", "model32_1stream", "model32"]


def case(name):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from passion_amd import ops, _lib as L
    dev = torch.device("cuda:0")
    if name.startswith("conv"):
        nl = 1 if name == "conv1" else 3
        C, S = (64, 16) if "deep" in name else (8, 32)
        torch.manual_seed(0)
        x0 = torch.randn(2, S, S, S, C, device=dev).to(torch.bfloat16).requires_grad_(True)
        ws = [(torch.randn(C, C, 3, 3, 3, device=dev) / (C * 27) ** 0.5).requires_grad_(True) for _ in range(nl)]
        bs = [torch.zeros(C, device=dev, requires_grad=True) for _ in range(nl)]
        cache = [{} for _ in range(nl)]

        side = [torch.cuda.Stream(), torch.cuda.Stream()] if "side" in name else []

        def chain(x):
            for i in range(nl):
                x = ops.conv_block([ops.SegIn(x, stats=ops.instnorm_stats(x.detach()))], ws[i], bs[i], S, S, S, ksize=3, stride=1,
                                   pad_mode=L.PAD_REFLECT, pack_cache=cache[i])
            return x.float().square().mean()

        def step():
            if not side:
                chain(x0).backward()
                return
            cur = torch.cuda.current_stream()
            outs = []
            for sd in side[:2 if name.endswith("x2") else 1]:      # the convs (and, by autograd, their backward) on a stream forked from the current one
                sd.wait_stream(cur)
                with torch.cuda.stream(sd):
                    outs.append(chain(x0))
                cur.wait_stream(sd)
            sum(outs).backward()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(3):
                step()
        torch.cuda.synchronize()
        ref = [w.grad.clone() for w in ws]
        for w in ws:
            w.grad = None
        x0.grad = None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            step()
        print("captured", flush=True)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print("replayed; dW equal to eager/3:", all(torch.allclose(w.grad, r / 3, rtol=1e-3, atol=1e-5) for w, r in zip(ws, ref)), flush=True)
        return
    from _util import build_product_model
    from passion_amd import train as ptrain
    from passion_amd.graph import GraphedStep
    from passion_amd.models import mmformer
    if name.endswith("1stream"):
        mmformer.ENC_STREAMS = False
    S = 32
    torch.manual_seed(5)
    m = build_product_model(S, small=False).to(dev)
    m.mask_type, m.use_passion, m.compute_dtype = "idt", False, torch.bfloat16
    m.train()
    opt = ptrain.make_optimizer(m, lr=1e-3)
    x, target, _ = ptrain.synthetic_batch(1, S, dev, seed=3)
    mask = torch.ones(1, 4, dtype=torch.bool, device=dev)
    step = GraphedStep(m, opt, 4.0, None, warm=3)
    for it in range(8):
        loss = step(x, mask, target.view(target.shape))
    torch.cuda.synchronize()
    print("model steps done, loss", float(loss), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        case(sys.argv[1])
    else:
        for c in CASES:
            env = dict(os.environ, PASSION_WGRAD_FORK="1")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), c], env=env, capture_output=True, text=True, timeout=600)
            tail = (r.stdout + r.stderr).strip().splitlines()[-3:]
            print(f"== {c}: rc {r.returncode}", " | ".join(tail), flush=True)
