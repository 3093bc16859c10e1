"""GraphedStep against the eager step on one GPU: bit-equality without dropout, timing with it.

    python tools/graph_step_check.py [S] [passion 0/1] [steps]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _util import build_product_model   # noqa: E402
from passion_amd import train as ptrain   # noqa: E402
from passion_amd.graph import GraphedStep   # noqa: E402


def make(S, passion, dropout):
    dev = torch.device("cuda:0")
    m = build_product_model(S, small=False).to(dev)
    m.mask_type, m.use_passion, m.compute_dtype = "idt", passion, torch.bfloat16
    m.train()
    if not dropout:
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    opt = ptrain.make_optimizer(m, lr=1e-3)
    state = ptrain.LossState(dev) if passion else None
    return m, opt, state


def run(S, passion, dropout, steps, graphed, masks):
    torch.manual_seed(5)
    m, opt, state = make(S, passion, dropout)
    dev = torch.device("cuda:0")
    x, target, _ = ptrain.synthetic_batch(1, S, dev, seed=3)
    step = GraphedStep(m, opt, 4.0, state, warm=3) if graphed else (lambda a, b, c: ptrain.train_step(m, opt, a, b, c, 4.0, state))
    losses = []
    for it in range(steps):
        losses.append(step(x, masks[it % len(masks)], target.view(target.shape)).clone())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        step(x, masks[it % len(masks)], target.view(target.shape))
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"  {'graph' if graphed else 'eager'}: host {1e3 * th / steps:.2f} ms/step, wall {1e3 * dt / steps:.2f} ms/step", flush=True)
    return torch.stack(losses).double().cpu(), {k: v.detach().clone() for k, v in m.state_dict().items()}, state


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    passion = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dev = torch.device("cuda:0")
    masks = [torch.tensor([[True, True, True, True]], device=dev), torch.tensor([[True, False, True, True]], device=dev),
             torch.tensor([[False, True, False, False]], device=dev)] if passion else [torch.ones(1, 4, dtype=torch.bool, device=dev)]
    print(f"S={S} passion={passion} dropout off", flush=True)
    le, pe, se = run(S, passion, False, steps, False, masks)
    lg, pg, sg = run(S, passion, False, steps, True, masks)
    print("  losses eager", [round(float(v), 6) for v in le])
    print("  losses graph", [round(float(v), 6) for v in lg])
    bad = [k for k in pe if not torch.equal(pe[k], pg[k])]
    print(f"  losses bit-equal: {torch.equal(le, lg)}; parameters that differ after {2 * steps} steps: {len(bad)} of {len(pe)}", flush=True)
    if passion:
        print("  imb epoch_dist_m eager / graph:", se.epoch_dist_m.tolist(), sg.epoch_dist_m.tolist())
    print(f"S={S} passion={passion} dropout on", flush=True)
    le, _, _ = run(S, passion, True, steps, False, masks)
    lg, _, _ = run(S, passion, True, steps, True, masks)
    lg2, _, _ = run(S, passion, True, steps, True, masks)
    print("  losses eager", [round(float(v), 5) for v in le])
    print("  losses graph", [round(float(v), 5) for v in lg])
    print(f"  two graph runs from the same seed bit-equal: {torch.equal(lg, lg2)}", flush=True)


if __name__ == "__main__":
    main()
