import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import torch
import test_parallel_gpu as T
dev = torch.device("cuda:0")
m = T._model(dev)
if os.environ.get("DIAG_F32"): m.compute_dtype = torch.float32
runs = []
for rep in range(3):
    T._grads(m, 0, dev)
    torch.cuda.synchronize()
    runs.append({k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters() if p.grad is not None})
for a, b in ((0, 1), (1, 2)):
    worst = sorted(((float((runs[a][k] - runs[b][k]).abs().max()) / (float(runs[a][k].abs().max()) + 1e-12), float(runs[a][k].abs().max()), k)
                    for k in runs[0] if not k.endswith("bias")), reverse=True)[:8]
    print(f"run{a} vs run{b}:", worst)
