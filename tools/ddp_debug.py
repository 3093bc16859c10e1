"""debug of tests/test_parallel_gpu.py: which parameters differ between the 2-rank reducer result and the sum of the two
ranks' gradients computed one after the other, and how (missing a rank's share? stale? garbage?)"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.multiprocessing as mp
import test_parallel_gpu as T

if __name__ == "__main__":
    dtype = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == "f32") else torch.bfloat16
    for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
        out = os.path.join(tempfile.mkdtemp(), "r0.pt")
        mp.spawn(T._worker, args=(2, T._free_port(), out, dtype), nprocs=2, join=True)
        got = torch.load(out)
        dev = torch.device("cuda:0")
        m = T._model(dev, dtype)
        per = []
        for rank in range(2):
            T._grads(m, rank, dev)
            per.append({k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters() if p.grad is not None})
        bad = 0
        for k in per[0]:
            w = per[0][k] + per[1][k]
            g = got["g"][k].float()
            e = float((g - w).abs().max()); ref = float(w.abs().max())
            if e > 2e-4 * ref + 1e-6:
                bad += 1
                e0 = float((g - per[0][k]).abs().max()); e1 = float((g - per[1][k]).abs().max()); e2 = float((g - 2 * per[0][k]).abs().max())
                print(f"  trial {trial}: {k}: err {e:.2e} of {ref:.2e}; |got - rank0| {e0:.2e}, |got - rank1| {e1:.2e}, |got - 2 rank0| {e2:.2e}")
        print(f"trial {trial}: {bad} parameters off", flush=True)
