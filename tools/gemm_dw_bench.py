"""weight-gradient GEMMs of the grouped token layers (ops.GroupedLinearFn.backward): one batched bmm with f32 output against one mm per
weight set.  python tools/gemm_dw_bench.py"""
import torch
dev = torch.device("cuda:0")


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for G, R, Cin, Cout in ((4, 512, 4096, 512), (4, 512, 512, 4096), (4, 512, 512, 1536), (4, 512, 512, 512), (4, 512, 128, 512)):
    g = torch.randn(G, R, Cout, device=dev).to(torch.bfloat16)
    x = torch.randn(G, R, Cin, device=dev).to(torch.bfloat16)
    a = t(lambda: torch.bmm(g.transpose(1, 2), x, out_dtype=torch.float32))
    b = t(lambda: [torch.mm(g[i].t(), x[i], out_dtype=torch.float32) for i in range(G)])
    c = t(lambda: torch.bmm(g.transpose(1, 2), x))
    gt = g.transpose(1, 2).contiguous()
    d = t(lambda: torch.bmm(gt, x, out_dtype=torch.float32))
    out = torch.empty(G, Cout, Cin, dtype=torch.float32, device=dev)
    e = t(lambda: [torch.mm(g[i].t(), x[i], out_dtype=torch.float32, out=out[i]) for i in range(G)])
    ref = torch.bmm(g.transpose(1, 2), x, out_dtype=torch.float32)
    same = all(torch.equal(ref[i], torch.mm(g[i].t(), x[i], out_dtype=torch.float32)) for i in range(G))
    print(f"dW G={G} R={R} Cin={Cin} Cout={Cout}: bmm f32-out {a:.1f} us | {G} x mm f32-out {b:.1f} us (into one buffer {e:.1f}) | bmm bf16-out {c:.1f} | bmm f32-out, "
          f"A contiguous {d:.1f}; mm == bmm bit for bit: {same}")
