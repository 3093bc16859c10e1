import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.multiprocessing as mp
import test_depth_gpu as T

def layers(m):
    enc = m.flair_encoder
    return [("e1_c2 s1 reflect", enc.e1_c2.conv, 1, True), ("e2_c1 s2", enc.e2_c1.conv, 2, True), ("RFM1.1 zero pad", m.decoder_fuse.RFM1.fusion_layer[1].conv, 1, True),
            ("d1_out 1x1x1", m.decoder_fuse.d1_out.conv, 1, True), ("seg_layer bare", m.decoder_fuse.seg_layer, 1, False)]

def worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      PASSION_DEVICE_INDEX="0", PASSION_DIST_BACKEND="gloo")
    from passion_amd import depth, parallel, ops
    ctx = parallel.init()
    shard = depth.DepthCtx()
    m = T._model(ctx.device)
    S = 16
    g = torch.Generator().manual_seed(1)
    t = torch.randn(1, S, 16, 16, 8, generator=g).to(ctx.device)
    Dl = S // world
    res = {}
    for name, mod, stride, norm in layers(m):
        tl = t[:, rank * Dl:(rank + 1) * Dl].clone().requires_grad_(True)
        mod.weight.grad = None; mod.bias.grad = None
        y = depth.conv(mod, depth.Slab(tl, shard), shard, S, norm=norm, out_dtype=None if norm else torch.float32).t
        gy = torch.randn(1, S // stride, 16 // stride, 16 // stride, y.shape[-1], generator=torch.Generator().manual_seed(2)).to(ctx.device)
        Do = Dl // stride
        y.backward(gy[:, rank * Do:(rank + 1) * Do].contiguous())
        res[name] = (y.detach().cpu(), tl.grad.cpu(), mod.weight.grad.cpu(), mod.bias.grad.cpu())
    tl = t[:, rank * Dl:(rank + 1) * Dl].clone().requires_grad_(True)
    u = depth.up(depth.Slab(tl, shard), 2, shard, S).t
    gu = torch.randn(1, 2 * S, 32, 32, 8, generator=torch.Generator().manual_seed(3)).to(ctx.device)
    u.backward(gu[:, rank * 2 * Dl:(rank + 1) * 2 * Dl].contiguous())
    res["up2"] = (u.detach().cpu(), tl.grad.cpu())
    torch.save(res, f"{out}.{rank}")
    parallel.barrier(ctx)
    torch.distributed.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = "/tmp/dd3"
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    parts = [torch.load(f"{out}.{r}") for r in range(2)]
    from passion_amd import ops
    from passion_amd.models.blocks import Feat, run_conv
    dev = torch.device("cuda:0")
    m = T._model(dev)
    S = 16
    t = torch.randn(1, S, 16, 16, 8, generator=torch.Generator().manual_seed(1)).to(dev)
    for name, mod, stride, norm in layers(m):
        ta = t.clone().requires_grad_(True)
        y = run_conv(mod, Feat(ta), norm=norm, out_dtype=None if norm else torch.float32).t
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2)).to(dev)
        y.backward(gy)
        ys = torch.cat([p[name][0] for p in parts], 1); dx = torch.cat([p[name][1] for p in parts], 1)
        dW = sum(p[name][2] for p in parts); db = sum(p[name][3] for p in parts)
        e = (dx - ta.grad.cpu()).abs()
        print(f"{name:18s} y {float((ys - y.detach().cpu()).abs().max()):.2e} dx {float(e.max()):.2e}/{float(ta.grad.abs().max()):.2e} per plane {[round(float(e[:, i].max()), 5) for i in range(S)]}"
              f" dW {float((dW - mod.weight.grad.cpu()).abs().max()):.2e}/{float(mod.weight.grad.abs().max()):.2e} db {float((db - mod.bias.grad.cpu()).abs().max()):.2e}")
    ta = t.clone().requires_grad_(True)
    u = ops.upsample(ta, 2); gu = torch.randn(u.shape, generator=torch.Generator().manual_seed(3)).to(dev); u.backward(gu)
    us = torch.cat([p["up2"][0] for p in parts], 1); dx = torch.cat([p["up2"][1] for p in parts], 1)
    e = (dx - ta.grad.cpu()).abs()
    print("up2 y", float((us - u.detach().cpu()).abs().max()), "dx", float(e.max()), [round(float(e[:, i].max()), 5) for i in range(S)])
