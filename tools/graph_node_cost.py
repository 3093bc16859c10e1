"""What does a HIP-graph kernel node cost on this stack (ROCm 7.2 / torch 2.10)?  N small dependent launches, eager vs
graph replay, on one stream and fanned over 4 streams; plus N independent short kernels (gaps between dependent nodes).

    python tools/graph_node_cost.py [N]
"""
import sys
import time

import torch


def chain(xs, n, streams=None):
    if streams is None:
        for i in range(n):
            xs[0].add_(1.0)
        return
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    for j, s in enumerate(streams):
        with torch.cuda.stream(s):
            for i in range(n // len(streams)):
                xs[j].add_(1.0)
    for s in streams:
        cur.wait_stream(s)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t_host = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, t_host


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    dev = torch.device("cuda:0")
    for numel in (1024, 1 << 22):
        xs = [torch.zeros(numel, device=dev) for _ in range(4)]
        streams = [torch.cuda.Stream() for _ in range(4)]
        for label, st in (("1 stream", None), ("4 streams", streams)):
            wall, host = timeit(lambda: chain(xs, n, st))
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                chain(xs, n, st)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                chain(xs, n, st)
            gw, gh = timeit(g.replay)
            print(f"numel {numel:8d} {label:9s} n={n}: eager wall {wall * 1e3:7.3f} ms (host {host * 1e3:7.3f}) = {wall / n * 1e6:6.2f} us/launch | "
                  f"graph wall {gw * 1e3:7.3f} ms (host {gh * 1e3:7.3f}) = {gw / n * 1e6:6.2f} us/node", flush=True)


if __name__ == "__main__":
    main()
