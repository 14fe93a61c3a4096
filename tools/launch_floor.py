"""Dispatch floor of a dependent chain of trivial kernels inside a captured graph (one stream)."""
import time, torch
x = torch.zeros(64, device="cuda")
big = torch.zeros(1 << 24, device="cuda")
for n, t in ((200, x), (200, big)):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            t.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                t.add_(1.0)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"numel {t.numel()}: {dt * 1e6 / n:.2f} us per kernel node ({n} nodes)")
