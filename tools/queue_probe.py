#!/usr/bin/env python3
"""What does a kernel on ANOTHER stream cost a dependent chain of small kernels on the current stream?  (round 4: one unrelated
one-thread kernel per step on a second stream costs the training step ~170 us unprofiled, profiles/r04_experiments_second_queue.txt.)
Chain = N dependent 8-us kernels (8 workgroups each) on the current stream: GPU-bound; `poke` = one one-thread kernel on the candidate stream in front of every chain.
Prints us per chain for: no poke, poke on the same stream, poke on each candidate stream (a high-priority one, pool streams, a
low-priority one)."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from multimodal_segmentation_project_amd._lib import call  # noqa: E402
from multimodal_segmentation_project_amd.trainer import _priority_stream  # noqa: E402


def run(n_chain, reps, poke_stream, flag, pflag, buf=None):
    main = torch.cuda.current_stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        if poke_stream is not None:
            call("mi3d_flag_set", pflag.data_ptr(), r, poke_stream.cuda_stream)
        for i in range(n_chain):
            call("mi3d_debug_occupy_cus", 8, 8, buf.data_ptr(), buf.numel(), main.cuda_stream)      # 8 workgroups x 8 us: GPU-bound chain
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    flag = torch.zeros(2, dtype=torch.int64, device=dev)
    pflag = torch.zeros(2, dtype=torch.int64, device=dev)
    buf = torch.zeros(4096, dtype=torch.float32, device=dev)
    n_chain, reps = 140, 60
    cands = [("none", None), ("same", torch.cuda.current_stream())]
    cands.append(("high", torch.cuda.Stream(device=dev, priority=-1)))
    for i in range(6):
        cands.append((f"pool{i}", torch.cuda.Stream(device=dev)))
    cands.append(("low", _priority_stream(dev, +1)))
    for name, st in cands:
        run(n_chain, 10, st, flag, pflag, buf)
        ts = [run(n_chain, reps, st, flag, pflag, buf) for _ in range(3)]
        print(f"{name:8s} {min(ts):8.1f} us per chain of {n_chain}   ({' '.join(f'{t:.1f}' for t in ts)})", flush=True)


if __name__ == "__main__":
    main()
