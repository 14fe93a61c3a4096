#!/usr/bin/env python3
"""Diagnostic (round 4): the exchange path on a 1-rank RCCL group, measured IN ONE PROCESS (no box-to-box noise, no first-use transients):
ms per step of the same TrainStep with (a) the full exchange path, (b) the same path with every collective replaced by nothing,
(c) no exchange path at all (do_comm off), alternating, three rounds.  python tools/comm_step.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import multimodal_segmentation_project_amd as mi  # noqa: E402
from multimodal_segmentation_project_amd.trainer import TrainStep  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = mi.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(dev).train()
    ts = TrainStep(model, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=torch.bfloat16, use_graph=False, force_comm=True)
    x, y = bench.synth(2, 96, 1234)
    ts.load_batch(x.to(dev), y.to(dev))
    real_avg = ts.comm.average_

    def run(steps=40):
        for _ in range(3):
            ts.step_static()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts.step_static()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    for rep in range(3):
        ts.do_comm = True
        ts.comm.average_ = real_avg
        full = run()
        ts.comm.average_ = lambda t: None
        skip = run()
        ts.comm.average_ = real_avg
        ts.do_comm = False
        off = run()
        print(f"exchange path {full:.4f}   without the collectives {skip:.4f}   no exchange path {off:.4f} ms/step", flush=True)
    ts.do_comm = True
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
