#!/usr/bin/env python3
"""Diagnostic (round 4): aux-stream weight gradients on / off IN ONE PROCESS (routes are switched through the ABI between timed
blocks; eager launches read them per call): ms per step, alternating, four rounds.  python tools/aux_step.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import multimodal_segmentation_project_amd as mi  # noqa: E402
from multimodal_segmentation_project_amd import _lib  # noqa: E402
from multimodal_segmentation_project_amd.trainer import TrainStep  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    model = mi.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(dev).train()
    ts = TrainStep(model, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=torch.bfloat16, use_graph=False, aux_wgrad=True)
    x, y = bench.synth(2, 96, 1234)
    ts.load_batch(x.to(dev), y.to(dev))

    def run(steps=40):
        for _ in range(3):
            ts.step_static()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts.step_static()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    for rep in range(4):
        out = []
        for name, routes in (("aux", {}), ("chain", {"no_defer_wgrad": 1}), ("aux, deep only", {"defer_mask": 4}),
                             ("aux, decoder only", {"defer_mask": 3})):
            for k, v in routes.items():
                _lib.set_route(k, v)
            out.append(f"{name} {run():.4f}")
            for k in routes:
                _lib.set_route(k, {"no_defer_wgrad": 0, "defer_mask": 7}[k])
        print("   ".join(out), flush=True)


if __name__ == "__main__":
    main()
