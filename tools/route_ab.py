#!/usr/bin/env python3
"""A/B of ROUTES in ONE process (round 4: +-3 us resolution; separate processes on one box differ by +-10 us, boxes by +-50 us).
Each argument is name=route:value[,route:value...] (empty = defaults); the routes are switched through mi3d_debug_set_route between
timed blocks of eager steps of bench.py's step object.  python tools/route_ab.py base= old=no_wide_store:1 [--rounds 4] [--steps 40]"""
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
    args = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 4
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 40
    size = int(sys.argv[sys.argv.index("--size") + 1]) if "--size" in sys.argv else 96
    batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 2
    cfgs = []
    for a in args:
        name, _, rs = a.partition("=")
        cfgs.append((name, {r.split(":")[0]: int(r.split(":")[1]) for r in rs.split(",") if r}))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if "--pool-main" in sys.argv:      # the step on a non-null stream (a CU-masked aux stream is a BLOCKING stream: it serialises with the null stream)
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    torch.manual_seed(0)
    x, y = bench.synth(batch, size, 1234)
    steps_by_cus = {}

    def step_for(cus):      # "aux_cus:N" is not a route: the aux stream is confined to N CUs per XCD (its own TrainStep object)
        if cus not in steps_by_cus:
            torch.manual_seed(0)
            model = mi.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(dev).train()
            t = TrainStep(model, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=torch.bfloat16, use_graph=False,
                          aux_wgrad=("--no-aux" not in sys.argv), aux_cus=cus)
            t.load_batch(x.to(dev), y.to(dev))
            steps_by_cus[cus] = t
        return steps_by_cus[cus]

    ts = step_for(0)

    def run(ts=ts):
        for _ in range(3):
            ts.step_static()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts.step_static()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    res = {n: [] for n, _ in cfgs}
    for _ in range(rounds):
        for name, routes in cfgs:
            routes = dict(routes)
            t = step_for(routes.pop("aux_cus", 0))
            saved = {k: _lib.get_route(k) for k in routes}
            for k, v in routes.items():
                _lib.set_route(k, v)
            res[name].append(run(t))
            for k, v in saved.items():
                _lib.set_route(k, v)
    for name, _ in cfgs:
        v = res[name]
        print(f"{name:16s} " + "  ".join(f"{t:.4f}" for t in v) + f"   mean {sum(v) / len(v):.4f} ms/step", flush=True)


if __name__ == "__main__":
    main()
