#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: 3D volumes/s, UNet3D 96^3 4-class training step
(forward + Dice/CE loss + backward [+ gradient all-reduce] + AdamW + metrics) per node at N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  `value` = whole-job volumes/s (global batch * K / max-over-ranks time), inputs resident
in HBM before timing.  `roofline` = the dominant kernel (measured live with HIP events on its own launches, same
shapes as in the step; the launch with the largest time per step) against the 8 TB/s HBM peak; `cpu_baseline` = the oracle's
vanilla-torch restatement timed on the host cores (rank 0, N=1 only, bounded sample, 16 and 8 threads).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_GB_PER_VOL_96_BF16 = 2.421   # SURVEY §8(d) / BASELINE.md §3: fwd+bwd compulsory traffic per 96^3 volume, bf16
ALGO_GB_PER_VOL_96_F32 = 4.708


def synth(n, s, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 1, s, s, s, generator=g)
    y = torch.randint(0, 4, (n, 1, s, s, s), generator=g)
    return x, y


def cpu_baseline(size, batch, max_seconds=24.0):
    """The oracle (oracle/torch_ref.py: vanilla torch.nn.functional restatement, fp32, validated against the reference-
    generated fixtures in tests/golden by tests/test_oracle_cpu.py) on the host cores, at the two thread counts BASELINE.md
    section 4 asks for: the box's CPU share (16 on a 1-GPU box) and 8 (the survey container's probe).  `value` is the faster."""
    from oracle import torch_ref
    import multimodal_segmentation_project_amd as mi
    torch.manual_seed(0)
    m = mi.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)     # CPU construction only (parameter shells)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    x, y = synth(batch, size, 1234)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # a 1-GPU box exposes all host cores but grants a 16-core share (oversubscribing 256 threads ran 14x slower)
    share = min(avail, int(os.environ.get("MI3D_CPU_BASELINE_THREADS", "16")))
    runs = {}
    for cores in sorted({share, min(share, 8)}, reverse=True):
        torch.set_num_threads(cores)
        times = []
        t_all = time.time()
        for it in range(4):
            t0 = time.time()
            logits, _, _ = torch_ref.unet3d_forward(sd, x, train=True)
            loss = torch_ref.seg_loss(logits, y, "combined")
            loss.backward()
            for v in sd.values():
                v.grad = None
            dt = time.time() - t0
            if it > 0:
                times.append(dt)
            if time.time() - t_all > max_seconds / 2 and times:
                break
        med = sorted(times)[len(times) // 2]
        runs[cores] = (batch / med, len(times))
    best = max(runs, key=lambda c: runs[c][0])
    return {"value": runs[best][0], "unit": "volumes/s", "cores": best, "kind": "port",
            "by_threads": {str(c): round(v[0], 4) for c, v in runs.items()},
            "validated_against": "tests/golden (fixtures produced by executing the reference; tests/test_oracle_cpu.py)",
            "sample": f"{runs[best][1]} fwd+loss+bwd steps of UNet3D {size}^3 N={batch} fp32 after 1 warm-up (median), per thread count"}


def _kernel_traffic(key, ms):
    """HBM bytes per launch of a roofline candidate from the committed PMC passes of this build (tools/collect_profiles.sh ->
    profiles/roofline_kernel_traffic.json: FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc runs of the whole step,
    corrected by tools/pmc_traffic.py).  A PMC pass cannot run inside this process; the committed value is reported only
    while the kernel still takes the time it took when the counters were collected (else null, not a stale number)."""
    tf = os.path.join(ROOT, "profiles", "roofline_kernel_traffic.json")
    try:
        rec = json.load(open(tf)).get("records", {}).get(key)
        if not rec:
            return None
        ref_ms = rec.get("ms_per_launch_when_measured")
        if ref_ms and abs(ms - ref_ms) > 0.25 * ref_ms:      # (a launch beside another stream's kernels varies by more than 10 % between boxes)
            return None
        return rec.get("hbm_bytes_per_launch")
    except Exception:   # noqa: BLE001
        return None


def roofline_candidates(size, batch):
    """The full-resolution conv launches of one step that can be the largest one, keyed on the LAYER (hook kinds: include/mi3d.h,
    mi3d_time_next_conv3_kernel).  Algorithmic bytes per launch = every operand of that launch once (bf16 activations, fp32 dW):
    weight gradient R x, R dy, W dW; input gradient / forward R in, W out; fused backward R dy, R x, W dx, W dW (both products of
    the layer share one read of dy in the ideal kernel).  `ordinal` = which dispatch of that kernel name inside one step
    (tools/collect_profiles.sh picks the PMC record with it)."""
    vox = batch * size ** 3
    c = []

    def add(key, kind, cin, cout, kernel, ordinal, layer, elems, flops_mul):
        c.append({"key": key, "kind": kind, "cin": cin, "cout": cout, "kernel": kernel, "ordinal": ordinal, "layer": layer,
                  "stream": "aux" if kind == 1 else "compute",
                  "bytes": vox * elems * 2 + (16 * 32 * 27 * 4 if kind in (0, 1) else 0), "flops": flops_mul * 27 * 32 * 16 * vox})
    add("wgrad_dec3_conv0", 1, 32, 16, "conv3_wgrad_mfma_kernel<1, 1, 27>", 1,
        "decoder.3.conv0 weight gradient (32->16 at full resolution; on the aux stream beside the deep-level chain)", 32 + 16, 2)
    add("dgrad_dec3_conv0", 3, 16, 32, "conv3_mfma_persist_kernel<2, 1, false>", 0,
        "decoder.3.conv0 input gradient (conv 16->32 at full resolution)", 16 + 32, 2)
    add("fwd_dec3_conv0", 2, 32, 16, "conv3_mfma_persist_kernel<1, 2, true>", 0,
        "decoder.3.conv0 forward (conv 32->16 at full resolution, BatchNorm partial sums fused)", 32 + 16, 2)
    add("bwd_dec3_conv0_fused", 0, 32, 16, "conv3_bwd_fused_persist_kernel<2, 1>", 0,
        "decoder.3.conv0 backward, input + weight gradient in one launch (single-stream route only)", 16 + 32 + 32, 4)
    c.append({"key": "bwd_enc0_conv1_fused", "kind": 0, "cin": 16, "cout": 16, "kernel": "conv3_bwd_fused_persist_kernel<1, 1>",
              "ordinal": 0, "stream": "compute",
              "layer": "encoder.0.conv1 backward, input + weight gradient in one launch (16->16 at full resolution)",
              "bytes": vox * 48 * 2 + 16 * 16 * 27 * 4, "flops": 4 * 27 * 16 * 16 * vox})
    return c


def _time_hooked(once, kind, cin, cout, iters):
    """ms per launch of the armed kernel inside `once()` (HIP events tightly around the kernel on ITS stream), or None when
    no launch of that kind / layer happens in it."""
    import ctypes as C
    from multimodal_segmentation_project_amd import _lib
    from multimodal_segmentation_project_amd._lib import call
    e0, e1 = C.c_void_p(), C.c_void_p()
    call("mi3d_timing_event_create", C.byref(e0))
    call("mi3d_timing_event_create", C.byref(e1))
    tot, n = 0.0, 0
    try:
        for _ in range(iters):
            call("mi3d_time_next_conv3_kernel", e0, e1, kind, cin, cout)
            once()
            if _lib.lib().mi3d_time_hook_fired() != 1:       # also disarms: the events are never left armed
                return None
            torch.cuda.synchronize()
            t = C.c_float()
            call("mi3d_event_elapsed_ms", e0, e1, C.byref(t))
            tot += t.value
            n += 1
    finally:
        _lib.lib().mi3d_time_hook_fired()
        call("mi3d_event_destroy", e0)
        call("mi3d_event_destroy", e1)
    return tot / n if n else None


def roofline_dominant(size, batch, dtype_code, iters=10, ts=None):
    """Dominant kernel = the launch with the largest time on the step's critical path (the compute stream), CHOSEN BY MEASUREMENT
    among the full-resolution conv launches (roofline_candidates): each is timed live with HIP events recorded tightly around THAT kernel on the stream it is
    launched on, inside real training steps of the TrainStep `ts` (the kernel with the step's own data, layout and neighbours:
    a weight gradient on the aux stream is timed while the chain's kernels run beside it).  The others are reported in
    `others`.  Without `ts` (exact fp32 path, other workloads): the per-operator backward call of the 32->16 layer."""
    import ctypes as C
    from multimodal_segmentation_project_amd import _lib
    from multimodal_segmentation_project_amd._lib import call, ptr, stream_ptr
    dev = "cuda"
    n, d = batch, size
    if ts is not None and dtype_code == 1:
        was = ts.use_graph
        ts.use_graph = False                   # eager launches: the hook needs the launch to happen in this call
        for _ in range(2):
            ts.step_static()
        rows = []
        for cd in roofline_candidates(size, batch):
            ms = _time_hooked(ts.step_static, cd["kind"], cd["cin"], cd["cout"], iters)
            if ms is None:
                continue
            ach = cd["bytes"] / (ms * 1e-3) / 1e9
            rows.append({"bound": "hbm", "kernel": cd["kernel"] + " = " + cd["layer"], "key": cd["key"], "stream": cd["stream"],
                         "measured": "inside training steps",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": _kernel_traffic(cd["key"], ms), "traffic_source": "profiles/roofline_kernel_traffic.json (PMC passes "
                         "of this build; null when the live time is off by > 10 % from the time the counters were collected at)",
                         "ms_per_launch": ms, "algorithmic_bytes_per_launch": cd["bytes"], "flops_per_launch": cd["flops"]})
        ts.use_graph = was
        if not rows:
            return None
        # dominant = the largest launch on the COMPUTE stream, i.e. on the step's critical path (one launch of its kernel per step:
        # its time is the kernel's average in the rocprofv3 summary of this command).  A weight gradient on the aux stream is
        # timed while the chain's kernels share the chip with it, so its time says more about the sharing than about the kernel:
        # it is listed under `others` with stream = "aux"
        rows.sort(key=lambda r: (r["stream"] != "compute", -r["ms_per_launch"]))
        top = rows[0]
        top["others"] = [{k: r[k] for k in ("key", "stream", "kernel", "ms_per_launch", "achieved", "frac", "traffic",
                                            "algorithmic_bytes_per_launch")} for r in rows[1:]]
        return top
    cin, cout = 32, 16
    esz = 2 if dtype_code == 1 else 4
    T = torch.bfloat16 if dtype_code == 1 else torch.float32
    vox = n * d ** 3
    algo_bytes = vox * (cout + cin + cin) * esz + cout * cin * 27 * 4
    x = torch.randn((n, d, d, d, cin), device=dev).to(T)
    dy = torch.randn((n, d, d, d, cout), device=dev).to(T)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05
    dx = torch.empty_like(x)
    dW, db = torch.empty_like(w), torch.empty(cout, device=dev)
    wsb = _lib.lib().mi3d_conv3_workspace_bytes(cin, cout, n, d, d, d)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    s = stream_ptr()

    def once():
        call("mi3d_conv3_backward", dtype_code, dtype_code, ptr(x), cin, cin, ptr(w), ptr(dy), cout, cout, ptr(dx), cin, ptr(dW),
             ptr(db), 0, n, d, d, d, ptr(ws), wsb, s)
    for _ in range(2):
        once()
    kernel = "conv3_bwd_fused_persist_kernel<2, 1> = conv3 backward 32->16 (input + weight gradient in one launch)"
    ms = _time_hooked(once, 0, cin, cout, iters) if dtype_code == 1 else None
    if ms is None:      # exact fp32 path / shapes without the persistent kernels: the whole operator call is timed
        kernel = "conv3 backward 32->16, whole operator call (" + ("fp32 direct kernels" if dtype_code == 0 else "generic kernels") + ")"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            once()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / iters
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "measured": "per-operator call",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "ms_per_launch": ms, "algorithmic_bytes_per_launch": algo_bytes,
            "flops_per_launch": 2 * 2 * 27 * cin * cout * vox}


class _SclkSampler:
    """Shader clock during the timed region: every 20 ms the current level ('*') of every card's pp_dpm_sclk; the busiest card
    (highest mean) is ours on a shared host.  Host-side file reads only: the step is a graph replay and does not wait on the host."""

    def __init__(self):
        import glob
        import threading
        self.files = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.samples = [[] for _ in self.files]
        self._stop = threading.Event()
        self._thr = threading.Thread(target=self._run, daemon=True)

    def _read(self, f):
        try:
            for line in open(f):
                if "*" in line:
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError):
            pass
        return None

    def _run(self):
        while not self._stop.is_set():
            for i, f in enumerate(self.files):
                v = self._read(f)
                if v is not None:
                    self.samples[i].append(v)
            self._stop.wait(0.02)

    def start(self):
        if self.files:
            self._thr.start()

    def stop(self):
        if not self.files:
            return None
        self._stop.set()
        self._thr.join(timeout=1.0)
        best = max((s for s in self.samples if s), key=lambda s: sum(s) / len(s), default=None)
        if not best:
            return None
        return {"mean": round(sum(best) / len(best), 1), "min": min(best), "max": max(best), "samples": len(best)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--batch", type=int, default=2, help="per-GPU batch (BASELINE config 2: 2)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--dropout", type=float, default=0.0, help="run_training.sh:31 ships --dropout_rate 0.0")
    ap.add_argument("--no-graph", action="store_true", help="(default since round 4) eager launches")
    ap.add_argument("--graph", action="store_true", help="one hipGraph per step on one stream (no aux-stream weight gradients)")
    ap.add_argument("--no-aux-wgrad", action="store_true",
                    help="keep every weight gradient on the data-gradient chain (default: the decoder / deep-level ones run on a second stream)")
    ap.add_argument("--chain-prio", default=os.environ.get("MI3D_BENCH_CHAIN_PRIO", "normal"), choices=["normal", "high"],
                    help="priority class of the stream the step runs on (the aux stream of the weight gradients has the lowest)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--sclk", action="store_true",
                    help="sample the shader clock (sysfs pp_dpm_sclk of the busiest card) during the timed steps -> \"sclk_mhz\" in the JSON "
                         "line: boxes and builds differ in clock, an A/B should say so")
    ap.add_argument("--roofline-only", action="store_true", help="run only the dominant-kernel loop (for rocprofv3 --pmc passes)")
    ap.add_argument("--graph-segments", action="store_true", help="world > 1 / --force-comm: replay one hipGraph per comm-free run of kernels")
    ap.add_argument("--force-comm", action="store_true",
                    help="N=1 only: create a 1-rank RCCL group and drive the DP bucket path (side stream, all-reduces, segmented graphs)")
    ap.add_argument("--serial-forwards", action="store_true",
                    help="distill / dann: run the two independent forwards one after the other instead of on two streams")
    ap.add_argument("--workload", default="train", choices=["train", "distill", "dann", "eval"],
                    help="train = the headline metric (BASELINE config 2/3); distill = config 5 step (student + frozen teacher); "
                         "dann = config 4 step (N source + N target volumes per GPU); eval = inference forward + loss + metrics")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif a.force_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        os.environ["MI3D_FORCE_COMM"] = "1"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)

    import multimodal_segmentation_project_amd as mi
    from multimodal_segmentation_project_amd.trainer import TrainStep

    if a.chain_prio == "high":
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    if a.roofline_only:
        print(json.dumps(roofline_dominant(a.size, a.batch, 1 if a.dtype == "bf16" else 0, iters=a.steps)))
        return
    torch.manual_seed(0)
    cdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    # one hipGraph per step at world 1.  With gradient exchange (world > 1) the step is launched eagerly: the two exchanges
    # (dp.bucket_ranges) cut the graph into segments, and eager launches measured faster than segmented replay (DESIGN.md §6:
    # 2.447 vs 2.517 ms on the 1-rank RCCL path; eager == full graph without communication); --graph-segments forces the segments
    # round 4: the default step launches eagerly with the weight gradients of the decoder / deep levels on an aux stream
    # (TrainStep aux_wgrad) -- at world 1 and at world > 1 alike, so the single-GPU number and the DP step use one launch mode.
    # --graph replays ONE hipGraph on one stream instead (the round-3 mode; this runtime's graph executor cannot take the fork)
    use_graph = a.graph and (not a.no_graph) and ((world == 1 and not a.force_comm) or a.graph_segments)
    if a.graph_segments and not a.no_graph:
        use_graph = True
    x, y = synth(a.batch, a.size, 1234 + rank)
    if a.workload == "dann":
        from multimodal_segmentation_project_amd import unet_dann
        from multimodal_segmentation_project_amd.dann import DomainDiscriminator
        from multimodal_segmentation_project_amd.trainer import DannStep
        model = unet_dann.UNet3D(in_channels=1, out_channels=4, dropout_rate=a.dropout).to(dev).train()
        torch.manual_seed(3)
        disc = DomainDiscriminator(256).to(dev).train()
        ts = DannStep(model, disc, loss="ce_tversky", lambda_domain=0.2, lr=1e-3, weight_decay=0.01, compute_dtype=cdt,
                      use_graph=use_graph, overlap_forwards=not a.serial_forwards)
        xt, _ = synth(a.batch, a.size, 4321 + rank)
        ts.load_batch(x.clamp(0, 1).to(dev), y.to(dev), ((xt - xt.min()) / (xt.max() - xt.min())).to(dev))
    else:
        model = mi.UNet3D(in_channels=1, out_channels=4, dropout_rate=a.dropout).to(dev).train()
        teacher = None
        if a.workload == "distill":
            torch.manual_seed(1)
            teacher = mi.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(dev).eval()
        ts = TrainStep(model, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=cdt, kd_teacher=teacher,
                       use_graph=use_graph, aux_wgrad=(False if a.no_aux_wgrad else None), overlap_teacher=not a.serial_forwards)
        ts.load_batch(x.to(dev), y.to(dev))
    if a.workload == "eval":
        xd, yd = x.to(dev), y.to(dev)
        ts.step_static = lambda: ts.evaluate(xd, yd)

    for _ in range(a.warmup):
        ts.step_static()
    sclk = _SclkSampler() if (a.sclk and rank == 0) else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if sclk:
        sclk.start()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = ts.step_static()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sclk_mhz = sclk.stop() if sclk else None
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    met = out.detach().cpu().tolist()

    if rank == 0:
        global_batch = a.batch * world
        ms = dt / a.steps * 1e3
        value = global_batch * a.steps / dt
        scale = (a.size / 96.0) ** 3
        algo_gb_step = (ALGO_GB_PER_VOL_96_BF16 if a.dtype == "bf16" else ALGO_GB_PER_VOL_96_F32) * scale * a.batch
        res = {
            "metric": "3D volumes/sec (96^3, 4-class) fwd+bwd per node", "value": value, "unit": "volumes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": (f"UNet3D(1->4, features 16/32/64/128) {a.size}^3 patch, per-GPU batch {a.batch}, "
                                    f"fwd + Dice/CE loss + bwd + AdamW + metrics, dropout {a.dropout}") if a.workload == "train"
                       else f"{a.workload} step, UNet3D {a.size}^3, per-GPU batch {a.batch}",
                       "global_batch": global_batch, "parallelism": f"dp{world}", "hipgraph": bool(ts.use_graph),
                       "comm_path": bool(ts.do_comm)},
            "final_step": {"loss": met[0], "iou": met[1], "dice": met[2], "acc": met[3]},
            "step_hbm_frac": algo_gb_step / (ms * 1e-3) / HBM_PEAK_GBS,
            "step_algorithmic_gb": algo_gb_step,
            **({"sclk_mhz": sclk_mhz} if sclk_mhz else {}),
        }
        if not a.no_roofline:
            res["roofline"] = roofline_dominant(a.size, a.batch, 1 if a.dtype == "bf16" else 0,
                                                ts=ts if (a.workload == "train" and a.dropout == 0.0 and not ts.do_comm) else None)
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.size, a.batch)
        print(json.dumps(res))
    sys.stdout.flush()
    if world > 1 or a.force_comm:
        dist.barrier()               # rank 0 may still be measuring its roofline kernel: leave together
        dist.destroy_process_group()


def _main_with_clean_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner on every
    rank when its first communicator is created), so the process runs with file descriptor 1 pointing at stderr and only
    the result lines reach the real stdout."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    out = os.fdopen(real, "w")
    py_stdout = sys.stdout
    sys.stdout = out
    try:
        main()
    finally:
        out.flush()
        sys.stdout = py_stdout


if __name__ == "__main__":
    _main_with_clean_stdout()
