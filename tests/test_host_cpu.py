"""Host-side logic that needs no GPU: the module surface (constructors, state_dict, attributes), the C-ABI
library (loads, exports and binds every symbol of include/mi3d.h with matching arity), plan queries, and loud
failure instead of any CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import multimodal_segmentation_project_amd as mi
from multimodal_segmentation_project_amd import _lib, engine, unet_dann
from multimodal_segmentation_project_amd.dann import DomainDiscriminator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    hdr = open(os.path.join(ROOT, "include", "mi3d.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|int64_t|const char\*)\s+(mi3d_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_library_exports_every_declared_symbol():
    fns = _header_functions()
    assert len(fns) >= 35
    lib = _lib.lib()
    for name, nargs in fns.items():
        assert hasattr(lib, name), f"libmi3d.so lacks {name}"
        assert name in _lib._SIGS, f"_lib.py does not bind {name}"
        assert len(_lib._SIGS[name][1]) == nargs, f"{name}: header has {nargs} args, binding {len(_lib._SIGS[name][1])}"
    assert set(_lib._SIGS) == set(fns)
    assert lib.mi3d_abi_version() == 5


def test_state_dict_surface_matches_reference(golden):
    g = golden("default_unet")
    for cls in (mi.UNet3D, unet_dann.UNet3D):
        torch.manual_seed(0)
        m = cls(in_channels=1, out_channels=4, dropout_rate=0.0)
        sd = m.state_dict()
        assert len(sd) == 136
        assert sorted(sd.keys()) == list(g["param_keys"])
        assert sum(p.numel() for p in m.parameters()) == 5647908
        assert sum(b.numel() for b in m.buffers()) == 2962
        assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == list(g["param_shapes"])
        assert [str(sd[k].dtype) for k in sorted(sd)] == list(g["param_dtypes"])
        # a vanilla-torch module tree with the reference's layout loads strictly in both directions
        m2 = cls(in_channels=1, out_channels=4)
        m2.load_state_dict(sd, strict=True)
        for a in ("encoder", "pool", "bottleneck", "upconvs", "decoder", "final_conv", "output_activation", "dropout_rate"):
            assert hasattr(m, a)
        assert list(dict(m.named_parameters()).keys()) == list(g["grad_names"])
    # freezing API used by train_unet.py:31-43 / finetune_ct.py:270-304
    for p in list(m.encoder.parameters()) + list(m.bottleneck.parameters()):
        p.requires_grad = False
    assert sum(p.requires_grad for p in m.parameters()) == 8 * 4 + 8 + 2


def test_default_constructor_signature():
    m = mi.UNet3D()
    assert m.final_conv.out_channels == 1 and m.dropout_rate == 0.1
    assert [b.double_conv[0].out_channels for b in m.encoder] == [16, 32, 64, 128]
    d = DomainDiscriminator(256)
    assert sorted(d.state_dict().keys()) == sorted(f"net.{i}.{s}" for i in (0, 3, 6, 8) for s in ("weight", "bias"))
    assert sum(p.numel() for p in d.parameters()) == 107074


def test_plan_queries_and_argument_errors():
    m = mi.UNet3D(in_channels=1, out_channels=4)
    x = torch.zeros(2, 1, 96, 96, 96)
    desc = engine.build_desc(m, x, torch.bfloat16)
    lib = _lib.lib()
    assert lib.mi3d_unet_num_params(C.byref(desc)) == 82
    assert lib.mi3d_unet_num_buffers(C.byref(desc)) == 54
    assert lib.mi3d_unet_num_segments(C.byref(desc)) == 10
    ws_bf16 = lib.mi3d_unet_workspace_bytes(C.byref(desc))
    desc32 = engine.build_desc(m, x, torch.float32)
    ws_f32 = lib.mi3d_unet_workspace_bytes(C.byref(desc32))
    assert 1e9 < ws_bf16 < 8e9 and ws_bf16 < ws_f32 < 16e9
    # sum over 18 dropout layers of N*C
    chans = [16, 16, 32, 32, 64, 64, 128, 128, 256, 256, 128, 128, 64, 64, 32, 32, 16, 16]
    assert lib.mi3d_unet_dropout_count(C.byref(desc)) == 2 * sum(chans)
    # segments cover every parameter exactly once
    seen = []
    r = (C.c_int * 4)()
    for seg in range(10):
        assert lib.mi3d_unet_segment_params(C.byref(desc), seg, r) == 0
        seen += list(range(r[0], r[1])) + (list(range(r[2], r[3])) if r[2] >= 0 else [])
    assert sorted(seen) == list(range(82))
    # shape the plan cannot run -> error code + message, no crash
    # sides not divisible by 16 are planned (models/unet.py:81-83 nearest-resize route) ...
    odd = engine.build_desc(m, torch.zeros(1, 1, 20, 20, 20), torch.float32)
    assert lib.mi3d_unet_workspace_bytes(C.byref(odd)) > 0
    # ... a volume with no voxel left at the bottleneck is not
    bad = engine.build_desc(m, torch.zeros(1, 1, 20, 12, 20), torch.float32)
    assert lib.mi3d_unet_workspace_bytes(C.byref(bad)) == 0
    assert b"too small" in lib.mi3d_last_error()


def test_no_cpu_fallback():
    m = mi.UNet3D(in_channels=1, out_channels=4)
    with pytest.raises(_lib.Mi3dError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 16, 16, 16))
    with pytest.raises(_lib.Mi3dError):
        mi.combined_loss(torch.zeros(1, 4, 4, 4, 4), torch.zeros(1, 1, 4, 4, 4, dtype=torch.long))
    with pytest.raises(_lib.Mi3dError):
        m.encoder[0].double_conv[0](torch.zeros(1, 1, 4, 4, 4))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "multimodal_segmentation_project_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn


def test_checkpoint_wire_format_roundtrip(tmp_path):
    """the reference's checkpoint dict (train_unet.py:477-486) and its tolerant loaders
    (finetune_ct.py:246-268 raw / wrapped, test_model.py:381-385 'module.' prefix)."""
    import torch
    import multimodal_segmentation_project_amd as mi
    from multimodal_segmentation_project_amd import checkpoint as ck
    torch.manual_seed(3)
    m = mi.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    with torch.no_grad():
        for b in m.buffers():
            if b.dtype.is_floating_point:
                b.add_(0.25)
    path = tmp_path / "best_model_x.pth"
    saved = ck.save_checkpoint(str(path), m, epoch=7, train_loss=1.5, val_loss=1.25, train_dice=0.5, val_dice=0.625,
                               encoder_frozen=True, task_loss=0.75)
    assert set(saved) >= {"epoch", "model_state_dict", "optimizer_state_dict", "train_loss", "val_loss", "train_dice",
                          "val_dice", "encoder_frozen", "task_loss"}
    assert len(saved["model_state_dict"]) == 136
    m2 = mi.UNet3D(in_channels=1, out_channels=4)
    meta, res = ck.load_model(m2, str(path))
    assert meta["epoch"] == 7 and meta["encoder_frozen"] is True and not res.missing_keys and not res.unexpected_keys
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert a.dtype == b.dtype and torch.equal(a, b), k
    # raw state_dict and DDP-prefixed state_dict
    m3 = mi.UNet3D(in_channels=1, out_channels=4)
    ck.load_model(m3, {"module." + k: v for k, v in m.state_dict().items()})
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m3.state_dict().values()))
    m4 = mi.UNet3D(in_channels=1, out_channels=4)
    ck.load_model(m4, {"model_state_dict": m.state_dict()})                 # distill_unet.py:256 format
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m4.state_dict().values()))
    # a plain torch.nn restatement with the reference's layout loads the same file strictly (interchange)
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import torch_ref
    sd = ck.extract_state_dict(torch.load(str(path)))
    x = torch.randn(1, 1, 16, 16, 16)
    logits, _, _ = torch_ref.unet3d_forward(sd, x, train=False)
    assert tuple(logits.shape) == (1, 4, 16, 16, 16) and torch.isfinite(logits).all()


def test_trainer_host_logic_without_gpu():
    """Host-side pieces of trainer.py that need no device: loss-name validation (train_unet.py:538 choices), the
    `param_groups` surface a torch LR scheduler drives (train_unet.py:381,442), and the constructor refusing CPU models."""
    from multimodal_segmentation_project_amd import trainer
    for name in ("combined", "ce", "dice", "tversky", "ce_tversky"):
        cfg = trainer._loss_cfg(name)
        assert cfg.w_kd == 0.0
    with pytest.raises(_lib.Mi3dError, match="unknown loss"):
        trainer._loss_cfg("focal")
    kd = trainer._loss_cfg("combined", kd_alpha=0.7, temperature=2.0)       # distillation_loss, utils/metrics.py:169-190
    assert abs(kd.w_ce - 0.21) < 1e-6 and abs(kd.w_reg - 0.49) < 1e-6 and abs(kd.w_kd - 0.3) < 1e-6 and kd.temperature == 2.0
    m = mi.UNet3D(in_channels=1, out_channels=4, features=[4, 8])
    opt = trainer.ArenaAdamW(m.parameters(), 1e-3, (0.9, 0.999), 1e-8, 0.01)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="max", patience=0, factor=0.1, min_lr=1e-6)
    sched.step(0.5)
    sched.step(0.4)
    assert abs(opt.param_groups[0]["lr"] - 1e-4) < 1e-12
    with pytest.raises(_lib.Mi3dError):
        opt.step()
    with pytest.raises(_lib.Mi3dError, match="no CPU fallback"):
        trainer.TrainStep(m)


def test_preprocess_and_infer_symbols_are_bound():
    """Round-2 ABI additions are declared, exported and bound with the right arity (no compute without a GPU)."""
    lib = _lib.lib()
    for name in ("mi3d_unet_infer", "mi3d_preprocess_ct", "mi3d_preprocess_mri", "mi3d_preprocess_mri_workspace_bytes",
                 "mi3d_remap_labels", "mi3d_scale"):
        assert hasattr(lib, name) and name in _lib._SIGS
    assert lib.mi3d_preprocess_mri_workspace_bytes() >= 1024 * 8
    from multimodal_segmentation_project_amd import preprocess as P
    with pytest.raises(_lib.Mi3dError, match="no CPU fallback"):
        P.preprocess_ct(torch.zeros(4, 4, 4))
    with pytest.raises(_lib.Mi3dError, match="no CPU fallback"):
        P.remap_labels(torch.zeros(4, 4, 4, dtype=torch.long), "amos_ct")


def test_kernel_roofline_tool_assigns_every_launch_of_the_committed_timeline(tmp_path):
    """tools/kernel_roofline.py walks the committed step timeline by kernel name; a renamed or new kernel family that it does not
    know would silently shift every later launch to the wrong layer (it happened once in round 3).  The committed round-3 timeline
    must come out with no unassigned launch and a positive algorithmic model for every conv launch."""
    import csv
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tl, pmc = os.path.join(root, "profiles", "r03_step_timeline.txt"), os.path.join(root, "profiles", "r03_pmc_step_traffic.json")
    out = tmp_path / "roofline.csv"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_roofline.py"), tl, pmc, "--out", str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert " 0 unassigned" in r.stdout, r.stdout.splitlines()[0]
    rows = list(csv.DictReader(open(out)))
    assert len(rows) >= 130
    for row in rows:
        if row["kernel"].startswith(("conv3_", "upconv_", "head_loss")):
            assert float(row["algo_MB"]) > 0 and row["layer"], row


@pytest.mark.parametrize("rnd", ["r03", "r04"])
def test_kernel_roofline_tool_on_every_committed_round(tmp_path, rnd):
    """The same check on each round's committed single-stream timeline (round 4 added kernel families: the paired pool kernels)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tl, pmc = os.path.join(root, "profiles", rnd + "_step_timeline.txt"), os.path.join(root, "profiles", rnd + "_pmc_step_traffic.json")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_roofline.py"), tl, pmc, "--out", str(tmp_path / "r.csv")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert " 0 unassigned" in r.stdout, r.stdout.splitlines()[0]


def test_bench_roofline_candidates_have_committed_traffic_records():
    """bench.py reports `roofline.traffic` from profiles/roofline_kernel_traffic.json (PMC passes cannot run inside the bench process):
    every candidate that can be the dominant launch of the default route has a record keyed like the candidate, with counter bytes
    no smaller than its algorithmic bytes, and the lookup honours its time guard."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    rec = json.load(open(os.path.join(root, "profiles", "roofline_kernel_traffic.json")))["records"]
    cands = {c["key"]: c for c in bench.roofline_candidates(96, 2)}
    assert {c["stream"] for c in cands.values()} == {"compute", "aux"}
    for key in ("wgrad_dec3_conv0", "dgrad_dec3_conv0", "fwd_dec3_conv0", "bwd_enc0_conv1_fused"):
        assert key in rec and rec[key]["algorithmic_bytes_per_launch"] == cands[key]["bytes"]
        assert rec[key]["hbm_bytes_per_launch"] >= 0.95 * cands[key]["bytes"]
        ms = rec[key]["ms_per_launch_when_measured"]
        assert bench._kernel_traffic(key, ms) == rec[key]["hbm_bytes_per_launch"]
        assert bench._kernel_traffic(key, 2.0 * ms) is None          # a kernel that changed: no stale number
    assert bench._kernel_traffic("no_such_kernel", 0.1) is None


def test_route_switches_are_read_once_and_set_through_the_abi():
    """Every kernel-selection switch lives in one struct (csrc/common.h MI3D_ROUTE_LIST): the environment is read at first use
    only, later changes go through mi3d_debug_set_route; unknown names fail loudly."""
    import os
    names = _lib.route_names()
    assert "no_defer_wgrad" in names and "ks_target" in names and len(names) == len(set(names))
    assert _lib.get_route("ks_target") == 128 and _lib.get_route("conv8") == 1
    os.environ["MI3D_NO_PERSIST"] = "1"          # after the first use: ignored
    try:
        assert _lib.get_route("no_persist") == 0
    finally:
        del os.environ["MI3D_NO_PERSIST"]
    with _lib.routes(no_persist=1, ks_target=64):
        assert _lib.get_route("no_persist") == 1 and _lib.get_route("ks_target") == 64
    assert _lib.get_route("no_persist") == 0 and _lib.get_route("ks_target") == 128
    with pytest.raises(_lib.Mi3dError):
        _lib.set_route("no_such_route", 1)
