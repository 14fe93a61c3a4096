"""Round-4 GPU parity tests (through the C ABI).

  * the bench's OWN path -- TrainStep(bf16, fused head, one hipGraph, weight gradients on the aux stream) -- pinned directly
    to the reference run of BASELINE config 2 (tests/golden/config2_96.npz; /root/reference/train_unet.py:220-232), instead
    of transitively through the autograd path;
  * the deferred-weight-gradient route (mi3d_unet_backward with an aux stream: the backward's critical path is the
    input-gradient chain alone, train_unet.py:225 fixes no order between a layer's two gradients) is BITWISE the
    single-stream route: same slab partition, same split-K factors, same kernels' K order.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import multimodal_segmentation_project_amd as mi  # noqa: F401
from multimodal_segmentation_project_amd import _lib
from multimodal_segmentation_project_amd.trainer import TrainStep
from multimodal_segmentation_project_amd.unet import UNet3D

from test_gpu_round2 import DEV, check_summary, default_model, relerr, synth


@pytest.mark.parametrize("dtype,graph", [(torch.bfloat16, True), (torch.bfloat16, False), (torch.float32, False)])
def test_trainstep_bench_path_vs_config2_96_reference_fixture(golden, dtype, graph):
    """bench.py's step object on bench.py's batch (synth(2, 96, 1234)), lr = 0 so that the parameters stay the fixture's:
    loss, Dice / IoU / accuracy, BatchNorm buffers after the step and all 82 gradient norms + the stored gradient slices, read
    from the gradient arena the fused AdamW consumes.  bf16: the north-star tolerances (loss 2e-3, Dice/IoU 1e-3, accuracy
    2e-3, gradients within 1.5x the reference's own autocast deviation); fp32: tight."""
    g = golden("config2_96")
    fp32 = dtype == torch.float32
    m = default_model().to(DEV).train()
    ts = TrainStep(m, loss="combined", lr=0.0, weight_decay=0.01, compute_dtype=dtype, use_graph=graph, keep_logits=False)
    assert (ts.aux_stream is not None) == (not graph)        # default: aux-stream weight gradients for eager launches only
    x, y = synth(2, 96, 1234)
    ts.load_batch(x.to(DEV), y.to(DEV))
    out = ts.step_static().cpu()
    assert ts._static["fused_head"] == (not fp32)
    np.testing.assert_allclose(float(out[0]), float(g["loss"]), rtol=2e-5 if fp32 else 2e-3)
    assert abs(float(out[1]) - float(g["iou"])) < (1e-5 if fp32 else 1e-3)
    assert abs(float(out[2]) - float(g["dice"])) < (1e-5 if fp32 else 1e-3)
    assert abs(float(out[3]) - float(g["acc"])) < (1e-5 if fp32 else 2e-3)
    for p, o in zip(ts.arena.params, ts.arena.offsets):      # the arena views ARE the .grad tensors check_summary reads
        assert p.grad is not None and p.grad.data_ptr() == ts.arena.g.data_ptr() + 4 * o
    check_summary(g, "", m, fp32, yard=g["autocast_bf16/grad_relerr"])
    torch.manual_seed(0)                     # parameters untouched by the lr = 0 update
    ref = UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    for (k, a), (_, b) in zip(m.named_parameters(), ref.named_parameters()):
        assert torch.equal(a.detach().cpu(), b.detach()), k


@pytest.mark.parametrize("size,graph", [(96, True), (96, False), (32, True), (48, False)])
def test_deferred_weight_gradients_are_bitwise_the_chain_route(size, graph):
    """TrainStep(aux_wgrad=True) (default: decoder full-resolution and all deep-level weight gradients on the aux stream, the
    chain runs the stand-alone input-gradient kernels) against aux_wgrad=False (every layer's fused launch on one stream), two
    steps each from the same initial state: metrics, every gradient, parameters after AdamW and BatchNorm buffers bit for bit.
    32^3 / 48^3: other level -> tiling assignments (which levels are 'deep')."""
    x, y = synth(2, size, 4321, blocky=True)
    res = []
    for aux in (True, False):
        m = default_model().to(DEV).train()
        ts = TrainStep(m, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=torch.bfloat16, use_graph=graph, aux_wgrad=aux)
        assert (ts.aux_stream is not None) == aux
        ts.load_batch(x.to(DEV), y.to(DEV))
        outs = [ts.step_static().clone() for _ in range(2)]
        torch.cuda.synchronize()
        res.append((outs, ts.arena.g.clone(), ts.arena.p.clone(), [b.clone() for b in m.buffers()]))
        spans = [(k, o, p.numel()) for (k, p), o in zip(m.named_parameters(), ts.arena.offsets)]
        ts.close()
    (o0, g0, p0, b0), (o1, g1, p1, b1) = res
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    assert torch.isfinite(g0).all() and float(g0.abs().max()) > 0
    if not torch.equal(g0, g1):
        bad = [(k, relerr(g0[o:o + n].cpu(), g1[o:o + n].cpu())) for k, o, n in spans if not torch.equal(g0[o:o + n], g1[o:o + n])]
        raise AssertionError(f"{len(bad)} gradient tensors differ between the routes: {bad[:8]}")
    assert torch.equal(p0, p1)
    for a, b in zip(b0, b1):
        assert torch.equal(a, b)


@pytest.mark.parametrize("size,graph,aux", [(96, False, True), (96, True, False), (32, False, True), (48, False, True), (64, True, False)])
def test_apply_on_load_is_bitwise_the_launched_apply(routes, size, graph, aux):
    """Deep levels: conv0's BatchNorm + ReLU + Dropout3d is applied in conv1's staging pass (forward; z1 written as a by-product),
    and a deferred layer's BatchNorm backward in the staging pass of its input-gradient conv (dy written as a by-product for the
    weight gradient on the aux stream) -- models/unet.py:12-18 fixes WHAT is computed, not in which launch.  Against the route
    with the bn_apply / bn_bwd_apply launches (the default; apply_on_load = 0): metrics, every gradient, parameters after AdamW, BatchNorm
    buffers incl. num_batches_tracked, bit for bit; with Dropout3d masks (per-sample scales enter the staging coefficients)."""
    if not _lib.lib().mi3d_debug_experiments():
        pytest.skip("apply on load is compiled into experiment builds only (make EXPERIMENTS=1): measured slower, off the product path")
    x, y = synth(2, size, 777, blocky=True)
    res = []
    for on in (1, 0):
        routes.set("apply_on_load", on)
        m = default_model().to(DEV).train()
        ts = TrainStep(m, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=torch.bfloat16, use_graph=graph, aux_wgrad=aux)
        ts.load_batch(x.to(DEV), y.to(DEV))
        outs = [ts.step_static().clone() for _ in range(2)]
        torch.cuda.synchronize()
        res.append((outs, ts.arena.g.clone(), ts.arena.p.clone(), [b.clone() for b in m.buffers()]))
        spans = [(k, o, p.numel()) for (k, p), o in zip(m.named_parameters(), ts.arena.offsets)]
        ts.close()
    routes.reset("apply_on_load")
    (o0, g0, p0, b0), (o1, g1, p1, b1) = res
    for a, b in zip(o0, o1):
        assert torch.equal(a, b), (a, b)
    assert torch.isfinite(g0).all() and float(g0.abs().max()) > 0
    if not torch.equal(g0, g1):
        bad = [(k, relerr(g0[o:o + n].cpu(), g1[o:o + n].cpu())) for k, o, n in spans if not torch.equal(g0[o:o + n], g1[o:o + n])]
        raise AssertionError(f"{len(bad)} gradient tensors differ between the routes: {bad[:8]}")
    assert torch.equal(p0, p1)
    for (k, _), a, b in zip(m.named_buffers(), b0, b1):
        assert torch.equal(a, b), k


def test_apply_on_load_with_dropout_masks(routes):
    """The same comparison with Dropout3d(p = 0.3) in train mode: the injected per-(sample, channel) scales must reach the staging
    coefficients of the right sample (tile -> sample) in both directions."""
    if not _lib.lib().mi3d_debug_experiments():
        pytest.skip("apply on load is compiled into experiment builds only (make EXPERIMENTS=1)")
    x, y = synth(2, 32, 778, blocky=True)
    res = []
    for on in (1, 0):
        routes.set("apply_on_load", on)
        torch.manual_seed(0)            # also the seed of the model's counter-based dropout stream: the same masks on both routes
        m = UNet3D(in_channels=1, out_channels=4, dropout_rate=0.3).to(DEV).train()
        ts = TrainStep(m, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=torch.bfloat16, use_graph=False, aux_wgrad=True)
        ts.load_batch(x.to(DEV), y.to(DEV))
        outs = [ts.step_static().clone() for _ in range(2)]
        torch.cuda.synchronize()
        res.append((outs, ts.arena.g.clone(), ts.arena.p.clone()))
        ts.close()
    routes.reset("apply_on_load")
    (o0, g0, p0), (o1, g1, p1) = res
    for a, b in zip(o0, o1):
        assert torch.equal(a, b), (a, b)
    assert torch.equal(g0, g1) and torch.equal(p0, p1)


@pytest.mark.parametrize("size", [96, 32])
def test_optimizer_tail_on_the_aux_stream_is_bitwise_the_serial_tail(routes, size):
    """Eager step with aux-stream weight gradients: AdamW over everything but the leading (full-resolution) encoder blocks and
    the re-pack of those weights run on the aux stream beside the end of the backward, the compute stream updates the leading
    blocks behind the join, and the next forward packs only them (desc.prepacked_from).  Against the default (opt_tail = 0: whole update and
    all packs on the compute stream): metrics of every step, parameters, AdamW moments, step count, BatchNorm buffers bit for
    bit over four steps -- including a step after the parameters were changed THROUGH TORCH between two steps (the arena's
    version counter must invalidate the pre-packed weights: stale packs would show in the very next loss)."""
    x, y = synth(2, size, 31, blocky=True)
    res = []
    for off in (0, 1):
        routes.set("opt_tail", 1 - off)
        m = default_model().to(DEV).train()
        ts = TrainStep(m, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=torch.bfloat16, use_graph=False, aux_wgrad=True)
        ts.load_batch(x.to(DEV), y.to(DEV))
        outs = [ts.step_static().clone() for _ in range(2)]
        if off == 0:
            assert ts._static.get("prepacked") is not None and ts._static["prepacked"][0] >= 1
        with torch.no_grad():
            m.bottleneck.double_conv[0].weight.mul_(1.25)          # a deep-level weight, re-packed by the aux tail route
            m.decoder[3].double_conv[4].weight.add_(0.01)
        outs += [ts.step_static().clone() for _ in range(2)]
        torch.cuda.synchronize()
        res.append((outs, ts.arena.p.clone(), ts.arena.m.clone(), ts.arena.v.clone(), ts.arena.step.clone(), [b.clone() for b in m.buffers()]))
        ts.close()
    routes.reset("opt_tail")
    (o0, p0, m0, v0, s0, b0), (o1, p1, m1, v1, s1, b1) = res
    for i, (a, b) in enumerate(zip(o0, o1)):
        assert torch.equal(a, b), (i, a, b)
    assert float(o0[2][0]) != float(o0[1][0])
    assert torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1) and torch.equal(s0, s1)
    for a, b in zip(b0, b1):
        assert torch.equal(a, b)


def test_wide_batchnorm_consumers_with_dropout_masks(routes):
    """Round 4: at levels 0-1 the BatchNorm apply (+ pool) passes finish the conv epilogue's partial rows themselves, as <= 256
    workgroups of 1024 threads (bn_apply_wide_kernel / bn_apply_pool_wide_kernel), instead of a finalize launch per layer
    (models/unet.py:12-14,16-18).  64^3 N=2 with Dropout3d(p = 0.3): 512 rows at level 0 (persistent convs) and 128 / 432-row
    layers below; against the finalize route the statistics are the same rows summed in double in another order, so the BatchNorm
    buffers agree to fp32 roundoff and everything downstream to bf16 re-rounding noise -- a wrong sample's dropout scale, a wrong
    window or a wrong coefficient would move the loss by percents."""
    x, y = synth(2, 64, 779, blocky=True)
    res = []
    for mode in (3, 0):
        routes.set("wide_bn", mode)
        torch.manual_seed(0)            # also the seed of the model's counter-based dropout stream: the same masks on both routes
        m = UNet3D(in_channels=1, out_channels=4, dropout_rate=0.3).to(DEV).train()
        ts = TrainStep(m, loss="combined", lr=0.0, weight_decay=0.01, compute_dtype=torch.bfloat16, use_graph=False)
        ts.load_batch(x.to(DEV), y.to(DEV))
        out = ts.step_static().clone()
        torch.cuda.synchronize()
        res.append((out.cpu(), ts.arena.g.clone().cpu(), {k: b.clone().cpu() for k, b in m.named_buffers()}))
        ts.close()
    routes.reset("wide_bn")
    (o0, g0, b0), (o1, g1, b1) = res
    assert abs(float(o0[0]) - float(o1[0])) < 2e-3 * abs(float(o1[0])), (o0, o1)
    assert (o0[1:] - o1[1:]).abs().max() < 2e-3, (o0, o1)
    # the first layer's statistics see identical inputs on both routes: equal to fp32 roundoff; deeper layers inherit bf16 re-rounding
    for k in b0:
        if "num_batches_tracked" in k:
            assert torch.equal(b0[k], b1[k]), k
        elif k.startswith("encoder.0.double_conv.1."):
            assert relerr(b0[k], b1[k]) < 1e-6, (k, relerr(b0[k], b1[k]))
        else:
            assert relerr(b0[k], b1[k]) < 5e-3, (k, relerr(b0[k], b1[k]))
    assert relerr(g0, g1) < 5e-2, relerr(g0, g1)


def test_cu_masked_aux_stream_is_bitwise_the_unmasked_one():
    """mi3d_stream_create_masked (hipExtStreamCreateWithCUMask): the aux stream of the deferred weight gradients confined to 16 of
    the 32 CUs of every XCD.  A masked stream is a BLOCKING stream (it serialises with the null stream), so the step runs on a
    pool stream here; where the weight gradients run changes nothing in what they compute."""
    x, y = synth(2, 48, 781, blocky=True)
    res = []
    with torch.cuda.stream(torch.cuda.Stream(device=DEV)):
        for cus in (0, 16):
            torch.manual_seed(0)
            m = UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(DEV).train()
            ts = TrainStep(m, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=torch.bfloat16, use_graph=False, aux_cus=cus)
            assert ts.aux_stream is not None
            if cus:
                assert getattr(ts.aux_stream, "mi3d_cus_per_xcd", 0) == cus and ts.aux_stream.mi3d_concurrent
            ts.load_batch(x.to(DEV), y.to(DEV))
            outs = [ts.step_static().clone() for _ in range(2)]
            torch.cuda.synchronize()
            res.append((outs, ts.arena.g.clone(), ts.arena.p.clone()))
            ts.close()
    (o0, g0, p0), (o1, g1, p1) = res
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    assert torch.equal(g0, g1) and torch.equal(p0, p1)


@pytest.mark.parametrize("size,p_drop", [(96, 0.0), (64, 0.3)])
def test_split_k_ticket_finishes_the_deep_forward_convs(routes, size, p_drop):
    """Round 4: a split-K forward conv of a training step (levels 3-4, models/unet.py:11,15 of encoder.3 / bottleneck / decoder.0)
    finishes itself -- the last of a tile's ks workgroups to take its ticket sums the fp32 partials in k order, stores bf16 y and
    writes the tile's BatchNorm partial row -- instead of a bn_stats_splitk launch.  (1) Whichever workgroup arrives last, the
    sums run in the same order: two independent runs are BITWISE equal (a missed release / acquire or a counter left non-zero
    would show here).  (2) Against the separate finishing pass: the first such layer sees identical inputs and produces the same
    y, so its batch statistics agree to fp32 roundoff (another partition of the same sum); everything downstream to bf16
    re-rounding noise."""
    x, y = synth(2, size, 782, blocky=True)

    def run(mode):
        routes.set("splitk_ticket", mode)
        torch.manual_seed(0)
        m = UNet3D(in_channels=1, out_channels=4, dropout_rate=p_drop).to(DEV).train()
        ts = TrainStep(m, loss="combined", lr=lr, weight_decay=0.01, compute_dtype=torch.bfloat16, use_graph=False)
        ts.load_batch(x.to(DEV), y.to(DEV))
        outs = [ts.step_static().clone().cpu() for _ in range(6)]
        torch.cuda.synchronize()
        r = (outs, ts.arena.g.clone().cpu(), ts.arena.p.clone().cpu(), {k: b.clone().cpu() for k, b in m.named_buffers()})
        ts.close()
        return r

    lr = 1e-3
    a, b = run(1), run(1)
    for u, v in zip(a[0], b[0]):
        assert torch.equal(u, v)
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k
    lr = 0.0                   # the comparison of the two routes: parameters fixed, so that only the kernels differ
    a, c = run(1), run(0)
    routes.reset("splitk_ticket")
    first = "encoder.3.double_conv.1." if size == 96 else None      # first split-K layer at 96^3 (identical inputs on both routes)
    # vs the finishing pass: first step's metrics, then the buffers after three steps
    assert abs(float(a[0][0][0]) - float(c[0][0][0])) < 2e-3 * abs(float(c[0][0][0])), (a[0][0], c[0][0])
    assert (a[0][0][1:] - c[0][0][1:]).abs().max() < 2e-3
    for k in a[3]:
        if "num_batches_tracked" in k:
            assert torch.equal(a[3][k], c[3][k]), k
        elif first and k.startswith(first):
            assert relerr(a[3][k], c[3][k]) < 1e-6, (k, relerr(a[3][k], c[3][k]))
        else:
            assert relerr(a[3][k], c[3][k]) < 5e-3, (k, relerr(a[3][k], c[3][k]))
    assert relerr(a[1], c[1]) < 5e-2, relerr(a[1], c[1])


def test_split_k_ticket_is_stable_over_many_steps():
    """The ticket's hand-off (write-through partial stores -> relaxed agent-scope ticket -> sc1 loads by the last arriver) under
    repetition: with lr = 0 and a fixed batch every step is the same computation, so 150 consecutive steps (900 ticket launches,
    ~50 000 tile episodes with whichever workgroup happens to arrive last) must return bitwise the same metrics and leave bitwise
    the same gradients.  A stale read or a lost arrival shows up as a different loss."""
    torch.manual_seed(0)
    m = UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(DEV).train()
    ts = TrainStep(m, loss="combined", lr=0.0, weight_decay=0.0, compute_dtype=torch.bfloat16, use_graph=False)
    x, y = synth(2, 96, 783, blocky=True)
    ts.load_batch(x.to(DEV), y.to(DEV))
    ref = ts.step_static().clone()
    g0 = ts.arena.g.clone()
    bad = 0
    for i in range(150):
        o = ts.step_static()
        if not torch.equal(o, ref):
            bad += 1
        if i % 50 == 49 and not torch.equal(ts.arena.g, g0):
            bad += 1
    torch.cuda.synchronize()
    ts.close()
    assert bad == 0, bad
