"""Round-3 GPU parity tests (through the C ABI): the bf16 VEC=8 elementwise kernels and the 1x1x1 head that the 96^3 step
actually launches, pinned per operator against the C oracle, and the remaining kernel-route switches at the headline shape.

Technique (as for the conv kernels in round 2): inputs are small DYADIC numbers that bf16 holds exactly, so that
  * integer-like sums (dbeta, the head's dW / db, the head's logits) are exact in fp32 whatever the summation order -> they are
    compared with `atol ~ 0`;
  * values that pass through invstd (not dyadic) are compared with the oracle's double-precision result to fp32 roundoff, and
    stored bf16 tensors to HALF A bf16 SPACING of it (<= 2^-8 relative: the kernels round, to nearest-even, an fp32 value that
    carries ~1e-7 of noise) -- i.e. every stored element is the correctly rounded result up to ties.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import multimodal_segmentation_project_amd as mi  # noqa: F401
from multimodal_segmentation_project_amd import _lib
from multimodal_segmentation_project_amd import metrics as M
from multimodal_segmentation_project_amd._lib import call, ptr
from multimodal_segmentation_project_amd.unet import UNet3D

DEV = "cuda:0"
HALF_ULP = 2.0 ** -8          # bf16: 8 significant bits -> spacing 2^-7 * 2^e, round-to-nearest error <= 2^-8 * 2^e <= 2^-8 |x|
EPS = 1e-5


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def dyadic(rng, shape, lo=-16, hi=16, den=8.0):
    return (rng.integers(lo, hi + 1, shape) / den).astype(np.float32)


def cl_bf16(a):
    """NCDHW float array -> channels-last bf16 device tensor [N][D][H][W][C] (values must be bf16-exact)."""
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 3, 4, 1))).to(DEV).bfloat16()
    assert torch.equal(t.float().cpu(), torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 3, 4, 1))))
    return t


def ncdhw(t):
    return t.float().cpu().numpy().transpose(0, 4, 1, 2, 3)


def close_bf16(got, ref, abs_tol, what):
    """stored bf16 value vs the exact value: half an ULP of the exact value + the fp32 noise of the expression."""
    bound = HALF_ULP * 1.001 * np.abs(ref) + abs_tol
    bad = np.abs(got.astype(np.float64) - ref.astype(np.float64)) > bound
    assert not bad.any(), (what, int(bad.sum()), float(np.abs(got - ref)[bad].max()))


BN_CASES = [
    # N, C, D, H, W, dropout p (0 = none)         which kernels
    (2, 16, 8, 16, 16, 0.0),      # VEC=8, small tensor: <= 128 partial rows finished in the consumers' prologues
    (2, 32, 6, 8, 8, 0.5),        # Dropout3d masks that differ per sample (scale 0 or 2)
    (1, 256, 6, 6, 6, 0.0),       # level-4 shape: 32 channel groups per row
    (2, 16, 32, 48, 48, 0.0),     # M*C > 2^21: the finalize-launch route (bn_stats_finalize / bn_bwd_finalize)
    (2, 64, 24, 24, 24, 0.5),     # level-2 shape of the 96^3 step, with masks
]


@pytest.mark.parametrize("case", BN_CASES)
def test_bn_relu_drop_bf16_vec8_per_op_vs_c_oracle(orc, case):
    """mi3d_bn_relu_drop_forward / _backward in bf16 with C % 8 == 0 = bn_stats / bn_apply / bn_bwd_reduce(_slab) /
    bn_bwd_apply <bf16, 8>, the instantiations that are 27 % of the 96^3 step: stat, running buffers, z, dy, dgamma, dbeta."""
    n, c, d, h, w, p = case
    rng = np.random.default_rng(c * 1000 + d)
    y = dyadic(rng, (n, c, d, h, w))
    # a per-channel offset so that means are not ~0 and (y - mean) cancels for real
    y += (rng.integers(-8, 9, (1, c, 1, 1, 1)) / 4.0).astype(np.float32)
    dz = dyadic(rng, (n, c, d, h, w), -8, 8, 4.0)
    gamma = (rng.random(c) + 0.5).astype(np.float32)
    beta = (rng.standard_normal(c) * 0.3).astype(np.float32)
    rm0, rv0 = (rng.standard_normal(c) * 0.1).astype(np.float32), (rng.random(c) + 0.5).astype(np.float32)
    scale = None
    if p > 0:
        scale = (rng.random((n, c)) >= p).astype(np.float32) / (1.0 - p)
        assert set(np.unique(scale).tolist()) == {0.0, 2.0}
        assert (scale[0] != scale[1]).any()
    m, v = n * d * h * w, d * h * w
    ycl, dzcl = cl_bf16(y), cl_bf16(dz)
    g_d, b_d = torch.from_numpy(gamma).to(DEV), torch.from_numpy(beta).to(DEV)
    rm, rv = torch.from_numpy(rm0).to(DEV), torch.from_numpy(rv0).to(DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    drop = torch.from_numpy(scale).to(DEV) if scale is not None else None
    ws = torch.zeros(_lib.lib().mi3d_bn_workspace_bytes(c), dtype=torch.uint8, device=DEV)
    stat = torch.empty(4 * c, device=DEV)
    z = torch.empty_like(ycl)
    call("mi3d_bn_relu_drop_forward", 1, ptr(ycl), c, c, m, v, ptr(g_d), ptr(b_d), ptr(rm), ptr(rv), ptr(nbt), 0.1, EPS, 1,
         ptr(drop), ptr(z), c, ptr(stat), ptr(ws), None)
    yhat, sm, si, rm_ref, rv_ref = orc.bn_train_fwd(y, gamma, beta, rm0, rv0, 0.1, EPS)
    st = stat.cpu().numpy().reshape(4, c)
    np.testing.assert_allclose(st[0], sm, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(st[1], si, rtol=2e-6)
    np.testing.assert_allclose(st[2], gamma * si, rtol=2e-6)
    np.testing.assert_allclose(rm.cpu().numpy(), rm_ref, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rv.cpu().numpy(), rv_ref, rtol=2e-6)
    assert int(nbt) == 1
    z_ref = orc.relu_drop_fwd(yhat, scale)
    close_bf16(ncdhw(z), z_ref, 2e-6, "z")
    # ---- backward, with the statistics the forward saved
    dy = torch.empty_like(ycl)
    dg, db = torch.full((c,), 7.0, device=DEV), torch.full((c,), -3.0, device=DEV)
    call("mi3d_bn_relu_drop_backward", 1, ptr(dzcl), c, ptr(ycl), c, c, m, v, ptr(stat), ptr(drop), ptr(dy), c, ptr(dg),
         ptr(db), 0, ptr(ws), None)
    # the oracle's chain: dropout/ReLU mask on the BN OUTPUT, then BatchNorm backward with the saved mean / invstd
    yhat_k = (y.astype(np.float64) * st[2].reshape(1, c, 1, 1, 1) + st[3].reshape(1, c, 1, 1, 1)).astype(np.float32)
    dyh = orc.relu_drop_bwd(yhat_k, dz, scale)
    dy_ref, dg_ref, db_ref = orc.bn_train_bwd(y, dyh, gamma, st[0], st[1])
    # dbeta = sum of dyadic numbers: exact in fp32 in any order
    np.testing.assert_allclose(db.cpu().numpy(), db_ref, rtol=0, atol=1e-4 * max(1.0, float(np.abs(db_ref).max()) * 1e-3))
    scale_g = np.sqrt(m) * 4.0
    np.testing.assert_allclose(dg.cpu().numpy(), dg_ref, rtol=2e-5, atol=2e-6 * scale_g)
    close_bf16(ncdhw(dy), dy_ref, 2e-5, "dy")
    # accumulate = 1 adds to the existing parameter gradients
    call("mi3d_bn_relu_drop_backward", 1, ptr(dzcl), c, ptr(ycl), c, c, m, v, ptr(stat), ptr(drop), ptr(dy), c, ptr(dg),
         ptr(db), 1, ptr(ws), None)
    np.testing.assert_allclose(db.cpu().numpy(), 2 * db_ref, rtol=0, atol=2e-4 * max(1.0, float(np.abs(db_ref).max()) * 1e-3))
    np.testing.assert_allclose(dg.cpu().numpy(), 2 * dg_ref, rtol=2e-5, atol=4e-6 * scale_g)


@pytest.mark.parametrize("case", [(2, 16, 8, 16, 32, 0.0), (2, 32, 12, 8, 16, 0.5), (1, 128, 12, 12, 12, 0.0)])
def test_bn_apply_pool_fused_bf16_per_op_vs_c_oracle(orc, case):
    """bn_apply_pool_kernel<bf16, 8> (the second half of every encoder block of the step): z as above AND pooled =
    MaxPool3d(2,2) of the STORED z (exact: a maximum of bf16 values), against the oracle's pool of the kernel's own z and
    against the two-launch route."""
    n, c, d, h, w, p = case
    rng = np.random.default_rng(c + d)
    y = dyadic(rng, (n, c, d, h, w))
    gamma = (rng.random(c) + 0.5).astype(np.float32) * np.where(rng.random(c) < 0.3, -1.0, 1.0).astype(np.float32)   # negative slopes too
    beta = (rng.standard_normal(c) * 0.3).astype(np.float32)
    scale = (rng.random((n, c)) >= p).astype(np.float32) / (1.0 - p) if p > 0 else None
    m, v = n * d * h * w, d * h * w
    ycl = cl_bf16(y)
    g_d, b_d = torch.from_numpy(gamma).to(DEV), torch.from_numpy(beta).to(DEV)
    drop = torch.from_numpy(scale).to(DEV) if scale is not None else None
    ws = torch.zeros(_lib.lib().mi3d_bn_workspace_bytes(c), dtype=torch.uint8, device=DEV)
    outs = []
    for fused in (True, False):
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        nbt = torch.zeros((), dtype=torch.int64, device=DEV)
        stat = torch.empty(4 * c, device=DEV)
        z = torch.empty_like(ycl)
        pooled = torch.empty((n, d // 2, h // 2, w // 2, c), device=DEV, dtype=torch.bfloat16)
        if fused:
            call("mi3d_bn_relu_drop_pool_forward", 1, ptr(ycl), c, c, n, d, h, w, ptr(g_d), ptr(b_d), ptr(rm), ptr(rv), ptr(nbt),
                 0.1, EPS, ptr(drop), ptr(z), c, ptr(pooled), c, ptr(stat), ptr(ws), None)
        else:
            call("mi3d_bn_relu_drop_forward", 1, ptr(ycl), c, c, m, v, ptr(g_d), ptr(b_d), ptr(rm), ptr(rv), ptr(nbt), 0.1, EPS, 1,
                 ptr(drop), ptr(z), c, ptr(stat), ptr(ws), None)
            call("mi3d_maxpool2_forward", 1, ptr(z), c, c, n, d, h, w, ptr(pooled), c, None)
        outs.append((z.clone(), pooled.clone(), stat.clone(), rm.clone(), rv.clone(), int(nbt)))
    for a, b in zip(outs[0][:5], outs[1][:5]):
        assert torch.equal(a, b)                                   # fused == two launches, bit for bit
    assert outs[0][5] == outs[1][5] == 1
    z, pooled = outs[0][0], outs[0][1]
    yhat, _, _, _, _ = orc.bn_train_fwd(y, gamma, beta, np.zeros(c, np.float32), np.ones(c, np.float32), 0.1, EPS)
    close_bf16(ncdhw(z), orc.relu_drop_fwd(yhat, scale), 2e-6, "z")
    np.testing.assert_array_equal(ncdhw(pooled), orc.maxpool2_fwd(ncdhw(z)))


@pytest.mark.parametrize("shape", [(2, 16, 16, 16), (1, 5, 7, 9), (2, 4, 6, 50)])
def test_conv1_head_bf16_per_op_exact(shape):
    """The final 1x1x1 conv (models/unet.py:62,87) through mi3d_conv1_forward / mi3d_conv1_backward in bf16 = conv1_fwd_kernel
    and conv1_bwd_mfma_kernel (+ its slab sum) of the training step.  Dyadic inputs: logits, dW and db are exact in fp32
    (compared with atol 1e-6), dz is the exactly computed sum rounded to bf16 once (bitwise)."""
    n, d, h, w = shape
    cin, cout = 16, 4
    v = d * h * w
    rng = np.random.default_rng(v)
    z = dyadic(rng, (n, cin, d, h, w), -8, 8, 8.0)
    wgt = dyadic(rng, (cout, cin), -4, 4, 4.0)
    bias = dyadic(rng, (cout,), -4, 4, 4.0)
    dl = dyadic(rng, (n, cout, d, h, w), -8, 8, 8.0)
    zcl = cl_bf16(z)
    w_d, b_d = torch.from_numpy(wgt.reshape(cout, cin, 1, 1, 1)).to(DEV), torch.from_numpy(bias).to(DEV)
    logits = torch.empty((n, cout, d, h, w), device=DEV)
    call("mi3d_conv1_forward", 1, ptr(zcl), cin, cin, ptr(w_d), ptr(b_d), ptr(logits), cout, n, v, None)
    ref = np.einsum("oc,ncdhw->nodhw", wgt.astype(np.float64), z.astype(np.float64)) + bias.reshape(1, cout, 1, 1, 1)
    np.testing.assert_allclose(logits.cpu().numpy(), ref, rtol=0, atol=1e-6)
    dl_d = torch.from_numpy(dl).to(DEV)
    dz = torch.empty_like(zcl)
    dW, db = torch.empty((cout, cin, 1, 1, 1), device=DEV), torch.empty(cout, device=DEV)
    wsb = _lib.lib().mi3d_conv1_workspace_bytes(cin, cout)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    call("mi3d_conv1_backward", 1, ptr(zcl), cin, cin, ptr(w_d), ptr(dl_d), cout, ptr(dz), cin, ptr(dW), ptr(db), 0, n, v, ptr(ws),
         wsb, None)
    dz_ref = np.einsum("oc,nodhw->ncdhw", wgt.astype(np.float64), dl.astype(np.float64))
    dz_ref_bf16 = torch.from_numpy(dz_ref.astype(np.float32)).bfloat16().float().numpy()      # one RNE rounding of the exact value
    np.testing.assert_array_equal(ncdhw(dz), dz_ref_bf16)
    dW_ref = np.einsum("nodhw,ncdhw->oc", dl.astype(np.float64), z.astype(np.float64))
    np.testing.assert_allclose(dW.cpu().numpy().reshape(cout, cin), dW_ref, rtol=0, atol=1e-6)
    np.testing.assert_allclose(db.cpu().numpy(), dl.astype(np.float64).sum(axis=(0, 2, 3, 4)), rtol=0, atol=1e-6)
    # accumulate
    call("mi3d_conv1_backward", 1, ptr(zcl), cin, cin, ptr(w_d), ptr(dl_d), cout, ptr(dz), cin, ptr(dW), ptr(db), 1, n, v, ptr(ws),
         wsb, None)
    np.testing.assert_allclose(dW.cpu().numpy().reshape(cout, cin), 2 * dW_ref, rtol=0, atol=2e-6)


def _synth(n, s, seed, blocky=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 1, s, s, s, generator=g)
    y = torch.randint(0, 4, (n, 1, s, s, s), generator=g)
    if blocky:
        zz, yy, xx = torch.meshgrid(torch.arange(s), torch.arange(s), torch.arange(s), indexing="ij")
        lab = ((zz // (s // 4)) + (yy // (s // 4)) + (xx // (s // 4))) % 4
        y = lab[None, None].expand(n, 1, s, s, s).contiguous().long()
        x = y.float() / 3.0 + 0.1 * x
    return x, y


def test_round2_routes_are_invisible_at_the_headline_shape(routes, golden):
    """96^3, N=2, bf16 -- the kernel routes added in round 2, each switched off in turn against the default build, per TENSOR:
      MI3D_NO_SMALL_BN        BatchNorm statistics finished by a finalize launch instead of the consumer's prologue: the SAME
                              partial rows summed in double in another order -> statistics equal to ~1e-7, everything
                              downstream within bf16 re-rounding noise
      MI3D_NO_DEFER_TAIL      split-K input gradients finished by their own pass instead of inside the next BatchNorm-backward
                              reduction: the same fp32 partials in the same order -> bitwise
      MI3D_NO_FUSED_BWD_BIG   level-1 weight gradient + input gradient in two launches instead of one: same bodies, another
      MI3D_NO_FUSED_BWD_P     slab partition (summation order) for the weight gradients only; same for the persistent pair."""
    torch.manual_seed(0)
    m = UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(DEV).train()
    m.compute_dtype = torch.bfloat16
    x, y = _synth(2, 96, 1234)
    x, y = x.to(DEV), y.to(DEV)

    def run():
        for p in m.parameters():
            p.grad = None
        o = m(x)
        l = M.combined_loss(o, y)
        l.backward()
        return l.item(), o.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}

    def noise_only(k):       # conv bias in front of train-mode BN: analytically zero, pure summation noise
        return k.endswith("double_conv.0.bias") or k.endswith("double_conv.4.bias")

    l0, o0, g0 = run()
    routes.set("no_defer_tail", 1)
    l1, o1, g1 = run()
    assert l1 == l0 and torch.equal(o1, o0)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    routes.reset("no_defer_tail")
    # round 3: which launch carries a weight-gradient slab sum changes nothing in the sum
    routes.set("no_upbwd_carry", 1)
    l1, o1, g1 = run()
    assert l1 == l0 and torch.equal(o1, o0)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    routes.reset("no_upbwd_carry")
    # round 4: a split-K gradient of a pooled tensor finished inside the MaxPool3d backward (same partials, same order, same rounding)
    routes.set("no_pool_splitk", 1)
    l1, o1, g1 = run()
    assert l1 == l0 and torch.equal(o1, o0)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    routes.reset("no_pool_splitk")
    # round 4: MaxPool3d backward with two threads per window (same first-maximum routing, same sums)
    routes.set("no_pool_pair", 1)
    l1, o1, g1 = run()
    assert l1 == l0 and torch.equal(o1, o0)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    routes.reset("no_pool_pair")
    for sw in ("no_fused_bwd_big", "no_fused_bwd_p"):
        routes.set(sw, 1)
        l2, o2, g2 = run()
        routes.reset(sw)
        assert l2 == l0 and torch.equal(o2, o0)
        for k in g0:
            if float(g0[k].double().norm()) < 1e-7 or noise_only(k):
                continue
            assert relerr(g2[k].cpu(), g0[k].cpu()) < 1e-5, (sw, k, relerr(g2[k].cpu(), g0[k].cpu()))
    routes.set("no_small_bn", 1)
    l3, o3, g3 = run()
    routes.reset("no_small_bn")
    # statistics differ in the last fp32 bit -> a fraction of the bf16 activations re-round by one spacing (0.4-0.8 %) in every
    # one of the 18 layers: logits move by ~0.5 % (measured 5.2e-3), gradients of the deep-level tensors (cancellation sums)
    # by more.  What this pins: no gross difference between the two routes, tensor by tensor.
    assert abs(l3 - l0) < 1e-3 * abs(l0)
    eo = relerr(o3.cpu(), o0.cpu())
    assert eo < 2e-2, eo
    worst, wk = 0.0, ""
    for k in g0:
        n0 = float(g0[k].double().norm())
        if n0 < 1e-7 or noise_only(k):
            continue
        e = relerr(g3[k].cpu(), g0[k].cpu())
        if e > worst:
            worst, wk = e, k
    print("MI3D_NO_SMALL_BN at 96^3 (blocky labels): logits relerr", eo, "worst per-tensor gradient relerr", worst, wk)
    # The bound, tensor by tensor, is the reference's OWN autocast yardstick (as check_summary uses it): on the batch of the
    # config-2 fixture the two routes may differ by no more than 1.5 x what the reference's bf16-autocast run differs from its
    # fp32 run on that tensor (floor 5 %) -- both routes are then equally good readings of the reference.
    g = golden("config2_96")
    yard = dict(zip(list(g["grad_names"]), g["autocast_bf16/grad_relerr"]))
    from test_gpu_round2 import synth as synth_cfg2
    x, y = (t.to(DEV) for t in synth_cfg2(2, 96, 1234))
    la, oa, ga = run()
    routes.set("no_small_bn", 1)
    lb, ob, gb = run()
    routes.reset("no_small_bn")
    assert abs(lb - la) < 1e-3 * abs(la)
    assert relerr(ob.cpu(), oa.cpu()) < 2e-2
    worst, wk, wtol = 0.0, "", 0.0
    for k in ga:
        if float(ga[k].double().norm()) < 1e-7 or noise_only(k):
            continue
        e = relerr(gb[k].cpu(), ga[k].cpu())
        tol = max(0.05, 1.5 * float(yard.get(k, 0.0)))
        if e / tol > (worst / wtol if wtol else 0.0):
            worst, wk, wtol = e, k, tol
        assert e < tol, (k, e, tol)
    print("MI3D_NO_SMALL_BN on the config-2 batch: worst per-tensor gradient relerr / its yardstick bound", worst, wtol, wk)
    # round 4: the apply pass finishes the conv epilogue's partial rows itself (levels 0-2; wide 1024-thread workgroups at levels
    # 0-1) instead of a finalize launch per layer: the same rows summed in double in another order, same bound as above
    routes.set("wide_bn", 0)
    lc, oc, gc = run()
    routes.reset("wide_bn")
    assert abs(lc - la) < 1e-3 * abs(la)
    assert relerr(oc.cpu(), oa.cpu()) < 2e-2
    worst, wk, wtol = 0.0, "", 0.0
    for k in ga:
        if float(ga[k].double().norm()) < 1e-7 or noise_only(k):
            continue
        e = relerr(gc[k].cpu(), ga[k].cpu())
        tol = max(0.05, 1.5 * float(yard.get(k, 0.0)))
        if e / tol > (worst / wtol if wtol else 0.0):
            worst, wk, wtol = e, k, tol
        assert e < tol, (k, e, tol)
    print("wide_bn=0 on the config-2 batch: logits relerr", relerr(oc.cpu(), oa.cpu()), "worst gradient relerr / bound", worst, wtol, wk)


def test_two_stream_forwards_are_bitwise_the_serial_order():
    """Configs 4 and 5: the two independent forwards of a step run on two streams (DannStep: source || target with the target's
    BatchNorm running-statistics update deferred and applied after the join; TrainStep distillation: student || frozen teacher).
    Everything a step leaves behind -- metrics, parameters after AdamW, gradients, BatchNorm buffers incl. num_batches_tracked
    -- must equal the serial order bit for bit, eagerly and under graph capture."""
    from multimodal_segmentation_project_amd import unet_dann
    from multimodal_segmentation_project_amd.dann import DomainDiscriminator
    from multimodal_segmentation_project_amd.trainer import DannStep, TrainStep
    xs, ys = _synth(2, 32, 11)
    xt, _ = _synth(2, 32, 12, blocky=False)

    def dann(overlap, graph):
        torch.manual_seed(0)
        seg = unet_dann.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(DEV).train()
        torch.manual_seed(3)
        disc = DomainDiscriminator(256).to(DEV).train()
        disc._mi3d_injected_drop_scales = [torch.ones(4, 256, device=DEV), torch.ones(4, 128, device=DEV)]      # no discriminator dropout
        ts = DannStep(seg, disc, loss="combined", lambda_domain=0.2, compute_dtype=torch.bfloat16, use_graph=graph,
                      overlap_forwards=overlap)
        mets = [ts.step(xs.to(DEV), ys.to(DEV), xt.to(DEV)).clone() for _ in range(3)]
        torch.cuda.synchronize()
        return mets, ts.arena.p.clone(), ts.arena.g.clone(), ts.disc_arena.p.clone(), [b.clone() for b in seg.buffers()]

    ref = dann(False, False)
    for overlap, graph in ((True, False), (True, True)):
        got = dann(overlap, graph)
        for a, b in zip(ref[0], got[0]):
            assert torch.equal(a, b), (overlap, graph)
        for a, b in zip(ref[1:4], got[1:4]):
            assert torch.equal(a, b), (overlap, graph)
        for a, b in zip(ref[4], got[4]):
            assert torch.equal(a, b), (overlap, graph)
    assert int(ref[4][2]) == 6                    # two forwards per step x three steps

    def distill(overlap, graph):
        torch.manual_seed(0)
        student = UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(DEV).train()
        torch.manual_seed(1)
        teacher = UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(DEV).eval()
        ts = TrainStep(student, kd_teacher=teacher, kd_alpha=0.7, kd_temperature=2.0, compute_dtype=torch.bfloat16, use_graph=graph,
                       overlap_teacher=overlap)
        mets = [ts.step(xs.to(DEV), ys.to(DEV)).clone() for _ in range(3)]
        torch.cuda.synchronize()
        return mets, ts.arena.p.clone(), ts.arena.g.clone()

    ref = distill(False, False)
    for overlap, graph in ((True, False), (True, True)):
        got = distill(overlap, graph)
        for a, b in zip(ref[0], got[0]):
            assert torch.equal(a, b), (overlap, graph)
        assert torch.equal(ref[1], got[1]) and torch.equal(ref[2], got[2]), (overlap, graph)


def test_eight_wave_conv_is_bit_identical_to_the_four_wave_kernel(routes):
    """conv3_mfma8_kernel (levels 1-4 forward / stand-alone input gradient: 8 waves, weights through LDS) computes every output
    element with the same K order and fp32 accumulation order as conv3_mfma_kernel -> identical bf16 outputs, ragged borders,
    both tile shapes, split-K included (MI3D_CONV8=0 selects the four-wave kernels)."""
    for (n, cin, cout, d, h, w) in [(2, 32, 32, 8, 16, 32), (1, 64, 32, 6, 17, 35), (2, 64, 64, 12, 12, 12), (1, 128, 256, 6, 6, 6),
                                    (1, 16, 32, 9, 20, 40)]:
        g = torch.Generator(device=DEV).manual_seed(cin + w)
        x = torch.randn(n, d, h, w, cin, device=DEV, generator=g).bfloat16()
        wgt = torch.randn(cout, cin, 3, 3, 3, device=DEV, generator=g) * 0.1
        b = torch.randn(cout, device=DEV, generator=g)
        wsb = _lib.lib().mi3d_conv3_workspace_bytes(cin, cout, n, d, h, w)
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        outs = []
        for v in ("0", "1"):
            routes.set("conv8", int(v))
            y = torch.empty(n, d, h, w, cout, device=DEV, dtype=torch.bfloat16)
            call("mi3d_conv3_forward", 1, 1, ptr(x), cin, cin, ptr(wgt), ptr(b), ptr(y), cout, cout, n, d, h, w, ptr(ws), wsb, None)
            outs.append(y.clone())
        routes.reset("conv8")
        assert torch.equal(outs[0], outs[1]), (n, cin, cout, d, h, w)


# ------------------------------------------------------------------------------------------------ head + loss in one pass
@pytest.mark.parametrize("case", [(2, 4, 16, 16, 16, "combined"), (1, 4, 5, 7, 9, "ce_tversky"), (2, 3, 4, 6, 50, "combined"),
                                  (3, 2, 3, 5, 37, "dice"), (2, 4, 8, 12, 20, "kd"), (1, 3, 5, 6, 11, "kd")])
def test_head_loss_fused_passes_against_the_unfused_operators(case):
    """mi3d_head_loss_forward / _backward (the training step's 1x1x1 head folded into the loss, models/unet.py:62,87 +
    utils/metrics.py:14-40) against the operator chain they replace: mi3d_conv1_forward -> mi3d_seg_loss_metrics_forward and
    mi3d_seg_loss_backward -> mi3d_conv1_backward.  The kept logits are the unfused head's bit for bit; the metrics come from
    integer counts (exact); the loss sums the same per-voxel terms in another order (a few ulp); given the SAME coefficients the
    fused backward is bitwise the unfused pair.  Ragged voxel counts, C < 4 and the distillation term (teacher logits from
    memory, distill_unet.py:107-115) included."""
    from multimodal_segmentation_project_amd.trainer import _loss_cfg
    n, c, d, h, w, loss = case
    cin, v = 16, d * h * w
    cfg = _loss_cfg("combined", 0.7, 2.0) if loss == "kd" else _loss_cfg(loss)
    assert _lib.lib().mi3d_head_loss_supported(1, cin, c, C.byref(cfg)) == 1
    g = torch.Generator().manual_seed(v + c)
    z = (torch.randn(n, v, cin, generator=g) * 1.5).bfloat16().to(DEV)
    wgt = (torch.randn(c, cin, generator=g) * 0.4).to(DEV)
    bias = (torch.randn(c, generator=g) * 0.2).to(DEV)
    lab = torch.randint(0, c, (n, v), generator=g).to(DEV)
    teach = (torch.randn(n, c, v, generator=g) * 1.2).to(DEV) if loss == "kd" else None
    lib = _lib.lib()
    lws = torch.empty(lib.mi3d_seg_loss_workspace_bytes(c), dtype=torch.uint8, device=DEV)
    mws = torch.empty(lib.mi3d_seg_metrics_workspace_bytes(c), dtype=torch.uint8, device=DEV)
    wsb = lib.mi3d_conv1_workspace_bytes(cin, c)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    scale = torch.tensor([0.5], device=DEV)
    # unfused
    logits = torch.empty((n, c, v), device=DEV)
    call("mi3d_conv1_forward", 1, ptr(z), cin, cin, ptr(wgt), ptr(bias), ptr(logits), c, n, v, None)
    met0, coef0 = torch.zeros(4, device=DEV), torch.zeros(_lib.LOSS_COEF_FLOATS, device=DEV)
    call("mi3d_seg_loss_metrics_forward", ptr(logits), ptr(lab), ptr(teach), n, c, d, v, C.byref(cfg), ptr(met0), ptr(coef0), ptr(met0[1:]),
         ptr(lws), ptr(mws), None)
    dl = torch.empty_like(logits)
    call("mi3d_seg_loss_backward", ptr(logits), ptr(lab), ptr(teach), n, c, v, C.byref(cfg), ptr(coef0), ptr(scale), ptr(dl), None)
    dz0 = torch.empty_like(z)
    dW0, db0 = torch.empty((c, cin), device=DEV), torch.empty(c, device=DEV)
    call("mi3d_conv1_backward", 1, ptr(z), cin, cin, ptr(wgt), ptr(dl), c, ptr(dz0), cin, ptr(dW0), ptr(db0), 0, n, v, ptr(ws), wsb, None)
    # fused
    met1, coef1 = torch.zeros(4, device=DEV), torch.zeros(_lib.LOSS_COEF_FLOATS, device=DEV)
    kept = torch.empty_like(logits)
    call("mi3d_head_loss_forward", ptr(z), cin, cin, ptr(wgt), ptr(bias), ptr(lab), ptr(teach), n, c, d, v, C.byref(cfg), ptr(met1),
         ptr(coef1), ptr(met1[1:]), ptr(lws), ptr(mws), ptr(kept), None)
    assert torch.equal(kept, logits)
    assert torch.equal(met1[1:], met0[1:])                                  # iou / dice / acc: from exact counts
    assert abs(float(met1[0]) - float(met0[0])) <= 2e-6 * abs(float(met0[0]))
    np.testing.assert_allclose(coef1.cpu().numpy(), coef0.cpu().numpy(), rtol=2e-6, atol=1e-12)
    dz1 = torch.empty_like(z)
    dW1, db1 = torch.empty((c, cin), device=DEV), torch.empty(c, device=DEV)
    call("mi3d_head_loss_backward", ptr(z), cin, cin, ptr(wgt), ptr(bias), ptr(lab), ptr(teach), n, c, v, C.byref(cfg), ptr(coef0),
         ptr(scale), ptr(dz1), cin, ptr(dW1), ptr(db1), 0, ptr(ws), wsb, None)
    assert torch.equal(dz1.view(torch.int16), dz0.view(torch.int16))
    assert torch.equal(dW1, dW0) and torch.equal(db1, db0)
    # without the optional logits copy the results do not change
    met2, coef2 = torch.zeros(4, device=DEV), torch.zeros(_lib.LOSS_COEF_FLOATS, device=DEV)
    call("mi3d_head_loss_forward", ptr(z), cin, cin, ptr(wgt), ptr(bias), ptr(lab), ptr(teach), n, c, d, v, C.byref(cfg), ptr(met2),
         ptr(coef2), ptr(met2[1:]), ptr(lws), ptr(mws), None, None)
    assert torch.equal(met2, met1) and torch.equal(coef2, coef1)


def test_train_step_with_fused_head_equals_the_unfused_step(routes):
    """TrainStep at 96^3 N=2 bf16 with the head folded into the loss (default) against MI3D_NO_HEAD_LOSS=1 (logits and dlogits
    through memory): metrics identical, loss to a few ulp, every updated parameter within the noise of one ulp of `coef`."""
    from multimodal_segmentation_project_amd.trainer import TrainStep
    x, y = _synth(2, 96, 77)
    res = []
    for fused in (True, False):
        if not fused:
            routes.set("no_head_loss", 1)
        torch.manual_seed(3)
        m = UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(DEV).train()
        ts = TrainStep(m, lr=1e-3, weight_decay=0.0, compute_dtype=torch.bfloat16)
        assert ts._prepare(x.to(DEV))["fused_head"] == fused
        out = ts.step(x.to(DEV), y.to(DEV)).cpu()
        grads = {k: p.grad.detach().clone().cpu() for k, p in m.named_parameters()}
        res.append((out, grads))
        ts.close()
    (o1, g1), (o0, g0) = res
    assert torch.equal(o1[1:], o0[1:])
    assert abs(float(o1[0]) - float(o0[0])) <= 2e-6 * abs(float(o0[0]))
    worst = 0.0
    for k in g0:
        if float(g0[k].double().norm()) < 1e-7 or k.endswith("double_conv.0.bias") or k.endswith("double_conv.4.bias"):
            continue
        worst = max(worst, relerr(g1[k], g0[k]))
    print("fused head vs unfused step at 96^3: worst per-tensor gradient relerr", worst)
    assert worst < 1e-3


def test_fused_head_is_refused_where_it_has_no_kernels():
    """fp32 activations, > 4 classes or features[0] != 16 have no fused head + loss kernels: mi3d_unet_head_loss_supported says
    so, the fused entry points fail loudly (no silent fallback inside the library), and TrainStep takes the four-call sequence."""
    from multimodal_segmentation_project_amd.trainer import TrainStep, _loss_cfg
    from multimodal_segmentation_project_amd._lib import Mi3dError
    lib = _lib.lib()
    cfg = _loss_cfg("combined")
    assert lib.mi3d_head_loss_supported(1, 16, 4, C.byref(cfg)) == 1
    assert lib.mi3d_head_loss_supported(0, 16, 4, C.byref(cfg)) == 0          # fp32
    assert lib.mi3d_head_loss_supported(1, 16, 5, C.byref(cfg)) == 0          # 5 classes
    assert lib.mi3d_head_loss_supported(1, 32, 4, C.byref(cfg)) == 0          # backward needs features[0] == 16
    kd = _loss_cfg("combined", 0.7, 2.0)
    assert lib.mi3d_head_loss_supported(1, 16, 4, C.byref(kd)) == 1           # distillation term: supported, needs teacher logits
    z = torch.zeros((1, 64, 16), dtype=torch.bfloat16, device=DEV)
    w, b = torch.zeros((5, 16), device=DEV), torch.zeros(5, device=DEV)
    lab = torch.zeros((1, 64), dtype=torch.int64, device=DEV)
    out, coef = torch.zeros(4, device=DEV), torch.zeros(_lib.LOSS_COEF_FLOATS, device=DEV)
    lws = torch.empty(lib.mi3d_seg_loss_workspace_bytes(5), dtype=torch.uint8, device=DEV)
    with pytest.raises(Mi3dError):
        call("mi3d_head_loss_forward", ptr(z), 16, 16, ptr(w), ptr(b), ptr(lab), None, 1, 5, 4, 64, C.byref(cfg), ptr(out), ptr(coef),
             None, ptr(lws), None, None, None)
    with pytest.raises(Mi3dError):       # distillation weight without teacher logits
        call("mi3d_head_loss_forward", ptr(z), 16, 16, ptr(w[:4]), ptr(b[:4]), ptr(lab), None, 1, 4, 4, 64, C.byref(kd), ptr(out),
             ptr(coef), None, ptr(lws), None, None, None)
    torch.manual_seed(0)
    m = UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(DEV).train()
    ts = TrainStep(m, lr=1e-3, compute_dtype=torch.float32)
    x, y = _synth(1, 16, 5)
    assert ts._prepare(x.to(DEV))["fused_head"] is False
    o = ts.step(x.to(DEV), y.to(DEV))
    assert torch.isfinite(o).all()
    ts.close()


def test_distillation_and_dann_steps_with_fused_head_equal_the_unfused_steps(routes):
    """TrainStep with a teacher (64^3, the student's body -> join with the teacher's stream -> head + loss with the distillation
    term in one pass) and DannStep (48^3, source head folded into the loss, target head not run at all) against
    MI3D_NO_HEAD_LOSS=1: metrics identical, loss to a few ulp, gradients within the noise of one ulp of `coef`."""
    from multimodal_segmentation_project_amd.trainer import TrainStep, DannStep
    from multimodal_segmentation_project_amd import unet_dann, dann
    x, y = _synth(2, 64, 5)
    res = []
    for fused in (True, False):
        if not fused:
            routes.set("no_head_loss", 1)
        torch.manual_seed(3)
        student = UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(DEV).train()
        torch.manual_seed(4)
        teacher = UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(DEV).eval()
        ts = TrainStep(student, lr=1e-3, weight_decay=0.0, kd_teacher=teacher, compute_dtype=torch.bfloat16)
        assert ts._prepare(x.to(DEV))["fused_head"] == fused
        out = ts.step(x.to(DEV), y.to(DEV)).cpu()
        res.append((out, {k: p.grad.detach().clone().cpu() for k, p in student.named_parameters()}))
        ts.close()
        routes.reset("no_head_loss")
    (o1, g1), (o0, g0) = res
    assert torch.equal(o1[1:], o0[1:]) and abs(float(o1[0]) - float(o0[0])) <= 2e-6 * abs(float(o0[0]))
    worst = max(relerr(g1[k], g0[k]) for k in g0 if float(g0[k].double().norm()) > 1e-7 and
                not (k.endswith("double_conv.0.bias") or k.endswith("double_conv.4.bias")))
    print("distillation, fused head vs unfused: worst per-tensor gradient relerr", worst)
    assert worst < 1e-3
    xs, ys = _synth(2, 48, 6)
    xt, _ = _synth(2, 48, 7)
    res = []
    for fused in (True, False):
        if not fused:
            routes.set("no_head_loss", 1)
        torch.manual_seed(3)
        seg = unet_dann.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(DEV).train()
        torch.manual_seed(5)
        disc = dann.DomainDiscriminator(256).to(DEV).train()
        for mod in disc.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        ds = DannStep(seg, disc, lr=1e-3, weight_decay=0.0, compute_dtype=torch.bfloat16)
        out = ds.step(xs.to(DEV), ys.to(DEV), xt.to(DEV)).cpu()
        res.append((out, {k: p.grad.detach().clone().cpu() for k, p in seg.named_parameters()}))
        ds.close()
        routes.reset("no_head_loss")
    (o1, g1), (o0, g0) = res
    assert torch.equal(o1[1:4], o0[1:4]) and abs(float(o1[0]) - float(o0[0])) <= 2e-6 * abs(float(o0[0]))
    worst = max(relerr(g1[k], g0[k]) for k in g0 if float(g0[k].double().norm()) > 1e-7 and
                not (k.endswith("double_conv.0.bias") or k.endswith("double_conv.4.bias")))
    print("DANN, fused head vs unfused: worst per-tensor gradient relerr", worst)
    assert worst < 1e-3
