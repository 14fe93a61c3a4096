"""Pin the CPU oracle (oracle/mi3d_oracle.c and oracle/torch_ref.py) to the golden vectors that
tools/gen_golden.py produced by executing the reference.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import torch_ref

LOSS_CASES = ["uniform", "absent2", "single0", "onehot", "c3_noncubic"]
# golden function name -> oracle loss kind
LOSS_FNS = {"combined": "combined", "default_fn": "combined", "tversky55": "tversky", "tversky_fn": "tversky",
            "ce_tversky73": "ce_tversky_default", "ce_tversky55": "ce_tversky", "dice": "dice"}


def _sd_from(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def test_c_oracle_doubleconv_chain(golden, orc):
    """conv3 -> BN(train) -> ReLU, twice, forward + full backward vs the reference's DoubleConv."""
    g = golden("doubleconv")
    sd = {k[4:]: v for k, v in g.items() if k.startswith("sd0/")}
    x = g["x"]
    pre = "double_conv."
    y1 = orc.conv3d_fwd(x, sd[pre + "0.weight"], sd[pre + "0.bias"])
    b1, m1, i1, rm1, rv1 = orc.bn_train_fwd(y1, sd[pre + "1.weight"], sd[pre + "1.bias"],
                                            sd[pre + "1.running_mean"], sd[pre + "1.running_var"])
    z1 = orc.relu_drop_fwd(b1)
    y2 = orc.conv3d_fwd(z1, sd[pre + "4.weight"], sd[pre + "4.bias"])
    b2, m2, i2, rm2, rv2 = orc.bn_train_fwd(y2, sd[pre + "5.weight"], sd[pre + "5.bias"],
                                            sd[pre + "5.running_mean"], sd[pre + "5.running_var"])
    z2 = orc.relu_drop_fwd(b2)
    np.testing.assert_allclose(z2, g["out"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(rm1, g["sd1/" + pre + "1.running_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv2, g["sd1/" + pre + "5.running_var"], rtol=1e-5, atol=1e-6)
    # backward
    gb2 = orc.relu_drop_bwd(b2, g["go"])
    gy2, gg2, gbt2 = orc.bn_train_bwd(y2, gb2, sd[pre + "5.weight"], m2, i2)
    gz1, gw2, gcb2 = orc.conv3d_bwd(z1, sd[pre + "4.weight"], gy2)
    gb1 = orc.relu_drop_bwd(b1, gz1)
    gy1, gg1, gbt1 = orc.bn_train_bwd(y1, gb1, sd[pre + "1.weight"], m1, i1)
    gx, gw1, gcb1 = orc.conv3d_bwd(x, sd[pre + "0.weight"], gy1)
    np.testing.assert_allclose(gx, g["gx"], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(gw1, g["grad/" + pre + "0.weight"], rtol=2e-3, atol=5e-5)
    np.testing.assert_allclose(gw2, g["grad/" + pre + "4.weight"], rtol=2e-3, atol=5e-5)
    np.testing.assert_allclose(gg1, g["grad/" + pre + "1.weight"], rtol=2e-3, atol=5e-5)
    np.testing.assert_allclose(gbt2, g["grad/" + pre + "5.bias"], rtol=2e-3, atol=5e-5)
    # conv bias grads are analytically zero in front of a train-mode BN: only roundoff remains
    assert np.abs(gcb1).max() < 1e-4 and np.abs(g["grad/" + pre + "0.bias"]).max() < 1e-4


@pytest.mark.parametrize("case", LOSS_CASES)
def test_c_oracle_losses(golden, orc, case):
    g = golden("losses_metrics")
    lg, lb = g[f"{case}/logits"], g[f"{case}/labels"]
    for fname, kind in LOSS_FNS.items():
        loss, grad = orc.seg_loss(lg, lb, kind)
        np.testing.assert_allclose(loss, g[f"{case}/{fname}/loss"], rtol=2e-5, atol=2e-6, err_msg=fname)
        np.testing.assert_allclose(grad, g[f"{case}/{fname}/grad"], rtol=2e-3, atol=2e-7, err_msg=fname)
    loss, grad = orc.seg_loss(lg, lb, "combined")
    np.testing.assert_allclose(loss, g[f"{case}/combined64/loss"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(grad, g[f"{case}/combined64/grad"], rtol=1e-4, atol=1e-9)
    for alpha, temp in ((0.7, 2.0), (0.3, 4.0)):
        loss, grad = orc.seg_loss(lg, lb, teacher=g[f"{case}/teacher"], kd_alpha=alpha, temperature=temp)
        np.testing.assert_allclose(loss, g[f"{case}/distill_a{alpha}_t{temp}/loss"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(grad, g[f"{case}/distill_a{alpha}_t{temp}/grad"], rtol=2e-3, atol=2e-7)


@pytest.mark.parametrize("case", LOSS_CASES + ["q1_d2", "nofg"])
def test_c_oracle_metrics(golden, orc, case):
    g = golden("losses_metrics")
    m = orc.seg_metrics(g[f"{case}/logits"], g[f"{case}/labels"])
    np.testing.assert_allclose(m["iou"], g[f"{case}/calculate_iou"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(m["dice"], g[f"{case}/calculate_dice"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(m["acc"], g[f"{case}/calculate_accuracy"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("case", LOSS_CASES)
def test_torch_ref_losses(golden, case):
    g = golden("losses_metrics")
    lg, lb = torch.from_numpy(g[f"{case}/logits"]), torch.from_numpy(g[f"{case}/labels"])
    for fname, kind in LOSS_FNS.items():
        z = lg.clone().requires_grad_(True)
        l = torch_ref.seg_loss(z, lb, kind)
        l.backward()
        np.testing.assert_allclose(l.item(), g[f"{case}/{fname}/loss"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(z.grad.numpy(), g[f"{case}/{fname}/grad"], rtol=2e-3, atol=2e-7)
    z = lg.clone().requires_grad_(True)
    l = torch_ref.seg_loss(z, lb, teacher=torch.from_numpy(g[f"{case}/teacher"]), kd_alpha=0.7, temperature=2.0)
    l.backward()
    np.testing.assert_allclose(l.item(), g[f"{case}/distill_a0.7_t2.0/loss"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(z.grad.numpy(), g[f"{case}/distill_a0.7_t2.0/grad"], rtol=2e-3, atol=2e-7)


def test_torch_ref_small_unet(golden):
    """Whole small net (non-cubic, odd channels): logits, loss, every grad, BN buffers, eval logits."""
    g = golden("small_unet")
    sd = _sd_from(g, "sd0/")
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    logits, _, upd = torch_ref.unet3d_forward(sd, x, train=True)
    loss = torch_ref.seg_loss(logits, y, "combined")
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-6)
    for k, v in sd.items():
        if v.requires_grad:
            np.testing.assert_allclose(v.grad.numpy(), g["grad/" + k], rtol=2e-3, atol=1e-5, err_msg=k)
    for k, v in upd.items():
        np.testing.assert_allclose(v.numpy(), g["sd1/" + k], rtol=1e-5, atol=1e-6, err_msg=k)
    sd_eval = {k: v.detach() for k, v in sd.items()}
    sd_eval.update(upd)
    with torch.no_grad():
        le, _, _ = torch_ref.unet3d_forward(sd_eval, x, train=False)
    np.testing.assert_allclose(le.numpy(), g["logits_eval"], rtol=1e-4, atol=1e-5)


def test_c_oracle_small_unet_forward(golden, orc):
    """The C oracle chained by hand through the small net (pool, upconv, cat, final 1x1x1) — forward."""
    g = golden("small_unet")
    sd = {k[4:]: v for k, v in g.items() if k.startswith("sd0/")}

    def dc(pre, x):
        for ci, bi in ((0, 1), (4, 5)):
            p = f"{pre}.double_conv."
            y = orc.conv3d_fwd(x, sd[p + f"{ci}.weight"], sd[p + f"{ci}.bias"])
            y, *_ = orc.bn_train_fwd(y, sd[p + f"{bi}.weight"], sd[p + f"{bi}.bias"],
                                     sd[p + f"{bi}.running_mean"], sd[p + f"{bi}.running_var"])
            x = orc.relu_drop_fwd(y)
        return x

    x = g["x"]
    s0 = dc("encoder.0", x)
    s1 = dc("encoder.1", orc.maxpool2_fwd(s0))
    b = dc("bottleneck", orc.maxpool2_fwd(s1))
    u = orc.convT2_fwd(b, sd["upconvs.0.weight"], sd["upconvs.0.bias"])
    d0 = dc("decoder.0", np.concatenate([s1, u], axis=1))
    u = orc.convT2_fwd(d0, sd["upconvs.1.weight"], sd["upconvs.1.bias"])
    d1 = dc("decoder.1", np.concatenate([s0, u], axis=1))
    logits = orc.conv3d_fwd(d1, sd["final_conv.weight"], sd["final_conv.bias"])
    np.testing.assert_allclose(logits, g["logits"], rtol=2e-4, atol=2e-5)
    loss, _ = orc.seg_loss(logits, g["y"], "combined", want_grad=False)
    np.testing.assert_allclose(loss, g["loss"], rtol=1e-5)


def test_c_oracle_pool_convT_backward(orc):
    """maxpool (with ties) and convT backward vs torch autograd (same op the reference calls)."""
    gen = torch.Generator().manual_seed(5)
    x = torch.randint(0, 3, (2, 3, 4, 6, 4), generator=gen).float().requires_grad_(True)   # many ties
    y = torch.nn.functional.max_pool3d(x, 2, 2)
    go = torch.randn(y.shape, generator=gen)
    y.backward(go)
    np.testing.assert_array_equal(orc.maxpool2_fwd(x.detach().numpy()), y.detach().numpy())
    np.testing.assert_allclose(orc.maxpool2_bwd(x.detach().numpy(), go.numpy()), x.grad.numpy(), atol=1e-6)
    x = torch.randn(2, 6, 3, 2, 4, generator=gen, requires_grad=True)
    w = torch.randn(6, 4, 2, 2, 2, generator=gen, requires_grad=True)
    b = torch.randn(4, generator=gen, requires_grad=True)
    y = torch.nn.functional.conv_transpose3d(x, w, b, stride=2)
    go = torch.randn(y.shape, generator=gen)
    y.backward(go)
    np.testing.assert_allclose(orc.convT2_fwd(x.detach().numpy(), w.detach().numpy(), b.detach().numpy()),
                               y.detach().numpy(), rtol=1e-4, atol=1e-5)
    gx, gw, gb = orc.convT2_bwd(x.detach().numpy(), w.detach().numpy(), go.numpy())
    np.testing.assert_allclose(gx, x.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gw, w.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gb, b.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_torch_ref_default_unet_and_dann(golden):
    """Default 5.65 M-parameter net: seeded init digest + 16^3 logits/loss/grad norms; one DANN step."""
    from multimodal_segmentation_project_amd.unet import UNet3D  # host-side mirror (construction is CPU torch)
    g = golden("default_unet")
    torch.manual_seed(0)
    m = UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    sd = m.state_dict()
    keys = sorted(sd.keys())
    assert keys == list(g["param_keys"])
    dig = np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys])
    np.testing.assert_allclose(dig, g["param_digest"], rtol=1e-12, atol=0)
    assert [str(tuple(sd[k].shape)) for k in keys] == list(g["param_shapes"])
    assert [str(sd[k].dtype) for k in keys] == list(g["param_dtypes"])
    sdf = {k: v.detach().clone() for k, v in sd.items()}
    for k, v in sdf.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn(2, 1, 16, 16, 16, generator=gen)
    y = torch.randint(0, 4, (2, 1, 16, 16, 16), generator=gen)
    logits, _, upd = torch_ref.unet3d_forward(sdf, x, train=True)
    loss = torch_ref.seg_loss(logits, y, "combined")
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), g["s16n2/logits"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(loss.item(), g["s16n2/loss"], rtol=1e-5)
    names = list(g["grad_names"])
    norms = np.array([float(sdf[k].grad.double().norm()) for k in names])
    # conv biases in front of BN have roundoff-only gradients: compare the others relatively
    big = g["s16n2/grad_norms"] > 1e-6
    np.testing.assert_allclose(norms[big], g["s16n2/grad_norms"][big], rtol=5e-3)


# ------------------------------------------------------------------------------------------------ round 2
def test_oracle_dropout_masks_match_reference(golden):
    """oracle/torch_ref.unet3d_forward(drop_scales=...) replays the Dropout3d masks the REFERENCE drew (recorded by
    forward hooks in tools/gen_golden.py::gen_dropout, p = 0.5, N = 3): logits, loss, every gradient, BN buffers."""
    import torch
    from oracle import torch_ref
    g = golden("dropout")
    sd = {k[len("small/sd0/"):]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("small/sd0/")}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    blocks = ["encoder.0", "encoder.1", "bottleneck", "decoder.0", "decoder.1"]
    drop = {b: tuple(torch.from_numpy(g[f"small/mask/{b}.double_conv.{i}"]) for i in (3, 7)) for b in blocks}
    for b in blocks:
        for sc in drop[b]:
            assert set(np.unique(sc.numpy()).tolist()) <= {0.0, 2.0}
    logits, _, upd = torch_ref.unet3d_forward(sd, torch.from_numpy(g["small/x"]), train=True, drop_scales=drop)
    loss = torch_ref.seg_loss(logits, torch.from_numpy(g["small/y"]), "combined")
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), g["small/logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(float(loss), g["small/loss"], rtol=1e-6)
    for k, v in sd.items():
        if v.requires_grad:
            np.testing.assert_allclose(v.grad.numpy(), g["small/grad/" + k], rtol=1e-3, atol=2e-6, err_msg=k)
    for k, v in upd.items():
        if "running" in k:
            np.testing.assert_allclose(v.numpy(), g["small/sd1/" + k], rtol=1e-5, atol=1e-6, err_msg=k)
    flat = torch_ref.plan_drop_scales(drop, 2)
    assert flat.numel() == 3 * 2 * (4 + 8 + 16 + 8 + 4)


def test_oracle_loop_reproduces_reference_accumulation_quirk(golden):
    """oracle/torch_ref.train_loop(zero_grad_quirk=True) == the reference's train_one_epoch executed under
    Accelerator(gradient_accumulation_steps=2) (fixture loops.npz accum/*): only the boundary micro-batch's gradient is
    applied; zero_grad_quirk=False (proper accumulation) gives different parameters."""
    import torch
    import multimodal_segmentation_project_amd as mi
    from oracle import torch_ref
    g = golden("loops")

    def synth(n, s, seed, blocky):
        gen = torch.Generator().manual_seed(seed)
        x = torch.randn(n, 1, s, s, s, generator=gen)
        y = torch.randint(0, 4, (n, 1, s, s, s), generator=gen)
        if blocky:
            zz, yy, xx = torch.meshgrid(torch.arange(s), torch.arange(s), torch.arange(s), indexing="ij")
            lab = ((zz // (s // 4)) + (yy // (s // 4)) + (xx // (s // 4))) % 4
            y = lab[None, None].expand(n, 1, s, s, s).contiguous().long()
            x = y.float() / 3.0 + 0.1 * x
        return x, y

    batches = [synth(2, 32, 500 + i, i % 2 == 0) for i in range(4)]
    torch.manual_seed(0)
    sd0 = {k: v.detach().clone() for k, v in mi.UNet3D(1, 4, dropout_rate=0.0).state_dict().items()}
    sd_q, losses, grads = torch_ref.train_loop(sd0, batches, accum=2, zero_grad_quirk=True)
    np.testing.assert_allclose(np.mean(losses), g["accum/result"][0], rtol=1e-5)
    keys = sorted(sd_q.keys())
    dig = np.array([[float(sd_q[k].double().sum()), float(sd_q[k].double().abs().sum())] for k in keys])
    sel = np.array([k.endswith(".weight") for k in keys])
    np.testing.assert_allclose(dig[sel, 1], g["accum/param_digest_after"][sel, 1], rtol=1e-5)
    for k, rn in zip(list(g["accum/grad_names"]), g["accum/grad_norms"]):
        if rn > 1e-6:
            assert abs(float(grads[k].double().norm()) - rn) / rn < 5e-3, k    # after one AdamW step: fp32 order noise
    sd_p, _, _ = torch_ref.train_loop(sd0, batches, accum=2, zero_grad_quirk=False)
    k = "decoder.3.double_conv.0.weight"
    assert float((sd_p[k] - sd_q[k]).abs().max()) > 1e-4


def test_oracle_odd_sizes_match_reference(golden):
    """Sides not divisible by 2^levels (models/unet.py:81-83 nearest-resize before the concat, MaxPool3d floor): the
    oracle's functional forward/backward against the reference-run fixture (6x10x7 volume)."""
    import torch
    from oracle import torch_ref
    g = golden("oddsize")
    sd = {k[4:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("sd0/")}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    logits, _, upd = torch_ref.unet3d_forward(sd, torch.from_numpy(g["x"]), train=True)
    loss = torch_ref.seg_loss(logits, torch.from_numpy(g["y"]), "combined")
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(float(loss), g["loss"], rtol=1e-6)
    for k, v in sd.items():
        if v.requires_grad:
            np.testing.assert_allclose(v.grad.numpy(), g["grad/" + k], rtol=1e-3, atol=2e-6, err_msg=k)


def test_oracle_preprocessing_matches_reference(golden):
    """oracle/preproc_ref.py (CT window, MRI z-score / percentile clip / min-max, AMOS / CHAOS label remaps) against the
    outputs of the reference's own CombinedDataset.__getitem__ (fixture preproc.npz)."""
    from oracle import preproc_ref
    g = golden("preproc")
    for name in g["names"]:
        name = str(name)
        img = preproc_ref.preprocess(g[f"{name}/image_in"], name)
        np.testing.assert_allclose(np.asarray(img, np.float32), g[f"{name}/image_out"], rtol=0, atol=1e-7, err_msg=name)
        np.testing.assert_array_equal(preproc_ref.remap_labels(g[f"{name}/label_in"], name), g[f"{name}/label_out"], err_msg=name)


def test_augment_restatement_is_self_consistent():
    """SURVEY §8 F4 augmentation chain (utils/dataloader.py:223-262).  MONAI is absent (parity unpinned): what CAN be
    checked on the CPU is the restatement against independent formulations of the same published algorithms."""
    from oracle import augment_ref as A
    rng = np.random.RandomState(5)
    # bias field: leggrid3d with MONAI's coefficient placement == explicit sum over (i, j, k), i + j + k <= 3
    shape = (6, 5, 7)
    coeff = rng.uniform(0, 0.1, A.n_bias_coeff()).tolist()
    assert A.n_bias_coeff() == 20
    field = A.bias_field(shape, coeff)
    P = [lambda x: np.ones_like(x), lambda x: x, lambda x: 0.5 * (3 * x * x - 1), lambda x: 0.5 * (5 * x ** 3 - 3 * x)]
    cs = [np.linspace(-1, 1, n, dtype=np.float32).astype(np.float64) for n in shape]
    want, c = np.zeros(shape), 0
    for i in range(4):
        for j in range(4 - i):
            for k in range(4 - i - j):
                want += coeff[c] * P[i](cs[0])[:, None, None] * P[j](cs[1])[None, :, None] * P[k](cs[2])[None, None, :]
                c += 1
    assert c == 20 and np.abs(field - want).max() < 1e-14
    # histogram shift == np.interp on the scaled control points
    img = rng.rand(1, 6, 5, 7).astype(np.float32) * 3 - 1
    ref = np.linspace(0, 1, 5)
    flt = np.array([0, 0.1, 0.45, 0.9, 1.0])
    got = A.histogram_shift(img, ref, flt)
    lo, hi = img.min(), img.max()
    assert np.abs(got - np.interp(img, ref * (hi - lo) + lo, flt * (hi - lo) + lo)).max() < 1e-5
    assert np.array_equal(A.histogram_shift(np.full((1, 2, 2, 2), 3.0, np.float32), ref, flt), np.full((1, 2, 2, 2), 3.0, np.float32))
    # contrast keeps the range and is the identity at gamma 1
    out = A.adjust_contrast(img, 1.0)
    assert np.abs(out - img).max() < 1e-5
    out = A.adjust_contrast(img, 0.7)
    assert abs(out.min() - lo) < 1e-6 and abs(out.max() - hi) < 1e-5 and (out >= img - 1e-6).all()
    # coarse dropout: both keys get the same boxes, every channel
    lab = rng.randint(1, 4, (1, 6, 5, 7))
    a = A.coarse_dropout(img, [(1, 0, 2), (3, 2, 4)], (2, 2, 3), 0.0)
    b = A.coarse_dropout(lab, [(1, 0, 2), (3, 2, 4)], (2, 2, 3), 0)
    assert ((a == 0) == (b == 0)).all() and (b == 0).sum() == 2 * 12


def test_augment_host_draws_match_restatement():
    """The product's host-side draws (augment.CombinedTransform.draw) and the oracle's follow the same stream layout."""
    from oracle import augment_ref as A
    from multimodal_segmentation_project_amd import augment
    tf = augment.combined_transform(prob=0.6, noise="host").set_random_state(1234)
    st = A.Streams(1234)
    fired = np.zeros(5, int)
    for _ in range(12):
        p = tf.draw((1, 20, 24, 18))
        q = A.draw_params(st, (1, 20, 24, 18), prob=0.6)
        for name, idx in (("bias_coeff", 0), ("noise", 1), ("gamma", 2), ("ref_cp", 3), ("hole_lo", 4)):
            a, b = getattr(p, name), q[name]
            assert (a is None) == (b is None), name
            if a is not None:
                fired[idx] += 1
                assert np.array_equal(np.asarray(a), np.asarray(b)), name
        if p.ref_cp is not None:
            assert np.array_equal(p.flt_cp, q["flt_cp"]) and (np.diff(p.flt_cp) >= 0).all()
        if p.hole_lo is not None:
            assert p.hole_size == q["hole_size"] == (16, 16, 16)
            assert all(0 <= lo[i] <= (20, 24, 18)[i] - 16 for lo in p.hole_lo for i in range(3))
    assert (fired > 0).all()


def test_oracle_dp_model_matches_the_references_ddp_run(golden):
    """SURVEY §8 E states the parity model of a data-parallel step: gradients = MEAN over ranks of the single-process
    gradients of the shards (BatchNorm statistics and Dice sums rank-local), metrics = mean of the per-rank values.  Here
    that model (oracle/torch_ref on the two shards) is pinned to what the REFERENCE does: tools/gen_golden.py::gen_dp2 ran
    train_unet.train_one_epoch in two gloo processes under accelerate's DDP wrap (fixture dp2.npz, plain1/)."""
    import torch
    import multimodal_segmentation_project_amd as mi
    from oracle import torch_ref
    g = golden("dp2")

    def synth(n, s, seed, blocky):
        gen = torch.Generator().manual_seed(seed)
        x = torch.randn(n, 1, s, s, s, generator=gen)
        y = torch.randint(0, 4, (n, 1, s, s, s), generator=gen)
        if blocky:
            zz, yy, xx = torch.meshgrid(torch.arange(s), torch.arange(s), torch.arange(s), indexing="ij")
            lab = ((zz // (s // 4)) + (yy // (s // 4)) + (xx // (s // 4))) % 4
            y = lab[None, None].expand(n, 1, s, s, s).contiguous().long()
            x = y.float() / 3.0 + 0.1 * x
        return x, y

    torch.manual_seed(0)
    sd0 = {k: v.detach().clone() for k, v in mi.UNet3D(1, 4, dropout_rate=0.0).state_dict().items()}
    grads, losses, bns = [], [], []
    for r in range(2):
        sd = {k: v.clone() for k, v in sd0.items()}
        for k, v in sd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
        x, y = synth(2, 32, 1000 + 10 * r, True)
        logits, _, upd = torch_ref.unet3d_forward(sd, x, train=True)
        loss = torch_ref.seg_loss(logits, y, "combined")
        loss.backward()
        grads.append({k: v.grad.clone() for k, v in sd.items() if v.requires_grad})
        losses.append(float(loss))
        bns.append(np.concatenate([upd[k].numpy().ravel() for k in sorted(upd) if "running" in k]))
    np.testing.assert_allclose(0.5 * (losses[0] + losses[1]), g["r0/plain1/result"][0], rtol=1e-5)
    for k, rn in zip(list(g["r0/plain1/grad_names"]), g["r0/plain1/grad_norms"]):
        if rn > 1e-6:
            mean = 0.5 * (grads[0][k] + grads[1][k])
            assert abs(float(mean.double().norm()) - rn) / rn < 2e-3, k
    for kk in g:
        if kk.startswith("r0/plain1/grad/"):
            k = kk[len("r0/plain1/grad/"):]
            ref = g[kk]
            if np.linalg.norm(ref) < 1e-6:
                continue
            mean = (0.5 * (grads[0][k] + grads[1][k])).numpy()
            got = mean[:ref.shape[0]] if ref.shape != mean.shape else mean
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-3, k
    # BatchNorm running statistics are the RANK-LOCAL ones on each rank (no SyncBatchNorm)
    for r in range(2):
        np.testing.assert_allclose(bns[r], g[f"r{r}/plain1/bn_after"], rtol=1e-4, atol=1e-6)
    # ... and the end-of-epoch boundary of a prepared loader (3 batches per rank, accumulation 2): steps after batches 2, 3
    np.testing.assert_array_equal(g["r0/loader/step_after_batch"], [2, 3])
    np.testing.assert_array_equal(g["r1/loader/step_after_batch"], [2, 3])
