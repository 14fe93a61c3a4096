"""Data-parallel TrainStep on the GPU with 2 ranks sharing cuda:0 over `gloo` (the 1-GPU box cannot run RCCL with two
ranks on one device): exercises the real kernels + the comm-stream/bucket logic.  Parity reference:
the mean of the single-process gradients of the two shards (NOT one batch-2N pass: BN stats and Dice are batch-
non-linear)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _synth(n, s, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 1, s, s, s, generator=g), torch.randint(0, 4, (n, 1, s, s, s), generator=g)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import multimodal_segmentation_project_amd as mi
        from multimodal_segmentation_project_amd.trainer import TrainStep
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        torch.manual_seed(50 + rank)                       # broadcast must make the ranks identical
        model = mi.UNet3D(1, 4, dropout_rate=0.0).to(dev).train()
        ts = TrainStep(model, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=torch.float32)
        x, y = _synth(1, 32, 900 + rank)
        p0 = ts.arena.p.clone()
        met = ts.step(x.to(dev), y.to(dev)).cpu()
        torch.cuda.synchronize()
        q.put((rank, "ok", p0.cpu().numpy(), ts.arena.g.cpu().numpy(), ts.arena.p.cpu().numpy(), met.numpy()))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, "err " + repr(e) + traceback.format_exc(), None, None, None, None))
    finally:
        dist.destroy_process_group()


def test_trainstep_dp2_matches_mean_of_shards():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    (_, _, p0a, ga, pa, ma), (_, _, p0b, gb, pb, mb) = res
    np.testing.assert_array_equal(p0a, p0b)                # C1: rank 0's parameters everywhere
    np.testing.assert_array_equal(ga, gb)                  # C2: identical averaged gradients
    np.testing.assert_array_equal(pa, pb)                  # identical AdamW update
    np.testing.assert_allclose(ma, mb)                     # C4: averaged metrics
    # reference: single-process gradients of each shard from the same parameters, averaged
    import multimodal_segmentation_project_amd as mi
    from multimodal_segmentation_project_amd import metrics as M
    from multimodal_segmentation_project_amd.dp import ParamArena
    dev = "cuda:0"
    grads, losses = [], []
    for r in range(2):
        torch.manual_seed(50)
        m = mi.UNet3D(1, 4, dropout_rate=0.0).to(dev).train()
        m.compute_dtype = torch.float32
        arena = ParamArena(m.parameters(), dev)
        arena.p.copy_(torch.from_numpy(p0a))
        for p in m.parameters():
            p.grad = None
        x, y = _synth(1, 32, 900 + r)
        o = m(x.to(dev))
        l = M.combined_loss(o, y.to(dev))
        l.backward()
        flat = torch.zeros_like(arena.p)
        for p, off in zip(m.parameters(), arena.offsets):
            flat[off:off + p.numel()] = p.grad.reshape(-1)
        grads.append(flat.cpu().numpy())
        losses.append(float(l))
    ref = 0.5 * (grads[0] + grads[1])
    err = np.linalg.norm(ga - ref) / np.linalg.norm(ref)
    assert err < 1e-5, err
    np.testing.assert_allclose(ma[0], 0.5 * (losses[0] + losses[1]), rtol=1e-5)


def _blocky():
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn(2, 1, 16, 16, 16, generator=gen)
    zz, yy, xx = torch.meshgrid(torch.arange(16), torch.arange(16), torch.arange(16), indexing="ij")
    lab = ((zz // 4) + (yy // 4) + (xx // 4)) % 4
    y = lab[None, None].expand(2, 1, 16, 16, 16).contiguous().long()
    return y.float() / 3.0 + 0.1 * x, y


def test_trainstep_trajectory_golden(golden):
    """TrainStep (native loop: fused loss, flat-arena AdamW kernel, on-device metrics) reproduces the reference's
    5-step AdamW trajectory (train_unet.py:378 defaults lr 1e-3, wd 0.01) on the learnable blocky problem."""
    import multimodal_segmentation_project_amd as mi
    from multimodal_segmentation_project_amd.trainer import TrainStep
    g = golden("default_unet")
    dev = "cuda:0"
    torch.manual_seed(0)
    model = mi.UNet3D(1, 4, dropout_rate=0.0).to(dev).train()
    ts = TrainStep(model, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=torch.float32)
    x, y = _blocky()
    losses, dices = [], []
    for _ in range(5):
        m = ts.step(x.to(dev), y.to(dev)).cpu()
        losses.append(float(m[0]))
        dices.append(float(m[2]))
    np.testing.assert_allclose(losses, g["traj/loss"], rtol=2e-3)
    np.testing.assert_allclose(dices, g["traj/dice"], atol=2e-3)
    # parameters after 5 steps: per-tensor digests of the reference (sum / abs-sum), via state_dict views of the arena
    sd = model.state_dict()
    keys = sorted(sd.keys())
    dig = np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys])
    ref = g["traj/param_digest_after5"]
    # conv biases in front of a train-mode BatchNorm have roundoff-only gradients; Adam turns that noise into lr-sized
    # steps in BOTH implementations, and BN running means / betas inherit it: compare the weight tensors (conv,
    # upconv, BN gamma: 45 tensors, 99.9 % of the parameters) tightly and the rest by absolute drift
    sel = np.array([k.endswith(".weight") for k in keys])
    np.testing.assert_allclose(dig[sel, 1], ref[sel, 1], rtol=2e-3)
    rest = np.array([(not k.endswith(".weight")) and "num_batches" not in k for k in keys])
    assert np.max(np.abs(dig[rest, 1] - ref[rest, 1]) / np.maximum(np.abs(ref[rest, 1]), 1.0)) < 0.05


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_trainstep_graph_equals_eager(dtype):
    """hipGraph replay of the captured step == eager launches, bitwise (same kernels, same order)."""
    import multimodal_segmentation_project_amd as mi
    from multimodal_segmentation_project_amd.trainer import TrainStep
    dev = "cuda:0"
    x, y = _synth(2, 32, 77)
    outs = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        model = mi.UNet3D(1, 4, dropout_rate=0.0).to(dev).train()
        ts = TrainStep(model, lr=1e-3, weight_decay=0.01, compute_dtype=dtype, use_graph=use_graph)
        ts.load_batch(x.to(dev), y.to(dev))
        mets = [ts.step_static().clone() for _ in range(4)]
        torch.cuda.synchronize()
        outs.append((torch.stack(mets).cpu(), ts.arena.p.clone().cpu(), int(ts.arena.step.item())))
    assert outs[0][2] == outs[1][2] == 4
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


def test_trainstep_distillation_and_eval(golden):
    """distill_unet.py:107-115 through TrainStep(kd_teacher=...): loss == golden; evaluate() runs in eval mode."""
    import multimodal_segmentation_project_amd as mi
    from multimodal_segmentation_project_amd.trainer import TrainStep
    g = golden("distill")
    dev = "cuda:0"
    torch.manual_seed(0)
    student = mi.UNet3D(1, 4, dropout_rate=0.0).to(dev).train()
    torch.manual_seed(1)
    teacher = mi.UNet3D(1, 4, dropout_rate=0.0)
    tsd = teacher.state_dict()
    off = 0
    for k in g["teacher_bn_keys"]:
        n = tsd[k].numel()
        tsd[k].copy_(torch.from_numpy(g["teacher_bn"][off:off + n]))
        off += n
    teacher = teacher.to(dev).eval()
    ts = TrainStep(student, lr=0.0, weight_decay=0.0, kd_teacher=teacher, kd_alpha=0.7, kd_temperature=2.0,
                   compute_dtype=torch.float32)
    x, y = _synth(2, 16, 1234)
    m = ts.step(x.to(dev), y.to(dev)).cpu()
    np.testing.assert_allclose(float(m[0]), g["loss"], rtol=5e-5)
    gn = np.array([float(p.grad.double().norm()) for p in student.parameters()])
    big = g["grad_norms"] > 1e-5
    np.testing.assert_allclose(gn[big], g["grad_norms"][big], rtol=5e-3)
    before = {k: v.clone() for k, v in student.state_dict().items() if "running" in k}
    ev = ts.evaluate(x.to(dev), y.to(dev)).cpu()
    assert torch.isfinite(ev).all()
    for k, v in student.state_dict().items():
        if "running" in k:
            assert torch.equal(v, before[k])          # eval forward must not touch running statistics


@pytest.mark.gpu
def test_optimizer_state_exports_in_torch_adamw_layout():
    """the fused arena AdamW state fills the reference checkpoint's optimizer_state_dict slot in a
    layout torch.optim.AdamW.load_state_dict accepts, and equals torch's own state after the same step."""
    import multimodal_segmentation_project_amd as mi
    from multimodal_segmentation_project_amd import checkpoint as ck
    from multimodal_segmentation_project_amd.trainer import TrainStep
    torch.manual_seed(0)
    dev = torch.device("cuda")
    m = mi.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(dev).train()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 1, 16, 16, 16, generator=g).to(dev)
    y = torch.randint(0, 4, (1, 1, 16, 16, 16), generator=g).to(dev)
    ts = TrainStep(m, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=torch.float32)
    ts.step(x, y)
    sd = ck.adamw_state_dict(ts)
    ref = mi.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=0.01)
    opt.load_state_dict(sd)                                   # layout accepted by torch
    assert len(sd["state"]) == 82 and float(sd["state"][0]["step"]) == 1.0
    # after one step from zero moments: exp_avg = (1-b1)*g, exp_avg_sq = (1-b2)*g^2
    grads = [p.grad.detach().cpu() for p in m.parameters()]
    for i in (0, 40, 81):
        assert torch.allclose(sd["state"][i]["exp_avg"], 0.1 * grads[i], rtol=1e-5, atol=1e-12)
        assert torch.allclose(sd["state"][i]["exp_avg_sq"], 0.001 * grads[i] ** 2, rtol=1e-4, atol=1e-20)


@pytest.mark.gpu
def test_frozen_encoder_step():
    """freezing encoder + bottleneck (train_unet.py:31-43) = no gradient, no AdamW/weight-decay update for
    them, the remaining gradients unchanged; unfreezing + reset_optimizer() trains everything again."""
    import multimodal_segmentation_project_amd as mi
    from multimodal_segmentation_project_amd.trainer import TrainStep
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 1, 16, 16, 16, generator=g).to(dev)
    y = torch.randint(0, 4, (2, 1, 16, 16, 16), generator=g).to(dev)

    def make():
        torch.manual_seed(0)
        m = mi.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(dev).train()
        return m, TrainStep(m, loss="combined", lr=1e-2, weight_decay=0.1, compute_dtype=torch.float32)

    m_all, ts_all = make()
    ts_all.step(x, y)
    m_frz, ts_frz = make()
    frozen = [p for mod in (m_frz.encoder, m_frz.bottleneck) for p in mod.parameters()]
    for p in frozen:
        p.requires_grad = False
    before = {k: v.detach().clone() for k, v in m_frz.named_parameters()}
    out = ts_frz.step(x, y)
    assert torch.isfinite(out).all()
    st = ts_frz._static
    assert st["nseg_run"] == len(m_frz.encoder) + 1 and len(st["opt_ranges"]) == 1       # final + decoders only
    n_frozen = len(frozen)
    for i, ((k, p), (_, q)) in enumerate(zip(m_frz.named_parameters(), m_all.named_parameters())):
        if i < n_frozen:
            assert torch.equal(p, before[k]), k                       # untouched: no update, no weight decay
        else:
            assert not torch.equal(p, before[k]), k
            assert torch.equal(p, q), k                               # same update as the unfrozen run
    a = ts_frz.arena
    lo, hi = a.range_of(0, n_frozen)
    assert float(a.m[lo:hi].abs().max()) == 0.0 and float(a.v[lo:hi].abs().max()) == 0.0
    assert int(a.step.item()) == 1
    # unfreeze (optimizer rebuilt in the reference): everything moves
    for p in frozen:
        p.requires_grad = True
    ts_frz.reset_optimizer()
    snap = {k: v.detach().clone() for k, v in m_frz.named_parameters()}
    ts_frz.step(x, y)
    assert all(not torch.equal(p, snap[k]) for k, p in m_frz.named_parameters() if p.dim() == 5)
    assert int(a.step.item()) == 1


@pytest.mark.gpu
def test_lr_change_reaches_a_captured_graph():
    """The learning rate is a kernel argument baked into the step graph: assigning TrainStep.lr must drop the graph."""
    import multimodal_segmentation_project_amd as mi
    from multimodal_segmentation_project_amd.trainer import TrainStep
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = mi.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(dev).train()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 1, 16, 16, 16, generator=g).to(dev)
    y = torch.randint(0, 4, (1, 1, 16, 16, 16), generator=g).to(dev)
    ts = TrainStep(m, loss="combined", lr=1e-3, weight_decay=0.0, compute_dtype=torch.float32, use_graph=True)
    ts.load_batch(x, y)
    ts.step_static()
    w0 = m.final_conv.weight.detach().clone()
    ts.lr = 0.0
    ts.step_static()
    assert torch.equal(m.final_conv.weight, w0)              # lr = 0, wd = 0: parameters frozen in place
    ts.lr = 1e-3
    ts.step_static()
    assert not torch.equal(m.final_conv.weight, w0)


@pytest.mark.gpu
def test_reference_volume_shape_192():
    """The reference trains/evaluates on 192^3 whole volumes, batch 1 (run_training.sh / test_model.py).  One bf16 train
    step + eval at that shape: finite, bitwise reproducible, and the train-mode Dice of the bf16 path within 1e-3 of the
    exact fp32 path on the same weights (BASELINE parity bar) -- size-independent properties at the real size."""
    import multimodal_segmentation_project_amd as mi
    from multimodal_segmentation_project_amd.trainer import TrainStep
    dev = torch.device("cuda")
    S = 192
    g = torch.Generator().manual_seed(1234)
    q = S // 4
    idx = torch.arange(S) // q
    lab = ((idx[:, None, None] + idx[None, :, None] + idx[None, None, :]) % 4).to(torch.int64)       # blocky labels
    y = lab[None, None].contiguous()
    x = (lab.float() / 3 + 0.1 * torch.randn(S, S, S, generator=g))[None, None].contiguous()
    x, y = x.to(dev), y.to(dev)

    def one(dtype):
        torch.manual_seed(0)
        m = mi.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(dev).train()
        ts = TrainStep(m, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=dtype)
        out = ts.step(x, y).cpu()
        ev = ts.evaluate(x, y).cpu()
        return out, ev

    a, ea = one(torch.bfloat16)
    b, eb = one(torch.bfloat16)
    assert torch.isfinite(a).all() and torch.isfinite(ea).all()
    assert torch.equal(a, b) and torch.equal(ea, eb)
    c, _ = one(torch.float32)
    assert abs(float(a[0]) - float(c[0])) < 5e-3 * abs(float(c[0]))          # loss
    assert abs(float(a[2]) - float(c[2])) < 1e-3                               # Dice


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_learnable_run_converges(dtype):
    """SURVEY §8(d) learnable parity run: blocky labels, image = label/3 + 0.1*noise.  The whole step (forward, Dice/CE,
    backward, fused AdamW, BN running statistics) must actually train: loss falls, train Dice and eval-mode Dice > 0.9."""
    import multimodal_segmentation_project_amd as mi
    from multimodal_segmentation_project_amd.trainer import TrainStep
    dev = torch.device("cuda")
    S = 32
    g = torch.Generator().manual_seed(7)
    idx = torch.arange(S) // (S // 4)
    lab = ((idx[:, None, None] + idx[None, :, None] + idx[None, None, :]) % 4).to(torch.int64)
    y = lab[None, None].repeat(2, 1, 1, 1, 1).contiguous()
    x = (y.float() / 3 + 0.1 * torch.randn(2, 1, S, S, S, generator=g)).contiguous()
    x, y = x.to(dev), y.to(dev)
    torch.manual_seed(0)
    m = mi.UNet3D(in_channels=1, out_channels=4, dropout_rate=0.0).to(dev).train()
    ts = TrainStep(m, loss="combined", lr=3e-3, weight_decay=0.01, compute_dtype=dtype, use_graph=True)
    ts.load_batch(x, y)
    hist = [ts.step_static().cpu().clone() for _ in range(60)]
    first, last = hist[0], hist[-1]
    assert all(torch.isfinite(h).all() for h in hist)
    assert float(last[0]) < 0.35 * float(first[0]), (first, last)
    assert float(last[2]) > 0.9, last                                   # train-mode Dice
    ev = ts.evaluate(x, y).cpu()
    assert float(ev[2]) > 0.9, ev                                       # eval mode: BN running statistics are usable


def _nccl1_worker(port, q):
    """1-rank RCCL group on the 1-GPU box: the bucket path (side stream, ReduceOp.AVG all-reduces, joins, segmented step
    graphs) runs for real and must be bitwise invisible."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import multimodal_segmentation_project_amd as mi
        from multimodal_segmentation_project_amd import unet_dann
        from multimodal_segmentation_project_amd.dann import DomainDiscriminator
        from multimodal_segmentation_project_amd.trainer import DannStep, TrainStep
        dev = torch.device("cuda", 0)
        x, y = _synth(2, 32, 77)
        xt, _ = _synth(2, 32, 78)
        outs = {}
        for name, kw in (("plain", {}), ("comm_eager", dict(force_comm=True)), ("comm_graph", dict(force_comm=True, use_graph=True)),
                         ("graph", dict(use_graph=True))):
            torch.manual_seed(0)
            model = mi.UNet3D(1, 4, dropout_rate=0.1).to(dev).train()
            ts = TrainStep(model, lr=1e-3, weight_decay=0.01, compute_dtype=torch.bfloat16, **kw)
            assert ts.do_comm == ("comm" in name)
            ts.load_batch(x.to(dev), y.to(dev))
            mets = [ts.step_static().clone() for _ in range(3)]
            torch.cuda.synchronize()
            nseg = len(next(iter(ts._static["graphs"].by_variant.values()))) if ts.use_graph else 0
            ts.sync_buffers()
            outs[name] = (torch.stack(mets).cpu(), ts.arena.p.clone().cpu(), nseg)
        base = outs["plain"]
        for name, o in outs.items():
            assert torch.equal(o[0], base[0]) and torch.equal(o[1], base[1]), name
        assert outs["graph"][2] == 1 and outs["comm_graph"][2] >= 3          # comm-free runs of kernels between all-reduces
        # DANN under forced communication (discriminator arena = the fifth bucket)
        douts = []
        for kw in ({}, dict(force_comm=True, use_graph=True)):
            torch.manual_seed(0)
            seg = unet_dann.UNet3D(1, 4, dropout_rate=0.0).to(dev).train()
            torch.manual_seed(3)
            disc = DomainDiscriminator(256).to(dev).train()
            st = DannStep(seg, disc, loss="ce_tversky", lambda_domain=0.2, compute_dtype=torch.bfloat16, **kw)
            st.load_batch(x.to(dev), y.to(dev), xt.to(dev))
            mets = [st.step_static().clone() for _ in range(2)]
            torch.cuda.synchronize()
            douts.append((torch.stack(mets).cpu(), st.arena.p.clone().cpu(), st.disc_arena.p.clone().cpu()))
        for a, b in zip(*douts):
            assert torch.equal(a, b)
        q.put("ok")
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put("err " + repr(e) + traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_forced_comm_path_over_rccl_is_bitwise_invisible():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl1_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=900)
    p.join(timeout=120)
    assert res == "ok", res


def _dp2_modes_worker(rank, world, port, q):
    """world 2 over gloo on one GPU: gradient accumulation + DP, frozen encoder + DP, DANN + DP, segmented graphs."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import multimodal_segmentation_project_amd as mi
        from multimodal_segmentation_project_amd import unet_dann
        from multimodal_segmentation_project_amd.dann import DomainDiscriminator
        from multimodal_segmentation_project_amd.trainer import DannStep, TrainStep
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        res = {}
        # (a) accumulation 2 under DP with segmented graphs: gradients are exchanged on the boundary micro-step only
        torch.manual_seed(50 + rank)
        model = mi.UNet3D(1, 4, dropout_rate=0.0).to(dev).train()
        ts = TrainStep(model, lr=1e-3, weight_decay=0.01, compute_dtype=torch.float32, grad_accum=2, use_graph=True)
        for i in range(2):
            x, y = _synth(1, 32, 900 + 10 * i + rank)
            met = ts.step(x.to(dev), y.to(dev)).clone()
        torch.cuda.synchronize()
        res["accum_g"], res["accum_p"], res["accum_met"] = ts.arena.g.cpu().numpy(), ts.arena.p.cpu().numpy(), met.cpu().numpy()
        # (b) frozen encoder + bottleneck under DP
        torch.manual_seed(60 + rank)
        model = mi.UNet3D(1, 4, dropout_rate=0.0).to(dev).train()
        for mod in (model.encoder, model.bottleneck):
            for p in mod.parameters():
                p.requires_grad = False
        ts = TrainStep(model, lr=1e-3, weight_decay=0.01, compute_dtype=torch.float32)
        x, y = _synth(1, 32, 950 + rank)
        ts.step(x.to(dev), y.to(dev))
        torch.cuda.synchronize()
        res["frz_g"], res["frz_p"] = ts.arena.g.cpu().numpy(), ts.arena.p.cpu().numpy()
        res["frz_sched"] = sorted(ts._static["comm_after"].items())
        # (c) DANN under DP
        torch.manual_seed(70 + rank)
        seg = unet_dann.UNet3D(1, 4, dropout_rate=0.0).to(dev).train()
        torch.manual_seed(80 + rank)
        disc = DomainDiscriminator(256).to(dev).train()
        st = DannStep(seg, disc, loss="combined", lambda_domain=0.2, lr=1e-3, weight_decay=0.01, compute_dtype=torch.float32)
        st.disc_p = [0.0, 0.0]
        xs, ys = _synth(2, 16, 700 + rank)
        xt, _ = _synth(2, 16, 800 + rank)
        p0, d0 = st.arena.p.clone(), st.disc_arena.p.clone()
        met = st.step(xs.to(dev), ys.to(dev), xt.to(dev)).clone()
        torch.cuda.synchronize()
        res["dann_p0"], res["dann_d0"] = p0.cpu().numpy(), d0.cpu().numpy()
        res["dann_g"], res["dann_dg"] = st.arena.g.cpu().numpy(), st.disc_arena.g.cpu().numpy()
        res["dann_met"] = met.cpu().numpy()
        q.put((rank, "ok", res))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, "err " + repr(e) + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def test_dp2_accumulation_frozen_and_dann():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp2_modes_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    a, b = res[0][2], res[1][2]
    for k in ("accum_g", "accum_p", "frz_g", "frz_p", "dann_g", "dann_dg", "dann_p0", "dann_d0"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)          # identical on both ranks after the exchange
    np.testing.assert_allclose(a["accum_met"], b["accum_met"])
    np.testing.assert_allclose(a["dann_met"], b["dann_met"])
    # frozen encoder: only the decoder bucket is exchanged (after the last run segment)
    assert len(a["frz_sched"]) == 1
    # DANN reference: single-process DannStep on each rank's shard from the broadcast parameters, gradients averaged
    import multimodal_segmentation_project_amd as mi
    from multimodal_segmentation_project_amd import unet_dann
    from multimodal_segmentation_project_amd.dann import DomainDiscriminator
    from multimodal_segmentation_project_amd.trainer import DannStep
    dev = "cuda:0"
    gs, dgs, mets = [], [], []
    for r in range(2):
        seg = unet_dann.UNet3D(1, 4, dropout_rate=0.0).to(dev).train()
        disc = DomainDiscriminator(256).to(dev).train()
        st = DannStep(seg, disc, loss="combined", lambda_domain=0.2, lr=0.0, weight_decay=0.0, compute_dtype=torch.float32)
        st.disc_p = [0.0, 0.0]
        st.arena.p.copy_(torch.from_numpy(a["dann_p0"]))
        st.disc_arena.p.copy_(torch.from_numpy(a["dann_d0"]))
        xs, ys = _synth(2, 16, 700 + r)
        xt, _ = _synth(2, 16, 800 + r)
        mets.append(st.step(xs.to(dev), ys.to(dev), xt.to(dev)).cpu().numpy().copy())
        gs.append(st.arena.g.cpu().numpy().copy())
        dgs.append(st.disc_arena.g.cpu().numpy().copy())
    ref, dref = 0.5 * (gs[0] + gs[1]), 0.5 * (dgs[0] + dgs[1])
    assert np.linalg.norm(a["dann_g"] - ref) / np.linalg.norm(ref) < 1e-5
    assert np.linalg.norm(a["dann_dg"] - dref) / np.linalg.norm(dref) < 1e-5
    np.testing.assert_allclose(a["dann_met"], 0.5 * (mets[0] + mets[1]), rtol=1e-5)


# ---------------------------------------------------------------------- A15 pinned to the reference's own DDP run (dp2.npz)
def _synth_b(n, s, seed, blocky):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 1, s, s, s, generator=g)
    y = torch.randint(0, 4, (n, 1, s, s, s), generator=g)
    if blocky:
        zz, yy, xx = torch.meshgrid(torch.arange(s), torch.arange(s), torch.arange(s), indexing="ij")
        lab = ((zz // (s // 4)) + (yy // (s // 4)) + (xx // (s // 4))) % 4
        y = lab[None, None].expand(n, 1, s, s, s).contiguous().long()
        x = y.float() / 3.0 + 0.1 * x
    return x, y


def _dp2_fixture_worker(rank, world, port, q):
    """The three scenarios of tools/gen_golden.py::gen_dp2 on the HIP path (fp32 compute), one rank of two."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import multimodal_segmentation_project_amd as mi
        from multimodal_segmentation_project_amd.trainer import TrainStep
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        out = {}

        def fresh(**kw):
            torch.manual_seed(0)
            model = mi.UNet3D(1, 4, dropout_rate=0.0).to(dev).train()
            return model, TrainStep(model, loss="combined", lr=1e-3, weight_decay=0.01, compute_dtype=torch.float32, **kw)

        def summ(tag, model, ts, mets):
            names = [k for k, _ in model.named_parameters()]
            offs = ts.arena.offsets
            g = ts.arena.g
            out[tag + "grad_norms"] = np.array([float(g[o:o + p.numel()].double().norm()) for o, p in zip(offs, ts.arena.params)])
            out[tag + "grad_names"] = np.array(names)
            for k, o, p in zip(names, offs, ts.arena.params):
                if k in ("final_conv.weight", "bottleneck.double_conv.4.weight", "encoder.0.double_conv.0.weight", "upconvs.0.bias"):
                    out[tag + "grad/" + k] = g[o:o + p.numel()].reshape(p.shape).cpu().numpy()
            sd = model.state_dict()
            bnk = sorted(k for k in sd if "running" in k)
            out[tag + "bn_after"] = np.concatenate([sd[k].cpu().numpy().ravel() for k in bnk])
            keys = sorted(sd.keys())
            out[tag + "param_digest_after"] = np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys])
            out["digest_keys"] = np.array(keys)
            out[tag + "result"] = np.mean(np.stack(mets), axis=0)          # train_one_epoch returns means over the batches

        data = [_synth_b(2, 32, 1000 + 10 * rank + i, i % 2 == 0) for i in range(2)]
        for tag, nb in (("plain1/", 1), ("plain2/", 2)):
            model, ts = fresh()
            mets = [ts.step(x.to(dev), y.to(dev)).cpu().numpy().copy() for x, y in data[:nb]]
            torch.cuda.synchronize()
            summ(tag, model, ts, mets)
            if tag == "plain2/":
                ts.sync_buffers()                 # what every rank evaluates / rank 0 saves with (DDP: rank 0's buffers)
                sd = model.state_dict()
                out["plain2/bn_synced"] = np.concatenate([sd[k].cpu().numpy().ravel() for k in sorted(k for k in sd if "running" in k)])
        # prepared-loader scenario: 3 batches per rank, accumulation 2, the reference's zero_grad quirk, step forced on the last
        model, ts = fresh(grad_accum=2, reference_zero_grad_quirk=True)
        mine = [_synth_b(2, 32, 1100 + b, b % 2 == 1) for b in range(6)][rank::2]
        steps, mets = [], []
        for i, (x, y) in enumerate(mine):
            before = int(ts.arena.step.item())
            mets.append(ts.step(x.to(dev), y.to(dev), last_batch=(i == len(mine) - 1)).cpu().numpy().copy())
            if int(ts.arena.step.item()) != before:
                steps.append(i + 1)
        torch.cuda.synchronize()
        out["loader/step_after_batch"] = np.array(steps)
        out["loader/seen_sum"] = np.array([float(x.double().sum()) for x, _ in mine])
        out["loader/micro_after"] = np.array(ts.micro)
        summ("loader/", model, ts, mets)
        q.put((rank, "ok", out))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, "err " + repr(e) + traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def test_dp2_matches_the_references_own_ddp_run(golden):
    """A15: TrainStep on two ranks vs tools/gen_golden.py::gen_dp2 -- the REFERENCE's train_one_epoch executed by two gloo
    processes under accelerate's DDP wrap (train_unet.py:309-312,384-386,221-238).  Pinned: gradients = mean over ranks
    (C2), returned metrics = mean of the per-rank values (C4), BatchNorm statistics rank-local with rank 0's buffers the ones
    that are saved / evaluated (C3), identical AdamW trajectories, and the end-of-epoch accumulation boundary of a prepared
    loader with len % accum != 0 (steps after batches 2 and 3 of 3, only the boundary micro-batch's gradient applied: Q2)."""
    g = golden("dp2")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp2_fixture_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    o0, o1 = res[0][2], res[1][2]

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))

    for tag in ("plain1/", "plain2/", "loader/"):
        # the reference's returned means (already gathered + averaged over ranks): loss, iou, dice, acc
        for o in (o0, o1):
            np.testing.assert_allclose(o[tag + "result"], g["r0/" + tag + "result"], rtol=2e-4, atol=2e-5)
        np.testing.assert_array_equal(o0[tag + "grad_norms"], o1[tag + "grad_norms"])          # all-reduced: bitwise equal
        names = list(g["r0/" + tag + "grad_names"])
        assert names == list(o0[tag + "grad_names"])
        # plain1: gradients at the INITIAL parameters (fp32 summation-order noise only).  The later gradients sit behind an AdamW
        # step whose first update is sign-like (g / sqrt(g^2)): noise-level gradient elements flip whole lr-sized updates in
        # both implementations, so those are compared by norm at 10 % (as tests/test_gpu_round2.py does for the accumulation loop)
        gtol = 1e-2 if tag == "plain1/" else 0.1
        for k, rn, gn in zip(names, g["r0/" + tag + "grad_norms"], o0[tag + "grad_norms"]):
            if rn < 1e-6:
                continue
            assert abs(gn - rn) / rn < gtol, (tag, k, gn, rn)
        for kk in o0:
            if tag == "plain1/" and kk.startswith(tag + "grad/"):
                ref = g["r0/" + kk]
                got = o0[kk][:ref.shape[0]] if ref.shape != o0[kk].shape else o0[kk]
                if np.linalg.norm(ref) > 1e-6:
                    assert rel(got, ref) < 1e-2, (kk, rel(got, ref))
        # BatchNorm buffers: rank 0's trajectory is the reference's rank 0 trajectory
        assert rel(o0[tag + "bn_after"], g["r0/" + tag + "bn_after"]) < 1e-4, tag
    # after ONE step the per-rank (local-statistics) buffers match on BOTH ranks; later DDP keeps overwriting rank 1's
    assert rel(o1["plain1/bn_after"], g["r1/plain1/bn_after"]) < 1e-4
    assert rel(o0["plain1/bn_after"], o1["plain1/bn_after"]) > 1e-3          # ... and they are NOT synchronised statistics
    np.testing.assert_array_equal(o0["plain2/bn_synced"], o1["plain2/bn_synced"])
    assert rel(o1["plain2/bn_synced"], g["r0/plain2/bn_after"]) < 1e-4
    # parameters after the optimizer steps (weights; biases in front of BN are roundoff-driven, see the trajectory test)
    sel = np.array([str(k).endswith(".weight") for k in o0["digest_keys"]])
    for tag in ("plain1/", "plain2/", "loader/"):
        ref, got = g["r0/" + tag + "param_digest_after"], o0[tag + "param_digest_after"]
        assert ref.shape == got.shape
        np.testing.assert_allclose(got[sel, 1], ref[sel, 1], rtol=2e-3)        # conv / upconv / BN-gamma tensors
    np.testing.assert_array_equal(o0["loader/step_after_batch"], g["r0/loader/step_after_batch"])
    np.testing.assert_array_equal(o1["loader/step_after_batch"], g["r1/loader/step_after_batch"])
    np.testing.assert_allclose(o0["loader/seen_sum"], g["r0/loader/seen_sum"], rtol=1e-6)      # same shards as accelerate dealt
    np.testing.assert_allclose(o1["loader/seen_sum"], g["r1/loader/seen_sum"], rtol=1e-6)
    assert int(o0["loader/micro_after"]) % 2 == 0                                           # next epoch starts a fresh window
