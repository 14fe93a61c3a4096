"""ctypes binding of oracle/mi3d_oracle.c (test infrastructure; numpy in, numpy out, NCDHW)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmi3d_oracle.so")
_lib = None

f32p = C.POINTER(C.c_float)
i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)


def build(force=False):
    src = os.path.join(_HERE, "mi3d_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(f32p)


def conv3d_fwd(x, w, b):
    x, w, b = _f(x), _f(w), _f(b)
    N, Cin, D, H, W = x.shape
    Cout, k = w.shape[0], w.shape[2]
    y = np.empty((N, Cout, D, H, W), np.float32)
    lib().orc_conv3d_fwd(_p(x), _p(w), _p(b), _p(y), N, Cin, Cout, D, H, W, k)
    return y


def conv3d_bwd(x, w, gy):
    x, w, gy = _f(x), _f(w), _f(gy)
    N, Cin, D, H, W = x.shape
    Cout, k = w.shape[0], w.shape[2]
    gx, gw, gb = np.empty_like(x), np.empty_like(w), np.empty((Cout,), np.float32)
    lib().orc_conv3d_bwd(_p(x), _p(w), _p(gy), _p(gx), _p(gw), _p(gb), N, Cin, Cout, D, H, W, k)
    return gx, gw, gb


def bn_train_fwd(x, gamma, beta, rmean, rvar, momentum=0.1, eps=1e-5):
    x = _f(x)
    N, Cc = x.shape[:2]
    V = int(np.prod(x.shape[2:]))
    y = np.empty_like(x)
    sm, si = np.empty(Cc, np.float32), np.empty(Cc, np.float32)
    rm, rv = _f(rmean).copy(), _f(rvar).copy()
    lib().orc_bn_train_fwd(_p(x), _p(_f(gamma)), _p(_f(beta)), _p(rm), _p(rv), C.c_double(momentum),
                           C.c_double(eps), _p(y), _p(sm), _p(si), N, Cc, C.c_int64(V))
    return y, sm, si, rm, rv


def bn_eval_fwd(x, gamma, beta, rmean, rvar, eps=1e-5):
    x = _f(x)
    N, Cc = x.shape[:2]
    V = int(np.prod(x.shape[2:]))
    y = np.empty_like(x)
    lib().orc_bn_eval_fwd(_p(x), _p(_f(gamma)), _p(_f(beta)), _p(_f(rmean)), _p(_f(rvar)), C.c_double(eps),
                          _p(y), N, Cc, C.c_int64(V))
    return y


def bn_train_bwd(x, gy, gamma, save_mean, save_invstd):
    x, gy = _f(x), _f(gy)
    N, Cc = x.shape[:2]
    V = int(np.prod(x.shape[2:]))
    gx, gg, gb = np.empty_like(x), np.empty(Cc, np.float32), np.empty(Cc, np.float32)
    lib().orc_bn_train_bwd(_p(x), _p(gy), _p(_f(gamma)), _p(_f(save_mean)), _p(_f(save_invstd)), _p(gx),
                           _p(gg), _p(gb), N, Cc, C.c_int64(V))
    return gx, gg, gb


def relu_drop_fwd(x, scale=None):
    x = _f(x)
    N, Cc = x.shape[:2]
    V = int(np.prod(x.shape[2:]))
    y = np.empty_like(x)
    lib().orc_relu_drop_fwd(_p(x), _p(_f(scale)), _p(y), N, Cc, C.c_int64(V))
    return y


def relu_drop_bwd(x, gy, scale=None):
    x, gy = _f(x), _f(gy)
    N, Cc = x.shape[:2]
    V = int(np.prod(x.shape[2:]))
    gx = np.empty_like(x)
    lib().orc_relu_drop_bwd(_p(x), _p(_f(scale)), _p(gy), _p(gx), N, Cc, C.c_int64(V))
    return gx


def maxpool2_fwd(x):
    x = _f(x)
    N, Cc, D, H, W = x.shape
    y = np.empty((N, Cc, D // 2, H // 2, W // 2), np.float32)
    lib().orc_maxpool2_fwd(_p(x), _p(y), N, Cc, D, H, W)
    return y


def maxpool2_bwd(x, gy):
    x, gy = _f(x), _f(gy)
    N, Cc, D, H, W = x.shape
    gx = np.empty_like(x)
    lib().orc_maxpool2_bwd(_p(x), _p(gy), _p(gx), N, Cc, D, H, W)
    return gx


def convT2_fwd(x, w, b):
    x, w, b = _f(x), _f(w), _f(b)
    N, Cin, D, H, W = x.shape
    Cout = w.shape[1]
    y = np.empty((N, Cout, 2 * D, 2 * H, 2 * W), np.float32)
    lib().orc_convT2_fwd(_p(x), _p(w), _p(b), _p(y), N, Cin, Cout, D, H, W)
    return y


def convT2_bwd(x, w, gy):
    x, w, gy = _f(x), _f(w), _f(gy)
    N, Cin, D, H, W = x.shape
    Cout = w.shape[1]
    gx, gw, gb = np.empty_like(x), np.empty_like(w), np.empty(Cout, np.float32)
    lib().orc_convT2_bwd(_p(x), _p(w), _p(gy), _p(gx), _p(gw), _p(gb), N, Cin, Cout, D, H, W)
    return gx, gw, gb


# loss-kind table shared with the product's host code by VALUE only (no import either way)
LOSS_KINDS = {
    # name: (w_ce, region_kind, w_reg, alpha, beta, eps)
    "combined": (1.0, 1, 1.0, 0.0, 0.0, 1e-5),          # utils/metrics.py:14-40
    "dice": (0.0, 1, 1.0, 0.0, 0.0, 1e-5),              # train_unet.py:186-198
    "tversky": (0.0, 2, 1.0, 0.5, 0.5, 1e-6),           # train_unet.py:182-185
    "ce_tversky": (0.3, 2, 0.7, 0.5, 0.5, 1e-6),        # train_unet.py:199-203 (alpha=beta=0.5)
    "ce_tversky_default": (0.3, 2, 0.7, 0.7, 0.3, 1e-6),  # utils/metrics.py:158-167 defaults
}


def seg_loss(logits, labels, kind="combined", teacher=None, kd_alpha=None, temperature=2.0, want_grad=True,
             params=None):
    lg = _f(logits)
    N, Cc = lg.shape[:2]
    V = int(np.prod(lg.shape[2:]))
    lb = np.ascontiguousarray(labels, dtype=np.int64).reshape(N, V)
    w_ce, rk, w_reg, a, b, eps = params if params is not None else LOSS_KINDS[kind]
    w_kd = 0.0
    t = None
    if teacher is not None:
        # utils/metrics.py:169-190: alpha*ce_tversky(default a,b) + (1-alpha)*T^2*KL
        w_ce, rk, w_reg, a, b, eps = LOSS_KINDS["ce_tversky_default"]
        w_ce, w_reg, w_kd = w_ce * kd_alpha, w_reg * kd_alpha, 1.0 - kd_alpha
        t = _f(teacher)
    loss = C.c_double(0.0)
    grad = np.empty_like(lg) if want_grad else None
    lib().orc_seg_loss(_p(lg), lb.ctypes.data_as(i64p), _p(t), N, Cc, C.c_int64(V), C.c_double(w_ce), rk,
                       C.c_double(w_reg), C.c_double(a), C.c_double(b), C.c_double(eps), C.c_double(w_kd),
                       C.c_double(temperature), C.byref(loss), _p(grad))
    return loss.value, grad


def seg_metrics(logits, labels):
    lg = _f(logits)
    N, Cc, D = lg.shape[:3]
    V = int(np.prod(lg.shape[2:]))
    lb = np.ascontiguousarray(labels, dtype=np.int64).reshape(N, V)
    out = np.zeros(3, np.float64)
    counts = np.zeros(3 * Cc + 1, np.int64)
    lib().orc_seg_metrics(_p(lg), lb.ctypes.data_as(i64p), N, Cc, D, C.c_int64(V), out.ctypes.data_as(f64p),
                          counts.ctypes.data_as(i64p))
    return {"iou": out[0], "dice": out[1], "acc": out[2], "counts": counts}


def gap_fwd(x):
    x = _f(x)
    N, Cc = x.shape[:2]
    V = int(np.prod(x.shape[2:]))
    y = np.empty((N, Cc), np.float32)
    lib().orc_gap_fwd(_p(x), _p(y), N, Cc, C.c_int64(V))
    return y


def linear_fwd(x, w, b):
    x, w, b = _f(x), _f(w), _f(b)
    M, K = x.shape
    No = w.shape[0]
    y = np.empty((M, No), np.float32)
    lib().orc_linear_fwd(_p(x), _p(w), _p(b), _p(y), M, K, No)
    return y


def linear_bwd(x, w, gy):
    x, w, gy = _f(x), _f(w), _f(gy)
    M, K = x.shape
    No = w.shape[0]
    gx, gw, gb = np.empty_like(x), np.empty_like(w), np.empty(No, np.float32)
    lib().orc_linear_bwd(_p(x), _p(w), _p(gy), _p(gx), _p(gw), _p(gb), M, K, No)
    return gx, gw, gb


def adamw_step(p, g, m, v, lr, b1, b2, eps, wd, step):
    p, m, v = _f(p).copy(), _f(m).copy(), _f(v).copy()
    g = _f(g)
    lib().orc_adamw_step(_p(p), _p(g), _p(m), _p(v), C.c_int64(p.size), C.c_double(lr), C.c_double(b1),
                         C.c_double(b2), C.c_double(eps), C.c_double(wd), C.c_int64(step))
    return p, m, v
