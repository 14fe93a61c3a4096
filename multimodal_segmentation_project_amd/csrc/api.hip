// api.hip — extern "C" wrappers (include/mi3d.h) around the per-operator launchers, losses, DANN head,
// optimizer and hipGraph helpers.  The whole-network entry points live in plan.hip.
#include "../../include/mi3d.h"
#include <stdlib.h>
#include <string.h>

#include "ops.h"

static_assert(MI3D_LOSS_COEF_FLOATS >= 2 * MI3D_MAX_CLASSES + 4, "coef buffer too small");

// ---- route switches (common.h): one struct, filled from the environment once, changed only through the ABI
namespace {
struct RouteEntry { const char* name; int Mi3dRoutes::*field; };
const RouteEntry ROUTE_TABLE[] = {
#define MI3D_ROUTE_ENTRY(name, dflt) {#name, &Mi3dRoutes::name},
    MI3D_ROUTE_LIST(MI3D_ROUTE_ENTRY)
#undef MI3D_ROUTE_ENTRY
};
constexpr int N_ROUTES = sizeof(ROUTE_TABLE) / sizeof(ROUTE_TABLE[0]);
Mi3dRoutes load_routes() {
    Mi3dRoutes r;
    for (int i = 0; i < N_ROUTES; i++) {
        char env[64] = "MI3D_";
        size_t n = 5;
        for (const char* c = ROUTE_TABLE[i].name; *c && n + 1 < sizeof(env); c++) env[n++] = (*c >= 'a' && *c <= 'z') ? *c - 32 : *c;
        env[n] = 0;
        const char* v = getenv(env);
        if (!v) continue;
        char* end = nullptr;
        long k = strtol(v, &end, 10);
        r.*(ROUTE_TABLE[i].field) = (end != v) ? (int)k : 1;
    }
    return r;
}
Mi3dRoutes& routes_mut() {
    static Mi3dRoutes r = load_routes();      // thread-safe one-time initialisation
    return r;
}
}  // namespace
const Mi3dRoutes& mi3d_routes() { return routes_mut(); }

static inline LossCfg to_cfg(const mi3d_loss_cfg* c) {
    LossCfg k;
    k.w_ce = c->w_ce; k.region_kind = c->region_kind; k.w_reg = c->w_reg; k.alpha = c->alpha; k.beta = c->beta;
    k.eps = c->eps; k.w_kd = c->w_kd; k.temp = c->temperature > 0.f ? c->temperature : 1.f;
    return k;
}

static inline bool use_mfma(int in_dtype, int out_dtype, int Cin, int Cout, int xcs, int ycs) {
    return in_dtype == MI3D_BF16 && out_dtype == MI3D_BF16 && conv3_mfma_supported(Cin, Cout, xcs, ycs) &&
           !mi3d_routes().force_direct;
}

extern "C" {

size_t mi3d_seg_loss_workspace_bytes(int C) { return seg_loss_ws_bytes(C); }
int mi3d_seg_loss_forward(const float* logits, const int64_t* labels, const float* teacher, int N, int C, int64_t V,
                          const mi3d_loss_cfg* cfg, float* loss_out, float* coef, void* workspace, void* stream) {
    MI3D_CHECK_ARG(logits && labels && cfg && loss_out && coef && workspace, "mi3d_seg_loss_forward: null pointer");
    return seg_loss_fwd(logits, labels, teacher, N, C, V, to_cfg(cfg), loss_out, coef, workspace, (hipStream_t)stream);
}
int mi3d_seg_loss_metrics_forward(const float* logits, const int64_t* labels, const float* teacher, int N, int C, int D,
                                  int64_t V, const mi3d_loss_cfg* cfg, float* loss_out, float* coef, float* metrics_out,
                                  void* loss_workspace, void* metrics_workspace, void* stream) {
    MI3D_CHECK_ARG(logits && labels && cfg && loss_out && coef && metrics_out && loss_workspace && metrics_workspace,
                   "mi3d_seg_loss_metrics_forward: null pointer");
    return seg_loss_fwd(logits, labels, teacher, N, C, V, to_cfg(cfg), loss_out, coef, loss_workspace, (hipStream_t)stream, D,
                        metrics_out, metrics_workspace);
}
int mi3d_seg_loss_backward(const float* logits, const int64_t* labels, const float* teacher, int N, int C, int64_t V,
                           const mi3d_loss_cfg* cfg, const float* coef, const float* grad_out, float* dlogits,
                           void* stream) {
    MI3D_CHECK_ARG(logits && labels && cfg && coef && dlogits, "mi3d_seg_loss_backward: null pointer");
    return seg_loss_bwd(logits, labels, teacher, N, C, V, to_cfg(cfg), coef, grad_out, dlogits, (hipStream_t)stream);
}
size_t mi3d_seg_metrics_workspace_bytes(int C) { return seg_metrics_ws_bytes(C); }
int mi3d_seg_metrics(const float* logits, const int64_t* labels, int N, int C, int D, int64_t V, float* out,
                     void* workspace, void* stream) {
    MI3D_CHECK_ARG(logits && labels && out && workspace, "mi3d_seg_metrics: null pointer");
    return seg_metrics(logits, labels, N, C, D, V, out, workspace, (hipStream_t)stream);
}
int mi3d_seg_class_counts(const float* logits, const int64_t* labels, int N, int C, int64_t V, int64_t* counts,
                          void* workspace, void* stream) {
    MI3D_CHECK_ARG(logits && labels && counts && workspace, "mi3d_seg_class_counts: null pointer");
    return seg_metrics(logits, labels, N, C, 0, V, nullptr, workspace, (hipStream_t)stream, counts);
}

int mi3d_linear_forward(const float* x, const float* w, const float* b, float* y, int M, int K, int Nout, int relu,
                        const float* drop, void* stream) {
    MI3D_CHECK_ARG(x && w && y && M > 0 && K > 0 && Nout > 0, "mi3d_linear_forward: bad arguments");
    return linear_fwd(x, w, b, y, M, K, Nout, relu, drop, (hipStream_t)stream);
}
int mi3d_linear_backward(const float* x, const float* w, const float* y, const float* gy, int M, int K, int Nout,
                         int relu, const float* drop, float* gx, float* gw, float* gb, int accumulate, float gx_scale,
                         float* workspace, void* stream) {
    MI3D_CHECK_ARG(x && w && y && gy && workspace, "mi3d_linear_backward: null pointer");
    return linear_bwd(x, w, y, gy, M, K, Nout, relu, drop, gx, gw, gb, accumulate, gx_scale, workspace, (hipStream_t)stream);
}
int mi3d_softmax_ce_rows(const float* logits, const int64_t* labels, int M, int C, float* loss, float* dlogits,
                         float scale, void* stream) {
    MI3D_CHECK_ARG(logits && labels, "mi3d_softmax_ce_rows: null pointer");
    return softmax_ce_rows(logits, labels, M, C, loss, dlogits, scale, (hipStream_t)stream);
}

int mi3d_scale(const float* x, float* y, int64_t n, float alpha, const float* alpha_dev, void* stream) {
    MI3D_CHECK_ARG(x && y && n >= 0, "mi3d_scale: bad arguments");
    return scale_f32(x, y, n, alpha, alpha_dev, (hipStream_t)stream);
}

int mi3d_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, float grad_scale, int64_t* step_dev, void* stream) {
    MI3D_CHECK_ARG(p && g && m && v && step_dev && n >= 0, "mi3d_adamw_step: bad arguments");
    return adamw_step(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, grad_scale, step_dev, (hipStream_t)stream);
}
int mi3d_adamw_apply(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, float grad_scale, int64_t* step_dev, int increment, void* stream) {
    MI3D_CHECK_ARG(step_dev && (n == 0 || (p && g && m && v)) && n >= 0, "mi3d_adamw_apply: bad arguments");
    return adamw_step(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, grad_scale, step_dev, (hipStream_t)stream, increment);
}
int mi3d_flag_set(int64_t* flag, int64_t value, void* stream) {
    MI3D_CHECK_ARG(flag, "mi3d_flag_set: null flag");
    return flag_set(flag, value, (hipStream_t)stream);
}
int mi3d_flag_wait(int64_t* flag, int64_t value, int64_t timeout_us, void* stream) {
    MI3D_CHECK_ARG(flag && timeout_us > 0, "mi3d_flag_wait: bad arguments");
    return flag_wait(flag, value, timeout_us, (hipStream_t)stream);
}
int mi3d_debug_occupy_cus(int workgroups, int microseconds, float* buf, int64_t n, void* stream) {
    return occupy_cus(workgroups, microseconds, buf, n, (hipStream_t)stream);
}
int mi3d_dropout_scales(float* out, int64_t n, float p, uint64_t* state_dev, void* stream) {
    MI3D_CHECK_ARG(out && state_dev && n >= 0 && p >= 0.f && p <= 1.f, "mi3d_dropout_scales: bad arguments");
    return dropout_scales(out, n, p, state_dev, (hipStream_t)stream);
}

// ---- per-operator entry points ------------------------------------------------------------------
size_t mi3d_conv3_workspace_bytes(int Cin, int Cout, int N, int D, int H, int W) {
    Geo g{N, D, H, W};
    size_t wg = conv3_direct_wgrad_ws_floats(Cin, Cout, g);
    if ((conv3_mfma_supported(Cin, Cout, 16, 16) || (Cin == 1 && Cout % 16 == 0)) && conv3_mfma_wgrad_ws_floats(Cin, Cout, g) > wg)
        wg = conv3_mfma_wgrad_ws_floats(Cin, Cout, g);
    // + the split-K scratch of the fused (input-gradient + weight-gradient) deep-level launch
    size_t sk = conv3_mfma_supported(Cin, Cout, 16, 16) ? conv3_mfma_splitk_floats(Cout, Cin, g) : 0;
    return (conv3_direct_pack_floats(Cin, Cout) + conv3_direct_pack_floats(Cout, Cin) + wg + sk + 64) * sizeof(float);
}
int mi3d_conv3_forward(int in_dtype, int out_dtype, const void* x, int xcs, int Cin, const float* w, const float* bias,
                       void* y, int ycs, int Cout, int N, int D, int H, int W, void* workspace, size_t workspace_bytes,
                       void* stream) {
    MI3D_CHECK_ARG(x && w && y && workspace, "mi3d_conv3_forward: null pointer");
    MI3D_CHECK_ARG(workspace_bytes >= mi3d_conv3_workspace_bytes(Cin, Cout, N, D, H, W), "mi3d_conv3_forward: workspace too small");
    float* wpf = (float*)workspace;
    float* wpd = wpf + conv3_direct_pack_floats(Cin, Cout);
    hipStream_t s = (hipStream_t)stream;
    if (use_mfma(in_dtype, out_dtype, Cin, Cout, xcs, ycs)) {       // bf16 implicit GEMM on the matrix cores
        MI3D_TRY(conv3_mfma_pack(w, Cin, Cout, wpf, wpd, Geo{N, D, H, W}, s));
        return conv3_mfma_fwd(x, xcs, Cin, wpf, bias, y, ycs, Cout, Geo{N, D, H, W}, nullptr, nullptr, s);
    }
    MI3D_TRY(conv3_direct_pack(w, Cin, Cout, wpf, wpd, s));
    return conv3_direct_fwd(in_dtype, out_dtype, x, xcs, Cin, wpf, bias, y, ycs, Cout, Geo{N, D, H, W}, s);
}
int mi3d_conv3_backward(int x_dtype, int dy_dtype, const void* x, int xcs, int Cin, const float* w, const void* dy,
                        int dycs, int Cout, void* dx, int dxcs, float* dW, float* db, int accumulate, int N, int D, int H,
                        int W, void* workspace, size_t workspace_bytes, void* stream) {
    MI3D_CHECK_ARG(x && w && dy && workspace, "mi3d_conv3_backward: null pointer");
    MI3D_CHECK_ARG(workspace_bytes >= mi3d_conv3_workspace_bytes(Cin, Cout, N, D, H, W), "mi3d_conv3_backward: workspace too small");
    Geo g{N, D, H, W};
    float* wpf = (float*)workspace;
    float* wpd = wpf + conv3_direct_pack_floats(Cin, Cout);
    float* slabs = wpd + conv3_direct_pack_floats(Cout, Cin);
    hipStream_t s = (hipStream_t)stream;
    // the same kernel choice as the whole-network plan (plan.hip block_backward), so the per-operator parity tests pin
    // the kernels the training step runs: both products of a layer in ONE fused launch where that exists
    if (dx && (dW || db) && x_dtype == MI3D_BF16 && use_mfma(dy_dtype, dy_dtype, Cout, Cin, dycs, dxcs) &&
        use_mfma(x_dtype, dy_dtype, Cin, Cout, xcs, dycs) && !mi3d_routes().api_unfused) {
        size_t wgf = conv3_mfma_wgrad_ws_floats(Cin, Cout, g);
        if (conv3_direct_wgrad_ws_floats(Cin, Cout, g) > wgf) wgf = conv3_direct_wgrad_ws_floats(Cin, Cout, g);
        float* skws = slabs + ((wgf + 63) & ~(size_t)63);
        if (conv3_mfma_bwd_fused_persist_ok(Cin, Cout, xcs, dycs, g)) {
            MI3D_TRY(conv3_mfma_pack(w, Cin, Cout, wpf, wpd, g, s));
            return conv3_mfma_bwd_fused_persist(x, xcs, Cin, dy, dycs, Cout, wpd, dx, dxcs, g, dW, db, accumulate, slabs, wgf, s);
        }
        if (conv3_mfma_bwd_fused_ok(Cin, Cout, xcs, dycs, dxcs, g)) {
            MI3D_TRY(conv3_mfma_pack(w, Cin, Cout, wpf, wpd, g, s));
            return conv3_mfma_bwd_fused(x, xcs, Cin, dy, dycs, Cout, wpd, dx, dxcs, g, dW, db, accumulate, slabs, wgf, skws, s);
        }
    }
    if (dx && use_mfma(dy_dtype, dy_dtype, Cout, Cin, dycs, dxcs)) {
        MI3D_TRY(conv3_mfma_pack(w, Cin, Cout, wpf, wpd, g, s));
        MI3D_TRY(conv3_mfma_fwd(dy, dycs, Cout, wpd, nullptr, dx, dxcs, Cin, g, nullptr, nullptr, s));      // skws NULL: single pass
    } else {
        MI3D_TRY(conv3_direct_pack(w, Cin, Cout, wpf, wpd, s));
        if (dx) MI3D_TRY(conv3_direct_fwd(dy_dtype, dy_dtype, dy, dycs, Cout, wpd, nullptr, dx, dxcs, Cin, g, s));
    }
    if (dW || db) {
        if (x_dtype == MI3D_F32 && dy_dtype == MI3D_BF16 && Cin == 1 && xcs == 1 && Cout % 16 == 0 && dycs % 8 == 0 &&
            !mi3d_routes().force_direct)
            MI3D_TRY(conv3_mfma_wgrad_c1((const float*)x, dy, dycs, Cout, g, dW, db, accumulate, slabs,
                                         conv3_mfma_wgrad_ws_floats(Cin, Cout, g), s));
        else if (use_mfma(x_dtype, dy_dtype, Cin, Cout, xcs, dycs) && dycs % 8 == 0)
            MI3D_TRY(conv3_mfma_wgrad(x, xcs, Cin, dy, dycs, Cout, g, dW, db, accumulate, slabs,
                                      conv3_mfma_wgrad_ws_floats(Cin, Cout, g), s));
        else
            MI3D_TRY(conv3_direct_wgrad(x_dtype, dy_dtype, x, xcs, Cin, dy, dycs, Cout, g, dW, db, accumulate, slabs,
                                        conv3_direct_wgrad_ws_floats(Cin, Cout, g), s));
    }
    return 0;
}

size_t mi3d_bn_workspace_bytes(int C) { return bn_ws_floats(C) * sizeof(float); }
int mi3d_bn_relu_drop_forward(int dtype, const void* y, int ycs, int C, int64_t M, int64_t V, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                              float momentum, float eps, int training, const float* drop, void* z, int zcs, float* stat,
                              void* workspace, void* stream) {
    MI3D_CHECK_ARG(y && gamma && beta && z && stat && workspace, "mi3d_bn_relu_drop_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    int small_rows = 0;
    if (training)
        MI3D_TRY(bn_train_stats(dtype, y, ycs, C, M, gamma, beta, running_mean, running_var, num_batches_tracked, momentum,
                                eps, stat, (float*)workspace, s, &small_rows));
    else {
        MI3D_CHECK_ARG(running_mean && running_var, "eval-mode BN needs running statistics");
        MI3D_TRY(bn_eval_stats(C, gamma, beta, running_mean, running_var, eps, stat, s));
    }
    BnSmall sm{(const float*)workspace, small_rows, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps};
    return bn_apply_relu_drop(dtype, y, ycs, C, M, V, stat, drop, z, zcs, s, small_rows > 0 ? &sm : nullptr);
}
int mi3d_bn_relu_drop_backward(int dtype, const void* dz, int dzcs, const void* y, int ycs, int C, int64_t M, int64_t V,
                               const float* stat, const float* drop, void* dy, int dycs, float* dgamma, float* dbeta,
                               int accumulate, void* workspace, void* stream) {
    MI3D_CHECK_ARG(dz && y && stat && dy && workspace, "mi3d_bn_relu_drop_backward: null pointer");
    return bn_bwd(dtype, dz, dzcs, y, ycs, C, M, V, stat, drop, dy, dycs, dgamma, dbeta, accumulate, (float*)workspace,
                  (hipStream_t)stream);
}
int mi3d_bn_relu_drop_pool_forward(int dtype, const void* y, int ycs, int C, int N, int D, int H, int W, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                   float momentum, float eps, const float* drop, void* z, int zcs, void* pooled, int pcs,
                                   float* stat, void* workspace, void* stream) {
    MI3D_CHECK_ARG(y && gamma && beta && z && pooled && stat && workspace, "mi3d_bn_relu_drop_pool_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    Geo g{N, D, H, W};
    int small_rows = 0;
    MI3D_TRY(bn_train_stats(dtype, y, ycs, C, g.M(), gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                            stat, (float*)workspace, s, &small_rows));
    BnSmall sm{(const float*)workspace, small_rows, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps};
    return bn_apply_relu_drop_pool(dtype, y, ycs, C, g, stat, drop, z, zcs, pooled, pcs, s, small_rows > 0 ? &sm : nullptr);
}
size_t mi3d_conv1_workspace_bytes(int Cin, int Cout) { return conv1_bwd_ws_floats(Cin, Cout) * sizeof(float); }
int mi3d_conv1_forward(int dtype, const void* z, int zcs, int Cin, const float* w, const float* bias, float* logits, int Cout,
                       int N, int64_t V, void* stream) {
    MI3D_CHECK_ARG(z && w && logits, "mi3d_conv1_forward: null pointer");
    MI3D_CHECK_ARG(dtype == MI3D_F32 || dtype == MI3D_BF16, "mi3d_conv1_forward: bad dtype %d", dtype);
    return conv1_fwd(dtype, z, zcs, Cin, w, bias, logits, Cout, N, V, (hipStream_t)stream);
}
int mi3d_head_loss_supported(int dtype, int Cin, int C, const mi3d_loss_cfg* cfg) {
    return cfg && head_loss_bwd_ok(dtype, nullptr, Cin, Cin, C, to_cfg(cfg), nullptr, Cin) ? 1 : 0;
}
int mi3d_head_loss_forward(const void* z, int zcs, int Cin, const float* w, const float* bias, const int64_t* labels,
                           const float* teacher, int N, int C, int D, int64_t V, const mi3d_loss_cfg* cfg, float* loss_out, float* coef,
                           float* metrics_out, void* loss_workspace, void* metrics_workspace, float* logits_opt, void* stream) {
    MI3D_CHECK_ARG(z && w && labels && cfg && loss_out && coef && loss_workspace, "mi3d_head_loss_forward: null pointer");
    return head_loss_fwd(z, zcs, Cin, w, bias, labels, teacher, N, C, V, to_cfg(cfg), loss_out, coef, loss_workspace, (hipStream_t)stream,
                         D, metrics_out, metrics_workspace, logits_opt);
}
int mi3d_head_loss_backward(const void* z, int zcs, int Cin, const float* w, const float* bias, const int64_t* labels,
                            const float* teacher, int N, int C, int64_t V, const mi3d_loss_cfg* cfg, const float* coef,
                            const float* grad_scale, void* dz, int dzcs, float* dW, float* db, int accumulate, void* workspace,
                            size_t workspace_bytes, void* stream) {
    MI3D_CHECK_ARG(z && w && labels && cfg && coef && dz && workspace, "mi3d_head_loss_backward: null pointer");
    MI3D_CHECK_ARG(workspace_bytes >= mi3d_conv1_workspace_bytes(Cin, C), "mi3d_head_loss_backward: workspace too small");
    return head_loss_bwd(z, zcs, Cin, w, bias, labels, teacher, C, to_cfg(cfg), coef, grad_scale, dz, dzcs, dW, db, accumulate,
                         (float*)workspace, N, V, (hipStream_t)stream);
}
int mi3d_conv1_backward(int dtype, const void* z, int zcs, int Cin, const float* w, const float* dlogits, int Cout, void* dz,
                        int dzcs, float* dW, float* db, int accumulate, int N, int64_t V, void* workspace,
                        size_t workspace_bytes, void* stream) {
    MI3D_CHECK_ARG(z && w && dlogits && dz && workspace, "mi3d_conv1_backward: null pointer");
    MI3D_CHECK_ARG(dtype == MI3D_F32 || dtype == MI3D_BF16, "mi3d_conv1_backward: bad dtype %d", dtype);
    MI3D_CHECK_ARG(workspace_bytes >= mi3d_conv1_workspace_bytes(Cin, Cout), "mi3d_conv1_backward: workspace too small");
    return conv1_bwd(dtype, z, zcs, Cin, w, dlogits, Cout, dz, dzcs, dW, db, accumulate, (float*)workspace, N, V,
                     (hipStream_t)stream);
}
int mi3d_maxpool2_forward(int dtype, const void* z, int zcs, int C, int N, int D, int H, int W, void* p, int pcs,
                          void* stream) {
    MI3D_CHECK_ARG(z && p, "mi3d_maxpool2_forward: null pointer");
    return maxpool2_fwd(dtype, z, zcs, C, Geo{N, D, H, W}, p, pcs, (hipStream_t)stream);
}
int mi3d_maxpool2_backward(int dtype, const void* dp, int dpcs, const void* z, int zcs, const void* dskip, int dskipcs,
                           void* dz, int dzcs, int C, int N, int D, int H, int W, void* stream) {
    MI3D_CHECK_ARG(dp && z && dz, "mi3d_maxpool2_backward: null pointer");
    return maxpool2_bwd(dtype, dp, dpcs, z, zcs, dskip, dskipcs, dz, dzcs, C, Geo{N, D, H, W}, (hipStream_t)stream);
}
size_t mi3d_upconv2_workspace_bytes(int Cin, int Cout, int N, int D, int H, int W) {
    Geo g{N, D, H, W};
    size_t wsf = upconv2_bwd_ws_floats(Cin, Cout, g);
    if (upconv2_mfma_supported(Cin, Cout, 8, 8) && upconv2_mfma_bwd_ws_floats(Cin, Cout, g) > wsf) wsf = upconv2_mfma_bwd_ws_floats(Cin, Cout, g);
    return (upconv2_pack_floats(Cin, Cout) + wsf) * sizeof(float);
}
int mi3d_upconv2_forward(int dtype, const void* x, int xcs, int Cin, const float* w, const float* bias, void* y, int ycs,
                         int Cout, int N, int D, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
    MI3D_CHECK_ARG(x && w && y && workspace, "mi3d_upconv2_forward: null pointer");
    MI3D_CHECK_ARG(workspace_bytes >= mi3d_upconv2_workspace_bytes(Cin, Cout, N, D, H, W), "mi3d_upconv2_forward: workspace too small");
    float* wf = (float*)workspace;
    float* wb = wf + (size_t)cdiv(Cout, 8) * Cin * 64;
    if (dtype == MI3D_BF16 && upconv2_mfma_supported(Cin, Cout, xcs, ycs) && !mi3d_routes().force_direct) {
        MI3D_TRY(upconv2_mfma_pack(w, Cin, Cout, workspace, (hipStream_t)stream));
        return upconv2_mfma_fwd(x, xcs, Cin, workspace, bias, y, ycs, Cout, Geo{N, D, H, W}, (hipStream_t)stream);
    }
    MI3D_TRY(upconv2_pack(w, Cin, Cout, wf, wb, (hipStream_t)stream));
    return upconv2_fwd(dtype, x, xcs, Cin, wf, bias, y, ycs, Cout, Geo{N, D, H, W}, (hipStream_t)stream);
}
int mi3d_upconv2_backward(int dtype, const void* x, int xcs, int Cin, const float* w, const void* gy, int gycs, int Cout,
                          void* dx, int dxcs, float* dW, float* db, int accumulate, int N, int D, int H, int W,
                          void* workspace, size_t workspace_bytes, void* stream) {
    MI3D_CHECK_ARG(x && w && gy && workspace, "mi3d_upconv2_backward: null pointer");
    MI3D_CHECK_ARG(workspace_bytes >= mi3d_upconv2_workspace_bytes(Cin, Cout, N, D, H, W), "mi3d_upconv2_backward: workspace too small");
    Geo g{N, D, H, W};
    float* wf = (float*)workspace;
    float* wb = wf + (size_t)cdiv(Cout, 8) * Cin * 64;
    float* slabs = (float*)workspace + upconv2_pack_floats(Cin, Cout);
    if (dtype == MI3D_BF16 && upconv2_mfma_supported(Cin, Cout, xcs, gycs) && (!dx || dxcs % 4 == 0) && !mi3d_routes().force_direct) {
        MI3D_TRY(upconv2_mfma_pack(w, Cin, Cout, workspace, (hipStream_t)stream));
        return upconv2_mfma_bwd(x, xcs, Cin, gy, gycs, Cout, workspace, dx, dxcs, dW, db, accumulate, slabs,
                                upconv2_mfma_bwd_ws_floats(Cin, Cout, g), g, (hipStream_t)stream);
    }
    MI3D_TRY(upconv2_pack(w, Cin, Cout, wf, wb, (hipStream_t)stream));
    return upconv2_bwd(dtype, x, xcs, Cin, gy, gycs, Cout, wb, dx, dxcs, dW, db, accumulate, slabs,
                       upconv2_bwd_ws_floats(Cin, Cout, g), g, (hipStream_t)stream);
}
int mi3d_ncdhw_to_ndhwc(int dtype, const float* src, void* dst, int dcs, int C, int N, int64_t V, void* stream) {
    MI3D_CHECK_ARG(src && dst, "mi3d_ncdhw_to_ndhwc: null pointer");
    return ncdhw_to_ndhwc(dtype, src, dst, dcs, C, N, V, (hipStream_t)stream);
}
int mi3d_ndhwc_to_ncdhw(int dtype, const void* src, int scs, float* dst, int C, int N, int64_t V, void* stream) {
    MI3D_CHECK_ARG(src && dst, "mi3d_ndhwc_to_ncdhw: null pointer");
    return ndhwc_to_ncdhw(dtype, src, scs, dst, C, N, V, (hipStream_t)stream);
}

int mi3d_debug_set_route(const char* name, int value) {
    MI3D_CHECK_ARG(name, "mi3d_debug_set_route: null name");
    for (int i = 0; i < N_ROUTES; i++)
        if (!strcmp(name, ROUTE_TABLE[i].name)) { routes_mut().*(ROUTE_TABLE[i].field) = value; return 0; }
    MI3D_CHECK_ARG(false, "mi3d_debug_set_route: unknown route '%s'", name);
}
int mi3d_debug_get_route(const char* name, int* value_out) {
    MI3D_CHECK_ARG(name && value_out, "mi3d_debug_get_route: null argument");
    for (int i = 0; i < N_ROUTES; i++)
        if (!strcmp(name, ROUTE_TABLE[i].name)) { *value_out = routes_mut().*(ROUTE_TABLE[i].field); return 0; }
    MI3D_CHECK_ARG(false, "mi3d_debug_get_route: unknown route '%s'", name);
}
int mi3d_debug_experiments(void) {
#ifdef MI3D_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
int mi3d_debug_route_count(void) { return N_ROUTES; }
const char* mi3d_debug_route_name(int i) { return (i >= 0 && i < N_ROUTES) ? ROUTE_TABLE[i].name : nullptr; }

int mi3d_event_create(void** event_out) {
    MI3D_CHECK_ARG(event_out, "mi3d_event_create: null output");
    hipEvent_t e;
    // no system-scope fence on record: by default hipEventRecord writes the caches back for the HOST's benefit, which costs the
    // (measured neutral on this runtime, 2.229 vs 2.237 ms on the exchange path).  These events only order streams of ONE device;
    // the producing kernel's own end-of-kernel release covers that.
    MI3D_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    *event_out = (void*)e;
    return 0;
}
int mi3d_stream_create(int priority_class, void** stream_out) {
    MI3D_CHECK_ARG(stream_out && priority_class >= -1 && priority_class <= 1, "mi3d_stream_create: bad arguments");
    int least = 0, greatest = 0;      // numerically: least = lowest priority (largest value), greatest = highest
    MI3D_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    int prio = priority_class < 0 ? greatest : priority_class > 0 ? least : (least + greatest) / 2;
    hipStream_t s;
    MI3D_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
    *stream_out = (void*)s;
    return 0;
}
int mi3d_stream_create_masked(int cus_per_xcd, int from_top, void** stream_out) {
    MI3D_CHECK_ARG(stream_out && cus_per_xcd >= 1 && cus_per_xcd <= 31, "mi3d_stream_create_masked: bad arguments");
    hipDeviceProp_t prop;
    int dev = 0;
    MI3D_HIP(hipGetDevice(&dev));
    MI3D_HIP(hipGetDeviceProperties(&prop, dev));
    MI3D_CHECK_ARG(prop.multiProcessorCount == 256, "mi3d_stream_create_masked: expects 8 XCDs x 32 CUs, device has %d CUs", prop.multiProcessorCount);
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int lo = from_top ? (32 - cus_per_xcd) * 8 : 0, hi = lo + cus_per_xcd * 8;
    for (int b = lo; b < hi; b++) mask[b >> 5] |= 1u << (b & 31);
    hipStream_t s;
    MI3D_HIP(hipExtStreamCreateWithCUMask(&s, 8, mask));
    *stream_out = (void*)s;
    return 0;
}
int mi3d_stream_destroy(void* stream) {
    if (stream) MI3D_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
int mi3d_timing_event_create(void** event_out) {
    MI3D_CHECK_ARG(event_out, "mi3d_timing_event_create: null output");
    hipEvent_t e;
    MI3D_HIP(hipEventCreate(&e));
    *event_out = (void*)e;
    return 0;
}
int mi3d_event_elapsed_ms(void* start_event, void* stop_event, float* ms_out) {
    MI3D_CHECK_ARG(start_event && stop_event && ms_out, "mi3d_event_elapsed_ms: null argument");
    MI3D_HIP(hipEventSynchronize((hipEvent_t)stop_event));
    MI3D_HIP(hipEventElapsedTime(ms_out, (hipEvent_t)start_event, (hipEvent_t)stop_event));
    return 0;
}
int mi3d_event_destroy(void* event) {
    if (event) MI3D_HIP(hipEventDestroy((hipEvent_t)event));
    return 0;
}

// ---- hipGraph helpers ---------------------------------------------------------------------------
int mi3d_graph_begin(void* stream) {
    MI3D_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return 0;
}
int mi3d_graph_end(void* stream, void** graph_exec_out) {
    MI3D_CHECK_ARG(graph_exec_out, "mi3d_graph_end: null output");
    hipGraph_t graph = nullptr;
    MI3D_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    MI3D_HIP(e);
    *graph_exec_out = (void*)exec;
    return 0;
}
int mi3d_graph_launch(void* graph_exec, void* stream) {
    MI3D_CHECK_ARG(graph_exec, "mi3d_graph_launch: null graph");
    MI3D_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return 0;
}
int mi3d_graph_destroy(void* graph_exec) {
    if (graph_exec) MI3D_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}

}  // extern "C"
