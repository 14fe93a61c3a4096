// ops.h — internal launcher API (C++), one function per kernel family.  Every launcher
//   * takes raw device pointers + explicit channel strides (see common.h for the layout),
//   * launches on the stream it is given and never synchronises, allocates or frees,
//   * returns 0, a negative argument error, or a positive hipError_t.
// The extern "C" boundary (include/mi3d.h, api.hip) and the whole-network plan (plan.hip) sit on top.
#pragma once
#include "common.h"
#include "slab_sum.h"

struct Geo {
    int N, D, H, W;
    int64_t V() const { return (int64_t)D * H * W; }
    int64_t M() const { return (int64_t)N * D * H * W; }
};

// ---- 3x3x3 convolution, direct (any channel count, any dtype; fp32 FMA) -------------- conv3_direct.hip
// Reference: nn.Conv3d(k=3,p=1) models/unet.py:11,15.  Weights arrive in torch layout (Cout,Cin,3,3,3) fp32.
size_t conv3_direct_pack_floats(int Cin, int Cout);          // size of ONE packed operand (fwd or dgrad)
int conv3_direct_pack(const float* w, int Cin, int Cout, float* wp_fwd, float* wp_dgrad, hipStream_t s,
                      const float* scale = nullptr);     // scale: per-output-channel factor of the forward operand
// y[v,co] = bias[co] + sum_{tap,ci} x[v+tap,ci] * wp ;  dgrad = same call with wp_dgrad, (Cin,Cout) swapped, bias NULL
int conv3_direct_fwd(int in_dtype, int out_dtype, const void* x, int xcs, int Cin, const float* wp,
                     const float* bias, void* y, int ycs, int Cout, Geo g, hipStream_t s, int relu = 0);
// dW[co,ci,tap] (+)= sum_v dy[v,co] x[v+tap,ci] ; db[co] (+)= sum_v dy[v,co].  Deterministic slab reduction.
size_t conv3_direct_wgrad_ws_floats(int Cin, int Cout, Geo g);
int conv3_direct_wgrad(int x_dtype, int dy_dtype, const void* x, int xcs, int Cin, const void* dy, int dycs,
                       int Cout, Geo g, float* dW, float* db, int accumulate, float* ws, size_t ws_floats,
                       hipStream_t s);

// ---- 3x3x3 convolution, MFMA implicit GEMM (bf16, Cin,Cout % 16 == 0) ------------------ conv3_mfma.hip
bool conv3_mfma_supported(int Cin, int Cout, int xcs, int ycs);
size_t conv3_mfma_pack_elems(int Cin, int Cout);             // bf16 elements of ONE packed operand
int conv3_mfma_pack(const float* w, int Cin, int Cout, void* wp_fwd, void* wp_dgrad, Geo g, hipStream_t s);
// All weight packs of a network in ONE launch (every kernel node of the step graph costs ~4.5 us of dispatch floor):
// jobs are appended on the host, the kernel finds its job from the block index.
// scale (conv3 only, may be NULL): per-output-channel factor folded into the FORWARD image (inference: BatchNorm scale)
struct PackJob { const float* w; void* a; void* b; int Cin, Cout, kind, mode_f, mode_d, blk0; const float* scale; };   // kind 0 conv3, 1 upconv
constexpr int MAX_PACK_JOBS = 32;
struct PackJobs { int n, nblocks; PackJob j[MAX_PACK_JOBS]; int* zero = nullptr; int nzero = 0; };      // zero: ints block 0 clears (split-K ticket counters)
int pack_all_add_conv3(PackJobs& J, const float* w, int Cin, int Cout, void* wp_fwd, void* wp_dgrad, Geo g,
                       const float* scale = nullptr);
int pack_all_add_upconv(PackJobs& J, const float* w, int Cin, int Cout, void* wp);
int pack_all_launch(const PackJobs& J, hipStream_t s);
int conv3_mfma_stat_blocks(int Cin, int Cout, Geo g);                           // partials written when `part` != NULL
// dgrad = same call with the dgrad pack and (Cin,Cout) swapped, bias NULL, part NULL
size_t conv3_mfma_splitk_floats(int Cin, int Cout, Geo g);    // K-split scratch for deep (small-M) layers, 0 = none
bool conv3_mfma_fuses_stats(int Cin, int Cout, Geo g);        // false -> caller runs bn_train_stats afterwards
// A tensor whose channels live in TWO planes of the same stride ("planar halves" of a skip/up concat buffer):
// 16-channel block b >= split is found `delta` elements further than the single-plane address would say.
// Default = ordinary single-plane tensor.  Only the persistent full-resolution kernels and wgrad honour it.
struct Halves { int split = 1 << 30; int64_t delta = 0; bool on() const { return delta != 0; } };
bool conv3_mfma_halves_ok(int Cin, int Cout, Geo g);          // forward (Cin,Cout) launch can take Halves x / y
// "Apply on load" (round 4; deep levels, conv3_mfma8_kernel): the conv's input tensor is a BatchNorm output that is computed in
// the staging pass from the raw tensor(s) instead of being written by its own launch.  mode 1: x = raw conv output y0 of the
// layer in front, input = relu(a*x + b) * drop (forward conv1 of a block); mode 2: x = dz, y2 = the layer's raw output,
// input = BatchNorm/ReLU/Dropout3d backward of dz (the input-gradient conv).  rows = the <= 128 partial rows
// [nrows][2][C] the statistics (mode 1) / reduction (mode 2) kernel left; side = where the transformed tensor is written
// as a by-product (z1 resp. dy, for the weight-gradient kernels), NULL = nowhere.
struct XfArgs {
    int mode = 0;
    const bf16* y2 = nullptr; int y2cs = 0;
    float* stat = nullptr;                       // [4][C]: written (mode 1) / read (mode 2)
    const float* rows = nullptr; int nrows = 0;
    int64_t M = 0;                               // elements per channel
    int C = 0;                                   // channels of the transformed tensor (= Cin of the conv)
    const float* gamma = nullptr; const float* beta = nullptr; float* rmean = nullptr; float* rvar = nullptr; int64_t* nbt = nullptr;
    float momentum = 0.f, eps = 0.f;
    float* dgamma = nullptr; float* dbeta = nullptr; int accumulate = 0;
    const float* drop = nullptr;                 // [N][C] Dropout3d scales or NULL
    bf16* side = nullptr; int side_cs = 0;
    // split-K ticket (round 4, conv3_mfma8_kernel<..., TK>): per (tile, output group) arrival counters (zero before the launch, left
    // zero by it) and the BatchNorm partial rows [tiles][2][Cout] the finishing workgroups write
    float* tk_rows = nullptr; int* tk_count = nullptr;
};
// The split-K forward launch of this layer can finish itself: the LAST of the ks workgroups of an (output tile, channel group) to
// arrive (one atomic ticket per workgroup on a counter only those ks workgroups touch) sums the ks fp32 partials in k order, adds
// the bias, stores bf16 y and writes the tile's BatchNorm partial row -- the bn_stats_splitk launch of the layer disappears and
// the consumer finishes <= 108 tile rows instead of 128 block rows.  Same bits in y as the separate finishing pass.
bool conv3_mfma_ticket_ok(int Cin, int Cout, Geo g);
constexpr int CONV3_TK_COUNTERS = 4096;         // ints of counter space a plan reserves
bool conv3_mfma_xform_ok(int Cin, int Cout, Geo g);           // the launch conv3_mfma_fwd(Cin, Cout, g) would make can take XfArgs
int conv3_mfma_fwd(const void* x, int xcs, int Cin, const void* wp, const float* bias, void* y, int ycs, int Cout,
                   Geo g, float* part, float* skws, hipStream_t s, Halves xh = Halves(), Halves yh = Halves(),
                   int* ks_deferred = nullptr, int relu = 0, int ks_target = 0, const XfArgs* xf = nullptr,
                   float* tk_rows = nullptr, int* tk_count = nullptr);
// ks_target > 0: split-K workgroup target of this launch (0 = the forward default).  The input-gradient convs of the backward
// use conv3_bwd_ks_target() in the fused launch AND when they run stand-alone, so both routes produce the same bits
int conv3_bwd_ks_target();
// relu != 0: y = max(0, conv + bias) -- the inference path, where BatchNorm is folded into (weights, bias)
// ks_deferred != NULL: a split-K launch leaves its fp32 partials in skws ([ks][M][Cout]) WITHOUT the finishing pass and
// reports ks there (0 = y was written as usual); the caller finishes (bn_train_stats_splitk, fused with the statistics)

size_t conv3_mfma_wgrad_ws_floats(int Cin, int Cout, Geo g);
int conv3_mfma_wgrad(const void* x, int xcs, int Cin, const void* dy, int dycs, int Cout, Geo g, float* dW, float* db,
                     int accumulate, float* ws, size_t ws_floats, hipStream_t s, Halves xh = Halves(), SlabJob* pend = nullptr,
                     int wg_target = 0);
// wg_target > 0: workgroup target of the launch (0 = two per CU).  conv3_mfma_bwd_wg_target = the target the layer's FUSED
// backward launch uses for its weight-gradient half (0: the layer has no fused launch): a stand-alone weight gradient launched
// with it cuts the tiles into the same slabs, i.e. sums in the same order and produces the same bits as the fused route
int conv3_mfma_bwd_wg_target(int Cin, int Cout, int xcs, int dycs, int dxcs, Geo g);
bool conv3_mfma_big_geo(Geo g);      // 16-wide tiles (levels 0-1 of a 96^3 net) vs the 8-wide deep-level tiling
// pend != NULL (here and below): the final slab sum is NOT launched; its job is returned for the caller to attach to
// the next kernel on the chain (bn_bwd) or to run with slab_job_launch

// deep levels: weight gradient and input-gradient conv of one layer in ONE launch (independent, latency-bound each)
bool conv3_mfma_bwd_fused_ok(int Cin, int Cout, int xcs, int dycs, int dxcs, Geo g);
int conv3_mfma_bwd_fused(const void* x, int xcs, int Cin, const void* dy, int dycs, int Cout, const void* wp_dgrad, void* dx,
                         int dxcs, Geo g, float* dW, float* db, int accumulate, float* wgws, size_t wgws_floats, float* skws,
                         hipStream_t s, SlabJob* pend = nullptr, int* ks_deferred = nullptr);
// ks_deferred != NULL: a split-K input gradient is LEFT as fp32 partials in skws ([ks][M][Cin], *ks_deferred = ks, dx not
// written) for bn_bwd(..., skp, ks) of the layer below to finish inside its reduction; 0 = dx was written as usual
bool conv3_mfma_bwd_fused_persist_ok(int Cin, int Cout, int xcs, int dycs, Geo g);
int conv3_mfma_bwd_fused_persist(const void* x, int xcs, int Cin, const void* dy, int dycs, int Cout, const void* wp_dgrad, void* dx,
                                 int dxcs, Geo g, float* dW, float* db, int accumulate, float* wgws, size_t wgws_floats,
                                 hipStream_t s, Halves xh = Halves(), Halves dxh = Halves(), SlabJob* pend = nullptr);
// first layer (Cin = 1, fp32 input) forward on the matrix cores (K = taps); optional BN partial sums like conv3_mfma_fwd
int conv3_c1_fwd_stat_blocks(Geo g);
int conv3_c1_fwd_mfma(const float* x, const float* w, const float* bias, void* y, int ycs, int Cout, Geo g, float* part,
                      hipStream_t s, const float* wscale = nullptr, int relu = 0);
int conv3_mfma_wgrad_c1(const float* x, const void* dy, int dycs, int Cout, Geo g, float* dW, float* db, int accumulate,
                        float* ws, size_t ws_floats, hipStream_t s, SlabJob* pend = nullptr);

// ---- BatchNorm3d + ReLU + Dropout3d ---------------------------------------------------------- bn.hip
// Reference: nn.BatchNorm3d / nn.ReLU(inplace) / nn.Dropout3d  models/unet.py:12-14,16-18.
// stat buffer layout: float[4][C] = {mean, invstd, a = gamma*invstd, b = beta - mean*a}
size_t bn_ws_floats(int C);
// small_rows != NULL and the tensor is small (deep levels): NO finalize launch -- *small_rows = number of partial rows left
// in ws (<= 128) and the caller hands them to bn_apply_relu_drop (BnSmall), which finishes them in its prologue;
// otherwise *small_rows = 0 and stat / the running statistics are final on return as before
int bn_train_stats(int dtype, const void* y, int ycs, int C, int64_t M, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                   float eps, float* stat, float* ws, hipStream_t s, int* small_rows = nullptr);
bool bn_small_ok(int C, int64_t M, int rows);       // a producer's `rows` partial rows can take the consumer-prologue route
// round 4: the `rows` partial rows of a conv epilogue can be handed to bn_apply_relu_drop(_pool) as a BnSmall (no finalize launch):
// <= 128 rows by the thin kernels' prologue, up to 1024 rows by the wide (1024-thread) kernels; needs C a power of two, 8..256,
// 16-byte aligned channels-last rows, and for the pooled pass the two-threads-per-window shape
bool bn_rows_route_ok(int C, int64_t M, int rows);
struct BnSmall {
    const float* part; int nrows;
    const float* gamma; const float* beta; float* running_mean; float* running_var; int64_t* num_batches_tracked;
    float momentum, eps;
};
// same finalize, fed by conv-epilogue partials part[nblk][2][C]
int bn_train_finalize(const float* part, int nblk, int C, int64_t M, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                      float eps, float* stat, hipStream_t s);
int bn_eval_stats(int C, const float* gamma, const float* beta, const float* running_mean,
                  const float* running_var, float eps, float* stat, hipStream_t s);
// Inference: eval-mode BatchNorm folded into the preceding conv, for ALL layers of a network in ONE launch:
//   scale[c] = gamma/sqrt(running_var + eps)   (multiplied into the filter at pack time)
//   fbias[c] = (conv_bias - running_mean)*scale + beta
struct BnFoldJob { const float* gamma; const float* beta; const float* rm; const float* rv; const float* conv_bias;
                   float* scale; float* fbias; int C; };
constexpr int MAX_FOLD_JOBS = 2 * (2 * 6 + 1);
struct BnFoldJobs { int n; float eps; BnFoldJob j[MAX_FOLD_JOBS]; };
int bn_fold_all(const BnFoldJobs& J, hipStream_t s);
// DEFERRED running-statistics update: a training forward called with momentum < 0 does not touch running_mean / running_var /
// num_batches_tracked; `running_mean` then points at double[2C] where the finalize step publishes (batch mean, unbiased batch
// variance) exactly as it computed them.  bn_deferred_apply performs, later and in the order the caller chooses, the very
// update the forward would have made -- same doubles, same expression, bit-identical buffers.  (Two forwards of one model
// on two streams, train_dann.py:268-272: their updates of the shared buffers must stay in source-then-target order.)
struct BnDeferJob { float* rm; float* rv; int64_t* nbt; const double* side; int C; };
struct BnDeferJobs { int n; float momentum; BnDeferJob j[MAX_FOLD_JOBS]; };
int bn_deferred_apply(const BnDeferJobs& J, hipStream_t s);
// z = drop[n,c] * relu(a*y + b)      (drop == NULL -> 1)
// small != NULL: batch statistics from small->part (prologue); stat[4][C] is then WRITTEN (kept for backward) and the
// running statistics / num_batches_tracked are updated here
int bn_apply_relu_drop(int dtype, const void* y, int ycs, int C, int64_t M, int64_t V, float* stat,
                       const float* drop, void* z, int zcs, hipStream_t s, const BnSmall* small = nullptr);
// the same pass fused with MaxPool3d(2,2): writes z AND pooled = max over each 2x2x2 window of z (even D, H, W only)
int bn_apply_relu_drop_pool(int dtype, const void* y, int ycs, int C, Geo g, float* stat, const float* drop, void* z, int zcs,
                            void* pooled, int pcs, hipStream_t s, const BnSmall* small = nullptr);
// dy = gamma*invstd*(dyh - mean(dyh) - xhat*mean(dyh*xhat)), dyh = dz*drop*[a*y+b > 0]; dgamma, dbeta (+)=
int bn_bwd(int dtype, const void* dz, int dzcs, const void* y, int ycs, int C, int64_t M, int64_t V,
           const float* stat, const float* drop, void* dy, int dycs, float* dgamma, float* dbeta,
           int accumulate, float* ws, hipStream_t s, const SlabJob* extra = nullptr, const float* skp = nullptr, int ks = 0,
           const SlabJob* extra2 = nullptr, int* reduce_only_rows = nullptr);
// reduce_only_rows != NULL and the tensor takes the "small" route: only the reduction is launched; *reduce_only_rows = the number
// of partial rows it leaves in ws ([rows][2][C]) for a consumer that applies on load (XfArgs mode 2), else 0 and nothing changes
bool bn_small_route(int C, int64_t M);       // the statistics / reduction of a (C, M) tensor leave <= 128 rows for the consumer's prologue
// skp != NULL: dz is still the ks fp32 split-K partials [ks][M][C] of the conv that produced it; the reduction sums and
// rounds them and WRITES dz (the split-K finishing launch of that conv is skipped by the caller)
// extra: a pending slab sum that rides in the reduction kernel's launch (extra blocks)
int slab_job_launch(const SlabJob& q, hipStream_t s);

// split-K conv finish (y = bf16(bias + sum_k skp[k][M][C])) fused with the batch statistics of the stored values
int bn_train_stats_splitk(const float* skp, int ks, const float* bias, void* y, int ycs, int C, int64_t M, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                          float momentum, float eps, float* stat, float* ws, hipStream_t s, int* small_rows = nullptr);

// ---- MaxPool3d(2,2) ------------------------------------------------------------------------ pool.hip
// Reference: models/unet.py:40,71.  g = INPUT geometry; odd sides floor like nn.MaxPool3d (last slice in no window).
int maxpool2_fwd(int dtype, const void* z, int zcs, int C, Geo g, void* p, int pcs, hipStream_t s);
// dz = dskip (or 0) + route(dp) to the first max in (d,h,w) scan order
int maxpool2_bwd(int dtype, const void* dp, int dpcs, const void* z, int zcs, const void* dskip, int dskipcs,
                 void* dz, int dzcs, int C, Geo g, hipStream_t s, const float* skp = nullptr, int ks = 0);
// skp / ks: dp has not been written -- it is still the ks fp32 split-K partials [ks][M/8][C] of the conv that produces it

// F.interpolate(x, size=...) nearest, models/unet.py:81-83 (volume sides not divisible by 2^levels).  gi = input
// geometry, go = output geometry; backward is the gather-form adjoint (deterministic)
int nearest_resize_fwd(int dtype, const void* x, int xcs, int C, Geo gi, void* y, int ycs, Geo go, hipStream_t s);
int nearest_resize_bwd(int dtype, const void* gy, int gycs, int C, Geo go, void* gx, int gxcs, Geo gi, hipStream_t s);

// ---- ConvTranspose3d(k=2,s=2) ------------------------------------------------------------ upconv.hip
// Reference: models/unet.py:56-58,79.  Weight torch layout (Cin,Cout,2,2,2).  g = INPUT geometry.
size_t upconv2_pack_floats(int Cin, int Cout);
int upconv2_pack(const float* w, int Cin, int Cout, float* wp_fwd, float* wp_bwd, hipStream_t s);
int upconv2_fwd(int dtype, const void* x, int xcs, int Cin, const float* wp_fwd, const float* bias, void* y,
                int ycs, int Cout, Geo g, hipStream_t s);
size_t upconv2_bwd_ws_floats(int Cin, int Cout, Geo g);
int upconv2_bwd(int dtype, const void* x, int xcs, int Cin, const void* gy, int gycs, int Cout,
                const float* wp_bwd, void* dx, int dxcs, float* dW, float* db, int accumulate, float* ws,
                size_t ws_floats, Geo g, hipStream_t s);

// MFMA versions (bf16, Cin % 32 == 0, Cout % 16 == 0)                                   upconv_mfma.hip
bool upconv2_mfma_supported(int Cin, int Cout, int xcs, int ycs);
size_t upconv2_mfma_pack_elems(int Cin, int Cout);           // bf16 elements (fwd + bwd images)
int upconv2_mfma_pack(const float* w, int Cin, int Cout, void* wp, hipStream_t s);
int upconv2_mfma_fwd(const void* x, int xcs, int Cin, const void* wp, const float* bias, void* y, int ycs, int Cout,
                     Geo g, hipStream_t s);
size_t upconv2_mfma_bwd_ws_floats(int Cin, int Cout, Geo g);
int upconv2_mfma_bwd(const void* x, int xcs, int Cin, const void* gy, int gycs, int Cout, const void* wp, void* dx,
                     int dxcs, float* dW, float* db, int accumulate, float* ws, size_t ws_floats, Geo g, hipStream_t s, SlabJob* pend = nullptr);

// ---- final 1x1x1 conv, losses, metrics ------------------------------------------------ head_loss.hip
// Reference: nn.Conv3d(16,4,1) models/unet.py:62,87 ; utils/metrics.py:14-40,65-129,137-190.
int conv1_fwd(int dtype, const void* z, int zcs, int Cin, const float* w, const float* bias, float* logits,
              int Cout, int N, int64_t V, hipStream_t s);
size_t conv1_bwd_ws_floats(int Cin, int Cout);
int conv1_bwd(int dtype, const void* z, int zcs, int Cin, const float* w, const float* dlogits, int Cout,
              void* dz, int dzcs, float* dW, float* db, int accumulate, float* ws, int N, int64_t V,
              hipStream_t s, SlabJob* pend = nullptr);

struct LossCfg {
    float w_ce;       // weight of mean cross-entropy
    int region_kind;  // 0 none, 1 soft-Dice (combined_loss), 2 Tversky
    float w_reg;      // weight of mean_{c>=1} region term
    float alpha, beta, eps;
    float w_kd;       // weight of T^2 * mean_{n,c,v} KL(teacher || student)   (0 = no distillation)
    float temp;
};
enum { MI3D_MAX_CLASSES = 8 };
size_t seg_loss_ws_bytes(int C);
// loss_out: device float[1]; coef: device float[2*MI3D_MAX_CLASSES + 4] consumed by seg_loss_bwd
int seg_loss_fwd(const float* logits, const int64_t* labels, const float* teacher, int N, int C, int64_t V,
                 LossCfg cfg, float* loss_out, float* coef, void* ws, hipStream_t s,
                 int D = 0, float* metrics_out = nullptr, void* metrics_ws = nullptr);
int seg_loss_bwd(const float* logits, const int64_t* labels, const float* teacher, int N, int C, int64_t V,
                 LossCfg cfg, const float* coef, const float* grad_out, float* dlogits, hipStream_t s);
// head + loss of the training step in one pass each way (logits / dlogits never written); `*_ok` says whether the shape has
// the fused kernels (bf16, Cin % 16 == 0, <= 4 classes; backward: Cin == 16); teacher: (N,C,V) float logits, needed iff cfg.w_kd != 0
bool head_loss_ok(int dtype, const void* z, int zcs, int Cin, int C, LossCfg cfg);
bool head_loss_bwd_ok(int dtype, const void* z, int zcs, int Cin, int C, LossCfg cfg, const void* dz, int dzcs);
int head_loss_fwd(const void* z, int zcs, int Cin, const float* w, const float* bias, const int64_t* labels, const float* teacher,
                  int N, int C, int64_t V, LossCfg cfg, float* loss_out, float* coef, void* ws, hipStream_t s, int D, float* metrics_out,
                  void* metrics_ws, float* logits_opt);
int head_loss_bwd(const void* z, int zcs, int Cin, const float* w, const float* bias, const int64_t* labels, const float* teacher, int C,
                  LossCfg cfg, const float* coef, const float* grad_out, void* dz, int dzcs, float* dW, float* db, int accumulate,
                  float* ws, int N, int64_t V, hipStream_t s, SlabJob* pend = nullptr);
size_t seg_metrics_ws_bytes(int C);
// out: device float[3] = {iou, dice, acc}; Q1 loop bound = first spatial dim D (utils/metrics.py:74,101)
int seg_metrics(const float* logits, const int64_t* labels, int N, int C, int D, int64_t V, float* out, void* ws,
                hipStream_t s, int64_t* counts_out = nullptr);

// ---- misc ---------------------------------------------------------------------------------- misc.hip
int ncdhw_to_ndhwc(int dtype, const float* src, void* dst, int dcs, int C, int N, int64_t V, hipStream_t s);
int ndhwc_to_ncdhw(int dtype, const void* src, int scs, float* dst, int C, int N, int64_t V, hipStream_t s);
// torch.mean(x, dim=[2,3,4]) models/unet_dann.py:79 ; backward adds scale*g[n,c]/V to every voxel
int gap_fwd(int dtype, const void* z, int zcs, int C, int N, int64_t V, float* out, hipStream_t s);
int gap_bwd(int dtype, const float* g, float scale, void* dz, int dzcs, int C, int N, int64_t V,
            int accumulate, hipStream_t s);
// nn.Linear (+ optional ReLU and dropout scale) train_dann.py:38-46 ; fp32, small M
int linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int K, int Nout, int relu,
               const float* drop, hipStream_t s);
int linear_bwd(const float* x, const float* w, const float* y, const float* gy, int M, int K, int Nout,
               int relu, const float* drop, float* gx, float* gw, float* gb, int accumulate, float gx_scale,
               float* ws, hipStream_t s);
int softmax_ce_rows(const float* logits, const int64_t* labels, int M, int C, float* loss, float* dlogits,
                    float scale, hipStream_t s);
// fused flat AdamW (torch.optim.AdamW semantics, train_unet.py:378); step_dev: device int64 step counter (incremented)
int adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
               float eps, float wd, float grad_scale, int64_t* step_dev, hipStream_t s, int increment = 1);
// Dropout3d channel masks: out[i] = (u_i >= p) ? 1/(1-p) : 0, counter-based RNG; state_dev = {seed, counter}
int dropout_scales(float* out, int64_t n, float p, uint64_t* state_dev, hipStream_t s);
int fill_f32(float* p, int64_t n, float v, hipStream_t s);
int flag_set(int64_t* flag, int64_t value, hipStream_t s);                      // flag[0] = value behind everything enqueued on s
int flag_wait(int64_t* flag, int64_t value, int64_t timeout_us, hipStream_t s);   // s continues when flag[0] >= value (flag[1] = value on time-out)
int occupy_cus(int wgs, int usec, float* buf, int64_t n, hipStream_t s);     // collective stand-in (bench.py --emulate-comm)
int scale_f32(const float* x, float* y, int64_t n, float a, const float* a_dev, hipStream_t s);   // y = a * (*a_dev|1) * x
int scale_add_f32(float* dst, const float* src, int64_t n, float a, float b, hipStream_t s);  // dst = a*dst + b*src
