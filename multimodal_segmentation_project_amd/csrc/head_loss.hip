// head_loss.hip — final 1x1x1 conv (channels-last -> NCDHW fp32 logits), the segmentation loss family and
// the per-step metrics.  All HBM-bound single-pass kernels with two-stage deterministic reductions.
//
// Reference: nn.Conv3d(features[0], out_channels, 1)  models/unet.py:62,87
//            combined_loss / tversky_loss / combined_ce_tversky_loss / distillation_loss  utils/metrics.py:14-40,137-190
//            'dice' variant train_unet.py:186-198
//            calculate_iou / calculate_dice / calculate_accuracy  utils/metrics.py:65-129 (incl. quirk Q1)
//
// Loss family (one parametrisation, see LossCfg):
//   L = w_ce*CE + w_reg*mean_{c>=1} R_c + w_kd*T^2*mean_{n,c,v} KL(p_t^T || p_s^T)
//   dL/dz_k = w_ce/M (p_k - [t=k]) + p_k (g_k - sum_c g_c p_c) + w_kd*T/(M*C) (ps_k - pt_k),
//   g_c = w_reg/(C-1) * (A_c [t=c] + B_c),  A_c, B_c = dR_c/dI_c-ish coefficients computed once from the sums.
#include "ops.h"

int slab_reduce(const float* slabs, int nslab, int64_t slab_sz, int64_t nW, float* dW, float* db, int accumulate,
                hipStream_t s);

namespace {
constexpr int BLK = 256;
constexpr int MAXC = MI3D_MAX_CLASSES;
constexpr int CINB = 16;    // input-channel block of the 1x1x1 conv kernels
constexpr int LOSS_MAXBLK = 1024;

// ------------------------------------------------------------------------------------------ conv 1x1x1
template <typename T, bool VEC>
__device__ __forceinline__ void load_cin_block(const T* zp, int nci, float (&zv)[CINB]) {
    if constexpr (VEC) {
        float a[8], b[8];
        ld8<T>(zp, a);
        ld8<T>(zp + 8, b);
#pragma unroll
        for (int i = 0; i < 8; i++) { zv[i] = a[i]; zv[8 + i] = b[i]; }
    } else {
#pragma unroll
        for (int i = 0; i < CINB; i++) zv[i] = i < nci ? to_f<T>(zp[i]) : 0.f;
    }
}

// logits[n][co][v] = bias[co] + sum_ci z[n,v,ci] * w[co][ci]       (Cout <= MAXC per blockIdx.y group)
template <typename T, bool VEC>
__global__ __launch_bounds__(BLK) void conv1_fwd_kernel(const T* __restrict__ z, int zcs, int Cin, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ logits, int Cout,
                                                        int N, int64_t V) {
    int64_t M = (int64_t)N * V;
    int co0 = blockIdx.y * MAXC;
    int nco = min(MAXC, Cout - co0);
    for (int64_t m = (int64_t)blockIdx.x * BLK + threadIdx.x; m < M; m += (int64_t)gridDim.x * BLK) {
        float acc[MAXC];
#pragma unroll
        for (int j = 0; j < MAXC; j++) acc[j] = (bias && j < nco) ? bias[co0 + j] : 0.f;
        for (int c0 = 0; c0 < Cin; c0 += CINB) {
            float zv[CINB];
            int nci = min(CINB, Cin - c0);
            load_cin_block<T, VEC>(z + m * zcs + c0, nci, zv);
#pragma unroll
            for (int j = 0; j < MAXC; j++) {
                if (j < nco) {
#pragma unroll
                    for (int i = 0; i < CINB; i++)
                        if (VEC || i < nci) acc[j] = fmaf(zv[i], w[(int64_t)(co0 + j) * Cin + c0 + i], acc[j]);
                }
            }
        }
        int64_t n = m / V, v = m - n * V;
#pragma unroll
        for (int j = 0; j < MAXC; j++)
            if (j < nco) logits[((int64_t)n * Cout + co0 + j) * V + v] = acc[j];
    }
}

// per block (blockIdx.y = 16-channel input block): dz[v][ci] = sum_co dl[co][v] w[co][ci];
// slab[blockIdx.x] = { dW[co][ci] partial, db[co] partial }
template <typename T, bool VEC, int NCO>
__global__ __launch_bounds__(BLK) void conv1_bwd_kernel(const T* __restrict__ z, int zcs, int Cin, const float* __restrict__ w,
                                                        const float* __restrict__ dl, int Cout, T* __restrict__ dz, int dzcs,
                                                        int N, int64_t V, float* __restrict__ slabs) {
    __shared__ float red[4][NCO * CINB + NCO];
    int64_t M = (int64_t)N * V;
    int c0 = blockIdx.y * CINB;
    int nci = min(CINB, Cin - c0);
    float aw[NCO][CINB], ab[NCO];
#pragma unroll
    for (int j = 0; j < NCO; j++) {
        ab[j] = 0.f;
#pragma unroll
        for (int i = 0; i < CINB; i++) aw[j][i] = 0.f;
    }
    for (int64_t m = (int64_t)blockIdx.x * BLK + threadIdx.x; m < M; m += (int64_t)gridDim.x * BLK) {
        int64_t n = m / V, v = m - n * V;
        float g[NCO], zv[CINB], o[CINB];
#pragma unroll
        for (int j = 0; j < NCO; j++) g[j] = j < Cout ? dl[((int64_t)n * Cout + j) * V + v] : 0.f;
        load_cin_block<T, VEC>(z + m * zcs + c0, nci, zv);
#pragma unroll
        for (int i = 0; i < CINB; i++) o[i] = 0.f;
#pragma unroll
        for (int j = 0; j < NCO; j++) {
            if (j < Cout) {
                ab[j] += g[j];
#pragma unroll
                for (int i = 0; i < CINB; i++) {
                    aw[j][i] = fmaf(g[j], zv[i], aw[j][i]);
                    if (VEC || i < nci) o[i] = fmaf(g[j], w[(int64_t)j * Cin + c0 + i], o[i]);
                }
            }
        }
        if (dz) {
            if constexpr (VEC) {
                float lo[8], hi[8];
#pragma unroll
                for (int i = 0; i < 8; i++) { lo[i] = o[i]; hi[i] = o[8 + i]; }
                st8<T>(dz + m * dzcs + c0, lo);
                st8<T>(dz + m * dzcs + c0 + 8, hi);
            } else {
#pragma unroll
                for (int i = 0; i < CINB; i++) if (i < nci) dz[m * dzcs + c0 + i] = from_f<T>(o[i]);
            }
        }
    }
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < NCO; j++) {
#pragma unroll
        for (int i = 0; i < CINB; i++) {
            float s = wave_sum(aw[j][i]);
            if (lane == 0) red[wave][j * CINB + i] = s;
        }
        float sb = wave_sum(ab[j]);
        if (lane == 0) red[wave][NCO * CINB + j] = sb;
    }
    __syncthreads();
    int64_t nW = (int64_t)Cout * Cin;
    float* slab = slabs + (int64_t)blockIdx.x * (nW + Cout);
    for (int idx = threadIdx.x; idx < NCO * CINB + NCO; idx += BLK) {
        float s = red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx];
        if (idx < NCO * CINB) {
            int j = idx / CINB, i = idx - j * CINB;
            if (j < Cout && i < nci) slab[(int64_t)j * Cin + c0 + i] = s;
        } else {
            int j = idx - NCO * CINB;
            if (j < Cout && blockIdx.y == 0) slab[nW + j] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------- seg loss
// per-voxel softmax helpers (C <= MAXC, fully unrolled with predicates)
__device__ __forceinline__ void softmax_c(const float (&z)[MAXC], int C, float inv_t, float (&p)[MAXC], float& lse) {
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; c++) if (c < C) mx = fmaxf(mx, z[c] * inv_t);
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; c++) { p[c] = c < C ? __expf(z[c] * inv_t - mx) : 0.f; se += p[c]; }
    float r = 1.f / se;
#pragma unroll
    for (int c = 0; c < MAXC; c++) p[c] *= r;
    lse = mx + __logf(se);
}

constexpr int NQ = 2 + 3 * MAXC;   // ce, kl, I[c], P[c], T[c]

__global__ __launch_bounds__(BLK) void seg_loss_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                           const float* __restrict__ teacher, int N, int C, int64_t V,
                                                           float inv_t, double* __restrict__ part) {
    __shared__ float red[4][NQ];
    int64_t M = (int64_t)N * V;
    float q[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) q[i] = 0.f;
    for (int64_t m = (int64_t)blockIdx.x * BLK + threadIdx.x; m < M; m += (int64_t)gridDim.x * BLK) {
        int64_t n = m / V, v = m - n * V;
        float z[MAXC], p[MAXC], lse;
#pragma unroll
        for (int c = 0; c < MAXC; c++) z[c] = c < C ? logits[((int64_t)n * C + c) * V + v] : 0.f;
        int t = (int)labels[m];
        softmax_c(z, C, 1.f, p, lse);
        float zt = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; c++) {
            if (c < C) {
                bool is = (c == t);
                zt = is ? z[c] : zt;
                q[2 + c] += is ? p[c] : 0.f;
                q[2 + MAXC + c] += p[c];
                q[2 + 2 * MAXC + c] += is ? 1.f : 0.f;
            }
        }
        q[0] += lse - zt;
        if (teacher) {
            float zt_[MAXC], ps[MAXC], pt[MAXC], ls, lt;
#pragma unroll
            for (int c = 0; c < MAXC; c++) zt_[c] = c < C ? teacher[((int64_t)n * C + c) * V + v] : 0.f;
            softmax_c(z, C, inv_t, ps, ls);
            softmax_c(zt_, C, inv_t, pt, lt);
            float kl = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; c++)
                if (c < C && pt[c] > 0.f) kl += pt[c] * ((zt_[c] * inv_t - lt) - (z[c] * inv_t - ls));
            q[1] += kl;
        }
    }
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        float s = wave_sum(q[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NQ)
        part[(int64_t)blockIdx.x * NQ + threadIdx.x] =
            (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x];
}

// one 1024-thread block: wave w sums quantities w, w+16 over the block partials (lane-strided doubles + wave tree:
// fixed order), then thread 0 computes the loss and the gradient coefficients
__global__ __launch_bounds__(1024) void seg_loss_finalize_kernel(const double* __restrict__ part, int nblk, int N, int C,
                                                                  int64_t V, LossCfg cfg, float* loss_out, float* coef) {
    __shared__ double sums[NQ];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int q = wave; q < NQ; q += 16) {
        double s = 0.0;
        for (int b = lane; b < nblk; b += 64) s += part[(int64_t)b * NQ + q];
        s = wave_sum_d(s);
        if (lane == 0) sums[q] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double M = (double)N * (double)V;
        double reg = 0.0;
        double sc = C > 1 ? (double)cfg.w_reg / (double)(C - 1) : 0.0;
        for (int c = 0; c < MAXC; c++) { coef[c] = 0.f; coef[MAXC + c] = 0.f; }
        for (int c = 1; c < C; c++) {
            double I = sums[2 + c], P = sums[2 + MAXC + c], T = sums[2 + 2 * MAXC + c], eps = cfg.eps;
            double A = 0.0, B = 0.0;
            if (cfg.region_kind == 1) {
                double U = P + T;
                reg += 1.0 - (2.0 * I + eps) / (U + eps);
                A = -2.0 / (U + eps);
                B = (2.0 * I + eps) / ((U + eps) * (U + eps));
            } else if (cfg.region_kind == 2) {
                double fp = P - I, fn = T - I, num = I + eps, den = I + cfg.alpha * fp + cfg.beta * fn + eps;
                reg += 1.0 - num / den;
                A = -(1.0 / den) + num / (den * den) * (1.0 - (double)cfg.alpha - (double)cfg.beta);
                B = num / (den * den) * (double)cfg.alpha;
            }
            coef[c] = (float)(A * sc);
            coef[MAXC + c] = (float)(B * sc);
        }
        if (C > 1) reg /= (double)(C - 1);
        double loss = (double)cfg.w_ce * sums[0] / M + (double)cfg.w_reg * reg +
                      (double)cfg.w_kd * (double)cfg.temp * (double)cfg.temp * sums[1] / (M * C);
        coef[2 * MAXC + 0] = (float)((double)cfg.w_ce / M);
        coef[2 * MAXC + 1] = (float)((double)cfg.w_kd * (double)cfg.temp / (M * C));
        coef[2 * MAXC + 2] = (float)loss;
        coef[2 * MAXC + 3] = 0.f;
        *loss_out = (float)loss;
    }
}

__global__ __launch_bounds__(BLK) void seg_loss_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                           const float* __restrict__ teacher, int N, int C, int64_t V,
                                                           float inv_t, const float* __restrict__ coef,
                                                           const float* __restrict__ grad_out, float* __restrict__ dlogits) {
    int64_t M = (int64_t)N * V;
    float go = grad_out ? grad_out[0] : 1.f;
    float A[MAXC], B[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++) { A[c] = coef[c]; B[c] = coef[MAXC + c]; }
    float ce_s = coef[2 * MAXC], kd_s = coef[2 * MAXC + 1];
    for (int64_t m = (int64_t)blockIdx.x * BLK + threadIdx.x; m < M; m += (int64_t)gridDim.x * BLK) {
        int64_t n = m / V, v = m - n * V;
        float z[MAXC], p[MAXC], g[MAXC], lse;
#pragma unroll
        for (int c = 0; c < MAXC; c++) z[c] = c < C ? logits[((int64_t)n * C + c) * V + v] : 0.f;
        int t = (int)labels[m];
        softmax_c(z, C, 1.f, p, lse);
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; c++) {
            g[c] = (c == t ? A[c] : 0.f) + B[c];
            dot += g[c] * p[c];
        }
        float kd[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; c++) kd[c] = 0.f;
        if (teacher) {
            float zt_[MAXC], ps[MAXC], pt[MAXC], ls, lt;
#pragma unroll
            for (int c = 0; c < MAXC; c++) zt_[c] = c < C ? teacher[((int64_t)n * C + c) * V + v] : 0.f;
            softmax_c(z, C, inv_t, ps, ls);
            softmax_c(zt_, C, inv_t, pt, lt);
#pragma unroll
            for (int c = 0; c < MAXC; c++) kd[c] = kd_s * (ps[c] - pt[c]);
        }
#pragma unroll
        for (int c = 0; c < MAXC; c++)
            if (c < C)
                dlogits[((int64_t)n * C + c) * V + v] = go * (ce_s * (p[c] - (c == t ? 1.f : 0.f)) + p[c] * (g[c] - dot) + kd[c]);
    }
}

// ------------------------------------------------------------------------------------------ metrics
__global__ __launch_bounds__(BLK) void seg_metrics_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                          int N, int C, int64_t V, unsigned long long* __restrict__ counts) {
    // counts: [0..C) n_inter, [MAXC..) n_pred, [2*MAXC..) n_tgt, [3*MAXC] n_correct
    int64_t M = (int64_t)N * V;
    unsigned ni[MAXC], np[MAXC], nt[MAXC], nc = 0;
#pragma unroll
    for (int c = 0; c < MAXC; c++) ni[c] = np[c] = nt[c] = 0;
    int64_t start = (int64_t)blockIdx.x * BLK, stride = (int64_t)gridDim.x * BLK;
    for (int64_t base = start; base < M; base += stride) {     // wave-uniform trip count for the ballots
        int64_t m = base + threadIdx.x;
        bool ok = m < M;
        int best = -1, t = -2;
        if (ok) {
            int64_t n = m / V, v = m - n * V;
            float bv = logits[((int64_t)n * C) * V + v];
            best = 0;
            for (int c = 1; c < C; c++) {
                float zc = logits[((int64_t)n * C + c) * V + v];
                if (zc > bv) { bv = zc; best = c; }
            }
            t = (int)labels[m];
        }
#pragma unroll
        for (int c = 0; c < MAXC; c++) {
            if (c < C) {
                ni[c] += __popcll(__ballot(best == c && t == c));
                np[c] += __popcll(__ballot(best == c));
                nt[c] += __popcll(__ballot(t == c));
            }
        }
        nc += __popcll(__ballot(ok && best == t));
    }
    // per-wave counts -> LDS -> one partial row per block (exact integers, no atomics)
    __shared__ unsigned red[4][3 * MAXC + 1];
    int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < MAXC; c++) { red[wave][c] = ni[c]; red[wave][MAXC + c] = np[c]; red[wave][2 * MAXC + c] = nt[c]; }
        red[wave][3 * MAXC] = nc;
    }
    __syncthreads();
    if (threadIdx.x < 3 * MAXC + 1)
        counts[(int64_t)blockIdx.x * (3 * MAXC + 1) + threadIdx.x] =
            (unsigned long long)red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Q1: the reference's class loop is range(1, pred.size(1)) AFTER argmax -> bound = first spatial dim D
__global__ __launch_bounds__(1024) void seg_metrics_finalize_kernel(const unsigned long long* part, int nblk, int N, int C,
                                                                    int D, int64_t V, float* out) {
    __shared__ unsigned long long counts[3 * MAXC + 1];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int q = wave; q < 3 * MAXC + 1; q += 16) {         // exact integer sums, wave-parallel
        unsigned long long s = 0;
        for (int b = lane; b < nblk; b += 64) s += part[(int64_t)b * (3 * MAXC + 1) + q];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) counts[q] = s;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float iou = 0.f, dice = 0.f;
    int valid = 0;
    for (int c = 1; c < D && c < C; c++) {
        unsigned long long nt = counts[2 * MAXC + c];
        if (nt > 0) {
            float inter = (float)counts[c];
            float sum = (float)(long long)(counts[MAXC + c] + nt);
            iou += (inter + 1e-5f) / ((sum - inter) + 1e-5f);
            dice += (2.f * inter + 1e-5f) / (sum + 1e-5f);
            valid++;
        }
    }
    float dv = (float)(valid > 1 ? valid : 1);
    out[0] = iou / dv;
    out[1] = dice / dv;
    out[2] = (float)((double)counts[3 * MAXC] / ((double)N * (double)V));
}

constexpr int METRIC_BLOCKS = 1024;
inline int sgrid(int64_t total, int cap) {
    int64_t w = (total + BLK - 1) / BLK;
    return (int)(w < 1 ? 1 : (w > cap ? cap : w));
}
inline bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }
constexpr int CONV1_NBLK = 512;
}  // namespace

int conv1_fwd(int dtype, const void* z, int zcs, int Cin, const float* w, const float* bias, float* logits, int Cout,
              int N, int64_t V, hipStream_t s) {
    MI3D_CHECK_ARG(Cin >= 1 && Cout >= 1, "conv1_fwd: bad channels");
    dim3 grid((unsigned)sgrid((int64_t)N * V, 4096), (unsigned)cdiv(Cout, MAXC));
    DISPATCH_T(dtype, T, {
        if (Cin % CINB == 0 && zcs % 8 == 0 && al16(z))
            conv1_fwd_kernel<T, true><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, bias, logits, Cout, N, V);
        else
            conv1_fwd_kernel<T, false><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, bias, logits, Cout, N, V);
        MI3D_LAUNCH_CHECK();
    });
    return 0;
}

size_t conv1_bwd_ws_floats(int Cin, int Cout) { return (size_t)CONV1_NBLK * ((size_t)Cin * Cout + Cout); }

int conv1_bwd(int dtype, const void* z, int zcs, int Cin, const float* w, const float* dlogits, int Cout, void* dz,
              int dzcs, float* dW, float* db, int accumulate, float* ws, int N, int64_t V, hipStream_t s) {
    MI3D_CHECK_ARG(Cout <= MAXC, "conv1_bwd: out_channels %d > %d unsupported", Cout, MAXC);
    int nblk = sgrid((int64_t)N * V, CONV1_NBLK);
    int64_t nW = (int64_t)Cin * Cout;
    // every slab element is written by exactly one (blockIdx.x, blockIdx.y) block
    dim3 grid((unsigned)nblk, (unsigned)cdiv(Cin, CINB));
    DISPATCH_T(dtype, T, {
        bool vec = Cin % CINB == 0 && zcs % 8 == 0 && al16(z) && (!dz || (dzcs % 8 == 0 && al16(dz)));
if (vec && Cout <= 4) conv1_bwd_kernel<T, true, 4><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, dlogits, Cout, (T*)dz, dzcs, N, V, ws);
        else if (vec) conv1_bwd_kernel<T, true, MAXC><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, dlogits, Cout, (T*)dz, dzcs, N, V, ws);
        else if (Cout <= 4) conv1_bwd_kernel<T, false, 4><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, dlogits, Cout, (T*)dz, dzcs, N, V, ws);
        else conv1_bwd_kernel<T, false, MAXC><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, dlogits, Cout, (T*)dz, dzcs, N, V, ws);
        MI3D_LAUNCH_CHECK();
    });
    return slab_reduce(ws, nblk, nW + Cout, nW, dW, db, accumulate, s);
}

size_t seg_loss_ws_bytes(int C) { return (size_t)LOSS_MAXBLK * NQ * sizeof(double); }

int seg_loss_fwd(const float* logits, const int64_t* labels, const float* teacher, int N, int C, int64_t V, LossCfg cfg,
                 float* loss_out, float* coef, void* ws, hipStream_t s) {
    MI3D_CHECK_ARG(C >= 1 && C <= MAXC, "seg_loss: %d classes unsupported (max %d)", C, MAXC);
    MI3D_CHECK_ARG(cfg.w_kd == 0.f || teacher, "seg_loss: distillation weight without teacher logits");
    int nblk = sgrid((int64_t)N * V / 4, LOSS_MAXBLK);
    const float* tch = cfg.w_kd != 0.f ? teacher : nullptr;
    seg_loss_fwd_kernel<<<nblk, BLK, 0, s>>>(logits, labels, tch, N, C, V, 1.f / cfg.temp, (double*)ws);
    MI3D_LAUNCH_CHECK();
    seg_loss_finalize_kernel<<<1, 1024, 0, s>>>((const double*)ws, nblk, N, C, V, cfg, loss_out, coef);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int seg_loss_bwd(const float* logits, const int64_t* labels, const float* teacher, int N, int C, int64_t V, LossCfg cfg,
                 const float* coef, const float* grad_out, float* dlogits, hipStream_t s) {
    MI3D_CHECK_ARG(C >= 1 && C <= MAXC, "seg_loss_bwd: %d classes unsupported", C);
    const float* tch = cfg.w_kd != 0.f ? teacher : nullptr;
    seg_loss_bwd_kernel<<<sgrid((int64_t)N * V, 4096), BLK, 0, s>>>(logits, labels, tch, N, C, V, 1.f / cfg.temp, coef,
                                                                    grad_out, dlogits);
    MI3D_LAUNCH_CHECK();
    return 0;
}

size_t seg_metrics_ws_bytes(int C) { return (size_t)METRIC_BLOCKS * (3 * MAXC + 1) * sizeof(unsigned long long); }

int seg_metrics(const float* logits, const int64_t* labels, int N, int C, int D, int64_t V, float* out, void* ws,
                hipStream_t s) {
    MI3D_CHECK_ARG(C >= 1 && C <= MAXC, "seg_metrics: %d classes unsupported", C);
    int nblk = sgrid((int64_t)N * V / 8, METRIC_BLOCKS);
    seg_metrics_kernel<<<nblk, BLK, 0, s>>>(logits, labels, N, C, V, (unsigned long long*)ws);
    MI3D_LAUNCH_CHECK();
    seg_metrics_finalize_kernel<<<1, 1024, 0, s>>>((const unsigned long long*)ws, nblk, N, C, D, V, out);
    MI3D_LAUNCH_CHECK();
    return 0;
}
