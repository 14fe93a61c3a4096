// bn.hip — BatchNorm3d (train/eval) fused with ReLU and Dropout3d, forward and backward.
// Reference semantics: nn.BatchNorm3d(C) -> nn.ReLU(inplace) -> nn.Dropout3d(p), models/unet.py:12-14,16-18.
//
// All kernels are HBM-bound streaming passes over a channels-last [M][C] activation (M = N*D*H*W):
//   stats   : R y                      -> per-block (sum, sumsq) partials -> finalize (double, fixed order)
//   apply   : R y, W z                 z = drop[n,c] * relu(a*y + b)
//   bwd red.: R dz, R y                -> (sum dyh, sum dyh*xhat) partials -> finalize
//   bwd app.: R dz, R y, W dy          dy = g*(dyh - c1 - xhat*c2)
// Per-channel reductions: lane-private fp32 accumulation -> LDS across the rows of a block -> one partial
// per block -> a finalize kernel that sums the partials in double in a fixed order (bitwise reproducible,
// no float atomics).
#include "ops.h"

namespace {

constexpr int BLK = 256;
constexpr int MAXBLK = 1024;
constexpr int FIN_T = 256;          // threads per channel in the finalize kernels

struct RowMap {
    int G;   // channel groups per row (C / VEC)
    int R;   // rows handled per block iteration (BLK / G)
};

template <int VEC> __device__ __forceinline__ bool row_map(int C, int& r, int& g, int& R) {
    int G = C / VEC;
    R = BLK / G;
    r = threadIdx.x / G;
    g = threadIdx.x - r * G;
    return r < R;
}

// reduce K lane-private quantities per channel over the block; result to part[blockIdx.x][k][c]
template <int VEC, int K> __device__ __forceinline__ void block_colreduce(const float (&acc)[K][VEC], int C, bool active,
                                                                           float* lds, float* part) {
    int G = C / VEC, R = BLK / G;
    int r = threadIdx.x / G, g = threadIdx.x - r * G;
    if (active) {
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
            for (int i = 0; i < VEC; i++) lds[((k * R + r) * C) + g * VEC + i] = acc[k][i];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < K * C; idx += BLK) {
        int k = idx / C, c = idx - k * C;
        float s = 0.f;
        for (int rr = 0; rr < R; rr++) s += lds[(k * R + rr) * C + c];
        part[((size_t)blockIdx.x * K + k) * C + c] = s;
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(BLK) void bn_stats_kernel(const T* __restrict__ y, int ycs, int C, int64_t M,
                                                       float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int r, g, R;
    bool active = row_map<VEC>(C, r, g, R);
    float acc[2][VEC];
#pragma unroll
    for (int i = 0; i < VEC; i++) acc[0][i] = acc[1][i] = 0.f;
    if (active) {
        for (int64_t row = (int64_t)blockIdx.x * R + r; row < M; row += (int64_t)gridDim.x * R) {
            float v[VEC];
            ldv<T, VEC>(y + row * ycs + g * VEC, v);
#pragma unroll
            for (int i = 0; i < VEC; i++) { acc[0][i] += v[i]; acc[1][i] += v[i] * v[i]; }
        }
    }
    block_colreduce<VEC, 2>(acc, C, active, lds, part);
}

// split-K convolutions (deep levels): the finishing pass y = bf16(bias + sum_k part[k]) runs HERE, fused with the
// statistics of the rounded values it stores -- one launch and one read of y less per layer
__global__ __launch_bounds__(BLK) void bn_stats_splitk_kernel(const float* __restrict__ skp, int ks, const float* __restrict__ bias,
                                                              bf16* __restrict__ y, int ycs, int C, int64_t M,
                                                              float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int r, g, R;
    bool active = row_map<8>(C, r, g, R);
    float acc[2][8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[0][i] = acc[1][i] = 0.f;
    if (active) {
        float bv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) bv[i] = bias ? bias[g * 8 + i] : 0.f;
        for (int64_t row = (int64_t)blockIdx.x * R + r; row < M; row += (int64_t)gridDim.x * R) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = bv[i];
            for (int k = 0; k < ks; k++) {
                const float* p = skp + ((int64_t)k * M + row) * C + g * 8;
                float4 u = *reinterpret_cast<const float4*>(p), w = *reinterpret_cast<const float4*>(p + 4);
                v[0] += u.x; v[1] += u.y; v[2] += u.z; v[3] += u.w; v[4] += w.x; v[5] += w.y; v[6] += w.z; v[7] += w.w;
            }
            st8<bf16>(y + row * ycs + g * 8, v);
#pragma unroll
            for (int i = 0; i < 8; i++) { float q = (float)(bf16)v[i]; acc[0][i] += q; acc[1][i] += q * q; }
        }
    }
    block_colreduce<8, 2>(acc, C, active, lds, part);
}

// one 64-lane block per channel: double sums of the partials in a fixed order
__global__ void bn_stats_finalize_kernel(const float* __restrict__ part, int nblk, int C, int64_t M,
                                         const float* gamma, const float* beta, float* running_mean,
                                         float* running_var, int64_t* nbt, float momentum, float eps,
                                         float* stat) {
    // FIN_T threads per channel: lane-strided double sums, wave tree, then the waves in fixed order
    __shared__ double ws_[FIN_T / 64][2];
    int c = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = lane; b < nblk; b += FIN_T) {
        s += (double)part[((size_t)b * 2 + 0) * C + c];
        q += (double)part[((size_t)b * 2 + 1) * C + c];
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if ((lane & 63) == 0) { ws_[lane >> 6][0] = s; ws_[lane >> 6][1] = q; }
    __syncthreads();
    if (lane == 0) {
        s = 0.0; q = 0.0;
#pragma unroll
        for (int w = 0; w < FIN_T / 64; w++) { s += ws_[w][0]; q += ws_[w][1]; }
        double mean = s / (double)M;
        double var = q / (double)M - mean * mean;
        if (var < 0.0) var = 0.0;
        double inv = 1.0 / sqrt(var + (double)eps);
        float a = (float)((double)gamma[c] * inv);
        stat[c] = (float)mean;
        stat[C + c] = (float)inv;
        stat[2 * C + c] = a;
        stat[3 * C + c] = (float)((double)beta[c] - mean * (double)gamma[c] * inv);
        if (momentum < 0.f) {          // deferred running-statistics update (ops.h bn_deferred_apply): publish the doubles
            double* side = reinterpret_cast<double*>(running_mean);
            if (side) { side[c] = mean; side[C + c] = M > 1 ? var * (double)M / (double)(M - 1) : var; }
        } else {
            if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            if (running_var) {
                double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
                running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
            }
            if (nbt && c == 0) *nbt += 1;
        }
    }
}

// Small (deep-level) tensors: the statistics kernel leaves only a few partial rows (<= SMALL_ROWS), and the consumer kernel
// finishes them in its prologue (every workgroup redundantly, fixed order, double) instead of a finalize launch of its own:
// every kernel node is a link of the step's dependent chain (~5 us each at these sizes), and an in-kernel "last workgroup
// finalizes" ticket costs as much as the launch it replaces on 8 XCDs (measured).  Workgroup 0 publishes stat[4][C] for the
// backward pass and updates the running statistics.
constexpr int SMALL_ROWS = 128;        // round 3 scan (ms/step): 64: 2.330, 128: 2.283, 256: 2.288, 512: 2.307
constexpr int MAXC_BN = 256;
struct BnPart {
    const float* part; int nrows; int64_t M;
    const float* gamma; const float* beta; float* running_mean; float* running_var; int64_t* nbt;
    float momentum, eps;
};
// sum the nrows partial rows [nrows][2][C] per (k, channel): out[k] for threads < C.  Requirement: C in {4..256} a power
// of two.  The flat array is read as float4, thread t taking elements 4t + 1024 j: its (k, channel quad) is the same for
// every j, the loads are independent (a scalar `acc += part[r]` loop pays an L2 round trip per row: measured 0.3 us each;
// only the last, partial 1024-element block is guarded), the per-thread sums are combined through LDS in a fixed order.
__device__ __forceinline__ void rows_sum(const float* __restrict__ part, int nrows, int C, double* red /* [BLK*4] */, double (&out)[2]) {
    int total4 = (nrows * 2 * C) >> 2, J = total4 >> 8;
    const f32x4* p4 = reinterpret_cast<const f32x4*>(part) + threadIdx.x;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int j = 0;
    for (; j + 8 <= J; j += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p4[(size_t)(j + u) * BLK];
#pragma unroll
        for (int u = 0; u < 8; u++) { acc[0] += (double)v[u][0]; acc[1] += (double)v[u][1]; acc[2] += (double)v[u][2]; acc[3] += (double)v[u][3]; }
    }
    for (; j < J; j++) {
        f32x4 v = p4[(size_t)j * BLK];
        acc[0] += (double)v[0]; acc[1] += (double)v[1]; acc[2] += (double)v[2]; acc[3] += (double)v[3];
    }
    if (J * BLK + (int)threadIdx.x < total4) {
        f32x4 v = p4[(size_t)J * BLK];
        acc[0] += (double)v[0]; acc[1] += (double)v[1]; acc[2] += (double)v[2]; acc[3] += (double)v[3];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) red[threadIdx.x * 4 + i] = acc[i];
    __syncthreads();
    if (threadIdx.x < C) {
        int q = C >> 2, c = threadIdx.x, lines = BLK / q;          // thread t holds line t / q (k = line & 1), channels 4*(t % q)..+3
#pragma unroll
        for (int k = 0; k < 2; k++) {
            double t = 0.0;
            for (int L = k; L < lines; L += 2) t += red[(L * q + (c >> 2)) * 4 + (c & 3)];
            out[k] = t;
        }
    }
}

// The same sum for a WIDE workgroup (round 4, NT = 1024 threads, <= 1024 rows): the consumers of the full-resolution layers run
// as a few fat workgroups (<= 2 per CU) so that the rows are pulled from the L2 a few hundred times instead of 2048 times, and
// the combine is two LDS stages (4 values per thread, then NT / 2C per (k, channel)) instead of one thread per channel walking
// NT / (C / 4) lines.  red: [NT * 4] doubles, reused for the stage-two partials.  Result for threads < C in out[2].
constexpr int WIDE_J = 8;
template <int NT>
__device__ __forceinline__ void rows_sum_wide(const float* __restrict__ part, int nrows, int C, double* red, double (&out)[2]) {
    // ALL of a thread's pieces are requested before the first is used: one trip to the L2 / memory (the rows were written by
    // another kernel's workgroups on all eight XCDs).  A counted loop of dependent round trips cost 6 us here.  The launcher
    // guarantees nrows * C <= WIDE_J * NT * 2 (at most WIDE_J pieces per thread).
    const int total4 = (nrows * 2 * C) >> 2;
    const f32x4* p4 = reinterpret_cast<const f32x4*>(part) + threadIdx.x;
    f32x4 v[WIDE_J];
#pragma unroll
    for (int u = 0; u < WIDE_J; u++) {
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (u * NT + (int)threadIdx.x < total4) v[u] = p4[(size_t)u * NT];
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < WIDE_J; u++) { acc[0] += (double)v[u][0]; acc[1] += (double)v[u][1]; acc[2] += (double)v[u][2]; acc[3] += (double)v[u][3]; }
    // flat view: thread t holds elements 4t .. 4t+3 of a [NT * 4 / 2C][2C] array whose column kc = k * C + channel
#pragma unroll
    for (int i = 0; i < 4; i++) red[threadIdx.x * 4 + i] = acc[i];
    __syncthreads();
    const int C2 = 2 * C, lines = (NT * 4) / C2, S = NT / C2;      // S threads per column, lines / S (= 4) values each
    const int kc = threadIdx.x % C2, sidx = threadIdx.x / C2;
    double t = 0.0;
    if (sidx < S)
        for (int L = sidx; L < lines; L += S) t += red[L * C2 + kc];
    __syncthreads();
    if (sidx < S) red[sidx * C2 + kc] = t;
    __syncthreads();
    if (threadIdx.x < C) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            double u = 0.0;
            for (int q = 0; q < S; q++) u += red[q * C2 + k * C + threadIdx.x];
            out[k] = u;
        }
    }
}

template <int NT = BLK>
__device__ __forceinline__ void bn_train_coeffs(const BnPart& t, int C, float* stat, float* ab /* [2][MAXC_BN] */, double* red) {
    double sq[2];
    if constexpr (NT == BLK) rows_sum(t.part, t.nrows, C, red, sq);
    else rows_sum_wide<NT>(t.part, t.nrows, C, red, sq);
    int c = threadIdx.x;
    if (c < C) {
        double mean = sq[0] / (double)t.M;
        double var = sq[1] / (double)t.M - mean * mean;
        if (var < 0.0) var = 0.0;
        double inv = 1.0 / sqrt(var + (double)t.eps);
        float a = (float)((double)t.gamma[c] * inv);
        float b = (float)((double)t.beta[c] - mean * (double)t.gamma[c] * inv);
        ab[c] = a;
        ab[MAXC_BN + c] = b;
        if (blockIdx.x == 0) {
            stat[c] = (float)mean;
            stat[C + c] = (float)inv;
            stat[2 * C + c] = a;
            stat[3 * C + c] = b;
            if (t.momentum < 0.f) {      // deferred running-statistics update: publish the doubles
                double* side = reinterpret_cast<double*>(t.running_mean);
                if (side) { side[c] = mean; side[C + c] = t.M > 1 ? var * (double)t.M / (double)(t.M - 1) : var; }
            } else {
                if (t.running_mean) t.running_mean[c] = (float)((1.0 - t.momentum) * t.running_mean[c] + t.momentum * mean);
                if (t.running_var) {
                    double unb = t.M > 1 ? var * (double)t.M / (double)(t.M - 1) : var;
                    t.running_var[c] = (float)((1.0 - t.momentum) * t.running_var[c] + t.momentum * unb);
                }
                if (t.nbt && c == 0) *t.nbt += 1;
            }
        }
    }
    __syncthreads();
}

__global__ void bn_eval_stats_kernel(int C, const float* gamma, const float* beta, const float* rm,
                                     const float* rv, float eps, float* stat) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double inv = 1.0 / sqrt((double)rv[c] + (double)eps);
    stat[c] = rm[c];
    stat[C + c] = (float)inv;
    stat[2 * C + c] = (float)((double)gamma[c] * inv);
    stat[3 * C + c] = (float)((double)beta[c] - (double)rm[c] * (double)gamma[c] * inv);
}

// Dropout3d scale of this thread's channel group for the sample `row` lies in: rows are visited in increasing order, so the
// VEC scales are (re)loaded only when a sample boundary is crossed (a per-row `drop[(row / V) * C + c]` costs a 32-bit
// division and VEC global loads per element: +0.14 ms/step at p = 0.1)
template <int VEC>
struct DropCache {
    float s[VEC];
    int64_t bound = 0;
    __device__ __forceinline__ DropCache() {
#pragma unroll
        for (int i = 0; i < VEC; i++) s[i] = 1.f;
    }
    __device__ __forceinline__ void at(const float* __restrict__ drop, int64_t row, int64_t V, int C, int c0) {
        if (drop && row >= bound) {
            int64_t n = (int64_t)((unsigned)row / (unsigned)V);
            bound = (n + 1) * V;
#pragma unroll
            for (int i = 0; i < VEC; i++) s[i] = drop[n * C + c0 + i];
        }
    }
};

// grid stride (gridDim*BLK) is a multiple of G = C/VEC (launcher guarantees it), so a thread's channel group is
// fixed: per-channel coefficients are loaded into registers ONCE instead of per element
template <typename T, int VEC, bool TRAIN>
__global__ __launch_bounds__(BLK) void bn_apply_kernel(const T* __restrict__ y, int ycs, int C, int64_t M, int64_t V,
                                                       float* __restrict__ stat, BnPart tr, const float* __restrict__ drop,
                                                       T* __restrict__ z, int zcs) {
    int G = C / VEC;
    int64_t gtid = (int64_t)blockIdx.x * BLK + threadIdx.x;
    int g = (int)(gtid % G);
    int64_t row = gtid / G, rstep = ((int64_t)gridDim.x * BLK) / G;
    float a[VEC], b[VEC];
    if constexpr (TRAIN) {
        __shared__ double red[BLK * 4];
        __shared__ float ab[2 * MAXC_BN];
        bn_train_coeffs(tr, C, stat, ab, red);
#pragma unroll
        for (int i = 0; i < VEC; i++) { a[i] = ab[g * VEC + i]; b[i] = ab[MAXC_BN + g * VEC + i]; }
    } else {
#pragma unroll
        for (int i = 0; i < VEC; i++) { a[i] = stat[2 * C + g * VEC + i]; b[i] = stat[3 * C + g * VEC + i]; }
    }
    DropCache<VEC> dc;
    for (; row < M; row += rstep) {
        float v[VEC], o[VEC];
        ldv<T, VEC>(y + row * ycs + g * VEC, v);
        dc.at(drop, row, V, C, g * VEC);
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            float t = fmaf(v[i], a[i], b[i]);
            t = t > 0.f ? t : 0.f;
            o[i] = t * dc.s[i];
        }
        stv<T, VEC>(z + row * zcs + g * VEC, o);
    }
}

// WIDE consumer (round 4): the layers whose producing conv leaves more than SMALL_ROWS partial rows (levels 0-1: 432-1024 rows) used
// to pay a finalize launch between the conv and this pass (a 5 us link of the dependent chain, 12 of them per forward).  Here the
// pass runs as <= 2 workgroups of 1024 threads per CU: every workgroup finishes the rows itself (rows_sum_wide; the rows come out of
// the L2 <= 512 times instead of once per 256-thread workgroup), with the loads of its first two rows already in flight, and
// workgroup 0 publishes stat[4][C] / the running statistics like the small route does.
constexpr int WNT = 1024;
constexpr int WIDE_ROWS = 1024;
// bf16 only (the plan takes this route in the bf16 path).  WPRE rows per thread are requested BEFORE the prologue (raw 16-byte
// pieces, 4 VGPRs each): at level 1 that is the whole tensor, at level 0 more than half of it -- the statistics are finished
// while the first loads are in flight, and the main loop keeps WPRE loads per thread in flight.
constexpr int WPRE = 8;
__global__ __launch_bounds__(WNT) void bn_apply_wide_kernel(const bf16* __restrict__ y, int ycs, int C, int64_t M, int64_t V,
                                                            float* __restrict__ stat, BnPart tr, const float* __restrict__ drop,
                                                            bf16* __restrict__ z, int zcs) {
    constexpr int VEC = 8;
    const int G = C / VEC;
    const int64_t gtid = (int64_t)blockIdx.x * WNT + threadIdx.x;
    const int g = (int)(gtid % G);
    int64_t row = gtid / G;
    const int64_t rstep = ((int64_t)gridDim.x * WNT) / G;
    bf16x8 raw[WPRE];
#pragma unroll
    for (int u = 0; u < WPRE; u++)
        if (row + u * rstep < M) raw[u] = *reinterpret_cast<const bf16x8*>(y + (row + u * rstep) * ycs + g * VEC);
    __shared__ double red[WNT * 4];
    __shared__ float ab[2 * MAXC_BN];
    bn_train_coeffs<WNT>(tr, C, stat, ab, red);
    float a[VEC], b[VEC];
#pragma unroll
    for (int i = 0; i < VEC; i++) { a[i] = ab[g * VEC + i]; b[i] = ab[MAXC_BN + g * VEC + i]; }
    DropCache<VEC> dc;
    while (row < M) {
#pragma unroll
        for (int u = 0; u < WPRE; u++) {
            const int64_t r = row + u * rstep;
            if (r < M) {
                dc.at(drop, r, V, C, g * VEC);
                bf16x8 o;
#pragma unroll
                for (int i = 0; i < VEC; i++) {
                    float t = fmaf((float)raw[u][i], a[i], b[i]);
                    t = t > 0.f ? t : 0.f;
                    o[i] = (bf16)(t * dc.s[i]);
                }
                *reinterpret_cast<bf16x8*>(z + r * zcs + g * VEC) = o;
            }
        }
        row += WPRE * rstep;
#pragma unroll
        for (int u = 0; u < WPRE; u++)
            if (row + u * rstep < M) raw[u] = *reinterpret_cast<const bf16x8*>(y + (row + u * rstep) * ycs + g * VEC);
    }
}

// Second half of an encoder block: the same apply pass, one thread per 2x2x2 pooling window and channel group — writes the
// eight activated voxels (the skip tensor) AND their maximum (MaxPool3d(2,2), models/unet.py:40,71), so the pooling launch and
// its re-read of the skip tensor disappear.  Even D, H, W only (every voxel lies in exactly one window).
// PAIR (round 4, VEC = 8, C / 8 a power of two <= 32): two threads per window -- thread (window, c, g) owns the four voxels (a, b, c),
// so the lanes of a wave touch one contiguous run per (a, b) instead of every other 32-B half; the halves exchange their maxima with
// one lane swap (the maximum of the same eight stored values: bit-identical)
template <typename T, int VEC, bool TRAIN, bool PAIR = false>
__global__ __launch_bounds__(BLK) void bn_apply_pool_kernel(const T* __restrict__ y, int ycs, int C, int N, int D, int H, int W,
                                                            float* __restrict__ stat, BnPart tr, const float* __restrict__ drop,
                                                            T* __restrict__ z, int zcs, T* __restrict__ pl, int pcs) {
    const int G = C / VEC, Do = D / 2, Ho = H / 2, Wo = W / 2;
    const unsigned gtid = blockIdx.x * BLK + threadIdx.x;
    const int g = (int)(gtid % (unsigned)G);
    [[maybe_unused]] const int pc = PAIR ? (int)((gtid / (unsigned)G) & 1u) : 0;
    constexpr int NK = PAIR ? 4 : 8;
    float a[VEC], b[VEC];
    if constexpr (TRAIN) {
        __shared__ double red[BLK * 4];
        __shared__ float ab[2 * MAXC_BN];
        bn_train_coeffs(tr, C, stat, ab, red);
#pragma unroll
        for (int i = 0; i < VEC; i++) { a[i] = ab[g * VEC + i]; b[i] = ab[MAXC_BN + g * VEC + i]; }
    } else {
#pragma unroll
        for (int i = 0; i < VEC; i++) { a[i] = stat[2 * C + g * VEC + i]; b[i] = stat[3 * C + g * VEC + i]; }
    }
    const unsigned TPW = (unsigned)G * (PAIR ? 2u : 1u);          // threads per window
    const unsigned windows = (unsigned)N * Do * Ho * Wo, wstep = (gridDim.x * BLK) / TPW;
    float ds[VEC];
    int dn = -1;
#pragma unroll
    for (int i = 0; i < VEC; i++) ds[i] = 1.f;
    for (unsigned win = gtid / TPW; win < windows; win += wstep) {
        unsigned r = win;
        const int wo = (int)(r % (unsigned)Wo); r /= (unsigned)Wo;
        const int ho = (int)(r % (unsigned)Ho); r /= (unsigned)Ho;
        const int d_o = (int)(r % (unsigned)Do);
        const int n = (int)(r / (unsigned)Do);
        if (drop && n != dn) {
            dn = n;
#pragma unroll
            for (int i = 0; i < VEC; i++) ds[i] = drop[(int64_t)n * C + g * VEC + i];
        }
        float v[NK][VEC], m[VEC];
#pragma unroll
        for (int q = 0; q < NK; q++) {
            const int k = PAIR ? 2 * q + pc : q;
            const int64_t off = (((int64_t)n * D + 2 * d_o + (k >> 2)) * H + 2 * ho + ((k >> 1) & 1)) * W + 2 * wo + (k & 1);
            ldv<T, VEC>(y + off * ycs + g * VEC, v[q]);
        }
#pragma unroll
        for (int i = 0; i < VEC; i++) m[i] = -INFINITY;
#pragma unroll
        for (int q = 0; q < NK; q++) {
            const int k = PAIR ? 2 * q + pc : q;
            const int64_t off = (((int64_t)n * D + 2 * d_o + (k >> 2)) * H + 2 * ho + ((k >> 1) & 1)) * W + 2 * wo + (k & 1);
            float o[VEC];
#pragma unroll
            for (int i = 0; i < VEC; i++) {
                float t = fmaf(v[q][i], a[i], b[i]);
                t = t > 0.f ? t : 0.f;
                o[i] = (float)(T)(t * ds[i]);             // the pooled value is the maximum of the STORED (rounded) values
                m[i] = o[i] > m[i] ? o[i] : m[i];
            }
            stv<T, VEC>(z + off * zcs + g * VEC, o);
        }
        if constexpr (PAIR) {
#pragma unroll
            for (int i = 0; i < VEC; i++) { float om = __shfl_xor(m[i], G, 64); m[i] = om > m[i] ? om : m[i]; }
            if (pc != 0) continue;
        }
        stv<T, VEC>(pl + ((((int64_t)n * Do + d_o) * Ho + ho) * Wo + wo) * pcs + g * VEC, m);
    }
}

// The pooled pass as a wide consumer (bf16, two threads per window): two windows (8 pieces) per thread are requested before the
// prologue finishes the statistics; same arithmetic per element as the kernel above.
constexpr int WPW = 2;
__global__ __launch_bounds__(WNT) void bn_apply_pool_wide_kernel(const bf16* __restrict__ y, int ycs, int C, int N, int D, int H, int W,
                                                                 float* __restrict__ stat, BnPart tr, const float* __restrict__ drop,
                                                                 bf16* __restrict__ z, int zcs, bf16* __restrict__ pl, int pcs) {
    constexpr int VEC = 8;
    const int G = C / VEC, Do = D / 2, Ho = H / 2, Wo = W / 2;
    const unsigned gtid = blockIdx.x * WNT + threadIdx.x;
    const int g = (int)(gtid % (unsigned)G);
    const int pc = (int)((gtid / (unsigned)G) & 1u);
    const unsigned TPW = (unsigned)G * 2u;
    const unsigned windows = (unsigned)N * Do * Ho * Wo, wstep = (gridDim.x * WNT) / TPW;
    unsigned win = gtid / TPW;
    bf16x8 raw[WPW][4];
    int base[WPW], smp[WPW];          // voxel index of the window's corner (+ this thread's x offset), its sample
    auto issue = [&]() {
#pragma unroll
        for (int u = 0; u < WPW; u++) {
            unsigned r = win + u * wstep;
            if (r < windows) {
                const int wo = (int)(r % (unsigned)Wo); r /= (unsigned)Wo;
                const int ho = (int)(r % (unsigned)Ho); r /= (unsigned)Ho;
                const int d_o = (int)(r % (unsigned)Do);
                smp[u] = (int)(r / (unsigned)Do);
                base[u] = ((smp[u] * D + 2 * d_o) * H + 2 * ho) * W + 2 * wo + pc;
#pragma unroll
                for (int q = 0; q < 4; q++)       // k = 2q + pc: (dz, dy) = (q >> 1, q & 1), dx = pc
                    raw[u][q] = *reinterpret_cast<const bf16x8*>(y + (int64_t)(base[u] + ((q >> 1) * H + (q & 1)) * W) * ycs + g * VEC);
            }
        }
    };
    issue();
    __shared__ double red[WNT * 4];
    __shared__ float ab[2 * MAXC_BN];
    bn_train_coeffs<WNT>(tr, C, stat, ab, red);
    float a[VEC], b[VEC];
#pragma unroll
    for (int i = 0; i < VEC; i++) { a[i] = ab[g * VEC + i]; b[i] = ab[MAXC_BN + g * VEC + i]; }
    float ds[VEC];
    int dn = -1;
#pragma unroll
    for (int i = 0; i < VEC; i++) ds[i] = 1.f;
    while (win < windows) {
#pragma unroll
        for (int u = 0; u < WPW; u++) {
            const unsigned w_ = win + u * wstep;
            if (w_ < windows) {                                   // uniform over a window's thread pair (the lane swap below)
                if (drop && smp[u] != dn) {
                    dn = smp[u];
#pragma unroll
                    for (int i = 0; i < VEC; i++) ds[i] = drop[(int64_t)dn * C + g * VEC + i];
                }
                float m[VEC];
#pragma unroll
                for (int i = 0; i < VEC; i++) m[i] = -INFINITY;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    bf16x8 o;
#pragma unroll
                    for (int i = 0; i < VEC; i++) {
                        float t = fmaf((float)raw[u][q][i], a[i], b[i]);
                        t = t > 0.f ? t : 0.f;
                        o[i] = (bf16)(t * ds[i]);
                        const float of = (float)o[i];              // the pooled value is the maximum of the STORED values
                        m[i] = of > m[i] ? of : m[i];
                    }
                    *reinterpret_cast<bf16x8*>(z + (int64_t)(base[u] + ((q >> 1) * H + (q & 1)) * W) * zcs + g * VEC) = o;
                }
                bf16x8 mo;
#pragma unroll
                for (int i = 0; i < VEC; i++) { float om = __shfl_xor(m[i], G, 64); mo[i] = (bf16)(om > m[i] ? om : m[i]); }
                if (pc == 0) *reinterpret_cast<bf16x8*>(pl + (int64_t)w_ * pcs + g * VEC) = mo;
            }
        }
        win += WPW * wstep;
        issue();
    }
}

// nred = blocks of the reduction proper (= gridDim.x unless a slab-sum job rides behind them, see bn_bwd).
// skp != NULL: dz arrives as the ks fp32 split-K partials of the input-gradient conv that produced it ([ks][M][C]); they
// are summed and rounded HERE and dz is written for the apply pass -- the split-K finishing launch disappears
template <typename T, int VEC>
__device__ __forceinline__ void bn_bwd_reduce_body(int nred, const T* __restrict__ dz, int dzcs, const T* __restrict__ y,
                                                   int ycs, int C, int64_t M, int64_t V,
                                                   const float* __restrict__ stat,
                                                   const float* __restrict__ drop, float* __restrict__ part,
                                                   const float* __restrict__ skp = nullptr, int ks = 0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int r, g, R;
    bool active = row_map<VEC>(C, r, g, R);
    float acc[2][VEC];
#pragma unroll
    for (int i = 0; i < VEC; i++) acc[0][i] = acc[1][i] = 0.f;
    if (active) {
        float mean[VEC], inv[VEC], a[VEC], b[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            mean[i] = stat[g * VEC + i]; inv[i] = stat[C + g * VEC + i];
            a[i] = stat[2 * C + g * VEC + i]; b[i] = stat[3 * C + g * VEC + i];
        }
        DropCache<VEC> dc;
        for (int64_t row = (int64_t)blockIdx.x * R + r; row < M; row += (int64_t)nred * R) {
            float yv[VEC], gv[VEC];
            ldv<T, VEC>(y + row * ycs + g * VEC, yv);
            if (skp) {
#pragma unroll
                for (int i = 0; i < VEC; i++) gv[i] = 0.f;
                for (int k = 0; k < ks; k++) {
                    const float* p = skp + ((int64_t)k * M + row) * C + g * VEC;
#pragma unroll
                    for (int i = 0; i < VEC; i++) gv[i] += p[i];
                }
#pragma unroll
                for (int i = 0; i < VEC; i++) gv[i] = round_to<T>(gv[i]);
                stv<T, VEC>(const_cast<T*>(dz) + row * dzcs + g * VEC, gv);
            } else
                ldv<T, VEC>(dz + row * dzcs + g * VEC, gv);
            dc.at(drop, row, V, C, g * VEC);
#pragma unroll
            for (int i = 0; i < VEC; i++) {
                float pre = fmaf(yv[i], a[i], b[i]);
                float m = pre > 0.f ? dc.s[i] : 0.f;
                float dyh = gv[i] * m;
                acc[0][i] += dyh;
                acc[1][i] += dyh * (yv[i] - mean[i]) * inv[i];
            }
        }
    }
    block_colreduce<VEC, 2>(acc, C, active, lds, part);
}

template <typename T, int VEC>
__global__ __launch_bounds__(BLK) void bn_bwd_reduce_kernel(const T* __restrict__ dz, int dzcs, const T* __restrict__ y,
                                                            int ycs, int C, int64_t M, int64_t V,
                                                            const float* __restrict__ stat,
                                                            const float* __restrict__ drop, float* __restrict__ part,
                                                            const float* __restrict__ skp, int ks) {
    bn_bwd_reduce_body<T, VEC>((int)gridDim.x, dz, dzcs, y, ycs, C, M, V, stat, drop, part, skp, ks);
}
// the same with a pending weight-gradient slab sum in the blocks behind the reduction's: one chain link less
template <typename T, int VEC>
__global__ __launch_bounds__(BLK) void bn_bwd_reduce_slab_kernel(int nred, const T* __restrict__ dz, int dzcs, const T* __restrict__ y,
                                                                 int ycs, int C, int64_t M, int64_t V,
                                                                 const float* __restrict__ stat,
                                                                 const float* __restrict__ drop, float* __restrict__ part, SlabJob job,
                                                                 SlabJob job2, const float* __restrict__ skp, int ks) {
    if ((int)blockIdx.x < nred) bn_bwd_reduce_body<T, VEC>(nred, dz, dzcs, y, ycs, C, M, V, stat, drop, part, skp, ks);
    else {
        int b = (int)blockIdx.x - nred;
        if (b < job.nblocks) slab_job_run(job, b);
        else slab_job_run(job2, b - job.nblocks);
    }
}

// coef[3][C] = {c1 = sum_dyh / M, c2 = sum_dyh_xhat / M, g = gamma*invstd(=a)}; dgamma, dbeta (+)=
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C, int64_t M,
                                       const float* stat, float* dgamma, float* dbeta, int accumulate, float* coef) {
    __shared__ double ws_[FIN_T / 64][2];
    int c = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = lane; b < nblk; b += FIN_T) {
        s += (double)part[((size_t)b * 2 + 0) * C + c];
        q += (double)part[((size_t)b * 2 + 1) * C + c];
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if ((lane & 63) == 0) { ws_[lane >> 6][0] = s; ws_[lane >> 6][1] = q; }
    __syncthreads();
    if (lane == 0) {
        s = 0.0; q = 0.0;
#pragma unroll
        for (int w = 0; w < FIN_T / 64; w++) { s += ws_[w][0]; q += ws_[w][1]; }
        coef[c] = (float)(s / (double)M);
        coef[C + c] = (float)(q / (double)M);
        coef[2 * C + c] = stat[2 * C + c];
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q;
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
    }
}

// dy = g*(dyh - c1 - xhat*c2) rewritten per channel as  dy = g*dyh + A*y + B  with
//   A = -g*c2*invstd,  B = g*(c2*invstd*mean - c1);  coefficients live in registers (fixed channel group per thread)
// SMALL: no finalize launch ran -- `coef` is the reduction's partial rows [nrows][2][C]; the prologue sums them (every
// workgroup, fixed order, double) and workgroup 0 writes dgamma / dbeta (+)=
template <typename T, int VEC, bool SMALL>
__global__ __launch_bounds__(BLK) void bn_bwd_apply_kernel(const T* __restrict__ dz, int dzcs, const T* __restrict__ y,
                                                           int ycs, int C, int64_t M, int64_t V,
                                                           const float* __restrict__ stat, const float* __restrict__ coef,
                                                           int nrows, float* dgamma, float* dbeta, int accumulate,
                                                           const float* __restrict__ drop, T* __restrict__ dy, int dycs) {
    int G = C / VEC;
    int64_t gtid = (int64_t)blockIdx.x * BLK + threadIdx.x;
    int g = (int)(gtid % G);
    int64_t row = gtid / G, rstep = ((int64_t)gridDim.x * BLK) / G;
    float a[VEC], b[VEC], gg[VEC], A[VEC], B[VEC];
    if constexpr (SMALL) {
        __shared__ double red[BLK * 4];
        __shared__ float cf[2 * MAXC_BN];
        double sq[2];
        rows_sum(coef, nrows, C, red, sq);
        if (threadIdx.x < C) {
            int c = threadIdx.x;
            cf[c] = (float)(sq[0] / (double)M);
            cf[MAXC_BN + c] = (float)(sq[1] / (double)M);
            if (blockIdx.x == 0) {
                if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)sq[1] : (float)sq[1];
                if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)sq[0] : (float)sq[0];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            int c = g * VEC + i;
            float mean = stat[c], inv = stat[C + c];
            a[i] = stat[2 * C + c]; b[i] = stat[3 * C + c];
            gg[i] = a[i];
            float k = gg[i] * cf[MAXC_BN + c] * inv;
            A[i] = -k;
            B[i] = k * mean - gg[i] * cf[c];
        }
    } else {
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            int c = g * VEC + i;
            float mean = stat[c], inv = stat[C + c];
            a[i] = stat[2 * C + c]; b[i] = stat[3 * C + c];
            gg[i] = coef[2 * C + c];
            float k = gg[i] * coef[C + c] * inv;
            A[i] = -k;
            B[i] = k * mean - gg[i] * coef[c];
        }
    }
    DropCache<VEC> dc;
    for (; row < M; row += rstep) {
        float yv[VEC], gv[VEC], o[VEC];
        ldv<T, VEC>(y + row * ycs + g * VEC, yv);
        ldv<T, VEC>(dz + row * dzcs + g * VEC, gv);
        dc.at(drop, row, V, C, g * VEC);
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            float pre = fmaf(yv[i], a[i], b[i]);
            float m = pre > 0.f ? dc.s[i] : 0.f;
            o[i] = fmaf(gg[i] * m, gv[i], fmaf(A[i], yv[i], B[i]));
        }
        stv<T, VEC>(dy + row * dycs + g * VEC, o);
    }
}

// inference: every BatchNorm of the network folded into its conv in one launch (block = one layer)
__global__ void bn_fold_all_kernel(BnFoldJobs J) {
    const BnFoldJob& j = J.j[blockIdx.x];
    for (int c = threadIdx.x; c < j.C; c += blockDim.x) {
        double a = (double)j.gamma[c] / sqrt((double)j.rv[c] + (double)J.eps);
        j.scale[c] = (float)a;
        j.fbias[c] = (float)(((double)(j.conv_bias ? j.conv_bias[c] : 0.f) - (double)j.rm[c]) * a + (double)j.beta[c]);
    }
}

// the deferred running-statistics updates of a whole network in one launch (block = one BatchNorm layer)
__global__ void bn_deferred_apply_kernel(BnDeferJobs J) {
    const BnDeferJob& j = J.j[blockIdx.x];
    for (int c = threadIdx.x; c < j.C; c += blockDim.x) {
        j.rm[c] = (float)((1.0 - J.momentum) * j.rm[c] + J.momentum * j.side[c]);
        j.rv[c] = (float)((1.0 - J.momentum) * j.rv[c] + J.momentum * j.side[j.C + c]);
    }
    if (threadIdx.x == 0 && j.nbt) *j.nbt += 1;
}

inline bool vec8_ok(int C, int cs_a, int cs_b, const void* pa, const void* pb, int esz) {
    return C % 8 == 0 && cs_a % 8 == 0 && cs_b % 8 == 0 && ((uintptr_t)pa % 16 == 0) && ((uintptr_t)pb % 16 == 0) &&
           (C / 8) <= BLK && esz > 0;
}

inline int reduce_grid(int64_t M, int R) {
    int64_t want = (M + (int64_t)R * 4 - 1) / ((int64_t)R * 4);       // ~4 rows per thread (round 3: 2 and 8 measured neutral) ...
    int64_t one = (M + R - 1) / R;                                      // ... but small tensors (deep levels) are a latency
    if (want < 256) want = one < 256 ? one : 256;                       // chain, not a stream: one row per thread then
    if (want < 1) want = 1;
    return (int)(want > MAXBLK ? MAXBLK : want);
}
// grid for the streaming kernels: gridDim*BLK must be a multiple of G (fixed channel group per thread)
inline int stream_grid(int64_t total, int G) {
    int64_t want = (total + BLK - 1) / BLK;
    if (want < 1) want = 1;
    if (want > 256 * 8) want = 256 * 8;          // round 3: 1024 / 4096 workgroups measured neutral
    int m = 1;
    while ((m * BLK) % G != 0) m++;            // G = 5 -> m = 5, powers of two -> m = 1
    want = (want + m - 1) / m * m;
    return (int)want;
}

}  // namespace

size_t bn_ws_floats(int C) { return (size_t)MAXBLK * 2 * C + 3 * (size_t)C; }

// "small" = a deep-level tensor whose reduction fits a few workgroups: <= SMALL_ROWS partial rows, finished by the consumer
constexpr int64_t SMALL_ELEMS = 2 << 20;
inline bool bn_small(int C, int64_t M) {
    return C >= 4 && C <= MAXC_BN && (C & (C - 1)) == 0 && M * C <= SMALL_ELEMS && !mi3d_routes().no_small_bn;
}
// number of partial rows for a small tensor
inline int bn_small_rows(int nblk, int C) { return nblk > SMALL_ROWS ? SMALL_ROWS : nblk; }
bool bn_small_ok(int C, int64_t M, int rows) { return bn_small(C, M) && rows >= 1 && rows <= SMALL_ROWS; }
bool bn_small_route(int C, int64_t M) { return bn_small(C, M); }
// round 4: `rows` partial rows written by a conv epilogue ([rows][2][C]) can be finished by the consumer (thin workgroups up to
// SMALL_ROWS rows, the wide kernels above that): no finalize launch between the conv and the apply pass
inline bool wide_shape_ok(int C) { return C >= 8 && C <= MAXC_BN && (C & (C - 1)) == 0; }
bool bn_rows_route_ok(int C, int64_t M, int rows) {
    const int mode = mi3d_routes().wide_bn;
    if (!(mode & 1) || rows < 1 || M * C >= (1ll << 31)) return false;
    if (rows <= SMALL_ROWS && rows < mi3d_routes().wide_min_rows) return C >= 4 && C <= MAXC_BN && (C & (C - 1)) == 0;
    return (mode & 2) && rows <= WIDE_ROWS && wide_shape_ok(C) && (int64_t)rows * C <= (int64_t)WIDE_J * WNT * 2;
}
inline int wide_grid(int64_t total_threads) {
    int want = (int)((total_threads + WNT - 1) / WNT), cap = mi3d_routes().wide_bn_wgs;
    if (cap < 1) cap = 1;
    return want < 1 ? 1 : (want > cap ? cap : want);
}

int bn_train_stats(int dtype, const void* y, int ycs, int C, int64_t M, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps, float* stat,
                   float* ws, hipStream_t s, int* small_rows) {
    MI3D_CHECK_ARG(C >= 1 && C <= BLK && M >= 1, "bn_train_stats: bad C=%d M=%lld", C, (long long)M);
    bool small = small_rows && bn_small(C, M);
    if (small_rows) *small_rows = 0;
    DISPATCH_T(dtype, T, {
        bool v8 = vec8_ok(C, ycs, ycs, y, y, sizeof(T));
        int G = v8 ? C / 8 : C, R = BLK / G;
        int nblk = reduce_grid(M, R);
        if (small) nblk = bn_small_rows(nblk, C);
        size_t lds = (size_t)2 * R * C * sizeof(float);
        if (v8) bn_stats_kernel<T, 8><<<nblk, BLK, lds, s>>>((const T*)y, ycs, C, M, ws);
        else bn_stats_kernel<T, 1><<<nblk, BLK, lds, s>>>((const T*)y, ycs, C, M, ws);
        MI3D_LAUNCH_CHECK();
        if (small) *small_rows = nblk;
        else {
            bn_stats_finalize_kernel<<<C, FIN_T, 0, s>>>(ws, nblk, C, M, gamma, beta, running_mean, running_var, nbt,
                                                     momentum, eps, stat);
            MI3D_LAUNCH_CHECK();
        }
    });
    return 0;
}

int bn_train_finalize(const float* part, int nblk, int C, int64_t M, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps, float* stat,
                      hipStream_t s) {
    bn_stats_finalize_kernel<<<C, FIN_T, 0, s>>>(part, nblk, C, M, gamma, beta, running_mean, running_var, nbt, momentum, eps, stat);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int bn_eval_stats(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                  float* stat, hipStream_t s) {
    bn_eval_stats_kernel<<<cdiv(C, 64), 64, 0, s>>>(C, gamma, beta, rm, rv, eps, stat);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int bn_apply_relu_drop(int dtype, const void* y, int ycs, int C, int64_t M, int64_t V, float* stat,
                       const float* drop, void* z, int zcs, hipStream_t s, const BnSmall* small) {
    MI3D_CHECK_ARG(C >= 1 && M >= 1, "bn_apply: bad shape");
    MI3D_CHECK_ARG(!drop || M < (1ll << 32), "bn_apply: dropout path needs M < 2^32 (32-bit sample index)");
    BnPart t{};
    if (small) {
        MI3D_CHECK_ARG(small->part && small->nrows >= 1 && small->nrows <= WIDE_ROWS && C <= MAXC_BN && small->gamma && small->beta,
                       "bn_apply: bad partial statistics");
        t = BnPart{small->part, small->nrows, M, small->gamma, small->beta, small->running_mean, small->running_var,
                   small->num_batches_tracked, small->momentum, small->eps};
    }
    DISPATCH_T(dtype, T, {
        bool v8 = vec8_ok(C, ycs, zcs, y, z, sizeof(T));
        if (small && (small->nrows > SMALL_ROWS || small->nrows >= mi3d_routes().wide_min_rows)) {
            if (v8 && dtype == MI3D_BF16 && wide_shape_ok(C) && (int64_t)small->nrows * C <= (int64_t)WIDE_J * WNT * 2) {
                bn_apply_wide_kernel<<<wide_grid(M * (C / 8)), WNT, 0, s>>>((const bf16*)y, ycs, C, M, V, stat, t, drop, (bf16*)z, zcs);
                MI3D_LAUNCH_CHECK();
                return 0;
            }
            MI3D_CHECK_ARG(small->nrows <= SMALL_ROWS, "bn_apply: %d partial rows need the wide kernel (C = %d, 16-byte rows)", small->nrows, C);
        }
        int grid = v8 ? stream_grid(M * (C / 8), C / 8) : stream_grid(M * C, C);
        if (v8 && small) bn_apply_kernel<T, 8, true><<<grid, BLK, 0, s>>>((const T*)y, ycs, C, M, V, stat, t, drop, (T*)z, zcs);
        else if (v8) bn_apply_kernel<T, 8, false><<<grid, BLK, 0, s>>>((const T*)y, ycs, C, M, V, stat, t, drop, (T*)z, zcs);
        else if (small) bn_apply_kernel<T, 1, true><<<grid, BLK, 0, s>>>((const T*)y, ycs, C, M, V, stat, t, drop, (T*)z, zcs);
        else bn_apply_kernel<T, 1, false><<<grid, BLK, 0, s>>>((const T*)y, ycs, C, M, V, stat, t, drop, (T*)z, zcs);
        MI3D_LAUNCH_CHECK();
    });
    return 0;
}

int bn_apply_relu_drop_pool(int dtype, const void* y, int ycs, int C, Geo g, float* stat, const float* drop, void* z, int zcs,
                            void* pooled, int pcs, hipStream_t s, const BnSmall* small) {
    const int64_t M = g.M();
    MI3D_CHECK_ARG(C >= 1 && M >= 1 && g.D % 2 == 0 && g.H % 2 == 0 && g.W % 2 == 0 && M * C < (1ll << 31),
                   "bn_apply_pool: needs even sides and fewer than 2^31 elements");
    BnPart t{};
    if (small) {
        MI3D_CHECK_ARG(small->part && small->nrows >= 1 && small->nrows <= WIDE_ROWS && C <= MAXC_BN && small->gamma && small->beta,
                       "bn_apply_pool: bad partial statistics");
        t = BnPart{small->part, small->nrows, M, small->gamma, small->beta, small->running_mean, small->running_var,
                   small->num_batches_tracked, small->momentum, small->eps};
    }
    DISPATCH_T(dtype, T, {
        bool v8 = vec8_ok(C, ycs, zcs, y, z, sizeof(T)) && pcs % 8 == 0 && ((uintptr_t)pooled % 16 == 0);
        int grid = v8 ? stream_grid(M / 8 * (C / 8), C / 8) : stream_grid(M / 8 * C, C);
        const T* yp = (const T*)y; T* zp = (T*)z; T* pp = (T*)pooled;
        const int G8 = C / 8;
        const bool pair = v8 && (G8 & (G8 - 1)) == 0 && G8 <= 32 && !mi3d_routes().no_pool_pair;
        if (pair) grid = stream_grid(M / 8 * G8 * 2, G8 * 2);
        if (small && (small->nrows > SMALL_ROWS || small->nrows >= mi3d_routes().wide_min_rows)) {
            if (pair && dtype == MI3D_BF16 && wide_shape_ok(C) && (int64_t)small->nrows * C <= (int64_t)WIDE_J * WNT * 2) {
                bn_apply_pool_wide_kernel<<<wide_grid(M / 8 * G8 * 2), WNT, 0, s>>>((const bf16*)y, ycs, C, g.N, g.D, g.H, g.W, stat, t, drop, (bf16*)z, zcs, (bf16*)pooled, pcs);
                MI3D_LAUNCH_CHECK();
                return 0;
            }
            MI3D_CHECK_ARG(small->nrows <= SMALL_ROWS, "bn_apply_pool: %d partial rows need the wide kernel (C = %d)", small->nrows, C);
        }
        if (pair && small) bn_apply_pool_kernel<T, 8, true, true><<<grid, BLK, 0, s>>>(yp, ycs, C, g.N, g.D, g.H, g.W, stat, t, drop, zp, zcs, pp, pcs);
        else if (pair) bn_apply_pool_kernel<T, 8, false, true><<<grid, BLK, 0, s>>>(yp, ycs, C, g.N, g.D, g.H, g.W, stat, t, drop, zp, zcs, pp, pcs);
        else if (v8 && small) bn_apply_pool_kernel<T, 8, true><<<grid, BLK, 0, s>>>(yp, ycs, C, g.N, g.D, g.H, g.W, stat, t, drop, zp, zcs, pp, pcs);
        else if (v8) bn_apply_pool_kernel<T, 8, false><<<grid, BLK, 0, s>>>(yp, ycs, C, g.N, g.D, g.H, g.W, stat, t, drop, zp, zcs, pp, pcs);
        else if (small) bn_apply_pool_kernel<T, 1, true><<<grid, BLK, 0, s>>>(yp, ycs, C, g.N, g.D, g.H, g.W, stat, t, drop, zp, zcs, pp, pcs);
        else bn_apply_pool_kernel<T, 1, false><<<grid, BLK, 0, s>>>(yp, ycs, C, g.N, g.D, g.H, g.W, stat, t, drop, zp, zcs, pp, pcs);
        MI3D_LAUNCH_CHECK();
    });
    return 0;
}

int bn_bwd(int dtype, const void* dz, int dzcs, const void* y, int ycs, int C, int64_t M, int64_t V,
           const float* stat, const float* drop, void* dy, int dycs, float* dgamma, float* dbeta, int accumulate,
           float* ws, hipStream_t s, const SlabJob* extra, const float* skp, int ks, const SlabJob* extra2, int* reduce_only_rows) {
    if (reduce_only_rows) *reduce_only_rows = 0;
    MI3D_CHECK_ARG(C >= 1 && C <= BLK && M >= 1, "bn_bwd: bad C=%d", C);
    MI3D_CHECK_ARG(!drop || M < (1ll << 32), "bn_bwd: dropout path needs M < 2^32 (32-bit sample index)");
    float* part = ws;
    float* coef = ws + (size_t)MAXBLK * 2 * C;
    bool small = bn_small(C, M);
    DISPATCH_T(dtype, T, {
        bool v8 = vec8_ok(C, ycs, dzcs, y, dz, sizeof(T)) && dycs % 8 == 0 && ((uintptr_t)dy % 16 == 0);
        MI3D_CHECK_ARG(!skp || (v8 && ks >= 1), "bn_bwd: split-K source needs the vector path");
        int G = v8 ? C / 8 : C, R = BLK / G;
        int nblk = reduce_grid(M, R);
        if (small) nblk = bn_small_rows(nblk, C);
        size_t lds = (size_t)2 * R * C * sizeof(float);
        if (extra && extra->nblocks > 0) {
            SlabJob j2 = (extra2 && extra2->nblocks > 0) ? *extra2 : SlabJob();
            int tot = nblk + extra->nblocks + j2.nblocks;
            if (v8) bn_bwd_reduce_slab_kernel<T, 8><<<tot, BLK, lds, s>>>(nblk, (const T*)dz, dzcs, (const T*)y, ycs, C, M, V, stat, drop, part, *extra, j2, skp, ks);
            else bn_bwd_reduce_slab_kernel<T, 1><<<tot, BLK, lds, s>>>(nblk, (const T*)dz, dzcs, (const T*)y, ycs, C, M, V, stat, drop, part, *extra, j2, nullptr, 0);
        } else if (v8) bn_bwd_reduce_kernel<T, 8><<<nblk, BLK, lds, s>>>((const T*)dz, dzcs, (const T*)y, ycs, C, M, V, stat, drop, part, skp, ks);
        else bn_bwd_reduce_kernel<T, 1><<<nblk, BLK, lds, s>>>((const T*)dz, dzcs, (const T*)y, ycs, C, M, V, stat, drop, part, nullptr, 0);
        MI3D_LAUNCH_CHECK();
        if (!small) {
            bn_bwd_finalize_kernel<<<C, FIN_T, 0, s>>>(part, nblk, C, M, stat, dgamma, dbeta, accumulate, coef);
            MI3D_LAUNCH_CHECK();
        }
        // apply on load: the consumer (the input-gradient conv) finishes the rows in ITS prologue, computes dy while staging
        // and writes dgamma / dbeta; nothing more to launch here
        if (reduce_only_rows && small && v8) { *reduce_only_rows = nblk; return 0; }
        int grid = v8 ? stream_grid(M * (C / 8), C / 8) : stream_grid(M * C, C);
        const T* dzp = (const T*)dz; const T* yp = (const T*)y; T* dyp = (T*)dy;
        if (small && v8) bn_bwd_apply_kernel<T, 8, true><<<grid, BLK, 0, s>>>(dzp, dzcs, yp, ycs, C, M, V, stat, part, nblk, dgamma, dbeta, accumulate, drop, dyp, dycs);
        else if (small) bn_bwd_apply_kernel<T, 1, true><<<grid, BLK, 0, s>>>(dzp, dzcs, yp, ycs, C, M, V, stat, part, nblk, dgamma, dbeta, accumulate, drop, dyp, dycs);
        else if (v8) bn_bwd_apply_kernel<T, 8, false><<<grid, BLK, 0, s>>>(dzp, dzcs, yp, ycs, C, M, V, stat, coef, 0, nullptr, nullptr, 0, drop, dyp, dycs);
        else bn_bwd_apply_kernel<T, 1, false><<<grid, BLK, 0, s>>>(dzp, dzcs, yp, ycs, C, M, V, stat, coef, 0, nullptr, nullptr, 0, drop, dyp, dycs);
        MI3D_LAUNCH_CHECK();
    });
    return 0;
}

int bn_train_stats_splitk(const float* skp, int ks, const float* bias, void* y, int ycs, int C, int64_t M, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                          float* stat, float* ws, hipStream_t s, int* small_rows) {
    MI3D_CHECK_ARG(C % 8 == 0 && C / 8 <= BLK && ycs % 8 == 0 && ((uintptr_t)y % 16 == 0) && ks >= 1 && M >= 1,
                   "bn_train_stats_splitk: unsupported shape C=%d ycs=%d", C, ycs);
    bool small = small_rows && bn_small(C, M);
    if (small_rows) *small_rows = 0;
    int G = C / 8, R = BLK / G;
    int64_t want = (M + R - 1) / R;                 // one row per thread: the ks fp32 partial reads dominate, spread them wide
    int nblk = (int)(want > MAXBLK ? MAXBLK : want);
    if (small) nblk = bn_small_rows(nblk, C);
    size_t lds = (size_t)2 * R * C * sizeof(float);
    bn_stats_splitk_kernel<<<nblk, BLK, lds, s>>>(skp, ks, bias, (bf16*)y, ycs, C, M, ws);
    MI3D_LAUNCH_CHECK();
    if (small) { *small_rows = nblk; return 0; }
    bn_stats_finalize_kernel<<<C, FIN_T, 0, s>>>(ws, nblk, C, M, gamma, beta, running_mean, running_var, nbt, momentum, eps, stat);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int bn_deferred_apply(const BnDeferJobs& J, hipStream_t s) {
    MI3D_CHECK_ARG(J.n >= 1 && J.n <= MAX_FOLD_JOBS && J.momentum >= 0.f, "bn_deferred_apply: bad job list");
    bn_deferred_apply_kernel<<<J.n, 256, 0, s>>>(J);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int bn_fold_all(const BnFoldJobs& J, hipStream_t s) {
    MI3D_CHECK_ARG(J.n >= 1 && J.n <= MAX_FOLD_JOBS, "bn_fold_all: bad job count %d", J.n);
    bn_fold_all_kernel<<<J.n, 256, 0, s>>>(J);
    MI3D_LAUNCH_CHECK();
    return 0;
}
