// misc.hip — small utility kernels: slab reduction, NCDHW<->channels-last conversion, global average pool,
// the DANN discriminator's Linear layers, row-wise softmax-CE, fused flat AdamW, Dropout3d mask RNG.
// Reference call sites: models/unet_dann.py:79 (GAP); train_dann.py:22-49,283 (GRL, MLP, domain CE);
// train_unet.py:378 (AdamW); models/unet.py:14,18 (Dropout3d).
#include <stdarg.h>
#include <stdio.h>

#include "ops.h"

static thread_local char g_err[512] = "";
void mi3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* mi3d_last_error(void) { return g_err; }

namespace {
constexpr int BLK = 256;
inline int sgrid(int64_t total, int cap = 2048) {
    int64_t w = (total + BLK - 1) / BLK;
    return (int)(w < 1 ? 1 : (w > cap ? cap : w));
}

// fixed-order parallel slab sum: block = EW elements x 256/EW slab groups (each thread sums every SG-th slab, then a
// tree).  EW = 4 for tiny slabs (1x1x1 head: 68 floats x 512 slabs) so the parallelism comes from the slab dimension.
template <int EW>
__global__ __launch_bounds__(BLK) void slab_reduce_kernel(const float* __restrict__ slabs, int nslab, int64_t slab_sz, int64_t nW,
                                                          float* __restrict__ dW, float* __restrict__ db, int accumulate) {
    constexpr int SG = BLK / EW;
    __shared__ float red[SG][EW];
    int e = threadIdx.x % EW, sg = threadIdx.x / EW;
    int64_t i = (int64_t)blockIdx.x * EW + e;
    float s = 0.f;
    if (i < slab_sz)
        for (int b = sg; b < nslab; b += SG) s += slabs[(int64_t)b * slab_sz + i];
    red[sg][e] = s;
    __syncthreads();
    if (sg == 0 && i < slab_sz) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < SG; k += 8)
            t += ((red[k][e] + red[k + 1][e]) + (red[k + 2][e] + red[k + 3][e])) +
                 ((red[k + 4][e] + red[k + 5][e]) + (red[k + 6][e] + red[k + 7][e]));
        if (i < nW) { if (dW) dW[i] = accumulate ? dW[i] + t : t; }
        else if (db) { db[i - nW] = accumulate ? db[i - nW] + t : t; }
    }
}

template <typename T>
__global__ void ncdhw_to_ndhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int dcs, int C, int N, int64_t V) {
    int64_t total = (int64_t)N * V * C;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLK) {
        int c = (int)(i % C); int64_t m = i / C; int64_t n = m / V, v = m - n * V;
        dst[m * dcs + c] = from_f<T>(src[((int64_t)n * C + c) * V + v]);
    }
}
template <typename T>
__global__ void ndhwc_to_ncdhw_kernel(const T* __restrict__ src, int scs, float* __restrict__ dst, int C, int N, int64_t V) {
    int64_t total = (int64_t)N * V * C;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLK) {
        int64_t v = i % V; int64_t r = i / V; int c = (int)(r % C); int64_t n = r / C;
        dst[i] = to_f<T>(src[(n * V + v) * scs + c]);
    }
}

// one block per (n, 64-channel group): lanes along channels, waves along voxels
template <typename T>
__global__ __launch_bounds__(BLK) void gap_fwd_kernel(const T* __restrict__ z, int zcs, int C, int64_t V, float* __restrict__ out) {
    __shared__ float red[4][64];
    int n = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int64_t v = wave; v < V; v += 4) s += to_f<T>(z[((int64_t)n * V + v) * zcs + c]);
    red[wave][threadIdx.x & 63] = s;
    __syncthreads();
    if (wave == 0 && c < C) {
        double t = (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x];
        out[(int64_t)n * C + c] = (float)(t / (double)V);
    }
}
template <typename T>
__global__ void gap_bwd_kernel(const float* __restrict__ g, float scale, T* __restrict__ dz, int dzcs, int C, int N, int64_t V,
                               int accumulate) {
    int64_t total = (int64_t)N * V * C;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLK) {
        int c = (int)(i % C); int64_t m = i / C; int64_t n = m / V;
        float add = scale * g[n * C + c] / (float)V;
        T* p = dz + m * dzcs + c;
        *p = from_f<T>(accumulate ? to_f<T>(*p) + add : add);
    }
}

// y[m][o] = act(b[o] + sum_k x[m][k] w[o][k]) * drop[m][o]; one wave per output element
__global__ __launch_bounds__(BLK) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int M, int K,
                                                         int Nout, int relu, const float* __restrict__ drop) {
    int wid = (blockIdx.x * BLK + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wid >= M * Nout) return;
    int m = wid / Nout, o = wid - m * Nout;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s = fmaf(x[(int64_t)m * K + k], w[(int64_t)o * K + k], s);
    s = wave_sum(s);
    if (lane == 0) {
        s += b ? b[o] : 0.f;
        if (relu) s = s > 0.f ? s : 0.f;
        if (drop) s *= drop[(int64_t)m * Nout + o];
        y[(int64_t)m * Nout + o] = s;
    }
}
// gpre[m][o] = gy * drop * [y > 0]  (in place into ws), then gx, gw, gb from gpre
__global__ void linear_bwd_pre_kernel(const float* __restrict__ y, const float* __restrict__ gy, int n, int relu,
                                      const float* __restrict__ drop, float* __restrict__ gpre) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    float g = gy[i];
    if (drop) g *= drop[i];
    if (relu && !(y[i] > 0.f)) g = 0.f;
    gpre[i] = g;
}
__global__ __launch_bounds__(BLK) void linear_bwd_x_kernel(const float* __restrict__ gpre, const float* __restrict__ w,
                                                           float* __restrict__ gx, int M, int K, int Nout, float scale) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= M * K) return;
    int m = i / K, k = i - m * K;
    float s = 0.f;
    for (int o = 0; o < Nout; o++) s = fmaf(gpre[(int64_t)m * Nout + o], w[(int64_t)o * K + k], s);
    gx[i] = s * scale;
}
__global__ __launch_bounds__(BLK) void linear_bwd_w_kernel(const float* __restrict__ gpre, const float* __restrict__ x,
                                                           float* __restrict__ gw, float* __restrict__ gb, int M, int K,
                                                           int Nout, int accumulate) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i < Nout * K) {
        int o = i / K, k = i - o * K;
        float s = 0.f;
        for (int m = 0; m < M; m++) s = fmaf(gpre[(int64_t)m * Nout + o], x[(int64_t)m * K + k], s);
        gw[i] = accumulate ? gw[i] + s : s;
    }
    if (i < Nout && gb) {
        float s = 0.f;
        for (int m = 0; m < M; m++) s += gpre[(int64_t)m * Nout + i];
        gb[i] = accumulate ? gb[i] + s : s;
    }
}

// nn.CrossEntropyLoss (mean) over M rows, C <= 64 classes; dlogits = scale * (softmax - onehot) / M
__global__ void softmax_ce_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int M, int C,
                                       float* loss, float* dlogits, float scale) {
    int lane = threadIdx.x;
    double tot = 0.0;
    for (int m = 0; m < M; m++) {
        float z = lane < C ? logits[(int64_t)m * C + lane] : -INFINITY;
        float mx = z;
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float e = lane < C ? expf(z - mx) : 0.f;
        float se = wave_sum(e);
        int t = (int)labels[m];
        float zt = __shfl(z, t, 64);
        tot += (double)(mx + logf(se) - zt);
        if (dlogits && lane < C) dlogits[(int64_t)m * C + lane] = scale * (e / se - (lane == t ? 1.f : 0.f)) / (float)M;
    }
    if (lane == 0 && loss) *loss = (float)(tot / (double)M);
}

// (Round 3: folding the step increment into this kernel -- last block to finish, atomicInc ticket -- made it 58 instead of 29 us:
// 4096 arrivals on one word serialise at ~88 per us.  The one-thread step_inc_kernel launch (4 us) stays.)
// 16-byte accesses, the next iteration's four loads in flight under the current update, and the first loads issued BEFORE the
// double-precision bias corrections (two pow + sqrt, ~1 us of dependent arithmetic that every wave repeats): round 3, 32.5 -> see
// DESIGN.  Per element the arithmetic is unchanged (bit-identical parameters).
__device__ __forceinline__ void adamw_one(float& pi, float gi, float& mi, float& vi, float lr, float b1, float b2, float eps, float wd,
                                          float grad_scale, float bc2s, float step_size) {
    gi = gi * grad_scale;
    pi = pi * (1.f - lr * wd);
    mi = b1 * mi + (1.f - b1) * gi;       // torch: exp_avg.lerp_(grad, 1-b1)
    vi = b2 * vi + (1.f - b2) * gi * gi;
    float denom = sqrtf(vi) / bc2s + eps;
    pi = pi - step_size * (mi / denom);
}
__global__ __launch_bounds__(BLK) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                                    float grad_scale, const int64_t* __restrict__ step_dev) {
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const int64_t n4 = vec ? n >> 2 : 0, stride = (int64_t)gridDim.x * BLK;
    int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
    float4 g4, p4, m4, v4;
    bool have = i < n4;
    if (have) {
        g4 = reinterpret_cast<const float4*>(g)[i]; p4 = reinterpret_cast<const float4*>(p)[i];
        m4 = reinterpret_cast<const float4*>(m)[i]; v4 = reinterpret_cast<const float4*>(v)[i];
    }
    double step = (double)(*step_dev + 1);
    float bc1 = (float)(1.0 - pow((double)b1, step));
    float bc2s = (float)sqrt(1.0 - pow((double)b2, step));
    float step_size = lr / bc1;
    while (have) {
        int64_t j = i + stride;
        bool have2 = j < n4;
        float4 g5, p5, m5, v5;
        if (have2) {
            g5 = reinterpret_cast<const float4*>(g)[j]; p5 = reinterpret_cast<const float4*>(p)[j];
            m5 = reinterpret_cast<const float4*>(m)[j]; v5 = reinterpret_cast<const float4*>(v)[j];
        }
        adamw_one(p4.x, g4.x, m4.x, v4.x, lr, b1, b2, eps, wd, grad_scale, bc2s, step_size);
        adamw_one(p4.y, g4.y, m4.y, v4.y, lr, b1, b2, eps, wd, grad_scale, bc2s, step_size);
        adamw_one(p4.z, g4.z, m4.z, v4.z, lr, b1, b2, eps, wd, grad_scale, bc2s, step_size);
        adamw_one(p4.w, g4.w, m4.w, v4.w, lr, b1, b2, eps, wd, grad_scale, bc2s, step_size);
        reinterpret_cast<float4*>(m)[i] = m4;
        reinterpret_cast<float4*>(v)[i] = v4;
        reinterpret_cast<float4*>(p)[i] = p4;
        i = j; have = have2; g4 = g5; p4 = p5; m4 = m5; v4 = v5;
    }
    // scalar tail (and the whole range when a pointer is not 16-byte aligned)
    for (int64_t k = n4 * 4 + (int64_t)blockIdx.x * BLK + threadIdx.x; k < n; k += stride) {
        float pi = p[k], mi = m[k], vi = v[k];
        adamw_one(pi, g[k], mi, vi, lr, b1, b2, eps, wd, grad_scale, bc2s, step_size);
        m[k] = mi; v[k] = vi; p[k] = pi;
    }
}
__global__ void step_inc_kernel(int64_t* step_dev) { *step_dev += 1; }

// counter-based RNG (splitmix64 finaliser over (seed, counter+i)); statistically Bernoulli(1-p), NOT torch's Philox stream
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void dropout_scales_kernel(float* __restrict__ out, int64_t n, float p, const uint64_t* __restrict__ state) {
    uint64_t seed = state[0], ctr = state[1];
    float keep_scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        uint64_t r = mix64(mix64(seed) ^ (ctr + (uint64_t)i));
        float u = (float)(r >> 40) * (1.f / 16777216.f);
        out[i] = u >= p ? keep_scale : 0.f;
    }
}
__global__ void dropout_advance_kernel(uint64_t* state, int64_t n) { state[1] += (uint64_t)n; }

__global__ void fill_kernel(float* p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) p[i] = v;
}
__global__ void scale_add_kernel(float* dst, const float* src, int64_t n, float a, float b) {
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) dst[i] = a * dst[i] + b * src[i];
}
// y = a * (*a_dev or 1) * x   (gradient reversal: a = -lambda; row-CE backward: a_dev = upstream gradient)
__global__ void scale_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float a, const float* __restrict__ a_dev) {
    float f = a * (a_dev ? *a_dev : 1.f);
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) y[i] = f * x[i];
}
}  // namespace

int scale_f32(const float* x, float* y, int64_t n, float a, const float* a_dev, hipStream_t s) {
    scale_kernel<<<sgrid(n), BLK, 0, s>>>(x, y, n, a, a_dev);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int slab_reduce(const float* slabs, int nslab, int64_t slab_sz, int64_t nW, float* dW, float* db, int accumulate,
                hipStream_t s) {
    if (slab_sz < 128) slab_reduce_kernel<1><<<(unsigned)slab_sz, BLK, 0, s>>>(slabs, nslab, slab_sz, nW, dW, db, accumulate);
    else if (slab_sz < 1024) slab_reduce_kernel<4><<<cdiv(slab_sz, 4), BLK, 0, s>>>(slabs, nslab, slab_sz, nW, dW, db, accumulate);
    else if (slab_sz < (16 << 10)) slab_reduce_kernel<8><<<cdiv(slab_sz, 8), BLK, 0, s>>>(slabs, nslab, slab_sz, nW, dW, db, accumulate);
    else slab_reduce_kernel<32><<<cdiv(slab_sz, 32), BLK, 0, s>>>(slabs, nslab, slab_sz, nW, dW, db, accumulate);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int ncdhw_to_ndhwc(int dtype, const float* src, void* dst, int dcs, int C, int N, int64_t V, hipStream_t s) {
    DISPATCH_T(dtype, T, { ncdhw_to_ndhwc_kernel<T><<<sgrid((int64_t)N * V * C, 4096), BLK, 0, s>>>(src, (T*)dst, dcs, C, N, V); });
    MI3D_LAUNCH_CHECK();
    return 0;
}
int ndhwc_to_ncdhw(int dtype, const void* src, int scs, float* dst, int C, int N, int64_t V, hipStream_t s) {
    DISPATCH_T(dtype, T, { ndhwc_to_ncdhw_kernel<T><<<sgrid((int64_t)N * V * C, 4096), BLK, 0, s>>>((const T*)src, scs, dst, C, N, V); });
    MI3D_LAUNCH_CHECK();
    return 0;
}

int gap_fwd(int dtype, const void* z, int zcs, int C, int N, int64_t V, float* out, hipStream_t s) {
    dim3 grid((unsigned)N, (unsigned)cdiv(C, 64));
    DISPATCH_T(dtype, T, { gap_fwd_kernel<T><<<grid, BLK, 0, s>>>((const T*)z, zcs, C, V, out); });
    MI3D_LAUNCH_CHECK();
    return 0;
}
int gap_bwd(int dtype, const float* g, float scale, void* dz, int dzcs, int C, int N, int64_t V, int accumulate,
            hipStream_t s) {
    DISPATCH_T(dtype, T, { gap_bwd_kernel<T><<<sgrid((int64_t)N * V * C), BLK, 0, s>>>(g, scale, (T*)dz, dzcs, C, N, V, accumulate); });
    MI3D_LAUNCH_CHECK();
    return 0;
}

int linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int K, int Nout, int relu,
               const float* drop, hipStream_t s) {
    linear_fwd_kernel<<<cdiv((int64_t)M * Nout * 64, BLK), BLK, 0, s>>>(x, w, b, y, M, K, Nout, relu, drop);
    MI3D_LAUNCH_CHECK();
    return 0;
}
int linear_bwd(const float* x, const float* w, const float* y, const float* gy, int M, int K, int Nout, int relu,
               const float* drop, float* gx, float* gw, float* gb, int accumulate, float gx_scale, float* ws,
               hipStream_t s) {
    linear_bwd_pre_kernel<<<cdiv((int64_t)M * Nout, BLK), BLK, 0, s>>>(y, gy, M * Nout, relu, drop, ws);
    MI3D_LAUNCH_CHECK();
    if (gx) {
        linear_bwd_x_kernel<<<cdiv((int64_t)M * K, BLK), BLK, 0, s>>>(ws, w, gx, M, K, Nout, gx_scale);
        MI3D_LAUNCH_CHECK();
    }
    if (gw) {
        linear_bwd_w_kernel<<<cdiv((int64_t)Nout * K, BLK), BLK, 0, s>>>(ws, x, gw, gb, M, K, Nout, accumulate);
        MI3D_LAUNCH_CHECK();
    }
    return 0;
}
int softmax_ce_rows(const float* logits, const int64_t* labels, int M, int C, float* loss, float* dlogits, float scale,
                    hipStream_t s) {
    MI3D_CHECK_ARG(C <= 64, "softmax_ce_rows: C=%d > 64", C);
    softmax_ce_rows_kernel<<<1, 64, 0, s>>>(logits, labels, M, C, loss, dlogits, scale);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
               float wd, float grad_scale, int64_t* step_dev, hipStream_t s, int increment) {
    if (n > 0) {
        adamw_kernel<<<sgrid((n + 3) / 4, 2048), BLK, 0, s>>>(p, g, m, v, n, lr, b1, b2, eps, wd, grad_scale, step_dev);
        MI3D_LAUNCH_CHECK();
    }
    if (increment) {
        step_inc_kernel<<<1, 1, 0, s>>>(step_dev);
        MI3D_LAUNCH_CHECK();
    }
    return 0;
}

int dropout_scales(float* out, int64_t n, float p, uint64_t* state_dev, hipStream_t s) {
    dropout_scales_kernel<<<sgrid(n, 256), BLK, 0, s>>>(out, n, p, state_dev);
    MI3D_LAUNCH_CHECK();
    dropout_advance_kernel<<<1, 1, 0, s>>>(state_dev, n);
    MI3D_LAUNCH_CHECK();
    return 0;
}

// Stand-in for a collective kernel on a 1-GPU box: `wgs` workgroups of 512 threads with the register footprint of an RCCL
// all-reduce kernel (128 VGPRs) hold their CU slots for `usec` microseconds (s_memrealtime: 100 MHz) and stream through
// `buf` meanwhile.  Used by bench.py --emulate-comm to measure what a resident collective costs the persistent grids.
// READ-ONLY on `buf` (the keep-alive used to be a conditional store, and callers pass live tensors): the
// loaded values are pinned with an empty asm that names them as an input.
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(128))) void occupy_kernel(const float* buf, int64_t n, int usec) {
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), ticks = (unsigned long long)usec * 100ull;
    float acc = 0.f;
    int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (buf && n > 0) { acc += buf[i % n]; i += (int64_t)gridDim.x * 512; }
        __builtin_amdgcn_s_sleep(8);
    }
    asm volatile("" :: "v"(acc));                      // keep the loads alive without writing anything
}
// ---- stream-to-stream ordering through a memory flag ------------------------------------------------------------------------
// A hardware cross-queue dependency (hipStreamWaitEvent on an event of another hardware queue) costs the WAITING stream 30-45 us on
// this runtime even when the event fired long ago (profiles/r04_experiments_dp_marks.txt: fork + join around NOTHING = +47 us per
// step).  Where the producer is known to finish early -- the big gradient bucket's all-reduce ends ~400 us before the backward does
// -- the consumer polls a counter in memory instead: the producing stream bumps it behind its last kernel (that kernel's end has
// made its writes visible device-wide), the consuming stream runs a one-wave kernel that returns as soon as the counter has reached
// the value.  Bounded: after `timeout_us` the waiter gives up and raises the error word (flag[1]), which the host checks lazily.
__global__ void flag_set_kernel(long long* flag, long long value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void flag_wait_kernel(long long* flag, long long value, long long timeout_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
        if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) > timeout_ticks) {
            __hip_atomic_store(flag + 1, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);       // timed out waiting for `value`
            return;
        }
        __builtin_amdgcn_s_sleep(4);
    }
}
int flag_set(int64_t* flag, int64_t value, hipStream_t s) {
    flag_set_kernel<<<1, 1, 0, s>>>((long long*)flag, (long long)value);
    MI3D_LAUNCH_CHECK();
    return 0;
}
int flag_wait(int64_t* flag, int64_t value, int64_t timeout_us, hipStream_t s) {
    flag_wait_kernel<<<1, 64, 0, s>>>((long long*)flag, (long long)value, (long long)timeout_us * 100);     // s_memrealtime: 100 MHz
    MI3D_LAUNCH_CHECK();
    return 0;
}

int occupy_cus(int wgs, int usec, float* buf, int64_t n, hipStream_t s) {
    MI3D_CHECK_ARG(wgs >= 1 && wgs <= 256 && usec >= 1 && usec <= 5000, "occupy_cus: wgs in [1,256], usec in [1,5000]");
    occupy_kernel<<<wgs, 512, 0, s>>>(buf, n, usec);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int fill_f32(float* p, int64_t n, float v, hipStream_t s) {
    fill_kernel<<<sgrid(n), BLK, 0, s>>>(p, n, v);
    MI3D_LAUNCH_CHECK();
    return 0;
}
int scale_add_f32(float* dst, const float* src, int64_t n, float a, float b, hipStream_t s) {
    scale_add_kernel<<<sgrid(n), BLK, 0, s>>>(dst, src, n, a, b);
    MI3D_LAUNCH_CHECK();
    return 0;
}
