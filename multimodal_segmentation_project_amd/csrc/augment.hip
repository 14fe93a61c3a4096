// augment.hip — the training-set augmentation chain of the input pipeline on the GPU:
// combined_transform() (utils/dataloader.py:223-262, used at train_unet.py:361) = MONAI RandBiasField -> RandGaussianNoise
// -> RandAdjustContrast -> RandHistogramShift -> RandCoarseDropout.  The host draws the random parameters (augment.py);
// this file is the per-voxel arithmetic, HBM-bound.
//
// Contrast and histogram shift each need the min / max of THEIR input over the whole volume, so the chain is cut into at
// most three elementwise stages — {bias, noise} | {contrast} | {hist} (+ holes on the last one) — one read + one write
// each; a stage also leaves per-block (min, max) of what it wrote, which the next stage's blocks reduce in their
// prologue (deterministic, no atomics, no finalize launch).  Only when the first active transform is contrast / hist
// does a read-only min/max pass run first.  Worst case (all five drawn): 3 reads + 3 writes of the volume.
#include "ops.h"
#include "../../include/mi3d.h"

namespace {
constexpr int BLK = 256;
constexpr int MAXBLK = 2048;

struct VolGeo { int C, D, H, W; int64_t n; };
struct Stage { int bias, noise, contrast, hist, holes, write, emit, stats_in, uniform_cp; };

__device__ __forceinline__ uint64_t mix64a(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float wave_minf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_maxf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// np.linspace(-1, 1, n, dtype=float32)[i] as a double: float64 arithmetic, last point exact, cast to float32
__device__ __forceinline__ double lin_coord(int i, int n) {
    if (n == 1) return -1.0;
    if (i == n - 1) return 1.0;
    return (double)(float)(-1.0 + (double)i * (2.0 / (double)(n - 1)));
}
__device__ __forceinline__ void legendre4(double x, double* p) {
    p[0] = 1.0; p[1] = x; p[2] = 0.5 * (3.0 * x * x - 1.0); p[3] = 0.5 * (5.0 * x * x * x - 3.0 * x);
}

// KIND: 0 read-only min/max or copy (+ holes), 1 bias / noise, 2 contrast, 3 histogram shift — each instantiation carries
// only its own arithmetic (the float64 bias field costs registers the others do not need)
enum { K_PLAIN = 0, K_BIASNOISE = 1, K_CONTRAST = 2, K_HIST = 3 };

// exp(f) in float64 for the bias field: 2^k * Taylor_11(r), |r| <= ln2 / 2 -> relative error < 1e-13 (the product with
// the voxel is rounded to float32 afterwards; libm's exp handles specials this path cannot produce and costs 3x more)
__device__ __forceinline__ double exp_field(double f) {
    const double k = rint(f * 1.4426950408889634);
    double r = fma(k, -0.6931471803691238, f);
    r = fma(k, -1.9082149292705877e-10, r);
    double q = 1.0 / 39916800.0;
    q = fma(q, r, 1.0 / 3628800.0); q = fma(q, r, 1.0 / 362880.0); q = fma(q, r, 1.0 / 40320.0);
    q = fma(q, r, 1.0 / 5040.0);    q = fma(q, r, 1.0 / 720.0);    q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0);      q = fma(q, r, 1.0 / 6.0);      q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);             q = fma(q, r, 1.0);
    return ldexp(q, (int)k);
}
// N(0, 1) for voxel `idx`: one 64-bit draw per PAIR of voxels, Box-Muller's cosine branch for the even and sine branch
// for the odd one (native log2 / sqrt / sin / cos units; sin and cos take their argument in revolutions)
__device__ __forceinline__ float normal_at(uint64_t seed_mixed, uint64_t idx) {
    uint64_t rr = mix64a(seed_mixed ^ (idx >> 1));
    float u1 = ((float)(rr >> 40) + 1.f) * (1.f / 16777216.f);            // (0, 1]
    float u2 = (float)((rr >> 16) & 0xffffffu) * (1.f / 16777216.f);      // [0, 1)
    float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1), ln = log2 * ln2
    return rad * ((idx & 1) ? __builtin_amdgcn_sinf(u2) : __builtin_amdgcn_cosf(u2));
}

template <int VEC, int KIND>
__global__ __launch_bounds__(BLK) void aug_stage_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        const float* __restrict__ noise, VolGeo g, Stage st, mi3d_aug_params p,
                                                        const float2* __restrict__ part_in, int n_part_in,
                                                        float2* __restrict__ part_out) {
    __shared__ float s_mm[2][BLK / 64];
    __shared__ float s_stat[2];
    __shared__ float s_xp[MI3D_AUG_MAX_CP], s_fp[MI3D_AUG_MAX_CP], s_m[MI3D_AUG_MAX_CP], s_b[MI3D_AUG_MAX_CP];
    const int tid = threadIdx.x;
    float mn = 0.f, mx = 0.f;
    if (st.stats_in) {                                   // min / max of this stage's input from the producer's partials
        float a = INFINITY, b = -INFINITY;
        for (int i = tid; i < n_part_in; i += BLK) { float2 q = part_in[i]; a = fminf(a, q.x); b = fmaxf(b, q.y); }
        a = wave_minf(a); b = wave_maxf(b);
        if ((tid & 63) == 0) { s_mm[0][tid >> 6] = a; s_mm[1][tid >> 6] = b; }
        __syncthreads();
        if (tid == 0) {
            s_stat[0] = fminf(fminf(s_mm[0][0], s_mm[0][1]), fminf(s_mm[0][2], s_mm[0][3]));
            s_stat[1] = fmaxf(fmaxf(s_mm[1][0], s_mm[1][1]), fmaxf(s_mm[1][2], s_mm[1][3]));
        }
        __syncthreads();
        mn = s_stat[0]; mx = s_stat[1];
        __syncthreads();
    }
    const float range = mx - mn;
    const bool hist = KIND == K_HIST && mn != mx;        // RandHistogramShift returns the image unchanged when flat
    if (hist) {
        if (tid < p.n_cp) { s_xp[tid] = p.ref_cp[tid] * range + mn; s_fp[tid] = p.flt_cp[tid] * range + mn; }
        __syncthreads();
        if (tid < p.n_cp - 1) {
            float m = (s_fp[tid + 1] - s_fp[tid]) / (s_xp[tid + 1] - s_xp[tid]);
            s_m[tid] = m;
            s_b[tid] = s_fp[tid] - m * s_xp[tid];
        }
        __syncthreads();
    }
    const float cden = range + 1e-7f;
    const uint64_t seed_mixed = mix64a(p.noise_seed);
    const bool uniform_cp = st.uniform_cp != 0;
    const float inv_step = (float)(p.n_cp - 1) / range;
    float omin = INFINITY, omax = -INFINITY;
    const int64_t nvec = g.n / VEC;
    const int Wv = g.W / VEC;
    const bool need_pos = (KIND == K_BIASNOISE && st.bias) || st.holes;
    for (int64_t iv = (int64_t)blockIdx.x * BLK + tid; iv < nvec; iv += (int64_t)gridDim.x * BLK) {
        int w0 = 0, h = 0, d = 0;
        if (need_pos) {                                   // 32-bit index arithmetic: the host checked n < 2^31
            unsigned r = (unsigned)iv / (unsigned)Wv;
            w0 = (int)((unsigned)iv - r * (unsigned)Wv) * VEC;
            unsigned r2 = r / (unsigned)g.H;
            h = (int)(r - r2 * (unsigned)g.H);
            d = (int)(r2 % (unsigned)g.D);
        }
        float v[VEC];
        if constexpr (VEC == 4) { float4 q = ((const float4*)in)[iv]; v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else v[0] = in[iv];
        if (KIND == K_BIASNOISE && st.bias) {
            double pd[4], ph[4], A[4] = {0.0, 0.0, 0.0, 0.0};
            legendre4(lin_coord(d, g.D), pd);
            legendre4(lin_coord(h, g.H), ph);
            int c = 0;                                    // host re-laid the coefficients out for degree 3: constant bounds
#pragma unroll
            for (int i = 0; i <= 3; i++)
#pragma unroll
                for (int j = 0; j <= 3 - i; j++) {
                    double pij = pd[i] * ph[j];
#pragma unroll
                    for (int k = 0; k <= 3 - i - j; k++) A[k] = fma(p.bias_coeff[c++], pij, A[k]);
                }
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                double pw[4];
                legendre4(lin_coord(w0 + e, g.W), pw);
                double f = A[0] + A[1] * pw[1] + A[2] * pw[2] + A[3] * pw[3];
                v[e] = (float)((double)v[e] * exp_field(f));
            }
        }
        if (KIND == K_BIASNOISE && st.noise) {
            if (noise) {
                if constexpr (VEC == 4) { float4 q = ((const float4*)noise)[iv]; v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w; }
                else v[0] += noise[iv];
            } else {
#pragma unroll
                for (int e = 0; e < VEC; e++)
                    v[e] += p.noise_mean + p.noise_std * normal_at(seed_mixed, (uint64_t)(iv * VEC + e));
            }
        }
        if (KIND == K_CONTRAST) {
            // base in [0, 1): x ** gamma = exp2(gamma * log2 x) on the native transcendental units; the absolute error of
            // the result is bounded by ~ulp(log2 x) * x**gamma * |log2 x| * gamma * ln 2 < 3e-8 for any x in (0, 1]
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                float t = (v[e] - mn) / cden;
                float y = t > 0.f ? __builtin_amdgcn_exp2f(p.gamma * __builtin_amdgcn_logf(t)) : 0.f;
                v[e] = y * range + mn;
            }
        }
        if (hist) {
            const int nm = p.n_cp - 1;
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                float x = v[e];
                int idx;                                  // searchsorted(xp, x, left) - 1, clipped to [0, nm - 1]
                if (uniform_cp) {                         // linspace control points: arithmetic guess, exact fix-up
                    idx = (int)((x - mn) * inv_step);
                    idx = idx < 0 ? 0 : (idx > nm - 1 ? nm - 1 : idx);
                    idx -= (idx > 0 && !(s_xp[idx] < x)) ? 1 : 0;
                    idx += (idx < nm - 1 && s_xp[idx + 1] < x) ? 1 : 0;
                } else {
                    idx = 0;
                    for (int i = 0; i < p.n_cp; i++) idx += s_xp[i] < x ? 1 : 0;
                    idx = idx - 1 < 0 ? 0 : (idx - 1 > nm - 1 ? nm - 1 : idx - 1);
                }
                float f = s_m[idx] * x + s_b[idx];
                f = x < s_xp[0] ? s_fp[0] : f;
                f = x > s_xp[nm] ? s_fp[nm] : f;
                v[e] = f;
            }
        }
        if (st.holes) {
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                bool inside = false;
                for (int q = 0; q < p.n_holes; q++)
                    inside |= d >= p.hole_lo[q][0] && d < p.hole_lo[q][0] + p.hole_size[0] && h >= p.hole_lo[q][1] &&
                              h < p.hole_lo[q][1] + p.hole_size[1] && w0 + e >= p.hole_lo[q][2] &&
                              w0 + e < p.hole_lo[q][2] + p.hole_size[2];
                v[e] = inside ? p.fill_value : v[e];
            }
        }
        if (st.emit) {
#pragma unroll
            for (int e = 0; e < VEC; e++) { omin = fminf(omin, v[e]); omax = fmaxf(omax, v[e]); }
        }
        if (st.write) {
            if constexpr (VEC == 4) ((float4*)out)[iv] = make_float4(v[0], v[1], v[2], v[3]);
            else out[iv] = v[0];
        }
    }
    if (st.emit) {
        omin = wave_minf(omin); omax = wave_maxf(omax);
        if ((tid & 63) == 0) { s_mm[0][tid >> 6] = omin; s_mm[1][tid >> 6] = omax; }
        __syncthreads();
        if (tid == 0)
            part_out[blockIdx.x] = make_float2(fminf(fminf(s_mm[0][0], s_mm[0][1]), fminf(s_mm[0][2], s_mm[0][3])),
                                               fmaxf(fmaxf(s_mm[1][0], s_mm[1][1]), fmaxf(s_mm[1][2], s_mm[1][3])));
    }
}

struct HoleLo { int32_t v[MI3D_AUG_MAX_HOLES * 3]; };
__global__ __launch_bounds__(BLK) void fill_boxes_i64_kernel(int64_t* __restrict__ lab, VolGeo g, int n_holes, int3 size, HoleLo lo,
                                                             int64_t fill) {
    const int64_t per = (int64_t)size.x * size.y * size.z;
    const int64_t total = per * n_holes * g.C;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLK) {
        int64_t r = i;
        const int w = (int)(r % size.z); r /= size.z;
        const int h = (int)(r % size.y); r /= size.y;
        const int d = (int)(r % size.x); r /= size.x;
        const int q = (int)(r % n_holes);
        const int c = (int)(r / n_holes);
        lab[(((int64_t)c * g.D + lo.v[q * 3 + 0] + d) * g.H + lo.v[q * 3 + 1] + h) * g.W + lo.v[q * 3 + 2] + w] = fill;
    }
}

template <int KIND>
int launch_kind(bool vec4, int nblk, hipStream_t s, const float* in, float* out, const float* noise, const VolGeo& g,
                const Stage& st, const mi3d_aug_params& p, const float2* pin, int npin, float2* pout) {
    if (vec4) aug_stage_kernel<4, KIND><<<nblk, BLK, 0, s>>>(in, out, noise, g, st, p, pin, npin, pout);
    else aug_stage_kernel<1, KIND><<<nblk, BLK, 0, s>>>(in, out, noise, g, st, p, pin, npin, pout);
    MI3D_LAUNCH_CHECK();
    return 0;
}
int launch_stage(bool vec4, int nblk, hipStream_t s, const float* in, float* out, const float* noise, const VolGeo& g,
                 const Stage& st, const mi3d_aug_params& p, const float2* pin, int npin, float2* pout) {
    if (st.bias || st.noise) return launch_kind<K_BIASNOISE>(vec4, nblk, s, in, out, noise, g, st, p, pin, npin, pout);
    if (st.contrast) return launch_kind<K_CONTRAST>(vec4, nblk, s, in, out, noise, g, st, p, pin, npin, pout);
    if (st.hist) return launch_kind<K_HIST>(vec4, nblk, s, in, out, noise, g, st, p, pin, npin, pout);
    return launch_kind<K_PLAIN>(vec4, nblk, s, in, out, noise, g, st, p, pin, npin, pout);
}

int holes_ok(int n_holes, const int32_t* lo, const int32_t* size, int D, int H, int W) {
    if (n_holes < 0 || n_holes > MI3D_AUG_MAX_HOLES) return 0;
    if (n_holes == 0) return 1;
    const int dim[3] = {D, H, W};
    for (int a = 0; a < 3; a++) {
        if (size[a] < 1 || size[a] > dim[a]) return 0;
        for (int q = 0; q < n_holes; q++)
            if (lo[q * 3 + a] < 0 || lo[q * 3 + a] + size[a] > dim[a]) return 0;
    }
    return 1;
}
}  // namespace

extern "C" {

size_t mi3d_augment_workspace_bytes(void) { return 2 * MAXBLK * sizeof(float2); }

int mi3d_augment(const float* in, float* out, const float* noise, int C, int D, int H, int W, const mi3d_aug_params* params,
                 void* workspace, size_t workspace_bytes, void* stream) {
    MI3D_CHECK_ARG(in && out && params && workspace && C >= 1 && D >= 1 && H >= 1 && W >= 1, "mi3d_augment: bad arguments");
    MI3D_CHECK_ARG((int64_t)C * D * H * W < (1ll << 31), "mi3d_augment: volume of 2^31 voxels or more");
    MI3D_CHECK_ARG(workspace_bytes >= mi3d_augment_workspace_bytes() && ((uintptr_t)workspace & 15) == 0,
                   "mi3d_augment: workspace too small or not 16-byte aligned");
    const mi3d_aug_params& p = *params;
    MI3D_CHECK_ARG(!p.do_bias || (p.bias_degree >= 0 && p.bias_degree <= 3), "mi3d_augment: bias degree must be 0..3");
    MI3D_CHECK_ARG(!p.do_hist || (p.n_cp >= 2 && p.n_cp <= MI3D_AUG_MAX_CP), "mi3d_augment: 2..16 control points");
    MI3D_CHECK_ARG(!p.do_noise || noise || p.noise_std >= 0.f, "mi3d_augment: negative noise std");
    MI3D_CHECK_ARG(holes_ok(p.n_holes, &p.hole_lo[0][0], p.hole_size, D, H, W), "mi3d_augment: hole outside the volume");
    if (p.do_hist)
        for (int i = 1; i < p.n_cp; i++)
            MI3D_CHECK_ARG(p.ref_cp[i] > p.ref_cp[i - 1], "mi3d_augment: reference control points must increase");
    hipStream_t s = (hipStream_t)stream;
    mi3d_aug_params pk = *params;                                     // kernel copy: bias coefficients in the degree-3 layout
    if (pk.do_bias && pk.bias_degree < 3) {
        double c3[MI3D_AUG_MAX_COEFF] = {};
        int src_i = 0;
        for (int i = 0; i <= pk.bias_degree; i++)
            for (int j = 0; j <= pk.bias_degree - i; j++)
                for (int k = 0; k <= pk.bias_degree - i - j; k++) {
                    int dst = 0;                                      // position of (i, j, k) in the degree-3 enumeration
                    for (int a = 0; a <= 3; a++)
                        for (int b = 0; b <= 3 - a; b++)
                            for (int cc = 0; cc <= 3 - a - b; cc++) {
                                if (a == i && b == j && cc == k) goto found;
                                dst++;
                            }
                found:
                    c3[dst] = params->bias_coeff[src_i++];
                }
        for (int i = 0; i < MI3D_AUG_MAX_COEFF; i++) pk.bias_coeff[i] = c3[i];
        pk.bias_degree = 3;
    }
    VolGeo g{C, D, H, W, (int64_t)C * D * H * W};
    const bool vec4 = (W % 4 == 0) && (((uintptr_t)in | (uintptr_t)out | (uintptr_t)noise) & 15) == 0;
    const int vec = vec4 ? 4 : 1;
    int64_t wb = (g.n / vec + BLK - 1) / BLK;
    const int nblk = (int)(wb < 1 ? 1 : (wb > MAXBLK ? MAXBLK : wb));
    float2* part[2] = {(float2*)workspace, (float2*)workspace + MAXBLK};

    // stages in chain order; `src` follows the data (in for the first stage that reads, out afterwards)
    Stage stages[3] = {};
    int ns = 0;
    if (p.do_bias || p.do_noise) { stages[ns].bias = p.do_bias != 0; stages[ns].noise = p.do_noise != 0; ns++; }
    if (p.do_contrast) { stages[ns].contrast = 1; stages[ns].stats_in = 1; ns++; }
    if (p.do_hist) {
        stages[ns].hist = 1;
        stages[ns].stats_in = 1;
        bool uni = true;                                              // MONAI's reference points are linspace(0, 1, n)
        for (int i = 0; i < p.n_cp; i++) uni = uni && fabsf(p.ref_cp[i] - (float)i / (float)(p.n_cp - 1)) < 1e-6f;
        stages[ns].uniform_cp = uni;
        ns++;
    }
    if (ns == 0) { ns = 1; }                                          // holes only / nothing drawn: one copy pass
    stages[ns - 1].holes = p.n_holes > 0;
    for (int i = 0; i < ns; i++) { stages[i].write = 1; stages[i].emit = i + 1 < ns; }   // a later stage always wants stats
    if (!p.do_bias && !p.do_noise && !p.do_contrast && !p.do_hist && p.n_holes == 0 && in == out)
        return 0;                                                     // nothing drawn, in place: no pass at all
    const float* src = in;
    int cur = 0;
    if (stages[0].stats_in) {                                         // first transform is contrast / hist: read-only min/max
        Stage mm = {};
        mm.emit = 1;
        MI3D_TRY(launch_stage(vec4, nblk, s, src, out, nullptr, g, mm, pk, nullptr, 0, part[cur]));
    }
    for (int i = 0; i < ns; i++) {
        const float2* pin = stages[i].stats_in ? part[cur] : nullptr;
        float2* pout = part[cur ^ 1];
        MI3D_TRY(launch_stage(vec4, nblk, s, src, out, noise, g, stages[i], pk, pin, nblk, pout));
        src = out;
        cur ^= 1;
    }
    return 0;
}

int mi3d_fill_boxes_i64(int64_t* label, int C, int D, int H, int W, int n_holes, const int32_t* hole_lo,
                        const int32_t* hole_size, int64_t fill, void* stream) {
    MI3D_CHECK_ARG(label && C >= 1 && D >= 1 && H >= 1 && W >= 1 && (n_holes == 0 || (hole_lo && hole_size)),
                   "mi3d_fill_boxes_i64: bad arguments");
    MI3D_CHECK_ARG(holes_ok(n_holes, hole_lo, hole_size, D, H, W), "mi3d_fill_boxes_i64: hole outside the volume");
    if (n_holes == 0) return 0;
    VolGeo g{C, D, H, W, (int64_t)C * D * H * W};
    HoleLo lo = {};                                                   // hole_lo / hole_size are HOST arrays (a few ints)
    for (int i = 0; i < n_holes * 3; i++) lo.v[i] = hole_lo[i];
    const int64_t total = (int64_t)hole_size[0] * hole_size[1] * hole_size[2] * n_holes * C;
    int64_t wb = (total + BLK - 1) / BLK;
    fill_boxes_i64_kernel<<<(int)(wb > MAXBLK ? MAXBLK : wb), BLK, 0, (hipStream_t)stream>>>(
        label, g, n_holes, make_int3(hole_size[0], hole_size[1], hole_size[2]), lo, fill);
    MI3D_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
