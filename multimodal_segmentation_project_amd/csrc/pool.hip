// pool.hip — MaxPool3d(kernel 2, stride 2) forward / backward on channels-last activations.
// Reference: nn.MaxPool3d(2,2), models/unet.py:40,71.  Pure HBM streaming (8:1 read:write forward).
// Backward routes the pooled gradient to the FIRST maximum in (d,h,w) scan order (torch semantics) and, in
// the same pass, adds the gradient arriving through the skip connection (the torch.cat split of unet.py:84),
// so the encoder output gradient is produced with one read of each operand and one write.
#include "ops.h"

namespace {
constexpr int BLK = 256;

template <typename T, int VEC>
__global__ __launch_bounds__(BLK) void maxpool2_fwd_kernel(const T* __restrict__ z, int zcs, int C, int N, int D, int H, int W,
                                                           T* __restrict__ p, int pcs) {
    int G = C / VEC, Do = D / 2, Ho = H / 2, Wo = W / 2;
    int64_t total = (int64_t)N * Do * Ho * Wo * G;
    for (int64_t idx = (int64_t)blockIdx.x * BLK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * BLK) {
        // 32-bit index math (total < 2^31 checked by the launcher): a 64-bit div/mod costs ~80 instructions
        unsigned iu = (unsigned)idx, r = iu / (unsigned)G;
        int g = (int)(iu - r * (unsigned)G);
        int wo = (int)(r % (unsigned)Wo); r /= (unsigned)Wo; int ho = (int)(r % (unsigned)Ho); r /= (unsigned)Ho;
        int d_o = (int)(r % (unsigned)Do); int n = (int)(r / (unsigned)Do);
        float m[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) m[i] = -INFINITY;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    float v[VEC];
                    ldv<T, VEC>(z + ((((int64_t)n * D + 2 * d_o + a) * H + 2 * ho + b) * W + 2 * wo + c) * zcs + g * VEC, v);
#pragma unroll
                    for (int i = 0; i < VEC; i++) m[i] = v[i] > m[i] ? v[i] : m[i];
                }
        stv<T, VEC>(p + ((((int64_t)n * Do + d_o) * Ho + ho) * Wo + wo) * pcs + g * VEC, m);
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(BLK) void maxpool2_bwd_kernel(const T* __restrict__ dp, int dpcs, const T* __restrict__ z, int zcs,
                                                           const T* __restrict__ dskip, int dskipcs, T* __restrict__ dz,
                                                           int dzcs, int C, int N, int D, int H, int W,
                                                           const float* __restrict__ skp, int ks) {
    // skp != NULL: dp is still the ks fp32 split-K partials [ks][Mp][C] of the input-gradient conv that produced it; they are
    // summed in order and rounded to T here (what splitk_finish_kernel would have stored): its launch disappears
    int G = C / VEC, Do = D / 2, Ho = H / 2, Wo = W / 2;
    int64_t total = (int64_t)N * Do * Ho * Wo * G;
    for (int64_t idx = (int64_t)blockIdx.x * BLK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * BLK) {
        unsigned iu = (unsigned)idx, r = iu / (unsigned)G;
        int g = (int)(iu - r * (unsigned)G);
        int wo = (int)(r % (unsigned)Wo); r /= (unsigned)Wo; int ho = (int)(r % (unsigned)Ho); r /= (unsigned)Ho;
        int d_o = (int)(r % (unsigned)Do); int n = (int)(r / (unsigned)Do);
        float v[8][VEC], m[VEC], gp[VEC];
        int arg[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) { m[i] = -INFINITY; arg[i] = 0; }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int a = k >> 2, b = (k >> 1) & 1, c = k & 1;
            ldv<T, VEC>(z + ((((int64_t)n * D + 2 * d_o + a) * H + 2 * ho + b) * W + 2 * wo + c) * zcs + g * VEC, v[k]);
#pragma unroll
            for (int i = 0; i < VEC; i++) if (v[k][i] > m[i]) { m[i] = v[k][i]; arg[i] = k; }
        }
        if (skp) {
            const int64_t row = (((int64_t)n * Do + d_o) * Ho + ho) * Wo + wo, Mp = (int64_t)N * Do * Ho * Wo;
#pragma unroll
            for (int i = 0; i < VEC; i++) gp[i] = 0.f;
            for (int k = 0; k < ks; k++) {
                const float* pp = skp + ((int64_t)k * Mp + row) * C + g * VEC;
#pragma unroll
                for (int i = 0; i < VEC; i++) gp[i] += pp[i];
            }
#pragma unroll
            for (int i = 0; i < VEC; i++) gp[i] = round_to<T>(gp[i]);
        } else
            ldv<T, VEC>(dp + ((((int64_t)n * Do + d_o) * Ho + ho) * Wo + wo) * dpcs + g * VEC, gp);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int a = k >> 2, b = (k >> 1) & 1, c = k & 1;
            int64_t off = (((int64_t)n * D + 2 * d_o + a) * H + 2 * ho + b) * W + 2 * wo + c;
            float o[VEC];
            if (dskip) ldv<T, VEC>(dskip + off * dskipcs + g * VEC, o);
            else {
#pragma unroll
                for (int i = 0; i < VEC; i++) o[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < VEC; i++) o[i] += (arg[i] == k) ? gp[i] : 0.f;
            stv<T, VEC>(dz + off * dzcs + g * VEC, o);
        }
    }
}

// The same pass with TWO threads per window and channel group (round 4): thread (window, c, g) owns the four voxels (a, b, c) of
// the window, so for every (a, b) the lanes of a wave read / write one contiguous run of memory (lane stride 16 B) instead of every
// other 32-B half of it (the one-thread-per-window kernel: 41 us = 4.3 TB/s at full resolution).  The two halves of a window combine
// their (maximum, first index) through one lane exchange; ties and non-finite values resolve exactly like the sequential strict `>`
// scan over k = 4a + 2b + c (first occurrence of the maximum; index 0 when nothing exceeds -inf).  G = C / 8 a power of two <= 32.
template <typename T>
__global__ __launch_bounds__(BLK) void maxpool2_bwd_pair_kernel(const T* __restrict__ dp, int dpcs, const T* __restrict__ z, int zcs,
                                                                const T* __restrict__ dskip, int dskipcs, T* __restrict__ dz,
                                                                int dzcs, int C, int N, int D, int H, int W,
                                                                const float* __restrict__ skp, int ks) {
    constexpr int VEC = 8;
    int G = C / VEC, Do = D / 2, Ho = H / 2, Wo = W / 2;
    int64_t total = (int64_t)N * Do * Ho * Wo * G * 2;
    for (int64_t idx = (int64_t)blockIdx.x * BLK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * BLK) {
        unsigned iu = (unsigned)idx, r = iu / (unsigned)G;
        int g = (int)(iu - r * (unsigned)G);
        int c = (int)(r & 1u); r >>= 1;
        int wo = (int)(r % (unsigned)Wo); r /= (unsigned)Wo; int ho = (int)(r % (unsigned)Ho); r /= (unsigned)Ho;
        int d_o = (int)(r % (unsigned)Do); int n = (int)(r / (unsigned)Do);
        float v[4][VEC], m[VEC], gp[VEC];
        int arg[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) { m[i] = -INFINITY; arg[i] = 99; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int a = q >> 1, b = q & 1, k = 4 * a + 2 * b + c;
            ldv<T, VEC>(z + ((((int64_t)n * D + 2 * d_o + a) * H + 2 * ho + b) * W + 2 * wo + c) * zcs + g * VEC, v[q]);
#pragma unroll
            for (int i = 0; i < VEC; i++) if (v[q][i] > m[i]) { m[i] = v[q][i]; arg[i] = k; }
        }
        // the other half of the window: lane ^ G (G <= 32: same wave; both halves of a window are always active together)
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            float om = __shfl_xor(m[i], G, 64);
            int oa = __shfl_xor(arg[i], G, 64);
            bool take = (oa != 99) && (arg[i] == 99 || om > m[i] || (om == m[i] && oa < arg[i]));
            arg[i] = take ? oa : arg[i];
            if (arg[i] == 99) arg[i] = 0;
        }
        if (skp) {
            const int64_t row = (((int64_t)n * Do + d_o) * Ho + ho) * Wo + wo, Mp = (int64_t)N * Do * Ho * Wo;
#pragma unroll
            for (int i = 0; i < VEC; i++) gp[i] = 0.f;
            for (int k = 0; k < ks; k++) {
                const float* pp = skp + ((int64_t)k * Mp + row) * C + g * VEC;
#pragma unroll
                for (int i = 0; i < VEC; i++) gp[i] += pp[i];
            }
#pragma unroll
            for (int i = 0; i < VEC; i++) gp[i] = round_to<T>(gp[i]);
        } else
            ldv<T, VEC>(dp + ((((int64_t)n * Do + d_o) * Ho + ho) * Wo + wo) * dpcs + g * VEC, gp);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int a = q >> 1, b = q & 1, k = 4 * a + 2 * b + c;
            int64_t off = (((int64_t)n * D + 2 * d_o + a) * H + 2 * ho + b) * W + 2 * wo + c;
            float o[VEC];
            if (dskip) ldv<T, VEC>(dskip + off * dskipcs + g * VEC, o);
            else {
#pragma unroll
                for (int i = 0; i < VEC; i++) o[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < VEC; i++) o[i] += (arg[i] == k) ? gp[i] : 0.f;
            stv<T, VEC>(dz + off * dzcs + g * VEC, o);
        }
    }
}

// odd D/H/W: MaxPool3d floors (the last slice of an odd dimension is in no window) -> those voxels only receive the
// skip gradient
template <typename T>
__global__ __launch_bounds__(BLK) void maxpool2_bwd_border_kernel(const T* __restrict__ dskip, int dskipcs, T* __restrict__ dz, int dzcs,
                                                                  int C, int N, int D, int H, int W) {
    int De = D & ~1, He = H & ~1, We = W & ~1;
    int64_t total = (int64_t)N * D * H * W;
    for (int64_t v = (int64_t)blockIdx.x * BLK + threadIdx.x; v < total; v += (int64_t)gridDim.x * BLK) {
        int w = (int)(v % W); int64_t r = v / W; int h = (int)(r % H); r /= H; int d = (int)(r % D);
        if (d < De && h < He && w < We) continue;
        for (int c = 0; c < C; c++) dz[v * dzcs + c] = dskip ? dskip[v * dskipcs + c] : from_f<T>(0.f);
    }
}

// F.interpolate(x, size=skip.shape[2:]) of models/unet.py:81-83 (default mode 'nearest'): src = min(floor(dst * in/out), in-1),
// computed in float like torch.  Only reached when a volume side is not divisible by 2^levels.
__device__ __forceinline__ int nn_src(int dst, float scale, int in) {
    int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}
template <typename T>
__global__ __launch_bounds__(BLK) void nearest_resize_fwd_kernel(const T* __restrict__ x, int xcs, int C, int N, int Di, int Hi, int Wi,
                                                                 T* __restrict__ y, int ycs, int Do, int Ho, int Wo) {
    float sd = (float)Di / Do, sh = (float)Hi / Ho, sw = (float)Wi / Wo;
    int64_t total = (int64_t)N * Do * Ho * Wo;
    for (int64_t v = (int64_t)blockIdx.x * BLK + threadIdx.x; v < total; v += (int64_t)gridDim.x * BLK) {
        int w = (int)(v % Wo); int64_t r = v / Wo; int h = (int)(r % Ho); r /= Ho; int d = (int)(r % Do); int n = (int)(r / Do);
        int64_t src = (((int64_t)n * Di + nn_src(d, sd, Di)) * Hi + nn_src(h, sh, Hi)) * Wi + nn_src(w, sw, Wi);
        for (int c = 0; c < C; c++) y[v * ycs + c] = x[src * xcs + c];
    }
}
// gx[src] = sum of gy over the destination voxels that read src (gather form: deterministic, no atomics)
template <typename T>
__global__ __launch_bounds__(BLK) void nearest_resize_bwd_kernel(const T* __restrict__ gy, int gycs, int C, int N, int Do, int Ho, int Wo,
                                                                 T* __restrict__ gx, int gxcs, int Di, int Hi, int Wi) {
    float sd = (float)Di / Do, sh = (float)Hi / Ho, sw = (float)Wi / Wo;
    int64_t total = (int64_t)N * Di * Hi * Wi;
    for (int64_t v = (int64_t)blockIdx.x * BLK + threadIdx.x; v < total; v += (int64_t)gridDim.x * BLK) {
        int w = (int)(v % Wi); int64_t r = v / Wi; int h = (int)(r % Hi); r /= Hi; int d = (int)(r % Di); int n = (int)(r / Di);
        // candidate destinations per dimension: a window around src/scale
        int d0 = (int)((float)d / sd) - 1, h0 = (int)((float)h / sh) - 1, w0 = (int)((float)w / sw) - 1;
        for (int c = 0; c < C; c++) {
            float acc = 0.f;
            for (int a = d0 < 0 ? 0 : d0; a < Do && a <= d0 + 3; a++) {
                if (nn_src(a, sd, Di) != d) continue;
                for (int b = h0 < 0 ? 0 : h0; b < Ho && b <= h0 + 3; b++) {
                    if (nn_src(b, sh, Hi) != h) continue;
                    for (int e = w0 < 0 ? 0 : w0; e < Wo && e <= w0 + 3; e++) {
                        if (nn_src(e, sw, Wi) != w) continue;
                        acc += to_f<T>(gy[((((int64_t)n * Do + a) * Ho + b) * Wo + e) * gycs + c]);
                    }
                }
            }
            gx[v * gxcs + c] = from_f<T>(acc);
        }
    }
}

inline int sgrid(int64_t total) {
    int64_t w = (total + BLK - 1) / BLK;
    return (int)(w < 1 ? 1 : (w > 2048 ? 2048 : w));
}
inline bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }
}  // namespace

int maxpool2_fwd(int dtype, const void* z, int zcs, int C, Geo g, void* p, int pcs, hipStream_t s) {
    MI3D_CHECK_ARG(g.D >= 2 && g.H >= 2 && g.W >= 2, "maxpool2: volume %dx%dx%d too small", g.D, g.H, g.W);
    MI3D_CHECK_ARG(g.M() / 8 * C < (1ll << 31), "maxpool2: more than 2^31 pooled elements");
    int64_t nout = (int64_t)g.N * (g.D / 2) * (g.H / 2) * (g.W / 2);          // floor, like nn.MaxPool3d
    DISPATCH_T(dtype, T, {
        if (C % 8 == 0 && zcs % 8 == 0 && pcs % 8 == 0 && al16(z) && al16(p))
            maxpool2_fwd_kernel<T, 8><<<sgrid(nout * (C / 8)), BLK, 0, s>>>((const T*)z, zcs, C, g.N, g.D, g.H, g.W, (T*)p, pcs);
        else
            maxpool2_fwd_kernel<T, 1><<<sgrid(nout * C), BLK, 0, s>>>((const T*)z, zcs, C, g.N, g.D, g.H, g.W, (T*)p, pcs);
        MI3D_LAUNCH_CHECK();
    });
    return 0;
}

int maxpool2_bwd(int dtype, const void* dp, int dpcs, const void* z, int zcs, const void* dskip, int dskipcs, void* dz,
                 int dzcs, int C, Geo g, hipStream_t s, const float* skp, int ks) {
    if (ks <= 0) skp = nullptr;
    MI3D_CHECK_ARG(g.D >= 2 && g.H >= 2 && g.W >= 2, "maxpool2_bwd: volume too small");
    MI3D_CHECK_ARG(g.M() / 8 * C < (1ll << 31), "maxpool2_bwd: more than 2^31 pooled elements");
    int64_t nout = (int64_t)g.N * (g.D / 2) * (g.H / 2) * (g.W / 2);
    bool odd = (g.D | g.H | g.W) & 1;
    DISPATCH_T(dtype, T, {
        bool v8 = C % 8 == 0 && zcs % 8 == 0 && dpcs % 8 == 0 && dzcs % 8 == 0 && (!dskip || dskipcs % 8 == 0) &&
                  al16(z) && al16(dp) && al16(dz) && al16(dskip);
        const int G8 = C / 8;
        if (v8 && (G8 & (G8 - 1)) == 0 && G8 <= 32 && nout * G8 * 2 < (1ll << 31) && !mi3d_routes().no_pool_pair)
            maxpool2_bwd_pair_kernel<T><<<sgrid(nout * G8 * 2), BLK, 0, s>>>((const T*)dp, dpcs, (const T*)z, zcs, (const T*)dskip, dskipcs, (T*)dz, dzcs, C, g.N, g.D, g.H, g.W, skp, ks);
        else if (v8)
            maxpool2_bwd_kernel<T, 8><<<sgrid(nout * (C / 8)), BLK, 0, s>>>((const T*)dp, dpcs, (const T*)z, zcs, (const T*)dskip, dskipcs, (T*)dz, dzcs, C, g.N, g.D, g.H, g.W, skp, ks);
        else
            maxpool2_bwd_kernel<T, 1><<<sgrid(nout * C), BLK, 0, s>>>((const T*)dp, dpcs, (const T*)z, zcs, (const T*)dskip, dskipcs, (T*)dz, dzcs, C, g.N, g.D, g.H, g.W, skp, ks);
        MI3D_LAUNCH_CHECK();
        if (odd) {
            maxpool2_bwd_border_kernel<T><<<sgrid(g.M()), BLK, 0, s>>>((const T*)dskip, dskipcs, (T*)dz, dzcs, C, g.N, g.D, g.H, g.W);
            MI3D_LAUNCH_CHECK();
        }
    });
    return 0;
}

int nearest_resize_fwd(int dtype, const void* x, int xcs, int C, Geo gi, void* y, int ycs, Geo go, hipStream_t s) {
    MI3D_CHECK_ARG(gi.N == go.N && C >= 1, "nearest_resize_fwd: bad shapes");
    DISPATCH_T(dtype, T, {
        nearest_resize_fwd_kernel<T><<<sgrid(go.M()), BLK, 0, s>>>((const T*)x, xcs, C, gi.N, gi.D, gi.H, gi.W, (T*)y, ycs, go.D, go.H, go.W);
        MI3D_LAUNCH_CHECK();
    });
    return 0;
}
int nearest_resize_bwd(int dtype, const void* gy, int gycs, int C, Geo go, void* gx, int gxcs, Geo gi, hipStream_t s) {
    MI3D_CHECK_ARG(gi.N == go.N && C >= 1, "nearest_resize_bwd: bad shapes");
    DISPATCH_T(dtype, T, {
        nearest_resize_bwd_kernel<T><<<sgrid(gi.M()), BLK, 0, s>>>((const T*)gy, gycs, C, go.N, go.D, go.H, go.W, (T*)gx, gxcs, gi.D, gi.H, gi.W);
        MI3D_LAUNCH_CHECK();
    });
    return 0;
}
