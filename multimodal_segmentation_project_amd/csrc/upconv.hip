// upconv.hip — ConvTranspose3d(kernel 2, stride 2): non-overlapping, so it is a per-voxel GEMM
//   out[n, 2d+a, 2h+b, 2w+c, co] = bias[co] + sum_ci x[n,d,h,w,ci] * W[ci,co,a,b,c]
// followed by a 2x2x2 pixel shuffle.  Reference: nn.ConvTranspose3d(2f,f,2,stride=2), models/unet.py:56-58,79.
// The output is written straight into channels [C,2C) of the level's concat buffer (ycs = 2C), which makes
// torch.cat((skip, x), 1) of unet.py:84 free.
//
// fwd : thread = one input voxel x 8 output channels x 8 taps (64 fp32 accumulators), weights wave-uniform
//       through the scalar cache.  bwd-data: thread = one input voxel x 8 input channels.  bwd-weight:
//       register-tiled (4ci x 4co) x tap reduction over 32-voxel LDS tiles, deterministic slab reduction.
#include "ops.h"

int slab_reduce(const float* slabs, int nslab, int64_t slab_sz, int64_t nW, float* dW, float* db, int accumulate,
                hipStream_t s);

namespace {
constexpr int BLK = 256;

__global__ void upconv_pack_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ wf,
                                   float* __restrict__ wb) {
    // wf[((cob*Cin + ci)*8 + tap)*8 + j] = w[ci][cob*8+j][tap]
    // wb[((cib*8 + tap)*Cout + co)*8 + j] = w[cib*8+j][co][tap]
    int64_t nf = (int64_t)cdiv(Cout, 8) * Cin * 64, nb = (int64_t)cdiv(Cin, 8) * Cout * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nb; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < nf) {
            int j = i % 8; int64_t r = i / 8; int tap = r % 8; r /= 8; int ci = r % Cin; int cob = r / Cin;
            int co = cob * 8 + j;
            wf[i] = co < Cout ? w[((int64_t)ci * Cout + co) * 8 + tap] : 0.f;
        } else {
            int64_t k = i - nf;
            int j = k % 8; int64_t r = k / 8; int co = r % Cout; r /= Cout; int tap = r % 8; int cib = r / 8;
            int ci = cib * 8 + j;
            wb[k] = ci < Cin ? w[((int64_t)ci * Cout + co) * 8 + tap] : 0.f;
        }
    }
}

template <typename T, int CIC>
__global__ __launch_bounds__(BLK) void upconv_fwd_kernel(const T* __restrict__ x, int xcs, int Cin, const float* __restrict__ wf,
                                                         const float* __restrict__ bias, T* __restrict__ y, int ycs, int Cout,
                                                         int N, int D, int H, int W) {
    int64_t M = (int64_t)N * D * H * W;
    int64_t v = (int64_t)blockIdx.x * BLK + threadIdx.x;
    int cob = blockIdx.y;
    if (v >= M) return;
    float acc[8][8];
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[t][j] = 0.f;
    const float* wc = wf + (int64_t)cob * Cin * 64;
    for (int c0 = 0; c0 < Cin; c0 += CIC) {
        float xv[CIC];
        ldv<T, CIC>(x + v * xcs + c0, xv);
#pragma unroll
        for (int ci = 0; ci < CIC; ci++) {
            const float* wr = wc + (int64_t)(c0 + ci) * 64;
#pragma unroll
            for (int t = 0; t < 8; t++)
#pragma unroll
                for (int j = 0; j < 8; j++) acc[t][j] = fmaf(xv[ci], wr[t * 8 + j], acc[t][j]);
        }
    }
    int w_ = (int)(v % W); int64_t r = v / W; int h_ = (int)(r % H); r /= H; int d_ = (int)(r % D); int n = (int)(r / D);
    int co0 = cob * 8;
    bool vec = (co0 + 8 <= Cout) && (ycs % 8 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
#pragma unroll
    for (int t = 0; t < 8; t++) {
        int a = t >> 2, b = (t >> 1) & 1, c = t & 1;
        T* yp = y + ((((int64_t)n * 2 * D + 2 * d_ + a) * 2 * H + 2 * h_ + b) * 2 * W + 2 * w_ + c) * ycs + co0;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = acc[t][j] + ((bias && co0 + j < Cout) ? bias[co0 + j] : 0.f);
        if (vec) st8<T>(yp, o);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) if (co0 + j < Cout) yp[j] = from_f<T>(o[j]);
        }
    }
}

template <typename T, int COC>
__global__ __launch_bounds__(BLK) void upconv_bwd_data_kernel(const T* __restrict__ g, int gcs, int Cout,
                                                              const float* __restrict__ wb, T* __restrict__ dx, int dxcs,
                                                              int Cin, int N, int D, int H, int W) {
    int64_t M = (int64_t)N * D * H * W;
    int64_t v = (int64_t)blockIdx.x * BLK + threadIdx.x;
    int cib = blockIdx.y;
    if (v >= M) return;
    int w_ = (int)(v % W); int64_t r = v / W; int h_ = (int)(r % H); r /= H; int d_ = (int)(r % D); int n = (int)(r / D);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        int a = t >> 2, b = (t >> 1) & 1, c = t & 1;
        const T* gp = g + ((((int64_t)n * 2 * D + 2 * d_ + a) * 2 * H + 2 * h_ + b) * 2 * W + 2 * w_ + c) * gcs;
        const float* wt = wb + ((int64_t)cib * 8 + t) * Cout * 8;
        for (int c0 = 0; c0 < Cout; c0 += COC) {
            float gv[COC];
            ldv<T, COC>(gp + c0, gv);
#pragma unroll
            for (int co = 0; co < COC; co++) {
                const float* wr = wt + (int64_t)(c0 + co) * 8;
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] = fmaf(gv[co], wr[j], acc[j]);
            }
        }
    }
    int ci0 = cib * 8;
    T* dp = dx + v * dxcs + ci0;
    if ((ci0 + 8 <= Cin) && (dxcs % 8 == 0) && ((reinterpret_cast<uintptr_t>(dx) & 15) == 0)) st8<T>(dp, acc);
    else {
#pragma unroll
        for (int j = 0; j < 8; j++) if (ci0 + j < Cin) dp[j] = from_f<T>(acc[j]);
    }
}

constexpr int UV = 32;   // input voxels per LDS tile
constexpr int CB = 32;

template <typename T>
__global__ __launch_bounds__(BLK) void upconv_bwd_weight_kernel(const T* __restrict__ x, int xcs, int Cin,
                                                                const T* __restrict__ g, int gcs, int Cout, int N, int D,
                                                                int H, int W, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float xs[UV * CB];
    __shared__ __attribute__((aligned(16))) float gs[UV * 8 * CB];
    int ci0 = blockIdx.y * CB, co0 = blockIdx.z * CB;
    int ncib = min(CB, Cin - ci0), ncob = min(CB, Cout - co0);
    int CI4 = (ncib + 3) / 4, CO4 = (ncob + 3) / 4;
    int ntask = 8 * CI4 * CO4;   // <= 512 -> 2 per thread
    float acc[2][16];
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[k][i] = 0.f;
    float dbacc = 0.f;
    int64_t M = (int64_t)N * D * H * W;
    int64_t ntile = (M + UV - 1) / UV;
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < UV * CB; idx += BLK) {
            int vv = idx / CB, c = idx - vv * CB;
            int64_t v = tile * UV + vv;
            xs[idx] = (v < M && c < ncib) ? to_f<T>(x[v * xcs + ci0 + c]) : 0.f;
        }
        for (int idx = threadIdx.x; idx < UV * 8 * CB; idx += BLK) {
            int c = idx % CB; int r = idx / CB; int t = r % 8; int vv = r / 8;
            int64_t v = tile * UV + vv;
            float val = 0.f;
            if (v < M && c < ncob) {
                int w_ = (int)(v % W); int64_t q = v / W; int h_ = (int)(q % H); q /= H; int d_ = (int)(q % D); int n = (int)(q / D);
                int a = t >> 2, b = (t >> 1) & 1, cc = t & 1;
                val = to_f<T>(g[((((int64_t)n * 2 * D + 2 * d_ + a) * 2 * H + 2 * h_ + b) * 2 * W + 2 * w_ + cc) * gcs + co0 + c]);
            }
            gs[idx] = val;
        }
        __syncthreads();
        if (blockIdx.y == 0 && threadIdx.x < ncob) {
            float s = 0.f;
            for (int i = 0; i < UV * 8; i++) s += gs[i * CB + threadIdx.x];
            dbacc += s;
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            int task = threadIdx.x + k * BLK;
            if (task < ntask) {
                int co4 = task % CO4; int r = task / CO4; int ci4 = r % CI4; int t = r / CI4;
                for (int vv = 0; vv < UV; vv++) {
                    f32x4 xv = *reinterpret_cast<const f32x4*>(xs + vv * CB + ci4 * 4);
                    f32x4 gv = *reinterpret_cast<const f32x4*>(gs + (vv * 8 + t) * CB + co4 * 4);
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[k][i * 4 + j] = fmaf(xv[i], gv[j], acc[k][i * 4 + j]);
                }
            }
        }
    }
    int64_t nW = (int64_t)Cin * Cout * 8;
    float* slab = slabs + (int64_t)blockIdx.x * (nW + Cout);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        int task = threadIdx.x + k * BLK;
        if (task < ntask) {
            int co4 = task % CO4; int r = task / CO4; int ci4 = r % CI4; int t = r / CI4;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int ci = ci0 + ci4 * 4 + i, co = co0 + co4 * 4 + j;
                    if (ci < Cin && co < Cout) slab[((int64_t)ci * Cout + co) * 8 + t] = acc[k][i * 4 + j];
                }
        }
    }
    if (blockIdx.y == 0 && threadIdx.x < ncob) slab[nW + co0 + threadIdx.x] = dbacc;
}

inline int upw_nsb(int Cin, int Cout, Geo g) {
    int64_t ntile = (g.M() + UV - 1) / UV;
    int chan_blocks = cdiv(Cin, CB) * cdiv(Cout, CB);
    int64_t want = (512 + chan_blocks - 1) / chan_blocks;
    return (int)(ntile < want ? ntile : want);
}
inline bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }
}  // namespace

size_t upconv2_pack_floats(int Cin, int Cout) {
    return (size_t)cdiv(Cout, 8) * Cin * 64 + (size_t)cdiv(Cin, 8) * Cout * 64;   // fwd pack followed by bwd pack
}

int upconv2_pack(const float* w, int Cin, int Cout, float* wp_fwd, float* wp_bwd, hipStream_t s) {
    MI3D_CHECK_ARG(wp_bwd == wp_fwd + (size_t)cdiv(Cout, 8) * Cin * 64, "upconv2_pack: wp_bwd must follow wp_fwd");
    int64_t n = (int64_t)upconv2_pack_floats(Cin, Cout);
    upconv_pack_kernel<<<cdiv(n, 256) > 1024 ? 1024 : cdiv(n, 256), 256, 0, s>>>(w, Cin, Cout, wp_fwd, wp_bwd);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int upconv2_fwd(int dtype, const void* x, int xcs, int Cin, const float* wp_fwd, const float* bias, void* y, int ycs,
                int Cout, Geo g, hipStream_t s) {
    dim3 grid((unsigned)cdiv(g.M(), BLK), (unsigned)cdiv(Cout, 8));
    DISPATCH_T(dtype, T, {
        if (Cin % 8 == 0 && xcs % 8 == 0 && al16(x))
            upconv_fwd_kernel<T, 8><<<grid, BLK, 0, s>>>((const T*)x, xcs, Cin, wp_fwd, bias, (T*)y, ycs, Cout, g.N, g.D, g.H, g.W);
        else
            upconv_fwd_kernel<T, 1><<<grid, BLK, 0, s>>>((const T*)x, xcs, Cin, wp_fwd, bias, (T*)y, ycs, Cout, g.N, g.D, g.H, g.W);
        MI3D_LAUNCH_CHECK();
    });
    return 0;
}

size_t upconv2_bwd_ws_floats(int Cin, int Cout, Geo g) {
    return (size_t)upw_nsb(Cin, Cout, g) * ((size_t)Cin * Cout * 8 + Cout);
}

int upconv2_bwd(int dtype, const void* x, int xcs, int Cin, const void* gy, int gycs, int Cout, const float* wp_bwd,
                void* dx, int dxcs, float* dW, float* db, int accumulate, float* ws, size_t ws_floats, Geo g,
                hipStream_t s) {
    int nsb = upw_nsb(Cin, Cout, g);
    int64_t nW = (int64_t)Cin * Cout * 8;
    MI3D_CHECK_ARG(ws_floats >= (size_t)nsb * (nW + Cout), "upconv2_bwd: workspace too small");
    DISPATCH_T(dtype, T, {
        if (dx) {
            dim3 grid((unsigned)cdiv(g.M(), BLK), (unsigned)cdiv(Cin, 8));
            if (Cout % 8 == 0 && gycs % 8 == 0 && al16(gy))
                upconv_bwd_data_kernel<T, 8><<<grid, BLK, 0, s>>>((const T*)gy, gycs, Cout, wp_bwd, (T*)dx, dxcs, Cin, g.N, g.D, g.H, g.W);
            else
                upconv_bwd_data_kernel<T, 1><<<grid, BLK, 0, s>>>((const T*)gy, gycs, Cout, wp_bwd, (T*)dx, dxcs, Cin, g.N, g.D, g.H, g.W);
            MI3D_LAUNCH_CHECK();
        }
        dim3 gw((unsigned)nsb, (unsigned)cdiv(Cin, CB), (unsigned)cdiv(Cout, CB));
        upconv_bwd_weight_kernel<T><<<gw, BLK, 0, s>>>((const T*)x, xcs, Cin, (const T*)gy, gycs, Cout, g.N, g.D, g.H, g.W, ws);
        MI3D_LAUNCH_CHECK();
    });
    return slab_reduce(ws, nsb, nW + Cout, nW, dW, db, accumulate, s);
}
