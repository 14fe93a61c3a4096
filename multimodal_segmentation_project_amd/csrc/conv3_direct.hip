// conv3_direct.hip — 3x3x3 convolution (pad 1) as a direct LDS-tiled fp32-FMA kernel family.
// Reference: nn.Conv3d(Cin,Cout,3,padding=1) forward/backward, models/unet.py:11,15.
//
// This family handles ANY channel count and both storage types; it is the exact-fp32 path (parity runs) and
// the path for Cin = 1 (first layer, HBM-bound).  The bf16 layers with Cin,Cout % 16 == 0 use the MFMA
// implicit-GEMM kernels in conv3_mfma.hip instead.
//
//  fwd / dgrad : one workgroup = 4x8x8 output voxels x 16 output channels.  The (6x10x10) input halo tile is
//                staged through LDS in chunks of 8 input channels; weights are wave-uniform and are fetched
//                through the SCALAR cache (s_load) so the inner loop is v_fma(vgpr, sgpr) with no LDS weight
//                traffic.  dgrad is the same kernel on a flipped/transposed weight pack.
//  wgrad       : one workgroup = (<=32 co) x (<=32 ci) x 27 taps, sweeping 4x4x8-voxel tiles; each thread owns
//                up to seven 4x4 (co,ci) register tiles.  Partial results go to per-workgroup slabs that a
//                second kernel sums in a fixed order (deterministic, no float atomics).
#include "ops.h"

int slab_reduce(const float* slabs, int nslab, int64_t slab_sz, int64_t nW, float* dW, float* db, int accumulate,
                hipStream_t s);

namespace {

constexpr int BLK = 256;
constexpr int TZ = 4, TY = 8, TX = 8;          // fwd tile (output voxels)
constexpr int IZ = TZ + 2, IY = TY + 2, IX = TX + 2;
constexpr int COT = 16;                        // output channels per thread

__global__ void pack_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ wf,
                            float* __restrict__ wd, const float* __restrict__ scale) {
    // wf[((cb*Cin + ci)*27 + tap)*16 + j] = w[cb*16+j][ci][tap]
    // wd[((cb*Cout + co)*27 + tap)*16 + j] = w[co][cb*16+j][26-tap]
    int64_t nf = (int64_t)cdiv(Cout, COT) * COT * Cin * 27, nd = (int64_t)cdiv(Cin, COT) * COT * Cout * 27;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < nf) {
            int j = i % COT; int64_t r = i / COT; int tap = r % 27; r /= 27; int ci = r % Cin; int cb = r / Cin;
            int co = cb * COT + j;
            if (wf) wf[i] = co < Cout ? w[((int64_t)co * Cin + ci) * 27 + tap] * (scale ? scale[co] : 1.f) : 0.f;
        } else {
            int64_t k = i - nf;
            int j = k % COT; int64_t r = k / COT; int tap = r % 27; r /= 27; int co = r % Cout; int cb = r / Cout;
            int ci = cb * COT + j;
            if (wd) wd[k] = ci < Cin ? w[((int64_t)co * Cin + ci) * 27 + (26 - tap)] : 0.f;
        }
    }
}

// stage a (nz x ny x nx) voxel box starting at (z0,y0,x0) of sample n, channels [c0, c0+CH) of src (zero fill
// outside the volume / beyond Cvalid) into lds[vox*stride + c]
template <typename T, int CH, bool VEC8>
__device__ __forceinline__ void stage_box(const T* __restrict__ src, int cs, int Cvalid, int c0, float* lds, int stride,
                                          int n, int z0, int y0, int x0, int nz, int ny, int nx, int D, int H, int W) {
    int nvox = nz * ny * nx;
    if constexpr (VEC8) {
        constexpr int U = CH / 8;
        for (int idx = threadIdx.x; idx < nvox * U; idx += BLK) {
            int vox = idx / U, u = idx - vox * U;
            int x = vox % nx, t = vox / nx, y = t % ny, z = t / ny;
            int gz = z0 + z, gy = y0 + y, gx = x0 + x;
            float v[8];
            bool inb = gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W && (c0 + u * 8 < Cvalid);
            if (inb) ld8<T>(src + ((((int64_t)n * D + gz) * H + gy) * W + gx) * cs + c0 + u * 8, v);
            else {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = 0.f;
            }
            f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
            *reinterpret_cast<f32x4*>(lds + vox * stride + u * 8) = a;
            *reinterpret_cast<f32x4*>(lds + vox * stride + u * 8 + 4) = b;
        }
    } else {
        for (int idx = threadIdx.x; idx < nvox * CH; idx += BLK) {
            int vox = idx / CH, c = idx - vox * CH;
            int x = vox % nx, t = vox / nx, y = t % ny, z = t / ny;
            int gz = z0 + z, gy = y0 + y, gx = x0 + x;
            bool inb = gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W && (c0 + c < Cvalid);
            lds[vox * stride + c] = inb ? to_f<T>(src[((((int64_t)n * D + gz) * H + gy) * W + gx) * cs + c0 + c]) : 0.f;
        }
    }
}

// CIC = input channels staged per chunk: 8 (vector path, Cin % 8 == 0) or 1 (scalar path)
template <typename TI, typename TO, int CIC>
__global__ __launch_bounds__(BLK) void conv3_direct_kernel(const TI* __restrict__ x, int xcs, int Cin,
                                                           const float* __restrict__ wp, const float* __restrict__ bias,
                                                           TO* __restrict__ y, int ycs, int Cout, int D, int H, int W,
                                                           int tilesZ, int tilesY, int tilesX, int relu) {
    constexpr int STR = (CIC == 8) ? 12 : 1;   // LDS voxel stride (floats); 12 keeps b128 reads 16-B aligned
    __shared__ __attribute__((aligned(16))) float xs[IZ * IY * IX * STR];
    int tile = blockIdx.x;
    int tx_ = tile % tilesX; tile /= tilesX;
    int ty_ = tile % tilesY; tile /= tilesY;
    int tz_ = tile % tilesZ; int n = tile / tilesZ;
    int cb = blockIdx.y;
    int z0 = tz_ * TZ, y0 = ty_ * TY, x0 = tx_ * TX;
    int lx = threadIdx.x % TX, ly = (threadIdx.x / TX) % TY, lz = threadIdx.x / (TX * TY);

    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; j++) acc[j] = 0.f;

    for (int c0 = 0; c0 < Cin; c0 += CIC) {
        __syncthreads();
        stage_box<TI, CIC, CIC == 8>(x, xcs, Cin, c0, xs, STR, n, z0 - 1, y0 - 1, x0 - 1, IZ, IY, IX, D, H, W);
        __syncthreads();
        const float* wc = wp + ((int64_t)cb * Cin + c0) * 27 * COT;   // wave-uniform -> scalar loads
#pragma unroll
        for (int dz = 0; dz < 3; dz++)
#pragma unroll
            for (int dy = 0; dy < 3; dy++)
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {
                    int tap = (dz * 3 + dy) * 3 + dx;
                    const float* xp = xs + (((lz + dz) * IY + (ly + dy)) * IX + (lx + dx)) * STR;
                    if constexpr (CIC == 8) {
                        f32x4 a = *reinterpret_cast<const f32x4*>(xp), b = *reinterpret_cast<const f32x4*>(xp + 4);
                        float xv[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
                        for (int ci = 0; ci < 8; ci++) {
                            const float* wr = wc + ((int64_t)ci * 27 + tap) * COT;
#pragma unroll
                            for (int j = 0; j < COT; j++) acc[j] = fmaf(xv[ci], wr[j], acc[j]);
                        }
                    } else {
                        float xv = xp[0];
                        const float* wr = wc + (int64_t)tap * COT;
#pragma unroll
                        for (int j = 0; j < COT; j++) acc[j] = fmaf(xv, wr[j], acc[j]);
                    }
                }
    }
    int gz = z0 + lz, gy = y0 + ly, gx = x0 + lx;
    if (gz < D && gy < H && gx < W) {
        int co0 = cb * COT;
        if (bias) {
#pragma unroll
            for (int j = 0; j < COT; j++) if (co0 + j < Cout) acc[j] += bias[co0 + j];
        }
        if (relu) {                   // inference: BatchNorm folded into (weights, bias), ReLU here
#pragma unroll
            for (int j = 0; j < COT; j++) acc[j] = fmaxf(acc[j], 0.f);
        }
        TO* yp = y + ((((int64_t)n * D + gz) * H + gy) * W + gx) * ycs + co0;
        bool vec = (co0 + COT <= Cout) && (ycs % 8 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
        if (vec) {
            float lo[8], hi[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { lo[j] = acc[j]; hi[j] = acc[8 + j]; }
            st8<TO>(yp, lo);
            st8<TO>(yp + 8, hi);
        } else {
#pragma unroll
            for (int j = 0; j < COT; j++) if (co0 + j < Cout) yp[j] = from_f<TO>(acc[j]);
        }
    }
}

// ---------------------------------------------------------------------------------------------- wgrad
constexpr int WZ = 4, WY = 4, WX = 8;                 // wgrad tile (output voxels) = 128
constexpr int WIZ = WZ + 2, WIY = WY + 2, WIX = WX + 2;
constexpr int CB = 32;                                // channel block (both co and ci)
constexpr int MAXT = 7;                               // ceil(8*8*27 / 256)

template <typename TX_, typename TD, bool VX, bool VD>
__global__ __launch_bounds__(BLK) void conv3_wgrad_direct_kernel(const TX_* __restrict__ x, int xcs, int Cin,
                                                                 const TD* __restrict__ dy, int dycs, int Cout, int N,
                                                                 int D, int H, int W, int tilesZ, int tilesY, int tilesX,
                                                                 float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float dys[WZ * WY * WX * CB];
    __shared__ __attribute__((aligned(16))) float xs[WIZ * WIY * WIX * CB];
    int co0 = blockIdx.y * CB, ci0 = blockIdx.z * CB;
    int ncob = min(CB, Cout - co0), ncib = min(CB, Cin - ci0);
    int CO4 = (ncob + 3) / 4, CI4 = (ncib + 3) / 4;
    int ntask = 27 * CO4 * CI4;
    float acc[MAXT][16];
#pragma unroll
    for (int k = 0; k < MAXT; k++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[k][i] = 0.f;
    float dbacc = 0.f;
    int ntiles = N * tilesZ * tilesY * tilesX;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        int tx_ = t % tilesX; t /= tilesX;
        int ty_ = t % tilesY; t /= tilesY;
        int tz_ = t % tilesZ; int n = t / tilesZ;
        int z0 = tz_ * WZ, y0 = ty_ * WY, x0 = tx_ * WX;
        __syncthreads();
        stage_box<TD, CB, VD>(dy, dycs, Cout, co0, dys, CB, n, z0, y0, x0, WZ, WY, WX, D, H, W);
        stage_box<TX_, CB, VX>(x, xcs, Cin, ci0, xs, CB, n, z0 - 1, y0 - 1, x0 - 1, WIZ, WIY, WIX, D, H, W);
        __syncthreads();
        if (blockIdx.z == 0 && threadIdx.x < ncob) {
            float s = 0.f;
            for (int v = 0; v < WZ * WY * WX; v++) s += dys[v * CB + threadIdx.x];
            dbacc += s;
        }
#pragma unroll
        for (int k = 0; k < MAXT; k++) {
            int task = threadIdx.x + k * BLK;
            if (task < ntask) {
                int ci4 = task % CI4; int r = task / CI4; int co4 = r % CO4; int tap = r / CO4;
                int dx = tap % 3, dyy = (tap / 3) % 3, dz = tap / 9;
                const float* dp = dys + co4 * 4;
                const float* xp = xs + ((dz * WIY + dyy) * WIX + dx) * CB + ci4 * 4;
                for (int vz = 0; vz < WZ; vz++)
                    for (int vy = 0; vy < WY; vy++) {
#pragma unroll
                        for (int vx = 0; vx < WX; vx++) {
                            f32x4 g = *reinterpret_cast<const f32x4*>(dp + ((vz * WY + vy) * WX + vx) * CB);
                            f32x4 xv = *reinterpret_cast<const f32x4*>(xp + ((vz * WIY + vy) * WIX + vx) * CB);
#pragma unroll
                            for (int i = 0; i < 4; i++)
#pragma unroll
                                for (int j = 0; j < 4; j++) acc[k][i * 4 + j] = fmaf(g[i], xv[j], acc[k][i * 4 + j]);
                        }
                    }
            }
        }
    }
    int64_t slab_sz = (int64_t)Cout * Cin * 27 + Cout;
    float* slab = slabs + (int64_t)blockIdx.x * slab_sz;
#pragma unroll
    for (int k = 0; k < MAXT; k++) {
        int task = threadIdx.x + k * BLK;
        if (task < ntask) {
            int ci4 = task % CI4; int r = task / CI4; int co4 = r % CO4; int tap = r / CO4;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int co = co0 + co4 * 4 + i, ci = ci0 + ci4 * 4 + j;
                    if (co < Cout && ci < Cin) slab[((int64_t)co * Cin + ci) * 27 + tap] = acc[k][i * 4 + j];
                }
        }
    }
    if (blockIdx.z == 0 && threadIdx.x < ncob) slab[(int64_t)Cout * Cin * 27 + co0 + threadIdx.x] = dbacc;
}

inline int wgrad_nsb(int Cin, int Cout, Geo g) {
    int64_t ntiles = (int64_t)g.N * cdiv(g.D, WZ) * cdiv(g.H, WY) * cdiv(g.W, WX);
    int chan_blocks = cdiv(Cout, CB) * cdiv(Cin, CB);
    int64_t want = (512 + chan_blocks - 1) / chan_blocks;
    if (want < 1) want = 1;
    return (int)(ntiles < want ? ntiles : want);
}

}  // namespace

size_t conv3_direct_pack_floats(int Cin, int Cout) { return (size_t)cdiv(Cout, COT) * COT * Cin * 27; }

int conv3_direct_pack(const float* w, int Cin, int Cout, float* wp_fwd, float* wp_dgrad, hipStream_t s, const float* scale) {
    int64_t n = (int64_t)conv3_direct_pack_floats(Cin, Cout) + (int64_t)conv3_direct_pack_floats(Cout, Cin);
    pack_kernel<<<(int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256), 256, 0, s>>>(w, Cin, Cout, wp_fwd, wp_dgrad, scale);
    MI3D_LAUNCH_CHECK();
    return 0;
}

template <typename TI, typename TO>
static int launch_fwd(const void* x, int xcs, int Cin, const float* wp, const float* bias, void* y, int ycs, int Cout,
                      Geo g, hipStream_t s, int relu) {
    int tz = cdiv(g.D, TZ), ty = cdiv(g.H, TY), tx = cdiv(g.W, TX);
    dim3 grid((unsigned)(g.N * tz * ty * tx), (unsigned)cdiv(Cout, COT));
    bool v8 = Cin % 8 == 0 && xcs % 8 == 0 && ((uintptr_t)x % 16 == 0);
    if (v8)
        conv3_direct_kernel<TI, TO, 8><<<grid, BLK, 0, s>>>((const TI*)x, xcs, Cin, wp, bias, (TO*)y, ycs, Cout, g.D, g.H, g.W, tz, ty, tx, relu);
    else
        conv3_direct_kernel<TI, TO, 1><<<grid, BLK, 0, s>>>((const TI*)x, xcs, Cin, wp, bias, (TO*)y, ycs, Cout, g.D, g.H, g.W, tz, ty, tx, relu);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int conv3_direct_fwd(int in_dtype, int out_dtype, const void* x, int xcs, int Cin, const float* wp, const float* bias,
                     void* y, int ycs, int Cout, Geo g, hipStream_t s, int relu) {
    MI3D_CHECK_ARG(Cin >= 1 && Cout >= 1 && xcs >= Cin && ycs >= Cout, "conv3_direct_fwd: bad channels");
    MI3D_CHECK_ARG((int64_t)g.N * cdiv(g.D, TZ) * cdiv(g.H, TY) * cdiv(g.W, TX) < (1ll << 31), "conv3: grid too large");
    if (in_dtype == MI3D_F32 && out_dtype == MI3D_F32) return launch_fwd<float, float>(x, xcs, Cin, wp, bias, y, ycs, Cout, g, s, relu);
    if (in_dtype == MI3D_F32 && out_dtype == MI3D_BF16) return launch_fwd<float, bf16>(x, xcs, Cin, wp, bias, y, ycs, Cout, g, s, relu);
    if (in_dtype == MI3D_BF16 && out_dtype == MI3D_BF16) return launch_fwd<bf16, bf16>(x, xcs, Cin, wp, bias, y, ycs, Cout, g, s, relu);
    return launch_fwd<bf16, float>(x, xcs, Cin, wp, bias, y, ycs, Cout, g, s, relu);
}

size_t conv3_direct_wgrad_ws_floats(int Cin, int Cout, Geo g) {
    return (size_t)wgrad_nsb(Cin, Cout, g) * ((size_t)Cout * Cin * 27 + Cout);
}

template <typename TX_, typename TD>
static int launch_wgrad(const void* x, int xcs, int Cin, const void* dy, int dycs, int Cout, Geo g, float* ws, int nsb,
                        hipStream_t s) {
    int tz = cdiv(g.D, WZ), ty = cdiv(g.H, WY), tx = cdiv(g.W, WX);
    dim3 grid((unsigned)nsb, (unsigned)cdiv(Cout, CB), (unsigned)cdiv(Cin, CB));
    bool vx = xcs % 8 == 0 && Cin % 8 == 0 && ((uintptr_t)x % 16 == 0);
    bool vd = dycs % 8 == 0 && Cout % 8 == 0 && ((uintptr_t)dy % 16 == 0);
#define WG_LAUNCH(A, B) conv3_wgrad_direct_kernel<TX_, TD, A, B><<<grid, BLK, 0, s>>>((const TX_*)x, xcs, Cin, (const TD*)dy, dycs, Cout, g.N, g.D, g.H, g.W, tz, ty, tx, ws)
    if (vx && vd) WG_LAUNCH(true, true);
    else if (vx) WG_LAUNCH(true, false);
    else if (vd) WG_LAUNCH(false, true);
    else WG_LAUNCH(false, false);
#undef WG_LAUNCH
    MI3D_LAUNCH_CHECK();
    return 0;
}

int conv3_direct_wgrad(int x_dtype, int dy_dtype, const void* x, int xcs, int Cin, const void* dy, int dycs, int Cout,
                       Geo g, float* dW, float* db, int accumulate, float* ws, size_t ws_floats, hipStream_t s) {
    MI3D_CHECK_ARG(Cin >= 1 && Cout >= 1, "conv3_direct_wgrad: bad channels");
    int nsb = wgrad_nsb(Cin, Cout, g);
    size_t slab_sz = (size_t)Cout * Cin * 27 + Cout;
    MI3D_CHECK_ARG(ws_floats >= (size_t)nsb * slab_sz, "conv3_direct_wgrad: workspace too small (%zu < %zu floats)",
                   ws_floats, (size_t)nsb * slab_sz);
    int rc;
    if (x_dtype == MI3D_F32 && dy_dtype == MI3D_F32) rc = launch_wgrad<float, float>(x, xcs, Cin, dy, dycs, Cout, g, ws, nsb, s);
    else if (x_dtype == MI3D_F32) rc = launch_wgrad<float, bf16>(x, xcs, Cin, dy, dycs, Cout, g, ws, nsb, s);
    else if (dy_dtype == MI3D_F32) { mi3d_set_error("conv3_direct_wgrad: (bf16 x, f32 dy) unsupported"); return -1; }
    else rc = launch_wgrad<bf16, bf16>(x, xcs, Cin, dy, dycs, Cout, g, ws, nsb, s);
    MI3D_TRY(rc);
    int64_t nW = (int64_t)Cout * Cin * 27;
    return slab_reduce(ws, nsb, (int64_t)slab_sz, nW, dW, db, accumulate, s);
}
