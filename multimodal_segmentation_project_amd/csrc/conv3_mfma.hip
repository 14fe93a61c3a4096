// conv3_mfma.hip — 3x3x3 convolution (pad 1) as an implicit GEMM on the CDNA4 matrix cores, bf16 in / fp32 acc.
// Reference: nn.Conv3d(Cin,Cout,3,padding=1) forward + input-gradient, models/unet.py:11,15 (17 of the 18 convs:
// every layer with Cin % 16 == 0 and Cout % 16 == 0).  The same kernel computes dgrad on a flipped/transposed pack.
//
// GEMM view:  Y[co][v] = sum_k W[co][k] * X[k][v],   k = (chunk, tap, ci16)  (K = 28*Cin, tap 27 = zero pad)
//   v_mfma_f32_16x16x32_bf16  with  A = weights (16 co x 32 k),  B = activations (32 k x 16 voxels)
//   -> D[row = co][col = voxel]: each lane ends up with 4 consecutive output channels of ONE voxel = one 8-byte
//      channels-last store; 4 lanes complete a voxel's 16 channels (32 B), a wave writes 16 voxels.
//   one K-step (32 k) = 2 taps x 16 input channels: lane group g = lane>>4 -> tap 2s + (g>>1), channels 8(g&1)..+7,
//      so the B fragment of a lane is ONE 16-byte LDS read (ds_read_b128) of a channels-last halo tile.
//
// Workgroup = 4 waves, output tile = TZ x (TYB*BY) x (TXB*BX) voxels cut into 16-voxel M-blocks (BY x BX, BY*BX=16);
// the (tile+2)^3 halo of 16 input channels (32 B per voxel) is staged through LDS per 16-channel chunk with zero
// fill at the volume border; weights stream from a pre-packed, lane-ordered bf16 image (1 KB per wave-load, L2
// resident).  Forward epilogue fuses bias, bf16 rounding, and the per-channel (sum, sum^2) BatchNorm partials of the
// ROUNDED values (wave shuffles -> LDS -> one partial per workgroup; summed later in fixed order, no atomics).
//
// Bank behaviour (BX = 16): the 16 lanes ds_read_b128 services together read 16 distinct 16-B slots mod 256 B
// (stride 32 B, the two channel halves interleaved) -> conflict-free without padding.
#include "ops.h"

namespace {

constexpr int BLK = 256;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------ pack
// wf[(((chunk*14 + s)*COB + cob)*64 + lane)*8 + j] = W[cob*16 + (lane&15)][chunk*16 + 8*(g&1) + j][2s + (g>>1)]
// wd[(((chunk*14 + s)*CIB + cib)*64 + lane)*8 + j] = W[chunk*16 + 8*(g&1) + j][cib*16 + (lane&15)][26 - (2s + (g>>1))]
__global__ void pack_mfma_kernel(const float* __restrict__ w, int Cin, int Cout, bf16* __restrict__ wf, bf16* __restrict__ wd) {
    int64_t n = (int64_t)Cin * Cout * 28;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += (int64_t)gridDim.x * blockDim.x) {
        bool dg = i >= n;
        int64_t k = dg ? i - n : i;
        int j = k & 7; int lane = (k >> 3) & 63; int64_t r = k >> 9;
        int nob = (dg ? Cin : Cout) / 16;
        int ob = r % nob; r /= nob; int s = r % 14; int chunk = r / 14;
        int g = lane >> 4, tap = 2 * s + (g >> 1);
        int o = ob * 16 + (lane & 15), ic = chunk * 16 + 8 * (g & 1) + j;
        float v = 0.f;
        if (tap < 27) v = dg ? w[((int64_t)ic * Cin + o) * 27 + (26 - tap)] : w[((int64_t)o * Cin + ic) * 27 + tap];
        (dg ? wd : wf)[k] = (bf16)v;
    }
}

// ------------------------------------------------------------------------------------------------ kernel
template <int TZ, int TYB, int TXB, int BX, int COB, bool STATS>
__global__ __launch_bounds__(BLK) void conv3_mfma_kernel(const bf16* __restrict__ x, int xcs, int Cin,
                                                         const bf16* __restrict__ wp, const float* __restrict__ bias,
                                                         bf16* __restrict__ y, int ycs, int CoutTotal, int D, int H, int W,
                                                         int tilesZ, int tilesY, int tilesX, float* __restrict__ part) {
    constexpr int BY = 16 / BX;
    constexpr int TY = TYB * BY, TX = TXB * BX;
    constexpr int IZ = TZ + 2, IY = TY + 2, IX = TX + 2;
    static_assert(TZ == 4, "one z-slice of the tile per wave");
    constexpr int MB = TYB * TXB;                  // M-blocks per wave (wave w owns z-slice w of the tile)
    constexpr int NVOX = IZ * IY * IX;
    __shared__ __attribute__((aligned(16))) bf16 xs[NVOX * 16];
    __shared__ float red[4][COB][16][2];

    int tile = blockIdx.x;
    int tx_ = tile % tilesX; tile /= tilesX;
    int ty_ = tile % tilesY; tile /= tilesY;
    int tz_ = tile % tilesZ; int n = tile / tilesZ;
    int z0 = tz_ * TZ, y0 = ty_ * TY, x0 = tx_ * TX;
    int cobBase = blockIdx.y * COB;
    int nCobTotal = CoutTotal / 16;

    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int vn = lane & 15, g = lane >> 4;
    // per-lane LDS byte offset of its voxel inside an M-block + channel half
    int laneOff = (((vn / BX) * IX + (vn % BX)) * 16 + (g & 1) * 8) * 2 + wave * (IY * IX * 32);
    const char* xsb = reinterpret_cast<const char*>(xs);

    f32x4 acc[MB][COB];
#pragma unroll
    for (int r = 0; r < MB; r++)
#pragma unroll
        for (int c = 0; c < COB; c++) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    int nchunk = Cin / 16;
    for (int chunk = 0; chunk < nchunk; chunk++) {
        __syncthreads();
        // ---- stage the halo tile of this 16-channel chunk: 2 x 16-B pieces per voxel, zero fill outside the volume
        for (int idx = threadIdx.x; idx < NVOX * 2; idx += BLK) {
            int vox = idx >> 1, half = idx & 1;
            int ix = vox % IX, t = vox / IX, iy = t % IY, iz = t / IY;
            int gz = z0 - 1 + iz, gy = y0 - 1 + iy, gx = x0 - 1 + ix;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const bf16x8*>(x + ((((int64_t)n * D + gz) * H + gy) * W + gx) * xcs + chunk * 16 + half * 8);
            *reinterpret_cast<bf16x8*>(xs + vox * 16 + half * 8) = v;
        }
        __syncthreads();
        const bf16* wc = wp + ((int64_t)chunk * 14 * nCobTotal + cobBase) * 512 + lane * 8;
#pragma unroll
        for (int s = 0; s < 14; s++) {
            bf16x8 wf[COB];
#pragma unroll
            for (int c = 0; c < COB; c++) wf[c] = *reinterpret_cast<const bf16x8*>(wc + ((int64_t)s * nCobTotal + c) * 512);
            int t0 = 2 * s, t1 = (2 * s + 1 < 27) ? 2 * s + 1 : 26;
            int off0 = (((t0 / 9) * IY + ((t0 / 3) % 3)) * IX + (t0 % 3)) * 32;
            int off1 = (((t1 / 9) * IY + ((t1 / 3) % 3)) * IX + (t1 % 3)) * 32;
            int toff = laneOff + ((g >> 1) ? off1 : off0);
#pragma unroll
            for (int r = 0; r < MB; r++) {
                int rowOff = (((r / TXB) * BY) * IX + (r % TXB) * BX) * 32;     // compile-time after unrolling
                bf16x8 xf = *reinterpret_cast<const bf16x8*>(xsb + toff + rowOff);
#pragma unroll
                for (int c = 0; c < COB; c++) acc[r][c] = mfma16(wf[c], xf, acc[r][c]);
            }
        }
    }

    // ---- epilogue: bias, bf16 store (4 channels = 8 B per lane), BN partial statistics of the rounded values
    float s1[COB][4], s2[COB][4];
#pragma unroll
    for (int c = 0; c < COB; c++)
#pragma unroll
        for (int j = 0; j < 4; j++) s1[c][j] = s2[c][j] = 0.f;
#pragma unroll
    for (int r = 0; r < MB; r++) {
        int bz = wave, byb = r / TXB, bxb = r % TXB;
        int gz = z0 + bz, gy = y0 + byb * BY + vn / BX, gx = x0 + bxb * BX + vn % BX;
        bool ok = gz < D && gy < H && gx < W;
        bf16* yp = y + ((((int64_t)n * D + gz) * H + gy) * W + gx) * ycs + cobBase * 16 + g * 4;
#pragma unroll
        for (int c = 0; c < COB; c++) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = acc[r][c][j] + (bias ? bias[(cobBase + c) * 16 + g * 4 + j] : 0.f);
                o[j] = (bf16)v;
                if (STATS && ok) { float q = (float)o[j]; s1[c][j] += q; s2[c][j] += q * q; }
            }
            if (ok) *reinterpret_cast<bf16x4*>(yp + c * 16) = o;
        }
    }
    if constexpr (STATS) {
#pragma unroll
        for (int c = 0; c < COB; c++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float a = s1[c][j], b = s2[c][j];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (vn == 0) { red[wave][c][g * 4 + j][0] = a; red[wave][c][g * 4 + j][1] = b; }
            }
        __syncthreads();
        // part[((blk*2 + k)*C + ch]  (same layout bn_stats_finalize consumes); blk = blockIdx.x, channels of this group
        for (int idx = threadIdx.x; idx < COB * 16 * 2; idx += BLK) {
            int k = idx & 1, ch = idx >> 1;
            int c = ch / 16, cc = ch % 16;
            float v = red[0][c][cc][k] + red[1][c][cc][k] + red[2][c][cc][k] + red[3][c][cc][k];
            part[((int64_t)blockIdx.x * 2 + k) * CoutTotal + cobBase * 16 + ch] = v;
        }
    }
}

struct TileCfg { int tz, ty, tx; };

template <int TZ, int TYB, int TXB, int BX, int COB>
int launch_cfg(const bf16* x, int xcs, int Cin, const bf16* wp, const float* bias, bf16* y, int ycs, int Cout, Geo g,
               float* part, hipStream_t s) {
    constexpr int TY = TYB * (16 / BX), TX = TXB * BX;
    int tz = cdiv(g.D, TZ), ty = cdiv(g.H, TY), tx = cdiv(g.W, TX);
    dim3 grid((unsigned)(g.N * tz * ty * tx), (unsigned)(Cout / (16 * COB)));
    if (part)
        conv3_mfma_kernel<TZ, TYB, TXB, BX, COB, true><<<grid, BLK, 0, s>>>(x, xcs, Cin, wp, bias, y, ycs, Cout, g.D, g.H, g.W, tz, ty, tx, part);
    else
        conv3_mfma_kernel<TZ, TYB, TXB, BX, COB, false><<<grid, BLK, 0, s>>>(x, xcs, Cin, wp, bias, y, ycs, Cout, g.D, g.H, g.W, tz, ty, tx, nullptr);
    MI3D_LAUNCH_CHECK();
    return 0;
}

inline bool big_geo(Geo g) { return g.W >= 32 && g.H >= 16; }

}  // namespace

bool conv3_mfma_supported(int Cin, int Cout, int xcs, int ycs) {
    return Cin % 16 == 0 && Cout % 16 == 0 && xcs % 8 == 0 && ycs % 4 == 0 && Cin >= 16 && Cout >= 16;
}

size_t conv3_mfma_pack_elems(int Cin, int Cout) { return (size_t)Cin * Cout * 28; }   // one operand (fwd or dgrad)

int conv3_mfma_pack(const float* w, int Cin, int Cout, void* wp_fwd, void* wp_dgrad, hipStream_t s) {
    int64_t n = 2 * (int64_t)conv3_mfma_pack_elems(Cin, Cout);
    pack_mfma_kernel<<<cdiv(n, 256) > 2048 ? 2048 : cdiv(n, 256), 256, 0, s>>>(w, Cin, Cout, (bf16*)wp_fwd, (bf16*)wp_dgrad);
    MI3D_LAUNCH_CHECK();
    return 0;
}

// number of per-workgroup statistic partials the forward launch writes (0 if it would not run with stats)
int conv3_mfma_stat_blocks(Geo g) {
    if (big_geo(g)) return g.N * cdiv(g.D, 4) * cdiv(g.H, 8) * cdiv(g.W, 16);
    return g.N * cdiv(g.D, 4) * cdiv(g.H, 8) * cdiv(g.W, 8);
}

// y = conv(x, wp) (+ bias); part != NULL -> also write BN partial sums [nblk][2][Cout] of the rounded outputs
int conv3_mfma_fwd(const void* x, int xcs, int Cin, const void* wp, const float* bias, void* y, int ycs, int Cout, Geo g,
                   float* part, hipStream_t s) {
    MI3D_CHECK_ARG(conv3_mfma_supported(Cin, Cout, xcs, ycs), "conv3_mfma_fwd: unsupported channels %d->%d", Cin, Cout);
    MI3D_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0, "conv3_mfma_fwd: misaligned tensors");
    const bf16* xp = (const bf16*)x; const bf16* w = (const bf16*)wp; bf16* yp = (bf16*)y;
    bool two = Cout % 32 == 0;
    if (big_geo(g)) {
        if (two) return launch_cfg<4, 8, 1, 16, 2>(xp, xcs, Cin, w, bias, yp, ycs, Cout, g, part, s);
        return launch_cfg<4, 8, 1, 16, 1>(xp, xcs, Cin, w, bias, yp, ycs, Cout, g, part, s);
    }
    if (two) return launch_cfg<4, 2, 2, 4, 2>(xp, xcs, Cin, w, bias, yp, ycs, Cout, g, part, s);
    return launch_cfg<4, 2, 2, 4, 1>(xp, xcs, Cin, w, bias, yp, ycs, Cout, g, part, s);
}
