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
#include <stdlib.h>

#include <hip/hip_ext.h>

#include <utility>

#include "ops.h"
#include "slab_sum.h"

namespace {

constexpr int BLK = 256;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// K-step -> tap assignment.  mode 0 (natural): K-step s holds taps (2s, 2s+1).  mode 1 ("row reuse", persistent
// full-resolution kernel): taps are paired so that consecutive K-steps read the SAME LDS rows shifted by dy, which
// lets one fragment register serve up to three MFMAs (see conv3_mfma_persist_kernel):
//   s = 4dz+dy (dy<3): ((dz,dy,0),(dz,dy,1))   s = 4dz+3: ((dz,0,2),(dz,1,2))   s=12: ((0,2,2),(1,2,2))   s=13: ((2,2,2),pad)
MI3D_HD constexpr int ktap(int mode, int s, int h) {
    if (mode == 0) return 2 * s + h;
    if (s < 12) {
        int dz = s / 4, k = s % 4;
        return k < 3 ? dz * 9 + k * 3 + h : dz * 9 + h * 3 + 2;
    }
    if (s == 12) return h * 9 + 8;
    return h == 0 ? 26 : 27;
}

// ------------------------------------------------------------------------------------------------ pack
// wf[(((chunk*14 + s)*COB + cob)*64 + lane)*8 + j] = W[cob*16 + (lane&15)][chunk*16 + 8*(g&1) + j][2s + (g>>1)]
// wd[(((chunk*14 + s)*CIB + cib)*64 + lane)*8 + j] = W[chunk*16 + 8*(g&1) + j][cib*16 + (lane&15)][26 - (2s + (g>>1))]
__global__ void pack_mfma_kernel(const float* __restrict__ w, int Cin, int Cout, bf16* __restrict__ wf, bf16* __restrict__ wd,
                                 int mode_f, int mode_d) {
    int64_t n = (int64_t)Cin * Cout * 28;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += (int64_t)gridDim.x * blockDim.x) {
        bool dg = i >= n;
        int64_t k = dg ? i - n : i;
        int j = k & 7; int lane = (k >> 3) & 63; int64_t r = k >> 9;
        int nob = (dg ? Cin : Cout) / 16;
        int ob = r % nob; r /= nob; int s = r % 14; int chunk = r / 14;
        int g = lane >> 4, tap = ktap(dg ? mode_d : mode_f, s, g >> 1);
        int o = ob * 16 + (lane & 15), ic = chunk * 16 + 8 * (g & 1) + j;
        float v = 0.f;
        if (tap < 27) v = dg ? w[((int64_t)ic * Cin + o) * 27 + (26 - tap)] : w[((int64_t)o * Cin + ic) * 27 + tap];
        (dg ? wd : wf)[k] = (bf16)v;
    }
}

// One launch for every pack of the network.  conv3 unit = one (co-block, ci-block) pair of a layer: its 16 x (16*27)
// source floats are read as 16 contiguous 1.7 KB runs into LDS and both images (forward, dgrad) of the pair are
// written as 16-byte stores; upconv unit = 2048 packed elements (layout of upconv_mfma.hip, gathered directly).
constexpr int PK_LD = 16 * 27 + 1;
__global__ __launch_bounds__(BLK) void pack_all_kernel(PackJobs J) {
    __shared__ float wl[16 * PK_LD];
    if (blockIdx.x == 0 && J.zero)
        for (int i = threadIdx.x; i < J.nzero; i += BLK) J.zero[i] = 0;
    int b = blockIdx.x, ji = 0;
    while (ji + 1 < J.n && b >= J.j[ji + 1].blk0) ji++;
    const float* __restrict__ w = J.j[ji].w;
    int Cin = J.j[ji].Cin, Cout = J.j[ji].Cout, lb = b - J.j[ji].blk0;
    if (J.j[ji].kind == 0) {
        bf16* wf = (bf16*)J.j[ji].a; bf16* wd = (bf16*)J.j[ji].b;
        int mode_f = J.j[ji].mode_f, mode_d = J.j[ji].mode_d;
        int CIBN = Cin / 16, COBN = Cout / 16;
        int cib = lb % CIBN, cob = lb / CIBN;
        const float* __restrict__ sc = J.j[ji].scale;          // inference: per-output-channel BatchNorm scale (forward image)
        for (int idx = threadIdx.x; idx < 16 * 108; idx += BLK) {        // 16-byte loads: a row's 432 floats start 16-B aligned
            int row = idx / 108, k = (idx - row * 108) * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(w + ((int64_t)(cob * 16 + row) * Cin + cib * 16) * 27 + k);
            float f = sc ? sc[cob * 16 + row] : 1.f;
#pragma unroll
            for (int j = 0; j < 4; j++) wl[row * PK_LD + k + j] = v[j] * f;
        }
        __syncthreads();
        for (int q = threadIdx.x; q < 2 * 14 * 64; q += BLK) {
            int img = q / (14 * 64), r = q - img * (14 * 64);
            int s = r >> 6, lane = r & 63, g = lane >> 4, o = lane & 15;
            int tap = ktap(img ? mode_d : mode_f, s, g >> 1);
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                int ic = 8 * (g & 1) + j;
                float f = 0.f;
                if (tap < 27) f = img ? wl[ic * PK_LD + o * 27 + (26 - tap)] : wl[o * PK_LD + ic * 27 + tap];
                v[j] = (bf16)f;
            }
            bf16* dst = img ? wd + ((((int64_t)cob * 14 + s) * CIBN + cib) * 64 + lane) * 8
                            : wf + ((((int64_t)cib * 14 + s) * COBN + cob) * 64 + lane) * 8;
            *reinterpret_cast<bf16x8*>(dst) = v;
        }
    } else {
        bf16* wf = (bf16*)J.j[ji].a; bf16* wb = (bf16*)J.j[ji].b;
        int64_t n = (int64_t)Cin * Cout * 8;
        int KS = Cin / 32, COBN = Cout / 16, S = Cout / 4;
        int64_t k0 = ((int64_t)lb * BLK + threadIdx.x) * 8;
        if (k0 >= 2 * n) return;
        bool bw = k0 >= n;
        int64_t k = bw ? k0 - n : k0;
        int lane = (k >> 3) & 63, G = lane >> 4;
        int64_t r = k >> 9;
        bf16x8 v;
        if (!bw) {
            int ks = r % KS; r /= KS; int cb = r % COBN; int tap = r / COBN;
            int co = cb * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = (bf16)w[((int64_t)(32 * ks + 8 * G + j) * Cout + co) * 8 + tap];
        } else {
            int s = r % S; int cib = r / S;
            int kk0 = 32 * s + 8 * G, tap = kk0 / Cout, co0 = kk0 % Cout, ci = cib * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = (bf16)w[((int64_t)ci * Cout + co0 + j) * 8 + tap];
        }
        *reinterpret_cast<bf16x8*>((bw ? wb : wf) + k) = v;
    }
}

// ------------------------------------------------------------------------------------------------ kernel
// SPLITK: blockIdx.z owns a contiguous range of 16-channel input chunks and writes fp32 partial outputs
// part[kz][voxel][CoutTotal] (no bias / statistics); splitk_finish_kernel sums them.  Used for the deep levels
// where the spatial tile count alone cannot fill 256 CUs (M = N*V is small, K = 27*Cin is large).
// virtual block index / grid: the kernel bodies below are __device__ functions so that two of them can share one launch
// (conv3_bwd_fused_kernel: the input-gradient conv and the weight-gradient of a deep layer are independent and each
// only a dependent load -> MFMA -> store chain on a few hundred workgroups; run side by side they overlap)
// placed: x is already the tile index the caller wants (the body must not apply its own XCD remap)
struct Bid { int x, y, z, gx, gy, gz; bool placed = false; };
__device__ __forceinline__ Bid real_bid() {
    return Bid{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y, (int)gridDim.z};
}
// Workgroups are dealt round-robin over the 8 XCDs (private 4 MB L2 each): physical ids b, b+8, b+16 .. share an XCD and
// run at about the same time.  Bijective map [0,n) -> [0,n) that hands every XCD a CONTIGUOUS run of logical indices, so
// logically adjacent work (neighbouring tiles; the (co,ci)-block workgroups that re-read the same tile) meets in one L2.
__device__ __forceinline__ int xcd_contig(int b, int n) {
    int q8 = n / 8, r8 = n % 8, xcd = b % 8, idx = b / 8;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
}

template <int TZ, int TYB, int TXB, int BX, int COB, bool STATS, bool SPLITK, bool EXT_LDS = false>
__device__ __forceinline__ void conv3_mfma_body(Bid bid_, const bf16* __restrict__ x, int xcs, int Cin,
                                                const bf16* __restrict__ wp, const float* __restrict__ bias,
                                                bf16* __restrict__ y, int ycs, int CoutTotal, int D, int H, int W,
                                                int tilesZ, int tilesY, int tilesX, float* __restrict__ part,
                                                char* ext_lds = nullptr, int relu = 0) {
    constexpr int BY = 16 / BX;
    constexpr int TY = TYB * BY, TX = TXB * BX;
    constexpr int IZ = TZ + 2, IY = TY + 2, IX = TX + 2;
    static_assert(TZ == 4, "one z-slice of the tile per wave");
    constexpr int MB = TYB * TXB;                  // M-blocks per wave (wave w owns z-slice w of the tile)
    // LDS row pitch: 12 voxels instead of 10 for the 4 x 4-voxel M-blocks (bank conflicts, see conv3_mfma8_kernel)
    constexpr int IXP = BX == 4 ? 12 : IX;
    constexpr int NVOX = IZ * IY * IX, NVOXP = IZ * IY * IXP;
    // EXT_LDS: the tile lives in the caller's (dynamic) LDS block -- a fused launch shares one block between the bodies
    bf16* xs;
    float (*red)[COB][16][2];
    if constexpr (EXT_LDS) {
        xs = reinterpret_cast<bf16*>(ext_lds);
        red = reinterpret_cast<float (*)[COB][16][2]>(ext_lds + NVOXP * 32);
    } else {
        __shared__ __attribute__((aligned(16))) bf16 xs_s[NVOXP * 16];
        __shared__ float red_s[4][COB][16][2];
        xs = xs_s;
        red = red_s;
    }

    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2 each), so give every XCD a
    // CONTIGUOUS run of tiles -> halo voxels shared by neighbouring tiles hit in the same L2 (bijective remap)
    int tile = bid_.placed ? bid_.x : xcd_contig(bid_.x, bid_.gx);
    int tx_ = tile % tilesX; tile /= tilesX;
    int ty_ = tile % tilesY; tile /= tilesY;
    int tz_ = tile % tilesZ; int n = tile / tilesZ;
    int z0 = tz_ * TZ, y0 = ty_ * TY, x0 = tx_ * TX;
    int cobBase = bid_.y * COB;
    int nCobTotal = CoutTotal / 16;

    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int vn = lane & 15, g = lane >> 4;
    // per-lane LDS byte offset of its voxel inside an M-block + channel half
    int laneOff = (((vn / BX) * IXP + (vn % BX)) * 16 + (g & 1) * 8) * 2 + wave * (IY * IXP * 32);
    const char* xsb = reinterpret_cast<const char*>(xs);

    f32x4 acc[MB][COB];
#pragma unroll
    for (int r = 0; r < MB; r++)
#pragma unroll
        for (int c = 0; c < COB; c++) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging map: each thread moves NIT 16-byte pieces (voxel, channel half) per chunk; the voxel -> global offset map
    // is chunk-invariant, so it is computed once (-1 = outside the volume -> zero fill)
    constexpr int NIT = (NVOX * 2 + BLK - 1) / BLK;
    int soff[NIT], sdst[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        int idx = threadIdx.x + it * BLK;
        int vox = idx >> 1, half = idx & 1;
        int ix = vox % IX, t = vox / IX, iy = t % IY, iz = t / IY;
        sdst[it] = IXP == IX ? 0 : (t * IXP + ix) * 16 + half * 8;                 // element offset of the piece in the (padded) LDS tile
        int gz = z0 - 1 + iz, gy = y0 - 1 + iy, gx = x0 - 1 + ix;
        bool inb = idx < NVOX * 2 && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        soff[it] = inb ? ((gz * H + gy) * W + gx) * xcs + half * 8 : -1;
    }
    const bf16* xn = x + (int64_t)n * D * H * W * xcs;
    int nchunk = Cin / 16, chunk0 = 0;
    if (SPLITK) {
        int per = nchunk / bid_.gz;
        chunk0 = bid_.z * per;
        nchunk = chunk0 + per;
    }
    bf16x8 sv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        sv[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (soff[it] >= 0) sv[it] = *reinterpret_cast<const bf16x8*>(xn + soff[it] + chunk0 * 16);
    }
    // weight fragments stream through a register ring PD K-steps deep (L2-resident, 1 KB per wave-load): the load for
    // K-step s+PD is issued while K-step s computes, and the ring keeps running across chunk boundaries
    constexpr int PD = 7;      // must divide the 14 K-steps of a chunk so ring slots line up across chunks
    static_assert(14 % PD == 0, "ring depth must divide the K-steps per chunk");
    auto wptr = [&](int chunk, int s, int c) {
        return reinterpret_cast<const bf16x8*>(wp + (((int64_t)chunk * 14 + s) * nCobTotal + cobBase + c) * 512 + lane * 8);
    };
    bf16x8 wf[PD][COB];
#pragma unroll
    for (int s = 0; s < PD; s++)
#pragma unroll
        for (int c = 0; c < COB; c++) wf[s][c] = *wptr(chunk0, s, c);
    for (int chunk = chunk0; chunk < nchunk; chunk++) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            int idx = threadIdx.x + it * BLK;
            if (idx < NVOX * 2) *reinterpret_cast<bf16x8*>(xs + (IXP == IX ? idx * 8 : sdst[it])) = sv[it];
        }
        __syncthreads();
        // async-stage split: the next chunk's global loads are in flight while this chunk's MFMAs run
        if (chunk + 1 < nchunk) {
#pragma unroll
            for (int it = 0; it < NIT; it++)
                if (soff[it] >= 0) sv[it] = *reinterpret_cast<const bf16x8*>(xn + soff[it] + (chunk + 1) * 16);
        }
        // K loop: the weight ring advances once per K-step; LDS fragment reads are software-pipelined FG M-blocks
        // (one sub-step) ahead of the MFMAs that consume them
        constexpr int FG = MB < 4 ? MB : 4;
        constexpr int SUBS = MB / FG, NSUB = 14 * SUBS;
        auto frag_off = [&](int s) {
            int t0 = 2 * s, t1 = (2 * s + 1 < 27) ? 2 * s + 1 : 26;
            int off0 = (((t0 / 9) * IY + ((t0 / 3) % 3)) * IXP + (t0 % 3)) * 32;
            int off1 = (((t1 / 9) * IY + ((t1 / 3) % 3)) * IXP + (t1 % 3)) * 32;
            return laneOff + ((g >> 1) ? off1 : off0);
        };
        auto row_off = [&](int r) { return (((r / TXB) * BY) * IXP + (r % TXB) * BX) * 32; };
        bf16x8 xf[2][FG];
        {
            int toff = frag_off(0);
#pragma unroll
            for (int r = 0; r < FG; r++) xf[0][r] = *reinterpret_cast<const bf16x8*>(xsb + toff + row_off(r));
        }
        bf16x8 wcur[COB];
#pragma unroll
        for (int u = 0; u < NSUB; u++) {
            int s = u / SUBS, h = u % SUBS;
            if (h == 0) {
#pragma unroll
                for (int c = 0; c < COB; c++) wcur[c] = wf[s % PD][c];
                // refill this ring slot with the fragment PD K-steps ahead (possibly in the next chunk)
                if (s + PD < 14) {
#pragma unroll
                    for (int c = 0; c < COB; c++) wf[s % PD][c] = *wptr(chunk, s + PD, c);
                } else if (chunk + 1 < nchunk) {
#pragma unroll
                    for (int c = 0; c < COB; c++) wf[s % PD][c] = *wptr(chunk + 1, s + PD - 14, c);
                }
            }
            if (u + 1 < NSUB) {
                int s1 = (u + 1) / SUBS, h1 = (u + 1) % SUBS;
                int toff = frag_off(s1);
#pragma unroll
                for (int r = 0; r < FG; r++)
                    xf[(u + 1) & 1][r] = *reinterpret_cast<const bf16x8*>(xsb + toff + row_off(h1 * FG + r));
            }
#pragma unroll
            for (int r = 0; r < FG; r++)
#pragma unroll
                for (int c = 0; c < COB; c++) acc[h * FG + r][c] = mfma16(wcur[c], xf[u & 1][r], acc[h * FG + r][c]);
        }
    }

    if constexpr (SPLITK) {
        // fp32 partial tile: part[kz][voxel][CoutTotal], 4 channels (16 B) per lane
        int64_t Mtot = (int64_t)(bid_.gx / (tilesZ * tilesY * tilesX)) * D * H * W;
        float* pk = part + (int64_t)bid_.z * Mtot * CoutTotal;
#pragma unroll
        for (int r = 0; r < MB; r++) {
            int bz = wave, byb = r / TXB, bxb = r % TXB;
            int gz = z0 + bz, gy = y0 + byb * BY + vn / BX, gx = x0 + bxb * BX + vn % BX;
            if (gz < D && gy < H && gx < W) {
                float* pp = pk + ((((int64_t)n * D + gz) * H + gy) * W + gx) * CoutTotal + cobBase * 16 + g * 4;
#pragma unroll
                for (int c = 0; c < COB; c++) *reinterpret_cast<f32x4*>(pp + c * 16) = acc[r][c];
            }
        }
        return;
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
        bf16x4 oc[COB];
#pragma unroll
        for (int c = 0; c < COB; c++) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = acc[r][c][j] + (bias ? bias[(cobBase + c) * 16 + g * 4 + j] : 0.f);
                if (relu & 1) v = fmaxf(v, 0.f);       // inference: BatchNorm folded into (weights, bias), ReLU here
                o[j] = (bf16)v;
                if (STATS && ok) { float q = (float)o[j]; s1[c][j] += q; s2[c][j] += q * q; }
            }
            oc[c] = o;
        }
        if constexpr (COB == 2) {
            if (relu & 2) {          // wide store: see conv3_mfma8_kernel
                typedef unsigned __attribute__((ext_vector_type(2))) u32x2;
                typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
                u32x2 u0 = __builtin_bit_cast(u32x2, oc[0]), u1 = __builtin_bit_cast(u32x2, oc[1]);
                u32x2 p0 = __builtin_amdgcn_permlane16_swap(u0[0], u1[0], false, false);
                u32x2 p1 = __builtin_amdgcn_permlane16_swap(u0[1], u1[1], false, false);
                u32x4 wv = {p0[0], p1[0], p0[1], p1[1]};
                if (ok) *reinterpret_cast<u32x4*>(yp - g * 4 + (g & 1) * 16 + (g >> 1) * 8) = wv;
                continue;
            }
        }
#pragma unroll
        for (int c = 0; c < COB; c++)
            if (ok) *reinterpret_cast<bf16x4*>(yp + c * 16) = oc[c];
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
            part[((int64_t)bid_.x * 2 + k) * CoutTotal + cobBase * 16 + ch] = v;
        }
    }
}

template <int TZ, int TYB, int TXB, int BX, int COB, bool STATS, bool SPLITK>
__global__ __launch_bounds__(BLK) void conv3_mfma_kernel(const bf16* __restrict__ x, int xcs, int Cin,
                                                         const bf16* __restrict__ wp, const float* __restrict__ bias,
                                                         bf16* __restrict__ y, int ycs, int CoutTotal, int D, int H, int W,
                                                         int tilesZ, int tilesY, int tilesX, float* __restrict__ part, int relu) {
    conv3_mfma_body<TZ, TYB, TXB, BX, COB, STATS, SPLITK>(real_bid(), x, xcs, Cin, wp, bias, y, ycs, CoutTotal, D, H, W, tilesZ,
                                                          tilesY, tilesX, part, nullptr, relu);
}


// ------------------------------------------------------------------------------------------ eight-wave forward variant
constexpr size_t conv8_lds(int TY, int TX, int COB) {        // TX == 8 <=> BX == 4 tilings: LDS row pitch 12 voxels (see the kernel)
    return (size_t)6 * (TY + 2) * (TX == 8 ? 12 : TX + 2) * 32 + (size_t)14 * COB * 1024 + (size_t)8 * COB * 16 * 2 * 4;
}
constexpr size_t CONV8_XF_LDS = 256 * 6 * sizeof(float);     // XF coefficient table behind the kernel's own LDS block
// Levels 1-4 run ONE tile per workgroup with <= 2 workgroups per CU (432 tiles at level 1), so the serial chain of one
// workgroup (tile loads -> LDS -> K loop -> stores) IS the kernel time.  This variant halves that chain per wave: 8 waves,
// wave w owns z-slice w & 3 and HALF of the slice's M-blocks (w >> 2); staging is spread over 512 threads; the chunk's weight
// fragments are staged through LDS once per workgroup (no per-wave register ring, no vector-memory wait inside the K loop),
// which keeps the kernel under 128 VGPRs = 4 waves per SIMD.  Same arithmetic per output element as conv3_mfma_body (same
// K order, same fp32 accumulation) -> bit-identical outputs; the statistic partials sum 8 instead of 4 wave rows.
//
// XF != 0 ("apply on load", round 4, small-geometry tiling only): the input tensor is a BatchNorm output that was never
// written -- the staging pass computes it from the raw tensor(s) between the global load and the LDS store, and the launch
// that used to do that (bn_apply / bn_bwd_apply: one link of the deep-level chain each) disappears:
//   XF = 1  forward conv1 of a block:  z1 = relu(a*y0 + b) * drop      (x = y0 of conv0; a, b from conv0's statistics)
//   XF = 2  input gradient:            dy = g*m*dz + A*y + B            (x = dz, xf.y2 = the layer's own raw output y)
// The per-channel coefficients come from the <= 128 partial rows the statistics / reduction kernel left (summed in the
// prologue, in double, like the "small BatchNorm" consumers bn_apply_kernel<TRAIN> / bn_bwd_apply_kernel<SMALL> do); the
// workgroups of tile 0 / output group 0 publish stat[4][C] + running statistics (XF = 1) or dgamma / dbeta (XF = 2) for the
// channel chunks they own.  Output group 0 also WRITES the transformed tensor for the voxels inside its tile (xf.side): the
// weight-gradient kernels of the backward read it (z1 resp. dy) exactly as before.  Element for element the arithmetic is
// bn_apply_kernel's / bn_bwd_apply_kernel's (same fmaf order, same rounding), so both routes give the same bits.
template <int TZ, int TYB, int TXB, int BX, int COB, bool STATS, bool SPLITK, int XF = 0, bool TK = false>
__global__ __launch_bounds__(512, XF == 2 ? 2 : 4) void conv3_mfma8_kernel(const bf16* __restrict__ x, int xcs, int Cin,
                                                          const bf16* __restrict__ wp, const float* __restrict__ bias,
                                                          bf16* __restrict__ y, int ycs, int CoutTotal, int D, int H, int W,
                                                          int tilesZ, int tilesY, int tilesX, float* __restrict__ part, int relu,
                                                          XfArgs xf) {
    constexpr int NT = 512;
    constexpr int BY = 16 / BX;
    constexpr int TY = TYB * BY, TX = TXB * BX;
    constexpr int IZ = TZ + 2, IY = TY + 2, IX = TX + 2;
    // LDS row pitch in voxels.  With 4 x 4-voxel M-blocks (BX = 4) a 10-voxel pitch puts rows 0 and 3 of a block on the same
    // banks of a ds_read_b128 lane group (SQ_LDS_BANK_CONFLICT 24-29 % of the LDS-active cycles of these kernels); 12 voxels
    // (384 B = 128 mod 256) makes the 16 pieces of every lane group tile the 256-byte bank row exactly
    constexpr int IXP = BX == 4 ? 12 : IX;
    static_assert(TZ == 4, "one z-slice of the tile per wave pair");
    constexpr int MB = TYB * TXB, MBW = MB / 2;          // M-blocks per slice / per wave
    static_assert(MB % 2 == 0 && MBW <= 4, "two waves share a slice");
    constexpr int NVOX = IZ * IY * IX, NVOXP = IZ * IY * IXP;
    extern __shared__ __attribute__((aligned(16))) char lds8[];
    bf16* xs = reinterpret_cast<bf16*>(lds8);
    bf16* wl = xs + NVOXP * 16;                                   // [14][COB][64 lanes][8] of the current chunk
    float (*red)[COB][16][2] = reinterpret_cast<float (*)[COB][16][2]>(lds8 + (NVOXP * 16 + 14 * COB * 512) * 2);
    Bid bid_ = real_bid();
    int tile = xcd_contig(bid_.x, bid_.gx);
    [[maybe_unused]] const int tile_id = tile;
    int tx_ = tile % tilesX; tile /= tilesX;
    int ty_ = tile % tilesY; tile /= tilesY;
    int tz_ = tile % tilesZ; int n = tile / tilesZ;
    int z0 = tz_ * TZ, y0 = ty_ * TY, x0 = tx_ * TX;
    int cobBase = bid_.y * COB;
    int nCobTotal = CoutTotal / 16;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int zs = wave & 3, hb = wave >> 2;
    int vn = lane & 15, g = lane >> 4;
    int laneOff = (((vn / BX) * IXP + (vn % BX)) * 16 + (g & 1) * 8) * 2 + zs * (IY * IXP * 32);
    const char* xsb = reinterpret_cast<const char*>(xs);
    f32x4 acc[MBW][COB];
#pragma unroll
    for (int r = 0; r < MBW; r++)
#pragma unroll
        for (int c = 0; c < COB; c++) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NIT = (NVOX * 2 + NT - 1) / NT;
    int soff[NIT], sdst[NIT];
    [[maybe_unused]] int svox[NIT], own = 0;                // XF: voxel index of the piece (second raw tensor, side tensor), "this workgroup writes it" bits
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        int idx = threadIdx.x + it * NT;
        int vox = idx >> 1, half = idx & 1;
        int ix = vox % IX, t = vox / IX, iy = t % IY, iz = t / IY;
        sdst[it] = IXP == IX ? 0 : (t * IXP + ix) * 16 + half * 8;              // element offset of the piece in the (padded) LDS tile
        int gz = z0 - 1 + iz, gy = y0 - 1 + iy, gx = x0 - 1 + ix;
        bool inb = idx < NVOX * 2 && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        soff[it] = inb ? ((gz * H + gy) * W + gx) * xcs + half * 8 : -1;
        if constexpr (XF != 0) {
            svox[it] = (gz * H + gy) * W + gx;
            if (inb && iz >= 1 && iz <= TZ && iy >= 1 && iy <= TY && ix >= 1 && ix <= TX && bid_.y == 0 && xf.side) own |= 1 << it;
        }
    }
    const bf16* xn = x + (int64_t)n * D * H * W * xcs;
    int nchunk = Cin / 16, chunk0 = 0;
    if (SPLITK) {
        int per = nchunk / bid_.gz;
        chunk0 = bid_.z * per;
        nchunk = chunk0 + per;
    }
    bf16x8 sv[NIT];
    [[maybe_unused]] bf16x8 sv2[NIT];
    [[maybe_unused]] const bf16* x2n = nullptr;
    [[maybe_unused]] bf16* siden = nullptr;
    // XF coefficient table [k][256 channels of this workgroup's chunks]: k = a, b, d (Dropout3d scale), and for XF = 2 g, A, B
    [[maybe_unused]] float* cft = reinterpret_cast<float*>(lds8 + conv8_lds(TY, TX, COB));
    if constexpr (XF != 0) {
        x2n = xf.y2 + (int64_t)n * D * H * W * xf.y2cs;
        siden = xf.side ? xf.side + (int64_t)n * D * H * W * xf.side_cs : nullptr;
    }
    constexpr int NWI = (14 * COB * 64 + NT - 1) / NT;
    bf16x8 wv[NWI];
    auto load_chunk = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            sv[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (soff[it] >= 0) sv[it] = *reinterpret_cast<const bf16x8*>(xn + soff[it] + chunk * 16);
            if constexpr (XF == 2) {
                sv2[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (soff[it] >= 0) sv2[it] = *reinterpret_cast<const bf16x8*>(x2n + svox[it] * xf.y2cs + (threadIdx.x & 1) * 8 + chunk * 16);
            }
        }
#pragma unroll
        for (int i = 0; i < NWI; i++) {
            int q = threadIdx.x + i * NT;
            if (q < 14 * COB * 64) {
                int f = q >> 6, s_ = f / COB, c_ = f - s_ * COB;
                wv[i] = *reinterpret_cast<const bf16x8*>(wp + (((int64_t)chunk * 14 + s_) * nCobTotal + cobBase + c_) * 512 + (q & 63) * 8);
            }
        }
    };
    load_chunk(chunk0);                                   // (XF: in flight under the coefficient prologue)
    if constexpr (XF != 0) {
        const int C = xf.C, c0 = chunk0 * 16, nloc = (nchunk - chunk0) * 16;      // channels of the input tensor owned here
        // sum the partial rows [nrows][2][C] for (k, c) in [0,2) x [c0, c0 + nloc): 512 / (2 nloc) row groups, then the groups
        // in fixed order.  (Sums of <= 128 fp32 values of one sign pattern in double: the order does not change the result.)
        double* red8 = reinterpret_cast<double*>(lds8);                               // the tile area is free before the first stage
        const int npair = 2 * nloc, ngrp = NT / npair;
        {
            const int pair = threadIdx.x % npair, grp = threadIdx.x / npair;
            double acc_ = 0.0;
            if (grp < ngrp) {
                const int k = pair / nloc, cc = pair - k * nloc;
                for (int r = grp; r < xf.nrows; r += ngrp) acc_ += (double)xf.rows[((size_t)r * 2 + k) * C + c0 + cc];
                red8[grp * npair + pair] = acc_;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < nloc) {
            const int cc = threadIdx.x, c = c0 + cc;
            double s0 = 0.0, s1 = 0.0;
            for (int q = 0; q < ngrp; q++) { s0 += red8[q * npair + cc]; s1 += red8[q * npair + nloc + cc]; }
            const bool pub = xcd_contig(bid_.x, bid_.gx) == 0 && bid_.y == 0;        // tile 0, output group 0: one publisher per chunk range
            const float dsc = xf.drop ? xf.drop[(size_t)n * C + c] : 1.f;
            if constexpr (XF == 1) {
                double mean = s0 / (double)xf.M;
                double var = s1 / (double)xf.M - mean * mean;
                if (var < 0.0) var = 0.0;
                double inv = 1.0 / sqrt(var + (double)xf.eps);
                float a = (float)((double)xf.gamma[c] * inv);
                float b = (float)((double)xf.beta[c] - mean * (double)xf.gamma[c] * inv);
                cft[cc] = a; cft[256 + cc] = b; cft[512 + cc] = dsc;
                if (pub) {
                    xf.stat[c] = (float)mean; xf.stat[C + c] = (float)inv; xf.stat[2 * C + c] = a; xf.stat[3 * C + c] = b;
                    if (xf.momentum < 0.f) {
                        double* side = reinterpret_cast<double*>(xf.rmean);
                        if (side) { side[c] = mean; side[C + c] = xf.M > 1 ? var * (double)xf.M / (double)(xf.M - 1) : var; }
                    } else {
                        if (xf.rmean) xf.rmean[c] = (float)((1.0 - xf.momentum) * xf.rmean[c] + xf.momentum * mean);
                        if (xf.rvar) {
                            double unb = xf.M > 1 ? var * (double)xf.M / (double)(xf.M - 1) : var;
                            xf.rvar[c] = (float)((1.0 - xf.momentum) * xf.rvar[c] + xf.momentum * unb);
                        }
                        if (xf.nbt && c == 0) *xf.nbt += 1;
                    }
                }
            } else {
                const float mean = xf.stat[c], inv = xf.stat[C + c], a = xf.stat[2 * C + c], b = xf.stat[3 * C + c];
                const float cf0 = (float)(s0 / (double)xf.M), cf1 = (float)(s1 / (double)xf.M);
                const float gg = a, k = gg * cf1 * inv;
                cft[cc] = a; cft[256 + cc] = b; cft[512 + cc] = dsc;
                cft[768 + cc] = gg; cft[1024 + cc] = -k; cft[1280 + cc] = k * mean - gg * cf0;
                if (pub) {
                    if (xf.dgamma) xf.dgamma[c] = xf.accumulate ? xf.dgamma[c] + (float)s1 : (float)s1;
                    if (xf.dbeta) xf.dbeta[c] = xf.accumulate ? xf.dbeta[c] + (float)s0 : (float)s0;
                }
            }
        }
        // (the first chunk's barrier in front of the LDS stores orders the table writes before their first use)
    }
    auto frag_off = [&](int s) {
        int t0 = 2 * s, t1 = (2 * s + 1 < 27) ? 2 * s + 1 : 26;
        int off0 = (((t0 / 9) * IY + ((t0 / 3) % 3)) * IXP + (t0 % 3)) * 32;
        int off1 = (((t1 / 9) * IY + ((t1 / 3) % 3)) * IXP + (t1 % 3)) * 32;
        return laneOff + ((g >> 1) ? off1 : off0);
    };
    auto row_off = [&](int r) { int rg = hb * MBW + r; return (((rg / TXB) * BY) * IXP + (rg % TXB) * BX) * 32; };
    for (int chunk = chunk0; chunk < nchunk; chunk++) {
        __syncthreads();
        if constexpr (XF != 0) {
            // this thread's 8 channels of the chunk: (chunk - chunk0) * 16 + (threadIdx.x & 1) * 8 (NT is even: `half` is fixed);
            // four channels at a time, so that the coefficient registers stay few (the kernel lives under 128 VGPRs)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const float* ct = cft + (chunk - chunk0) * 16 + (threadIdx.x & 1) * 8 + q * 4;
                const f32x4 ca = *reinterpret_cast<const f32x4*>(ct), cb = *reinterpret_cast<const f32x4*>(ct + 256),
                            cd = *reinterpret_cast<const f32x4*>(ct + 512);
                [[maybe_unused]] f32x4 cg, cA, cB;
                if constexpr (XF == 2) {
                    cg = *reinterpret_cast<const f32x4*>(ct + 768); cA = *reinterpret_cast<const f32x4*>(ct + 1024);
                    cB = *reinterpret_cast<const f32x4*>(ct + 1280);
                }
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    if (soff[it] < 0) continue;                  // outside the volume: the conv's zero padding, not a transformed zero
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if constexpr (XF == 1) {
                            float t = fmaf((float)sv[it][q * 4 + i], ca[i], cb[i]);
                            t = t > 0.f ? t : 0.f;
                            sv[it][q * 4 + i] = (bf16)(t * cd[i]);
                        } else {
                            const float yv = (float)sv2[it][q * 4 + i], gv = (float)sv[it][q * 4 + i];
                            const float pre = fmaf(yv, ca[i], cb[i]);
                            const float m = pre > 0.f ? cd[i] : 0.f;
                            sv[it][q * 4 + i] = (bf16)fmaf(cg[i] * m, gv, fmaf(cA[i], yv, cB[i]));
                        }
                    }
                }
            }
            if (own) {
#pragma unroll
                for (int it = 0; it < NIT; it++)
                    if (own & (1 << it))
                        *reinterpret_cast<bf16x8*>(siden + svox[it] * xf.side_cs + (threadIdx.x & 1) * 8 + chunk * 16) = sv[it];
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            int idx = threadIdx.x + it * NT;
            if (idx < NVOX * 2) *reinterpret_cast<bf16x8*>(xs + (IXP == IX ? idx * 8 : sdst[it])) = sv[it];
        }
#pragma unroll
        for (int i = 0; i < NWI; i++) {
            int q = threadIdx.x + i * NT;
            if (q < 14 * COB * 64) *reinterpret_cast<bf16x8*>(wl + q * 8) = wv[i];
        }
        __syncthreads();
        if (chunk + 1 < nchunk) load_chunk(chunk + 1);        // in flight under this chunk's K loop
        bf16x8 xq[2][MBW];
        {
            int toff = frag_off(0);
#pragma unroll
            for (int r = 0; r < MBW; r++) xq[0][r] = *reinterpret_cast<const bf16x8*>(xsb + toff + row_off(r));
        }
#pragma unroll
        for (int s = 0; s < 14; s++) {
            bf16x8 wcur[COB];
#pragma unroll
            for (int c = 0; c < COB; c++) wcur[c] = *reinterpret_cast<const bf16x8*>(wl + ((s * COB + c) * 64 + lane) * 8);
            if (s + 1 < 14) {
                int toff = frag_off(s + 1);
#pragma unroll
                for (int r = 0; r < MBW; r++) xq[(s + 1) & 1][r] = *reinterpret_cast<const bf16x8*>(xsb + toff + row_off(r));
            }
#pragma unroll
            for (int r = 0; r < MBW; r++)
#pragma unroll
                for (int c = 0; c < COB; c++) acc[r][c] = mfma16(wcur[c], xq[s & 1][r], acc[r][c]);
        }
    }
    if constexpr (SPLITK) {
        int64_t Mtot = (int64_t)(bid_.gx / (tilesZ * tilesY * tilesX)) * D * H * W;
        float* pk = part + (int64_t)bid_.z * Mtot * CoutTotal;
#pragma unroll
        for (int r = 0; r < MBW; r++) {
            int rg = hb * MBW + r;
            int byb = rg / TXB, bxb = rg % TXB;
            int gz = z0 + zs, gy = y0 + byb * BY + vn / BX, gx = x0 + bxb * BX + vn % BX;
            if (gz < D && gy < H && gx < W) {
                float* pp = pk + ((((int64_t)n * D + gz) * H + gy) * W + gx) * CoutTotal + cobBase * 16 + g * 4;
#pragma unroll
                for (int c = 0; c < COB; c++) {
                    if constexpr (TK) {          // write-through (agent-coherent) store: the finishing workgroup sits on another XCD
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(pp + c * 16), "v"(acc[r][c]) : "memory");
                    } else
                        *reinterpret_cast<f32x4*>(pp + c * 16) = acc[r][c];
                }
            }
        }
        if constexpr (!TK) return;
        else {
            // ---- split-K ticket: the last of this (tile, output group)'s ks workgroups finishes the tile.  Release the partials
            // (agent scope: the other workgroups sit on other XCDs), take a ticket on a counter only these ks workgroups touch,
            // and if it is the last one acquire and sum -- k = 0 .. ks-1 in order, whichever workgroup does it: same bits every run
            // (the "I am last" flag lives in the kernel's one LDS array: a second __shared__ object can cost the K loop its
            // counted waits -- cdna_hip_programming.md, three .s-level traps (a))
            volatile int& s_last = *reinterpret_cast<volatile int*>(&red[0][0][0][0]);
            const int ks = bid_.gz;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the write-through stores above have been acknowledged
            __syncthreads();
            if (threadIdx.x == 0) {
                int* cnt = xf.tk_count + (tile_id * bid_.gy + bid_.y);
                const int old = atomicAdd(cnt, 1);
                const int lastv = old == ks - 1;
                s_last = lastv;
                if (lastv) atomicExch(cnt, 0);                      // left zero for the next launch that uses the counters
            }
            __syncthreads();
            const bool last = s_last != 0;
            __syncthreads();                                           // (red is reused below)
            if (!last) return;
            constexpr int NVT = TZ * TY * TX, G8 = COB * 2;          // voxels of the tile, 8-channel groups of this workgroup
            static_assert(NT % G8 == 0, "a thread keeps its channel group");
            const int grp = threadIdx.x % G8, ch0 = cobBase * 16 + grp * 8;
            float bv8[8], s1[8], s2[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { bv8[i] = bias ? bias[ch0 + i] : 0.f; s1[i] = s2[i] = 0.f; }
            for (int idx = threadIdx.x; idx < NVT * G8; idx += NT) {
                const int vox = idx / G8;
                const int vx = vox % TX, vy = (vox / TX) % TY, vz = vox / (TX * TY);
                const int gz = z0 + vz, gy = y0 + vy, gx = x0 + vx;
                if (gz >= D || gy >= H || gx >= W) continue;
                const int64_t row = (((int64_t)n * D + gz) * H + gy) * W + gx;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = bv8[i];
                // four k's requested at a time (a counted loop of dependent round trips cost 35 us per launch); added in k order
                for (int k0 = 0; k0 < ks; k0 += 4) {
                    f32x4 u[4], w4[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        u[j] = w4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (k0 + j < ks) {           // loads that bypass this XCD's L2 (agent-coherent), no acquire fence
                            const float* pq = part + ((int64_t)(k0 + j) * Mtot + row) * CoutTotal + ch0;
                            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(u[j]) : "v"(pq) : "memory");
                            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w4[j]) : "v"(pq + 4) : "memory");
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(w4[0]), "+v"(w4[1]), "+v"(w4[2]), "+v"(w4[3]) :: "memory");
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (k0 + j < ks) {
                            v[0] += u[j][0]; v[1] += u[j][1]; v[2] += u[j][2]; v[3] += u[j][3];
                            v[4] += w4[j][0]; v[5] += w4[j][1]; v[6] += w4[j][2]; v[7] += w4[j][3];
                        }
                }
                bf16x8 o;
#pragma unroll
                for (int i = 0; i < 8; i++) { o[i] = (bf16)v[i]; const float q = (float)o[i]; s1[i] += q; s2[i] = fmaf(q, q, s2[i]); }
                *reinterpret_cast<bf16x8*>(y + row * ycs + ch0) = o;
            }
            // lanes of one channel group (lane % G8) -> wave total -> the 8 waves in fixed order
#pragma unroll
            for (int i = 0; i < 8; i++) {
#pragma unroll
                for (int o_ = G8; o_ < 64; o_ <<= 1) { s1[i] += __shfl_xor(s1[i], o_, 64); s2[i] += __shfl_xor(s2[i], o_, 64); }
            }
            __syncthreads();                                           // (the K loop's LDS reads are long done; red is free)
            if (lane < G8) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    red[wave][(lane * 8 + i) / 16][(lane * 8 + i) % 16][0] = s1[i];
                    red[wave][(lane * 8 + i) / 16][(lane * 8 + i) % 16][1] = s2[i];
                }
            }
            __syncthreads();
            for (int idx = threadIdx.x; idx < COB * 16 * 2; idx += NT) {
                const int k = idx & 1, ch = idx >> 1;
                float v = 0.f;
#pragma unroll
                for (int w_ = 0; w_ < 8; w_++) v += red[w_][ch / 16][ch % 16][k];
                xf.tk_rows[((int64_t)bid_.x * 2 + k) * CoutTotal + cobBase * 16 + ch] = v;
            }
            return;
        }
    }
    float s1[COB][4], s2[COB][4];
#pragma unroll
    for (int c = 0; c < COB; c++)
#pragma unroll
        for (int j = 0; j < 4; j++) s1[c][j] = s2[c][j] = 0.f;
#pragma unroll
    for (int r = 0; r < MBW; r++) {
        int rg = hb * MBW + r;
        int byb = rg / TXB, bxb = rg % TXB;
        int gz = z0 + zs, gy = y0 + byb * BY + vn / BX, gx = x0 + bxb * BX + vn % BX;
        bool ok = gz < D && gy < H && gx < W;
        bf16* yp = y + ((((int64_t)n * D + gz) * H + gy) * W + gx) * ycs + cobBase * 16 + g * 4;
        bf16x4 oc[COB];
#pragma unroll
        for (int c = 0; c < COB; c++) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = acc[r][c][j] + (bias ? bias[(cobBase + c) * 16 + g * 4 + j] : 0.f);
                if (relu & 1) v = fmaxf(v, 0.f);
                o[j] = (bf16)v;
                if (STATS && ok) { float q = (float)o[j]; s1[c][j] += q; s2[c][j] += q * q; }
            }
            oc[c] = o;
        }
        if constexpr (COB == 2) {
            if (relu & 2) {
                // wide store (round 4): the lane's two 4-channel pieces (output blocks 0 and 1 of ONE voxel) trade halves with the
                // neighbouring 16-lane rows (v_permlane16_swap); lane (vn, g) then holds channels (g & 1) * 16 + (g >> 1) * 8 .. + 7:
                // one 16-B store per lane, the voxel's 64 B written by four lanes
                typedef unsigned __attribute__((ext_vector_type(2))) u32x2;
                typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
                u32x2 u0 = __builtin_bit_cast(u32x2, oc[0]), u1 = __builtin_bit_cast(u32x2, oc[1]);
                u32x2 p0 = __builtin_amdgcn_permlane16_swap(u0[0], u1[0], false, false);
                u32x2 p1 = __builtin_amdgcn_permlane16_swap(u0[1], u1[1], false, false);
                u32x4 wv = {p0[0], p1[0], p0[1], p1[1]};
                if (ok) *reinterpret_cast<u32x4*>(yp - g * 4 + (g & 1) * 16 + (g >> 1) * 8) = wv;
                continue;
            }
        }
#pragma unroll
        for (int c = 0; c < COB; c++)
            if (ok) *reinterpret_cast<bf16x4*>(yp + c * 16) = oc[c];
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
        for (int idx = threadIdx.x; idx < COB * 16 * 2; idx += NT) {
            int k = idx & 1, ch = idx >> 1;
            int c = ch / 16, cc = ch % 16;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; w++) v += red[w][c][cc][k];
            part[((int64_t)bid_.x * 2 + k) * CoutTotal + cobBase * 16 + ch] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ persistent variant
// Full-resolution layers (16->16, 32->16, 16->32: 60 % of all conv FLOPs and most of the bytes).  Same math and tile
// (4 x 8 x 16 voxels) as conv3_mfma_kernel, restructured around what the profile showed (instruction-issue bound,
// ~1100 instructions per 112 MFMAs, most of it per-workgroup setup):
//   * a workgroup is PERSISTENT and strides over tiles: the staging map (piece -> relative offset) and ALL weight
//     fragments (NCH*14*COB x 16 B per lane, <= 112 VGPRs) are set up once and stay in registers;
//   * interior tiles take a check-free staging path (one add per 16-B piece); only border tiles test coordinates;
//   * the next tile's (or chunk's) global loads are issued before the MFMA loop of the current one (T14 split);
//   * BatchNorm partial sums accumulate in registers across tiles -> ONE partial row per workgroup.
template <int COB, int NCH, bool STATS, bool EXT_LDS = false>
__device__ __forceinline__ void conv3_mfma_persist_body(Bid bid_, const bf16* __restrict__ x, int xcs,
                                                        const bf16* __restrict__ wp, const float* __restrict__ bias,
                                                        bf16* __restrict__ y, int ycs, int D, int H, int W,
                                                        int tilesZ, int tilesY, int tilesX, int ntiles,
                                                        float* __restrict__ part, Halves xh, Halves yh, char* ext_lds = nullptr,
                                                        int relu = 0) {
    constexpr int TZ = 4, TY = 8, TX = 16, IZ = 6, IY = 10, IX = 18, MB = 8;
    constexpr int NVOX = IZ * IY * IX, NIT = (NVOX * 2 + BLK - 1) / BLK;
    constexpr int CoutTotal = COB * 16;
    // weights: resident in registers when they are 56 VGPRs (16->16), otherwise resident in LDS (28 KB) so that two
    // workgroups still fit per CU (registers AND LDS)
    constexpr bool WLDS = true;      // registers go to the F/G row fragments; weights (14-28 KB) live in LDS
    constexpr int NWF = NCH * 14 * COB;
    bf16* xs; bf16* wl;
    float (*red)[COB][16][2];
    if constexpr (EXT_LDS) {              // fused launch: one dynamic LDS block shared with the other body
        xs = reinterpret_cast<bf16*>(ext_lds);
        wl = xs + NVOX * 16;
        red = reinterpret_cast<float (*)[COB][16][2]>(ext_lds + (NVOX * 16 + NWF * 512) * 2);
    } else {
        __shared__ __attribute__((aligned(16))) bf16 xs_s[NVOX * 16];
        __shared__ __attribute__((aligned(16))) bf16 wl_s[WLDS ? NWF * 512 : 8];
        __shared__ float red_s[4][COB][16][2];
        xs = xs_s; wl = wl_s; red = red_s;
    }
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int vn = lane & 15, g = lane >> 4;
    int laneOff = ((vn * 16 + (g & 1) * 8) * 2) + wave * (IY * IX * 32);
    const char* xsb = reinterpret_cast<const char*>(xs);

    // ---- once per workgroup: staging map and resident weights
    int rel[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        int idx = threadIdx.x + it * BLK;
        int vox = idx >> 1, half = idx & 1;
        int ix = vox % IX, t = vox / IX, iy = t % IY, iz = t / IY;
        rel[it] = ((iz * H + iy) * W + ix) * xcs + half * 8;
    }
    bool lastValid = threadIdx.x + (NIT - 1) * BLK < NVOX * 2;
    // halo coordinates of this thread's pieces, 12 bits each (ix:5 | iy:4 | iz:3), two per register: border tiles test
    // coordinates with a few bit-field ops instead of re-deriving them by constant division per piece and tile
    unsigned pk[(NIT + 1) / 2];
#pragma unroll
    for (int i = 0; i < (NIT + 1) / 2; i++) pk[i] = 0;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        int vox = (threadIdx.x + it * BLK) >> 1;
        int ix = vox % IX, t = vox / IX, iy = t % IY, iz = t / IY;
        if (iz > 7) iz = 7;                                    // only the invalid tail pieces of the last iteration
        pk[it >> 1] |= (unsigned)(ix | (iy << 5) | (iz << 9)) << ((it & 1) * 16);
    }
    bf16x8 wf[WLDS ? 1 : NCH][WLDS ? 1 : 14][COB];
    if constexpr (!WLDS) {
#pragma unroll
        for (int ch = 0; ch < NCH; ch++)
#pragma unroll
            for (int s = 0; s < 14; s++)
#pragma unroll
                for (int c = 0; c < COB; c++)
                    wf[ch][s][c] = *reinterpret_cast<const bf16x8*>(wp + (((int64_t)ch * 14 + s) * COB + c) * 512 + lane * 8);
    }
    float bv[COB][4];
#pragma unroll
    for (int c = 0; c < COB; c++)
#pragma unroll
        for (int j = 0; j < 4; j++) bv[c][j] = bias ? bias[c * 16 + g * 4 + j] : 0.f;
    float s1[COB][4], s2[COB][4];
#pragma unroll
    for (int c = 0; c < COB; c++)
#pragma unroll
        for (int j = 0; j < 4; j++) s1[c][j] = s2[c][j] = 0.f;

    bf16x8 sv[NIT];
    // tile order: z fastest (z-neighbours share 2 of 6 halo slices), then y, then x
    auto tile_origin = [&](int tile, int& n, int& z0, int& y0, int& x0) {
        int tz_ = tile % tilesZ; tile /= tilesZ;
        int ty_ = tile % tilesY; tile /= tilesY;
        int tx_ = tile % tilesX; n = tile / tilesX;
        z0 = tz_ * TZ; y0 = ty_ * TY; x0 = tx_ * TX;
    };
    auto load_pieces = [&](int tile, int chunk) {
        int n, z0, y0, x0;
        tile_origin(tile, n, z0, y0, x0);
        // element offset of halo voxel (0,0,0) of this tile (may point before the volume for border tiles: only
        // dereferenced where the coordinate test passes)
        int64_t base = ((((int64_t)n * D + (z0 - 1)) * H + (y0 - 1)) * W + (x0 - 1)) * xcs + chunk * 16 +
                       (chunk >= xh.split ? xh.delta : 0);
        const bf16* xb = x + base;
        bool interior = z0 >= 1 && z0 + TZ + 1 <= D && y0 >= 1 && y0 + TY + 1 <= H && x0 >= 1 && x0 + TX + 1 <= W;
        if (interior) {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                if (it < NIT - 1 || lastValid) sv[it] = *reinterpret_cast<const bf16x8*>(xb + rel[it]);
            }
        } else {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                unsigned c = pk[it >> 1] >> ((it & 1) * 16);
                unsigned gz = (unsigned)(z0 - 1) + ((c >> 9) & 7u), gy = (unsigned)(y0 - 1) + ((c >> 5) & 15u),
                         gx = (unsigned)(x0 - 1) + (c & 31u);
                bool inb = (it < NIT - 1 || lastValid) && gz < (unsigned)D && gy < (unsigned)H && gx < (unsigned)W;
                sv[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (inb) sv[it] = *reinterpret_cast<const bf16x8*>(xb + rel[it]);
            }
        }
    };

    // XCD-aware start: workgroups b, b+8, b+16.. share an XCD (round-robin dispatch) -> give them ADJACENT tiles so
    // the halo voxels neighbouring tiles share are served by that XCD's L2 instead of the fabric (bijective remap)
    int tile;
    {
        int nwg = bid_.gx, bid = bid_.x, q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8, idx = bid / 8;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    if (tile < ntiles) load_pieces(tile, 0);
    // weights -> LDS AFTER the first tile's loads are in flight (both are cold fetches; the tile loop's first barrier orders
    // these LDS writes before any fragment read)
    if constexpr (WLDS) {
        for (int i = threadIdx.x; i < NWF * 64; i += BLK)
            *reinterpret_cast<bf16x8*>(wl + i * 8) = *reinterpret_cast<const bf16x8*>(wp + (int64_t)i * 8);
    }
    for (; tile < ntiles; tile += bid_.gx) {
        f32x4 acc[MB][COB];
#pragma unroll
        for (int r = 0; r < MB; r++)
#pragma unroll
            for (int c = 0; c < COB; c++) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                int idx = threadIdx.x + it * BLK;
                if (it < NIT - 1 || idx < NVOX * 2) *reinterpret_cast<bf16x8*>(xs + idx * 8) = sv[it];
            }
            __syncthreads();
            if (ch + 1 < NCH) load_pieces(tile, ch + 1);
            else if (tile + bid_.gx < ntiles) load_pieces(tile + bid_.gx, 0);
            // K loop with LDS-row reuse (ktap mode 1).  For a fixed dz the three K-steps (dz,dy,dx0|dx1), dy = 0..2, of
            // M-block row y read halo row y+dy: 10 row fragments F[0..9] feed 24 MFMAs (2.4x fewer ds_read_b128); the
            // dx = 2 taps pair up over dy (G), the last two K-steps over dz.  70 instead of 112 LDS reads per chunk.
            auto W_ = [&](int s, int c) -> bf16x8 {
                if constexpr (WLDS) return *reinterpret_cast<const bf16x8*>(wl + ((ch * 14 + s) * COB + c) * 512 + lane * 8);
                else return wf[ch][s][c];
            };
            const int hx = (g >> 1) * 32;                       // lane half -> dx = 0 | 1
            const int hy = (g >> 1) * (IX * 32) + 64;           // lane half -> dy = 0 | 1 at dx = 2
            const int hz = (g >> 1) * (IY * IX * 32) + (2 * IX + 2) * 32;   // lane half -> dz = 0 | 1 at (dy,dx) = (2,2)
            bf16x8 F[10], G[8];
#pragma unroll
            for (int row = 0; row < 10; row++) F[row] = *reinterpret_cast<const bf16x8*>(xsb + laneOff + hx + row * (IX * 32));
#pragma unroll
            for (int dz = 0; dz < 3; dz++) {
#pragma unroll
                for (int r = 0; r < 8; r++)
                    G[r] = *reinterpret_cast<const bf16x8*>(xsb + laneOff + hy + (dz * IY + r) * (IX * 32));
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    bf16x8 wc_[COB];
#pragma unroll
                    for (int c = 0; c < COB; c++) wc_[c] = W_(dz * 4 + dy, c);
#pragma unroll
                    for (int r = 0; r < 8; r++)
#pragma unroll
                        for (int c = 0; c < COB; c++) acc[r][c] = mfma16(wc_[c], F[r + dy], acc[r][c]);
                }
                // next group's row fragments may overwrite F now (MFMAs read their operands at issue)
                if (dz < 2) {
#pragma unroll
                    for (int row = 0; row < 10; row++)
                        F[row] = *reinterpret_cast<const bf16x8*>(xsb + laneOff + hx + ((dz + 1) * IY + row) * (IX * 32));
                } else {
#pragma unroll
                    for (int r = 0; r < 8; r++) F[r] = *reinterpret_cast<const bf16x8*>(xsb + laneOff + hz + r * (IX * 32));
                }
                {
                    bf16x8 wc_[COB];
#pragma unroll
                    for (int c = 0; c < COB; c++) wc_[c] = W_(dz * 4 + 3, c);
#pragma unroll
                    for (int r = 0; r < 8; r++)
#pragma unroll
                        for (int c = 0; c < COB; c++) acc[r][c] = mfma16(wc_[c], G[r], acc[r][c]);
                }
            }
            // s = 12: ((0,2,2),(1,2,2)) fragments are in F[0..7]; s = 13: ((2,2,2), pad) -> G
#pragma unroll
            for (int r = 0; r < 8; r++)
                G[r] = *reinterpret_cast<const bf16x8*>(xsb + laneOff + (2 * IY * IX + 2 * IX + 2) * 32 + r * (IX * 32));
            {
                bf16x8 wc_[COB];
#pragma unroll
                for (int c = 0; c < COB; c++) wc_[c] = W_(12, c);
#pragma unroll
                for (int r = 0; r < 8; r++)
#pragma unroll
                    for (int c = 0; c < COB; c++) acc[r][c] = mfma16(wc_[c], F[r], acc[r][c]);
#pragma unroll
                for (int c = 0; c < COB; c++) wc_[c] = W_(13, c);
#pragma unroll
                for (int r = 0; r < 8; r++)
#pragma unroll
                    for (int c = 0; c < COB; c++) acc[r][c] = mfma16(wc_[c], G[r], acc[r][c]);
            }
        }
        // ---- epilogue of this tile
        int n, z0, y0, x0;
        tile_origin(tile, n, z0, y0, x0);
        int gz = z0 + wave, gx = x0 + vn;
        bf16* yrow = y + ((((int64_t)n * D + gz) * H + y0) * W + gx) * ycs + g * 4;
        bool okzx = gz < D && gx < W;
        if (relu & 2) {
            // wide stores (round 4): a lane holds 4 channels (8 B) of one voxel per M-block row; rows r and r + 1 trade halves through
            // v_permlane16_swap (odd 16-lane rows of the first operand <-> even rows of the second), after which lane (vn, g) holds 8
            // consecutive channels (g >> 1) * 8 .. + 7 of voxel vn in row r + (g & 1): ONE 16-B store instead of two 8-B ones
            bf16* ywide = y + ((((int64_t)n * D + gz) * H + y0 + (g & 1)) * W + gx) * ycs + (g >> 1) * 8;
#pragma unroll
            for (int r = 0; r < MB; r += 2) {
                const bool okw = okzx && (y0 + r + (g & 1)) < H;
#pragma unroll
                for (int c = 0; c < COB; c++) {
                    bf16x4 oa, ob;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float va = acc[r][c][j] + bv[c][j], vb = acc[r + 1][c][j] + bv[c][j];
                        if (relu & 1) { va = fmaxf(va, 0.f); vb = fmaxf(vb, 0.f); }
                        oa[j] = (bf16)va; ob[j] = (bf16)vb;
                        if (STATS) {
                            float qa = (okzx && (y0 + r) < H) ? (float)oa[j] : 0.f, qb = (okzx && (y0 + r + 1) < H) ? (float)ob[j] : 0.f;
                            s1[c][j] += qa; s2[c][j] = fmaf(qa, qa, s2[c][j]);
                            s1[c][j] += qb; s2[c][j] = fmaf(qb, qb, s2[c][j]);
                        }
                    }
                    typedef unsigned __attribute__((ext_vector_type(2))) u32x2;
                    typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
                    u32x2 ua = __builtin_bit_cast(u32x2, oa), ub = __builtin_bit_cast(u32x2, ob);
                    u32x2 s0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
                    u32x2 s1_ = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
                    u32x4 w = {s0[0], s1_[0], s0[1], s1_[1]};
                    if (okw) *reinterpret_cast<u32x4*>(ywide + (int64_t)r * W * ycs + c * 16 + (c >= yh.split ? yh.delta : 0)) = w;
                }
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < MB; r++) {
            bool ok = okzx && (y0 + r) < H;
#pragma unroll
            for (int c = 0; c < COB; c++) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float v = acc[r][c][j] + bv[c][j];
                    if (relu) v = fmaxf(v, 0.f);
                    o[j] = (bf16)v;
                    if (STATS) { float q = ok ? (float)o[j] : 0.f; s1[c][j] += q; s2[c][j] = fmaf(q, q, s2[c][j]); }
                }
                if (ok) *reinterpret_cast<bf16x4*>(yrow + (int64_t)r * W * ycs + c * 16 + (c >= yh.split ? yh.delta : 0)) = o;
            }
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
        for (int idx = threadIdx.x; idx < COB * 16 * 2; idx += BLK) {
            int k = idx & 1, chn = idx >> 1;
            int c = chn / 16, cc = chn % 16;
            float v = (red[0][c][cc][k] + red[1][c][cc][k]) + (red[2][c][cc][k] + red[3][c][cc][k]);
            part[((int64_t)bid_.x * 2 + k) * CoutTotal + chn] = v;
        }
    }
}

template <int COB, int NCH, bool STATS>
__global__ __launch_bounds__(BLK, 2) void conv3_mfma_persist_kernel(const bf16* __restrict__ x, int xcs,
                                                                    const bf16* __restrict__ wp, const float* __restrict__ bias,
                                                                    bf16* __restrict__ y, int ycs, int D, int H, int W,
                                                                    int tilesZ, int tilesY, int tilesX, int ntiles,
                                                                    float* __restrict__ part, Halves xh, Halves yh, int relu) {
    conv3_mfma_persist_body<COB, NCH, STATS>(real_bid(), x, xcs, wp, bias, y, ycs, D, H, W, tilesZ, tilesY, tilesX, ntiles, part, xh, yh,
                                             nullptr, relu);
}

#ifdef MI3D_EXPERIMENTS      // default-off experiment routes are compiled only into experiment builds (make EXPERIMENTS=1)
// ------------------------------------------------------------------------ persistent variant with asynchronous staging
// Cout = 16 forward layers at full resolution (16 -> 16, 32 -> 16).  Same tile, K-step order and epilogue as the kernel
// above; what changes is how a workgroup's phases relate.  There, a chunk is  barrier | LDS writes | barrier | issue the next
// loads | MFMAs | stores  in series (stamps: the MFMA phase is half of a wave's time) and only the partner workgroup on the
// CU hides any of it.  Here the halo tile goes global -> LDS by DMA (buffer_load ... lds: no staging registers, no ds_write,
// out-of-bounds pieces arrive as zeros from the buffer bounds check) into the OTHER of two LDS tiles while the MFMAs of the
// current chunk run, with ONE barrier per chunk; the registers the staging freed hold all weight fragments (no weight reads
// from LDS), and the fragment reads are a hand-pipelined stream (ring of 7, counted lgkmcnt waits).
typedef __attribute__((address_space(3))) void lds_void_t;

template <int NCH>
struct DmaMma {
    static constexpr int IY = 10, IX = 18, RS = IX * 32, PS = IY * IX * 32;
    static constexpr int NFRAG = 70, PF = 6, NR = PF + 1;
    const char *pA, *pB, *pH, *pI;
    bf16x8 R[NR];
    template <int I>
    __device__ __forceinline__ bf16x8 fload() const {
        if constexpr (I < 54) {
            constexpr int dz = I / 18, k = I % 18;
            if constexpr (k < 10) return *reinterpret_cast<const bf16x8*>(pA + (dz * IY + k) * RS);
            else return *reinterpret_cast<const bf16x8*>(pB + (dz * IY + (k - 10)) * RS);
        } else if constexpr (I < 62) {
            return *reinterpret_cast<const bf16x8*>(pH + (I - 54) * RS);
        } else {
            return *reinterpret_cast<const bf16x8*>(pI + (I - 62) * RS);
        }
    }
    template <int CH, int I, typename Side>
    __device__ __forceinline__ void step(f32x4 (&acc)[8], const bf16x8 (&Wr)[NCH][14], Side& side) {
        // one DMA piece of the next chunk every 7 fragments: spread under the MFMAs instead of a burst in front of them
        if constexpr (I >= 2 && (I - 2) % 7 == 0 && (I - 2) / 7 < 9) side(std::integral_constant<int, (I - 2) / 7>{});
        if constexpr (I + PF < NFRAG) R[(I + PF) % NR] = fload<I + PF>();
        const bf16x8 f = R[I % NR];
        if constexpr (I < 54) {
            constexpr int dz = I / 18, k = I % 18;
            if constexpr (k < 10) {
                if constexpr (k < 8) acc[k] = mfma16(Wr[CH][dz * 4 + 0], f, acc[k]);
                if constexpr (k >= 1 && k - 1 < 8) acc[k - 1] = mfma16(Wr[CH][dz * 4 + 1], f, acc[k - 1]);
                if constexpr (k >= 2 && k - 2 < 8) acc[k - 2] = mfma16(Wr[CH][dz * 4 + 2], f, acc[k - 2]);
            } else {
                acc[k - 10] = mfma16(Wr[CH][dz * 4 + 3], f, acc[k - 10]);
            }
        } else if constexpr (I < 62) {
            acc[I - 54] = mfma16(Wr[CH][12], f, acc[I - 54]);
        } else {
            acc[I - 62] = mfma16(Wr[CH][13], f, acc[I - 62]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int... Is>
    __device__ __forceinline__ void prologue(std::integer_sequence<int, Is...>) { ((R[Is] = fload<Is>()), ...); }
    template <int CH, typename Side, int... Is>
    __device__ __forceinline__ void run(f32x4 (&acc)[8], const bf16x8 (&Wr)[NCH][14], Side& side, std::integer_sequence<int, Is...>) {
        (step<CH, Is>(acc, Wr, side), ...);
    }
    template <int CH, typename Side>
    __device__ __forceinline__ void chunk(f32x4 (&acc)[8], const bf16x8 (&Wr)[NCH][14], const char* tile, int laneOff, int g,
                                          Side& side) {
        pA = tile + laneOff + (g >> 1) * 32;                                   // dx = 0 | 1
        pB = tile + laneOff + (g >> 1) * RS + 64;                              // dy = 0 | 1 at dx = 2
        pH = tile + laneOff + (g >> 1) * PS + (2 * IX + 2) * 32;               // dz = 0 | 1 at (dy, dx) = (2, 2)
        pI = tile + laneOff + 2 * PS + 2 * RS + 64;                            // (2, 2, 2) | pad
        prologue(std::make_integer_sequence<int, PF>{});
        __builtin_amdgcn_sched_barrier(0);
        run<CH>(acc, Wr, side, std::make_integer_sequence<int, NFRAG>{});
    }
};

constexpr int DMA_NIT = 9, DMA_TILE_BYTES = DMA_NIT * BLK * 16;      // 36 864 B per LDS tile (6*10*18 voxels * 32 B = 34 560 used)
constexpr unsigned DMA_OOB = 0xffffff00u;                             // >= num_records: the bounds check returns zeros

template <int NCH, bool STATS>
__global__ __launch_bounds__(BLK, 2) void conv3_mfma_persist_dma_kernel(const bf16* __restrict__ x, int xcs,
                                                                        const bf16* __restrict__ wp, const float* __restrict__ bias,
                                                                        bf16* __restrict__ y, int ycs, int D, int H, int W,
                                                                        int tilesZ, int tilesY, int tilesX, int ntiles,
                                                                        float* __restrict__ part, Halves xh, Halves yh, int relu) {
    constexpr int TZ = 4, TY = 8, TX = 16, IY = 10, IX = 18, MB = 8, NVOX = 6 * IY * IX;
    extern __shared__ __attribute__((aligned(16))) char dma_lds[];
    float (*red)[16][2] = reinterpret_cast<float (*)[16][2]>(dma_lds + 2 * DMA_TILE_BYTES);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int vn = lane & 15, g = lane >> 4;
    const int laneOff = ((vn * 16 + (g & 1) * 8) * 2) + wave * (IY * IX * 32);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(x), (short)0, (int)DMA_OOB, 0x00020000);

    // staging map, tile-invariant: byte offset of this thread's pieces relative to halo voxel (0,0,0), and their halo
    // coordinates packed 12 bits each for the border tiles
    unsigned relb[DMA_NIT];
    unsigned pk[(DMA_NIT + 1) / 2];
#pragma unroll
    for (int i = 0; i < (DMA_NIT + 1) / 2; i++) pk[i] = 0;
#pragma unroll
    for (int it = 0; it < DMA_NIT; it++) {
        int idx = threadIdx.x + it * BLK;
        int vox = idx >> 1, half = idx & 1;
        int ix = vox % IX, t = vox / IX, iy = t % IY, iz = t / IY;
        relb[it] = idx < NVOX * 2 ? (unsigned)((((iz * H + iy) * W + ix) * xcs + half * 8) * 2) : DMA_OOB;
        if (iz > 7) iz = 7;
        pk[it >> 1] |= (unsigned)(ix | (iy << 5) | (iz << 9)) << ((it & 1) * 16);
    }
    // all weight fragments stay in registers (ktap mode 1 pack: [chunk][14 K-steps][lane][8])
    bf16x8 Wr[NCH][14];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++)
#pragma unroll
        for (int s_ = 0; s_ < 14; s_++) Wr[ch][s_] = *reinterpret_cast<const bf16x8*>(wp + ((int64_t)ch * 14 + s_) * 512 + lane * 8);
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) bv[j] = bias ? bias[g * 4 + j] : 0.f;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};

    auto tile_origin = [&](int tile, int& n, int& z0, int& y0, int& x0) {
        int tz_ = tile % tilesZ; tile /= tilesZ;
        int ty_ = tile % tilesY; tile /= tilesY;
        int tx_ = tile % tilesX; n = tile / tilesX;
        z0 = tz_ * TZ; y0 = ty_ * TY; x0 = tx_ * TX;
    };
    // DMA of (tile, chunk) into LDS tile `buf`: each wave-instruction lands 64 consecutive 16-byte pieces.  stage_prep
    // computes the nine byte offsets (out of bounds -> DMA_OOB), stage_piece<it> issues one piece
    unsigned soff[DMA_NIT];
    char* sdst = dma_lds;
    bool s_on = false;
    auto stage_prep = [&](int tile, int chunk, int buf) {
        int n, z0, y0, x0;
        tile_origin(tile, n, z0, y0, x0);
        const int64_t base = ((((int64_t)n * D + (z0 - 1)) * H + (y0 - 1)) * W + (x0 - 1)) * xcs + chunk * 16 +
                             (chunk >= xh.split ? xh.delta : 0);
        const unsigned baseb = (unsigned)(base * 2);         // may wrap below zero for border tiles: only used where in bounds
        const bool interior = z0 >= 1 && z0 + TZ + 1 <= D && y0 >= 1 && y0 + TY + 1 <= H && x0 >= 1 && x0 + TX + 1 <= W;
        sdst = dma_lds + buf * DMA_TILE_BYTES + wave * 1024;
#pragma unroll
        for (int it = 0; it < DMA_NIT; it++) {
            unsigned off = relb[it] == DMA_OOB ? DMA_OOB : baseb + relb[it];
            if (!interior) {
                unsigned c = pk[it >> 1] >> ((it & 1) * 16);
                unsigned gz = (unsigned)(z0 - 1) + ((c >> 9) & 7u), gy = (unsigned)(y0 - 1) + ((c >> 5) & 15u),
                         gx = (unsigned)(x0 - 1) + (c & 31u);
                bool inb = gz < (unsigned)D && gy < (unsigned)H && gx < (unsigned)W;
                off = inb ? off : DMA_OOB;
            }
            soff[it] = off;
        }
        s_on = true;
    };
    auto stage_piece = [&](auto itc) {
        constexpr int it = decltype(itc)::value;
        if (s_on) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(sdst + it * (BLK * 16)), 16, soff[it], 0, 0, 0);
    };
    auto stage_all = [&]() {
        stage_piece(std::integral_constant<int, 0>{}); stage_piece(std::integral_constant<int, 1>{});
        stage_piece(std::integral_constant<int, 2>{}); stage_piece(std::integral_constant<int, 3>{});
        stage_piece(std::integral_constant<int, 4>{}); stage_piece(std::integral_constant<int, 5>{});
        stage_piece(std::integral_constant<int, 6>{}); stage_piece(std::integral_constant<int, 7>{});
        stage_piece(std::integral_constant<int, 8>{});
        static_assert(DMA_NIT == 9, "stage_all issues nine pieces");
    };

    int tile;
    {
        int nwg = gridDim.x, bid = blockIdx.x, q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8, idx = bid / 8;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    int buf = 0;
    if (tile < ntiles) { stage_prep(tile, 0, 0); stage_all(); }
    DmaMma<NCH> mm;
    for (; tile < ntiles; tile += gridDim.x) {
        f32x4 acc[MB];
#pragma unroll
        for (int r = 0; r < MB; r++) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto one_chunk = [&](auto chc) {
            constexpr int ch = decltype(chc)::value;
            // my pieces of the current tile have landed; after the barrier everyone's have, and everyone has finished
            // reading the other tile (the previous chunk's MFMAs) -> it may be overwritten by the next DMA
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            s_on = false;
            if (ch + 1 < NCH) stage_prep(tile, ch + 1, buf ^ 1);
            else if (tile + (int)gridDim.x < ntiles) stage_prep(tile + gridDim.x, 0, buf ^ 1);
            stage_all();
            auto no_side = [](auto) {};
            mm.template chunk<ch>(acc, Wr, dma_lds + buf * DMA_TILE_BYTES, laneOff, g, no_side);
            buf ^= 1;
        };
        one_chunk(std::integral_constant<int, 0>{});
        if constexpr (NCH > 1) one_chunk(std::integral_constant<int, NCH - 1>{});
        static_assert(NCH <= 2, "one or two 16-channel chunks");
        // ---- epilogue of this tile
        int n, z0, y0, x0;
        tile_origin(tile, n, z0, y0, x0);
        int gz = z0 + wave, gx = x0 + vn;
        bf16* yrow = y + ((((int64_t)n * D + gz) * H + y0) * W + gx) * ycs + g * 4 + (0 >= yh.split ? yh.delta : 0);
        bool okzx = gz < D && gx < W;
#pragma unroll
        for (int r = 0; r < MB; r++) {
            bool ok = okzx && (y0 + r) < H;
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = acc[r][j] + bv[j];
                if (relu) v = fmaxf(v, 0.f);
                o[j] = (bf16)v;
                if (STATS) { float q = ok ? (float)o[j] : 0.f; s1[j] += q; s2[j] = fmaf(q, q, s2[j]); }
            }
            if (ok) *reinterpret_cast<bf16x4*>(yrow + (int64_t)r * W * ycs) = o;
        }
    }
    if constexpr (STATS) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float a = s1[j], b = s2[j];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
            if (vn == 0) { red[wave][g * 4 + j][0] = a; red[wave][g * 4 + j][1] = b; }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 32; idx += BLK) {
            int k = idx & 1, chn = idx >> 1;
            float v = (red[0][chn][k] + red[1][chn][k]) + (red[2][chn][k] + red[3][chn][k]);
            part[((int64_t)blockIdx.x * 2 + k) * 16 + chn] = v;
        }
    }
}

#endif  // MI3D_EXPERIMENTS

// eight-wave kernels (route conv8, default on): round 3 A/B level-1 forward convs 112 -> 103 us, deep 153 -> 143 us per step
// CU budget (mi3d_set_cu_budget): CUs the caller wants left free of persistent workgroups because a collective kernel is
// resident on them (data-parallel step: the gradient all-reduce runs beside the encoder backward).  A persistent grid sized
// for all 256 CUs would otherwise run a second, partial round on the CUs it has to share.
static thread_local int g_cu_budget = 0;
extern "C" int mi3d_set_cu_budget(int cus) { g_cu_budget = cus < 0 ? 0 : (cus > 128 ? 128 : cus); return 0; }
// compute units of the device (256 on MI355X), queried once per process; the plan sizes its partial-row buffers with the
// same number, so it must not change between mi3d_unet_workspace_bytes and the launches
inline int device_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8)
            v = 256;
        (void)hipGetLastError();
        return v;
    }();
    return n;
}
inline int persist_cus() { return device_cus() - g_cu_budget; }
// ---- measurement hook (bench.py `roofline`): HIP events tightly around ONE kernel launch, keyed on the LAYER -----------------
// kind 0 = fused full-resolution backward (input + weight gradient in one launch), 1 = stand-alone weight gradient,
// 2 = persistent conv with BatchNorm partial sums (training forward), 3 = persistent conv without (input gradient);
// cin / cout as the launcher sees them.  One-shot: the first matching launch of the calling thread takes the events.
struct TimeHook { hipEvent_t e0 = nullptr, e1 = nullptr; int kind = -1, cin = 0, cout = 0; bool armed = false, fired = false; };
static thread_local TimeHook g_hook;
inline bool time_hook_take(int kind, int cin, int cout, hipEvent_t& e0, hipEvent_t& e1) {
    if (!g_hook.armed || g_hook.kind != kind || g_hook.cin != cin || g_hook.cout != cout) return false;
    e0 = g_hook.e0; e1 = g_hook.e1;
    g_hook.armed = false; g_hook.fired = true;
    return true;
}

inline bool persist_ok(int Cin, int Cout, Geo g) {
    // 16 -> 32 (two co blocks, 256 VGPRs): only worth it with >= 2 tiles per workgroup; the one-tile-per-workgroup
    // generic kernel is faster below that (level 1 of the 96^3 net)
    int64_t nt = (int64_t)g.N * cdiv(g.D, 4) * cdiv(g.H, 8) * cdiv(g.W, 16);
    if (Cin == 16 && Cout == 32 && nt < 1024) return false;
    return g.W >= 32 && g.H >= 16 && ((Cin == 16 && Cout == 16) || (Cin == 32 && Cout == 16) || (Cin == 16 && Cout == 32)) &&
           !mi3d_routes().no_persist;
}
inline int persist_grid(int Cin, int Cout, Geo g) {
    int64_t nt = (int64_t)g.N * cdiv(g.D, 4) * cdiv(g.H, 8) * cdiv(g.W, 16);
    int want = 2 * persist_cus();      // 2 resident workgroups per CU
    return (int)(nt < want ? nt : want);
}

// y[v][c] = bf16(bias[c] + sum_k part[k][v][c]); 8 channels per thread
__device__ __forceinline__ void splitk_finish_body(int blk, int nblk, const float* __restrict__ part, int ksplit, int64_t M, int C,
                                                   const float* __restrict__ bias, bf16* __restrict__ y, int ycs, int relu = 0) {
    int G8 = C / 8;
    int64_t total = M * G8;
    for (int64_t idx = (int64_t)blk * BLK + threadIdx.x; idx < total; idx += (int64_t)nblk * BLK) {
        int64_t v = idx / G8;
        int c0 = (int)(idx - v * G8) * 8;
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = bias ? bias[c0 + j] : 0.f;
        for (int k = 0; k < ksplit; k++) {
            const float* p = part + ((int64_t)k * M + v) * C + c0;
            f32x4 u = *reinterpret_cast<const f32x4*>(p), w = *reinterpret_cast<const f32x4*>(p + 4);
            a[0] += u[0]; a[1] += u[1]; a[2] += u[2]; a[3] += u[3]; a[4] += w[0]; a[5] += w[1]; a[6] += w[2]; a[7] += w[3];
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = fmaxf(a[j], 0.f);
        }
        st8<bf16>(y + v * ycs + c0, a);
    }
}

__global__ __launch_bounds__(BLK) void splitk_finish_kernel(const float* __restrict__ part, int ksplit, int64_t M, int C,
                                                            const float* __restrict__ bias, bf16* __restrict__ y, int ycs, int relu) {
    splitk_finish_body((int)blockIdx.x, (int)gridDim.x, part, ksplit, M, C, bias, y, ycs, relu);
}

template <int TZ, int TYB, int TXB, int BX, int COB>
int launch_cfg(const bf16* x, int xcs, int Cin, const bf16* wp, const float* bias, bf16* y, int ycs, int Cout, Geo g,
               float* part, int ksplit, float* skws, hipStream_t s, bool defer_finish = false, int relu = 0,
               const XfArgs* xf = nullptr, float* tk_rows = nullptr, int* tk_count = nullptr) {
    constexpr int TY = TYB * (16 / BX), TX = TXB * BX;
    int tz = cdiv(g.D, TZ), ty = cdiv(g.H, TY), tx = cdiv(g.W, TX);
    dim3 grid((unsigned)(g.N * tz * ty * tx), (unsigned)(Cout / (16 * COB)), (unsigned)ksplit);
    if (tk_rows) {       // split-K launch that finishes itself (ticket per (tile, output group)); eight-wave kernel only
        MI3D_CHECK_ARG(ksplit > 1 && tk_count && mi3d_routes().conv8 != 0 && (!xf || xf->mode == 0) &&
                       (int64_t)grid.x * grid.y <= CONV3_TK_COUNTERS && ycs % 8 == 0 && ((uintptr_t)y % 16) == 0,
                       "conv3_mfma: this launch cannot take the split-K ticket");
        XfArgs xa;
        xa.tk_rows = tk_rows; xa.tk_count = tk_count;
        constexpr size_t lds_tk = conv8_lds(TY, TX, COB);
        MI3D_SET_MAX_LDS_ONCE((&conv3_mfma8_kernel<TZ, TYB, TXB, BX, COB, false, true, 0, true>), lds_tk + CONV8_XF_LDS);
        conv3_mfma8_kernel<TZ, TYB, TXB, BX, COB, false, true, 0, true><<<grid, 512, lds_tk, s>>>(x, xcs, Cin, wp, bias, y, ycs, Cout, g.D, g.H, g.W,
                                                                                                tz, ty, tx, skws, 0, xa);
        MI3D_LAUNCH_CHECK();
        return 0;
    }
    // eight-wave variant (see conv3_mfma8_kernel); MI3D_CONV8=0 selects the four-wave kernels
    static_assert((TYB * TXB) % 2 == 0, "tile shapes used here have an even number of M-blocks per slice");
    const bool w8 = mi3d_routes().conv8 != 0;
    constexpr size_t lds8 = conv8_lds(TY, TX, COB);
#define LC8X(ST_, SK_, XF_, PART_, BIAS_, RELU_)                                                                               \
    do {                                                                                                                       \
        MI3D_SET_MAX_LDS_ONCE((&conv3_mfma8_kernel<TZ, TYB, TXB, BX, COB, ST_, SK_, XF_>), lds8 + CONV8_XF_LDS);               \
        conv3_mfma8_kernel<TZ, TYB, TXB, BX, COB, ST_, SK_, XF_><<<grid, 512, lds8 + (XF_ ? CONV8_XF_LDS : 0), s>>>(x, xcs, Cin, wp, BIAS_, y, ycs, Cout, g.D, g.H, g.W, tz, ty, tx, PART_, RELU_, xf ? *xf : XfArgs()); \
    } while (0)
    // apply on load: only the small-geometry tiling with two output blocks carries the transform (TX == 8, COB == 2)
    constexpr bool XF_CFG = TX == 8 && COB == 2;
    const int xfm = xf ? xf->mode : 0;
    MI3D_CHECK_ARG(xfm == 0 || (XF_CFG && w8 && xf->C == Cin && Cin <= 256 && xf->rows && xf->nrows > 0 && xf->stat &&
                                (xfm == 1 || (xfm == 2 && xf->y2))),
                   "conv3_mfma: this launch cannot apply BatchNorm on load");
#ifdef MI3D_EXPERIMENTS      // the apply-on-load kernels are a measured-slower experiment: experiment builds only
#define LC8(ST_, SK_, PART_, BIAS_, RELU_)                                                                                     \
    do {                                                                                                                       \
        if constexpr (XF_CFG) {                                                                                                \
            if (xfm == 1) { LC8X(ST_, SK_, 1, PART_, BIAS_, RELU_); break; }                                                    \
            if (xfm == 2) { LC8X(ST_, SK_, 2, PART_, BIAS_, RELU_); break; }                                                    \
        }                                                                                                                      \
        LC8X(ST_, SK_, 0, PART_, BIAS_, RELU_);                                                                                \
    } while (0)
#else
#define LC8(ST_, SK_, PART_, BIAS_, RELU_) LC8X(ST_, SK_, 0, PART_, BIAS_, RELU_)
#endif
    if (ksplit > 1) {
        if (w8) LC8(false, true, skws, nullptr, 0);
        else conv3_mfma_kernel<TZ, TYB, TXB, BX, COB, false, true><<<grid, BLK, 0, s>>>(x, xcs, Cin, wp, nullptr, y, ycs, Cout, g.D, g.H, g.W, tz, ty, tx, skws, 0);
        MI3D_LAUNCH_CHECK();
        if (defer_finish) return 0;
        int64_t tot = g.M() * (Cout / 8);
        splitk_finish_kernel<<<cdiv(tot, BLK) > 2048 ? 2048 : cdiv(tot, BLK), BLK, 0, s>>>(skws, ksplit, g.M(), Cout, bias, y, ycs, relu & 1);
    } else if (part) {
        if (w8) LC8(true, false, part, bias, relu);
        else conv3_mfma_kernel<TZ, TYB, TXB, BX, COB, true, false><<<grid, BLK, 0, s>>>(x, xcs, Cin, wp, bias, y, ycs, Cout, g.D, g.H, g.W, tz, ty, tx, part, relu & 1);
    } else {
        if (w8) LC8(false, false, nullptr, bias, relu);
        else conv3_mfma_kernel<TZ, TYB, TXB, BX, COB, false, false><<<grid, BLK, 0, s>>>(x, xcs, Cin, wp, bias, y, ycs, Cout, g.D, g.H, g.W, tz, ty, tx, nullptr, relu & 1);
    }
#undef LC8
#undef LC8X
    MI3D_LAUNCH_CHECK();
    return 0;
}

inline bool big_geo(Geo g) { return g.W >= 32 && g.H >= 16; }

// K-split factor: only for the small-geometry configuration, when the (tile x cout-group) grid is below ~one
// workgroup per CU; power of two dividing the chunk count
inline int pick_ksplit(int Cin, int Cout, Geo g, int target_override = 0) {
    if (big_geo(g)) return 1;
    int64_t wgs = (int64_t)g.N * cdiv(g.D, 4) * cdiv(g.H, 8) * cdiv(g.W, 8) * (Cout / (Cout % 32 == 0 ? 32 : 16));
    int nchunk = Cin / 16, k = 1;
    // split-K workgroup target.  Round 2 (four-wave kernels): 128 / 256 / 512 / 1024 -> 2.375 / 2.350 / 2.395 / 2.421 ms.  Round 3
    // (eight-wave kernels: a workgroup's chain is half as long, fewer and fatter workgroups win; more layers keep their BatchNorm
    // partial sums in the conv epilogue): 64 / 96 / 128 / 192 / 256 / 384 / 512 -> 2.252 / 2.229 / 2.219 / 2.221 / 2.263 / 2.265 / 2.286 ms
    const int target = mi3d_routes().ks_target;
    const int tgt = target_override > 0 ? (target_override < target ? target_override : target) : target;
    while (wgs * k < tgt && k * 2 <= nchunk && nchunk % (k * 2) == 0 && k < 16) k *= 2;
    return k;
}

}  // namespace

bool conv3_mfma_supported(int Cin, int Cout, int xcs, int ycs) {
    return Cin % 16 == 0 && Cout % 16 == 0 && xcs % 8 == 0 && ycs % 4 == 0 && Cin >= 16 && Cout >= 16;
}

size_t conv3_mfma_pack_elems(int Cin, int Cout) { return (size_t)Cin * Cout * 28; }   // one operand (fwd or dgrad)

// g = the geometry the packs will be used at: it decides (with the channel counts) whether the forward / dgrad launch
// runs the persistent row-reuse kernel, whose K-step -> tap order differs (ktap mode 1)
int conv3_mfma_pack(const float* w, int Cin, int Cout, void* wp_fwd, void* wp_dgrad, Geo g, hipStream_t s) {
    int64_t n = 2 * (int64_t)conv3_mfma_pack_elems(Cin, Cout);
    int mode_f = persist_ok(Cin, Cout, g) ? 1 : 0, mode_d = persist_ok(Cout, Cin, g) ? 1 : 0;
    pack_mfma_kernel<<<cdiv(n, 256) > 2048 ? 2048 : cdiv(n, 256), 256, 0, s>>>(w, Cin, Cout, (bf16*)wp_fwd, (bf16*)wp_dgrad,
                                                                             mode_f, mode_d);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int pack_all_add_conv3(PackJobs& J, const float* w, int Cin, int Cout, void* wp_fwd, void* wp_dgrad, Geo g, const float* scale) {
    MI3D_CHECK_ARG(J.n < MAX_PACK_JOBS && Cin % 16 == 0 && Cout % 16 == 0, "pack_all: too many jobs / bad channels");
    MI3D_CHECK_ARG(((uintptr_t)w % 16) == 0, "pack_all: conv weight tensor must be 16-byte aligned");
    PackJob& j = J.j[J.n++];
    j = PackJob{w, wp_fwd, wp_dgrad, Cin, Cout, 0, persist_ok(Cin, Cout, g) ? 1 : 0, persist_ok(Cout, Cin, g) ? 1 : 0, J.nblocks, scale};
    J.nblocks += (Cin / 16) * (Cout / 16);
    return 0;
}
int pack_all_add_upconv(PackJobs& J, const float* w, int Cin, int Cout, void* wp) {
    MI3D_CHECK_ARG(J.n < MAX_PACK_JOBS, "pack_all: too many jobs");
    int64_t n = (int64_t)Cin * Cout * 8;
    PackJob& j = J.j[J.n++];
    j = PackJob{w, wp, (bf16*)wp + n, Cin, Cout, 1, 0, 0, J.nblocks};
    J.nblocks += (int)cdiv(2 * n, (int64_t)BLK * 8);
    return 0;
}
int pack_all_launch(const PackJobs& J, hipStream_t s) {
    if (J.n == 0) return 0;
    pack_all_kernel<<<J.nblocks, BLK, 0, s>>>(J);
    MI3D_LAUNCH_CHECK();
    return 0;
}

// number of per-workgroup statistic partials the forward launch writes
int conv3_mfma_stat_blocks(int Cin, int Cout, Geo g) {
    if (persist_ok(Cin, Cout, g)) return persist_grid(Cin, Cout, g);
    if (big_geo(g)) return g.N * cdiv(g.D, 4) * cdiv(g.H, 8) * cdiv(g.W, 16);
    return g.N * cdiv(g.D, 4) * cdiv(g.H, 8) * cdiv(g.W, 8);
}

// fp32 scratch the K-split path needs for this layer (0 = single-pass kernel)
size_t conv3_mfma_splitk_floats(int Cin, int Cout, Geo g) {
    int k = pick_ksplit(Cin, Cout, g);
    return k > 1 ? (size_t)k * g.M() * Cout : 0;
}
// true when a forward launch with `part` fills the BN partials itself (single-pass kernel)
bool conv3_mfma_fuses_stats(int Cin, int Cout, Geo g) { return pick_ksplit(Cin, Cout, g) == 1; }

// y = conv(x, wp) (+ bias); part != NULL -> also write BN partial sums [nblk][2][Cout] of the rounded outputs
// (only when conv3_mfma_fuses_stats); skws = K-split scratch (conv3_mfma_splitk_floats) or NULL to force single pass
bool conv3_mfma_halves_ok(int Cin, int Cout, Geo g) { return persist_ok(Cin, Cout, g); }

int conv3_bwd_ks_target() {
    int kst = mi3d_routes().ks_target_bwd;    // pick_ksplit clamps it to the forward target (the planned split-K scratch)
    return kst < 1 ? 1 : kst;
}

bool conv3_mfma_xform_ok(int Cin, int Cout, Geo g) {
#ifndef MI3D_EXPERIMENTS
    return false;
#endif
    return !big_geo(g) && !persist_ok(Cin, Cout, g) && Cout % 32 == 0 && Cin % 16 == 0 && Cin <= 256 && mi3d_routes().conv8 != 0;
}

bool conv3_mfma_ticket_ok(int Cin, int Cout, Geo g) {
    if (!mi3d_routes().splitk_ticket || mi3d_routes().conv8 == 0 || big_geo(g) || persist_ok(Cin, Cout, g)) return false;
    if (pick_ksplit(Cin, Cout, g) <= 1) return false;
    const int64_t wgs = (int64_t)g.N * cdiv(g.D, 4) * cdiv(g.H, 8) * cdiv(g.W, 8) * (Cout / (Cout % 32 == 0 ? 32 : 16));
    return wgs <= CONV3_TK_COUNTERS;
}

int conv3_mfma_fwd(const void* x, int xcs, int Cin, const void* wp, const float* bias, void* y, int ycs, int Cout, Geo g,
                   float* part, float* skws, hipStream_t s, Halves xh, Halves yh, int* ks_deferred, int relu, int ks_target,
                   const XfArgs* xf, float* tk_rows, int* tk_count) {
    if (ks_deferred) *ks_deferred = 0;
    MI3D_CHECK_ARG(!tk_rows || (conv3_mfma_ticket_ok(Cin, Cout, g) && skws && ks_target == 0), "conv3_mfma_fwd: no split-K ticket for %d->%d here", Cin, Cout);
    MI3D_CHECK_ARG(!xf || xf->mode == 0 || conv3_mfma_xform_ok(Cin, Cout, g), "conv3_mfma_fwd: no apply-on-load kernel for %d->%d here", Cin, Cout);
    MI3D_CHECK_ARG((!xh.on() && !yh.on()) || persist_ok(Cin, Cout, g), "conv3_mfma_fwd: planar halves need the persistent kernel");
    MI3D_CHECK_ARG(conv3_mfma_supported(Cin, Cout, xcs, ycs), "conv3_mfma_fwd: unsupported channels %d->%d", Cin, Cout);
    MI3D_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0, "conv3_mfma_fwd: misaligned tensors");
    const bf16* xp = (const bf16*)x; const bf16* w = (const bf16*)wp; bf16* yp = (bf16*)y;
    bool two = Cout % 32 == 0;
    if (persist_ok(Cin, Cout, g)) {
        int tz = cdiv(g.D, 4), ty = cdiv(g.H, 8), tx = cdiv(g.W, 16), nt = g.N * tz * ty * tx, grid = persist_grid(Cin, Cout, g);
        // bit 1 of the relu word: 16-byte epilogue stores (two M-block rows trade halves through v_permlane16_swap)
        if (!(mi3d_routes().no_wide_store & 1) && ycs % 8 == 0 && ((uintptr_t)y % 16) == 0 && yh.delta % 8 == 0) relu |= 2;
#define PK(COB_, NCH_)                                                                                                         \
        do {                                                                                                                   \
            hipEvent_t tev0 = nullptr, tev1 = nullptr;                                                                          \
            if (time_hook_take(part ? 2 : 3, Cin, Cout, tev0, tev1)) {                                                          \
                if (part) hipExtLaunchKernelGGL((conv3_mfma_persist_kernel<COB_, NCH_, true>), dim3(grid), dim3(BLK), 0, s, tev0, tev1, 0, xp, xcs, w, bias, yp, ycs, g.D, g.H, g.W, tz, ty, tx, nt, part, xh, yh, relu); \
                else hipExtLaunchKernelGGL((conv3_mfma_persist_kernel<COB_, NCH_, false>), dim3(grid), dim3(BLK), 0, s, tev0, tev1, 0, xp, xcs, w, bias, yp, ycs, g.D, g.H, g.W, tz, ty, tx, nt, (float*)nullptr, xh, yh, relu); \
            }                                                                                                                   \
            else if (part) conv3_mfma_persist_kernel<COB_, NCH_, true><<<grid, BLK, 0, s>>>(xp, xcs, w, bias, yp, ycs, g.D, g.H, g.W, tz, ty, tx, nt, part, xh, yh, relu); \
            else conv3_mfma_persist_kernel<COB_, NCH_, false><<<grid, BLK, 0, s>>>(xp, xcs, w, bias, yp, ycs, g.D, g.H, g.W, tz, ty, tx, nt, nullptr, xh, yh, relu); \
        } while (0)
        // Cout = 16: asynchronous-staging variant (the tensor must be addressable with 32-bit byte offsets)
        // OFF by default: the kernels are 6-9 % faster, but with them in the step the part runs at 2350 instead of
        // 2392 MHz (rocm-smi during bench.py, at LOWER package power) and the step gets 15-40 us slower
#ifdef MI3D_EXPERIMENTS
        const bool dma = Cout == 16 && (Cin == 16 || Cin == 32) && mi3d_routes().conv_dma &&
                         (size_t)g.M() * (size_t)(xh.on() ? 2 * xcs : xcs) * 2 < (size_t)DMA_OOB;
        if (dma) {
            size_t lds = 2 * (size_t)DMA_TILE_BYTES + 4 * 16 * 2 * sizeof(float);
#define PD(NCH_)                                                                                                              \
            do {                                                                                                              \
                if (part) { MI3D_SET_MAX_LDS_ONCE((&conv3_mfma_persist_dma_kernel<NCH_, true>), lds);                         \
                    conv3_mfma_persist_dma_kernel<NCH_, true><<<grid, BLK, lds, s>>>(xp, xcs, w, bias, yp, ycs, g.D, g.H, g.W, tz, ty, tx, nt, part, xh, yh, relu & 1); } \
                else { MI3D_SET_MAX_LDS_ONCE((&conv3_mfma_persist_dma_kernel<NCH_, false>), lds);                             \
                    conv3_mfma_persist_dma_kernel<NCH_, false><<<grid, BLK, lds, s>>>(xp, xcs, w, bias, yp, ycs, g.D, g.H, g.W, tz, ty, tx, nt, nullptr, xh, yh, relu & 1); } \
            } while (0)
            if (Cin == 16) PD(1); else PD(2);
#undef PD
            MI3D_LAUNCH_CHECK();
            return 0;
        }
#endif  // MI3D_EXPERIMENTS
        if (Cin == 16 && Cout == 16) PK(1, 1);
        else if (Cin == 32) PK(1, 2);
        else PK(2, 1);
#undef PK
        MI3D_LAUNCH_CHECK();
        return 0;
    }
    int ks = skws ? pick_ksplit(Cin, Cout, g, ks_target) : 1;
    // bit 1 of the relu word: 16-byte epilogue stores in the eight-wave kernels (two output blocks per workgroup)
    if (!(mi3d_routes().no_wide_store & 2) && ycs % 8 == 0 && ((uintptr_t)y % 16) == 0 && Cout % 32 == 0) relu |= 2;
    if (ks > 1) MI3D_CHECK_ARG(ycs % 8 == 0 && ((uintptr_t)y % 16) == 0, "conv3_mfma_fwd: split-K needs 16-B aligned output rows");
    if (big_geo(g)) {
        // (round 4: ONE 16-channel output block per workgroup at the 16-wide levels -- twice the workgroups, half the chain each, the
        // input tile staged twice -- measured slower: forward 98 -> 114 us/step, input gradients 80 -> 93; profiles/r04_experiments_misc.txt)
        if (two) return launch_cfg<4, 8, 1, 16, 2>(xp, xcs, Cin, w, bias, yp, ycs, Cout, g, part, 1, nullptr, s, false, relu);
        return launch_cfg<4, 8, 1, 16, 1>(xp, xcs, Cin, w, bias, yp, ycs, Cout, g, part, 1, nullptr, s, false, relu);
    }
    if (tk_rows) {      // the launch finishes its split-K sums and the BatchNorm partial rows itself (nothing deferred)
        if (two) return launch_cfg<4, 2, 2, 4, 2>(xp, xcs, Cin, w, bias, yp, ycs, Cout, g, nullptr, ks, skws, s, false, 0, nullptr, tk_rows, tk_count);
        return launch_cfg<4, 2, 2, 4, 1>(xp, xcs, Cin, w, bias, yp, ycs, Cout, g, nullptr, ks, skws, s, false, 0, nullptr, tk_rows, tk_count);
    }
    bool defer = ks > 1 && ks_deferred;
    if (defer) *ks_deferred = ks;
    if (two) return launch_cfg<4, 2, 2, 4, 2>(xp, xcs, Cin, w, bias, yp, ycs, Cout, g, part, ks, skws, s, defer, relu, xf);
    return launch_cfg<4, 2, 2, 4, 1>(xp, xcs, Cin, w, bias, yp, ycs, Cout, g, part, ks, skws, s, defer, relu);
}

// =================================================================================================== wgrad
// dW[co][ci][tap] = sum_v dy[v][co] * x[v+tap][ci]   as   D[co][ci] += A[co][k] * B[k][ci],  k = voxel.
// Both operands are channels-last in LDS ([voxel][16 ch], 32 B per voxel), i.e. K is the STRIDED index, so the
// fragments are fetched with the CDNA4 transposing LDS read ds_read_b64_tr_b16 (4 voxels x 16 channels per 16-lane
// group -> each lane gets 4 consecutive voxels of ITS channel); two reads make one 8-deep MFMA fragment.
// One K-step = 32 voxels = 2 tile rows x 16 x-positions; wave w owns row pair w of every z-slice of the 4x8x16 tile.
// A workgroup owns CO_B x CI_B 16-channel blocks x NT taps and sweeps tiles persistently with its NT*CO_B*CI_B
// accumulator tiles in registers; waves are reduced through LDS once at the end and the result goes to one slab
// per spatial workgroup (fixed-order slab sum afterwards: deterministic, no float atomics).  The (NT, #slabs) choice
// keeps slab traffic below activation traffic: many slabs for the tiny-weight full-resolution layers, few slabs and
// more tap/channel groups for the weight-heavy deep layers.  Zero-filled staging makes ragged volumes exact.
namespace {

constexpr int WTZ = 4, WTY = 8, WTX = 16, WIZ = 6, WIY = 10, WIX = 18;
constexpr int WNV = WTZ * WTY * WTX, WNH = WIZ * WIY * WIX;

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x8 tr_frag(const char* base, int byteoff) {
    // two transposed reads: voxels +0..3 and +4..7 (128 B further) of this lane group's 8-voxel run
    auto* p0 = (lds_bf16x4*)(base + byteoff);
    auto* p1 = (lds_bf16x4*)(base + byteoff + 128);
    bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p0);
    bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p1);
    return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// reduce NTILE per-wave accumulator tiles across the 4 waves through LDS; `emit(idx, f32x4 sum)` is called by
// wave (idx % 4) for tile idx
template <int NTILE, typename Emit>
__device__ __forceinline__ void reduce_waves(f32x4 (&acc)[NTILE], float* red, int wave, int lane, Emit emit) {
#pragma unroll
    for (int base = 0; base < NTILE; base += 4) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (base + j < NTILE) *reinterpret_cast<f32x4*>(red + ((j * 4 + wave) * 64 + lane) * 4) = acc[base + j];
        __syncthreads();
        int idx = base + wave;
        if (idx < NTILE) {
            f32x4 s = *reinterpret_cast<f32x4*>(red + ((wave * 4 + 0) * 64 + lane) * 4);
#pragma unroll
            for (int w = 1; w < 4; w++) {
                f32x4 t = *reinterpret_cast<f32x4*>(red + ((wave * 4 + w) * 64 + lane) * 4);
                s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
            }
            emit(idx, s);
        }
    }
}

template <int CO_B, int CI_B, int NT>
__device__ __forceinline__ void conv3_wgrad_body(Bid bid_, const bf16* __restrict__ x, int xcs, int Cin,
                                                 const bf16* __restrict__ dy, int dycs, int Cout, int N, int D,
                                                 int H, int W, int tilesZ, int tilesY, int tilesX, int TG,
                                                 float* __restrict__ slabs, Halves xh, int xcd_tiles = 0) {
    // xcd_tiles (full-resolution layers, round 4): slab workgroups b, b + 8, ... run on one XCD (round-robin dispatch) and take
    // ADJACENT tiles (xcd_contig), so that the x-halo voxels neighbouring tiles share are served by that XCD's L2.  Before, tile =
    // slab index put neighbouring tiles on neighbouring XCDs and every halo was fetched twice (decoder.3.conv0: 261 MB for 170 MB)
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    bf16* dys = reinterpret_cast<bf16*>(lds_raw);                 // [CO_B][WNV][16]
    bf16* xs = dys + CO_B * WNV * 16;                             // [CI_B][WNH][16]
    const char* dysb = reinterpret_cast<const char*>(dys);
    const char* xsb = reinterpret_cast<const char*>(xs);
    int sb = bid_.x / TG, tg = bid_.x - sb * TG, nsb = bid_.gx / TG;
    int co0 = bid_.y * CO_B * 16, ci0 = bid_.z * CI_B * 16;
    int lane = threadIdx.x & 63;
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    int laneA = (((2 * wave + (G >> 1)) * WTX + 8 * (G & 1) + q) * 16 + 4 * p) * 2;
    int laneB = (((2 * wave + (G >> 1)) * WIX + 8 * (G & 1) + q) * 16 + 4 * p) * 2;
    // taps of this workgroup: NT = 27 -> all; NT = 9 -> the nine (dz, dx) taps of ONE dy (= tap group), so that every
    // workgroup still sees all three dz and can reuse an x fragment of halo slice zz for the output slices zz-dz
    constexpr int NY = NT / 9;                  // dy values handled here (3 or 1)
    static_assert(NT == 27 || NT == 9, "tap grouping");
    constexpr int NTILE = NT * CO_B * CI_B;
    f32x4 acc[NTILE];
#pragma unroll
    for (int i = 0; i < NTILE; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbs[CO_B];
#pragma unroll
    for (int a = 0; a < CO_B; a++) dbs[a] = 0.f;
    bool do_db = (tg == 0 && bid_.z == 0);

    int ntiles = N * tilesZ * tilesY * tilesX;
    // Staging map, tile-invariant and computed ONCE (round 3: the per-tile div/mod + bounds chains of 13 loads were ~800 vector
    // instructions per tile against 108 MFMAs -- the kernel was bound by its address arithmetic):
    //   dy   slot it = z-slice it of the 4x8x16 tile; thread t -> voxel (iy, ix) = ((t >> 1) / 16, (t >> 1) % 16), channel half t & 1
    //   x    the 6 halo slices as 3 slice pairs; pair p = slices 2p, 2p + 1 = 360 voxels = 3 slots of 120 voxels (threads 0..239)
    // Per tile only the origin moves: validity in y / x is one compare pair per slot KIND, in z one scalar-ish compare per slot, and
    // every address is a uniform base + a fixed 32-bit lane offset.  The LDS image ([voxel][16 channels]) is unchanged.
    static_assert(CO_B == 1 && CI_B == 1 && BLK == 256 && WTY * WTX * 2 == BLK && WIY * WIX * 2 == 360, "staging map");
    constexpr int NA = WTZ, NXJ = 3, NXP = WIZ / 2, NB = NXP * NXJ;
    const int hv = threadIdx.x >> 1, hf = threadIdx.x & 1;
    const int a_iy = hv / WTX, a_ix = hv % WTX;
    const unsigned a_rel = (unsigned)(((a_iy * W + a_ix) * dycs + hf * 8) * 2);         // bytes from the tile origin
    const bool b_act = threadIdx.x < 240;
    int b_iy[NXJ], b_ix[NXJ], b_iz[NXJ];
    unsigned b_rel[NXJ];
#pragma unroll
    for (int j = 0; j < NXJ; j++) {
        int v = 120 * j + (b_act ? hv : 0);
        b_iz[j] = v / (WIY * WIX);
        int r = v % (WIY * WIX);
        b_iy[j] = r / WIX;
        b_ix[j] = r % WIX;
        b_rel[j] = (unsigned)((((b_iz[j] * H + b_iy[j]) * W + b_ix[j]) * xcs + hf * 8) * 2);   // bytes from the halo origin of the pair
    }
    bf16x8 va[NA], vb[NB];
    auto load_tile = [&](int tile) {
        int t = tile;
        int tx_ = t % tilesX; t /= tilesX;
        int ty_ = t % tilesY; t /= tilesY;
        int tz_ = t % tilesZ; int n = t / tilesZ;
        int z0 = tz_ * WTZ, y0 = ty_ * WTY, x0 = tx_ * WTX;
        const char* dyn = reinterpret_cast<const char*>(dy + ((((int64_t)n * D + z0) * H + y0) * W + x0) * dycs + co0);
        // halo origin (z0 - 1, y0 - 1, x0 - 1): may lie in front of the sample; only in-range lanes dereference
        const char* xn = reinterpret_cast<const char*>(x + ci0 + ((ci0 >> 4) >= xh.split ? xh.delta : 0)) +
                         ((((int64_t)n * D + (z0 - 1)) * H + (y0 - 1)) * W + (x0 - 1)) * xcs * 2;
        const int64_t a_zs = (int64_t)H * W * dycs * 2, b_zs = (int64_t)H * W * xcs * 2;
        bool a_ok = y0 + a_iy < H && x0 + a_ix < W;
#pragma unroll
        for (int it = 0; it < NA; it++) {
            va[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (a_ok && z0 + it < D) va[it] = *reinterpret_cast<const bf16x8*>(dyn + it * a_zs + a_rel);
        }
        bool b_ok[NXJ];
#pragma unroll
        for (int j = 0; j < NXJ; j++)
            b_ok[j] = b_act && (unsigned)(y0 - 1 + b_iy[j]) < (unsigned)H && (unsigned)(x0 - 1 + b_ix[j]) < (unsigned)W;
#pragma unroll
        for (int pz = 0; pz < NXP; pz++)
#pragma unroll
            for (int j = 0; j < NXJ; j++) {
                vb[pz * NXJ + j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (b_ok[j] && (unsigned)(z0 - 1 + 2 * pz + b_iz[j]) < (unsigned)D)
                    vb[pz * NXJ + j] = *reinterpret_cast<const bf16x8*>(xn + 2 * pz * b_zs + b_rel[j]);
            }
    };
    // prefetch the next tile into registers while computing (only where the register file has room for it)
    constexpr bool PF = (NT * CO_B * CI_B + NA + NB) * 4 <= 300;
    const int tile0 = xcd_tiles ? xcd_contig(sb, nsb) : sb;
    if (PF && tile0 < ntiles) load_tile(tile0);
    for (int tile = tile0; tile < ntiles; tile += nsb) {
        if (!PF) load_tile(tile);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NA; it++) *reinterpret_cast<bf16x8*>(dys + (threadIdx.x + it * BLK) * 8) = va[it];
        if (b_act) {
#pragma unroll
            for (int it = 0; it < NB; it++)
                *reinterpret_cast<bf16x8*>(xs + ((it / NXJ) * 720 + (it % NXJ) * 240 + threadIdx.x) * 8) = vb[it];
        }
        __syncthreads();
        if (PF && tile + nsb < ntiles) load_tile(tile + nsb);     // next tile's loads fly under this tile's MFMAs
        // dy fragments of all four output slices of this wave's row pair
        bf16x8 A[WTZ][CO_B];
#pragma unroll
        for (int z = 0; z < WTZ; z++)
#pragma unroll
            for (int a = 0; a < CO_B; a++) {
                A[z][a] = tr_frag(dysb, laneA + z * (WTY * WTX * 32) + a * (WNV * 32));
                if (do_db) {
#pragma unroll
                    for (int j = 0; j < 8; j++) dbs[a] += (float)A[z][a][j];
                }
            }
        // x fragments are walked by HALO slice zz = z + dz: one fragment (zz, dy, dx, ci-block) feeds the up to three
        // (z, dz) pairs with z + dz = zz  ->  6*9 instead of 4*27 transposed LDS reads per ci-block (2x fewer).
        // They stream through a PFD-deep register ring (2*PFD tr-reads in flight ahead of the MFMAs).
        constexpr int NBF = WIZ * NY * 3 * CI_B, PFD = 4;
        auto bfrag = [&](int t) {
            int b_ = t % CI_B, r = t / CI_B;
            int dx = r % 3; r /= 3;
            int dyi = r % NY, zz = r / NY;
            int dy = (NT == 27) ? dyi : tg;
            return tr_frag(xsb, laneB + ((zz * WIY + dy) * WIX + dx) * 32 + b_ * (WNH * 32));
        };
        bf16x8 Bq[PFD];
#pragma unroll
        for (int t = 0; t < PFD; t++) Bq[t] = bfrag(t);
#pragma unroll
        for (int t = 0; t < NBF; t++) {
            bf16x8 Bc = Bq[t % PFD];
            if (t + PFD < NBF) Bq[t % PFD] = bfrag(t + PFD);
            int b_ = t % CI_B, r = t / CI_B;
            int dx = r % 3; r /= 3;
            int dyi = r % NY, zz = r / NY;
#pragma unroll
            for (int dz = 0; dz < 3; dz++) {
                int z = zz - dz;
                if (z >= 0 && z < WTZ) {
                    int i = (dz * NY + dyi) * 3 + dx;          // accumulator slot of tap (dz, dy, dx)
#pragma unroll
                    for (int a = 0; a < CO_B; a++)
                        acc[(i * CO_B + a) * CI_B + b_] = mfma16(A[z][a], Bc, acc[(i * CO_B + a) * CI_B + b_]);
                }
            }
        }
    }
    int64_t nW = (int64_t)Cout * Cin * 27;
    float* slab = slabs + (int64_t)sb * (nW + Cout);
    float* red = reinterpret_cast<float*>(lds_raw);
    // slab layout = MFMA-native: tile (tap, co16-block, ci16-block) -> 64 lanes x 4 floats, so every store is a
    // coalesced 1 KB wave write; slab_reduce_mfma_kernel un-permutes into torch's (Cout,Cin,3,3,3) once at the end
    int COBN = Cout / 16, CIBN = Cin / 16;
    reduce_waves<NTILE>(acc, red, wave, lane, [&](int idx, f32x4 sum) {
        int b = idx % CI_B, a = (idx / CI_B) % CO_B, i = idx / (CI_B * CO_B);
        int dx = i % 3, dyi = (i / 3) % NY, dz = i / (3 * NY);
        int tap = dz * 9 + ((NT == 27) ? dyi : tg) * 3 + dx;
        int64_t tileIdx = ((int64_t)tap * COBN + (co0 / 16 + a)) * CIBN + (ci0 / 16 + b);
        *reinterpret_cast<f32x4*>(slab + tileIdx * 256 + lane * 4) = sum;
    });
    if (do_db) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < CO_B; a++) {
            float sv = dbs[a];
            sv += __shfl_xor(sv, 16, 64);
            sv += __shfl_xor(sv, 32, 64);
            if (lane < 16) red[(a * 4 + wave) * 16 + lane] = sv;
        }
        __syncthreads();
        if (threadIdx.x < CO_B * 16) {
            int a = threadIdx.x / 16, c = threadIdx.x % 16;
            slab[nW + co0 + threadIdx.x] = (red[(a * 4 + 0) * 16 + c] + red[(a * 4 + 1) * 16 + c]) +
                                           (red[(a * 4 + 2) * 16 + c] + red[(a * 4 + 3) * 16 + c]);
        }
    }
}

template <int CO_B, int CI_B, int NT>
__global__ __launch_bounds__(BLK, (NT * CO_B * CI_B <= 27) ? 2 : 1) void conv3_wgrad_mfma_kernel(const bf16* __restrict__ x, int xcs, int Cin,
                                                               const bf16* __restrict__ dy, int dycs, int Cout, int N, int D,
                                                               int H, int W, int tilesZ, int tilesY, int tilesX, int TG,
                                                               float* __restrict__ slabs, Halves xh, int xcd_tiles,
                                                               int pgx, int pgy, int pgz) {
    // pgx > 0: flat 1-D grid of pgx * pgy * pgz workgroups, PLACED like the weight-gradient half of the fused launch -- the
    // (co, ci)-block workgroups of one slab (they re-read the same dy / x tiles) and neighbouring slabs sit on one XCD (round 4:
    // the stand-alone kernel carries every deferred weight gradient of the aux-stream route; with the natural 3-D grid the
    // blocks of one slab were spread over all eight L2s)
    if (pgx > 0) {
        int f = xcd_contig((int)blockIdx.x, pgx * pgy * pgz), G = pgy * pgz, yz = f % G;
        conv3_wgrad_body<CO_B, CI_B, NT>(Bid{f / G, yz % pgy, yz / pgy, pgx, pgy, pgz}, x, xcs, Cin, dy, dycs, Cout, N, D, H, W, tilesZ, tilesY,
                                         tilesX, TG, slabs, xh, 0);
        return;
    }
    conv3_wgrad_body<CO_B, CI_B, NT>(real_bid(), x, xcs, Cin, dy, dycs, Cout, N, D, H, W, tilesZ, tilesY, tilesX, TG, slabs, xh, xcd_tiles);
}

// Horizontal fusion for the deep levels: workgroups [0, nwg_w) run the weight gradient of a layer, the rest its
// input-gradient conv (small-geometry tiling, COB = 2, optionally split-K).  Flat 1-D grid, virtual 3-D indices.
struct FusedArgs {
    // weight gradient: x = layer input, dy
    const bf16* wx; int wxcs, wCin; const bf16* wdy; int wdycs, wCout; int tZ, tY, tX; float* slabs; int wgx, wgy, wgz;
    // input gradient conv: x = dy, packed dgrad weights, y = dx (or split-K partials)
    const bf16* dxin; int dxcs_in, dCin; const bf16* dwp; bf16* dyout; int dycs_out, dCout; int dtZ, dtY, dtX; float* dpart;
    int dgx, dgy, dgz;
    int N, D, H, W;
    int flags;                // bit 1: 16-byte epilogue stores in the input-gradient half
};
// Full-resolution variant: the input-gradient conv is the persistent kernel body.  Both halves are persistent with ONE
// workgroup per CU each, so every CU runs one MFMA-heavy dgrad workgroup beside one staging/LDS-heavy wgrad workgroup
// for the whole launch (two workgroups of the same kind contend for the same unit in the same phase).
struct FusedPArgs {
    const bf16* wx; int wxcs, wCin; const bf16* wdy; int wdycs, wCout; int tZ, tY, tX; float* slabs; int wgx, wgy, wgz; Halves wxh;
    const bf16* dxin; int dxcs_in; const bf16* dwp; bf16* dyout; int dycs_out; int ptZ, ptY, ptX, pnt, pgrid; Halves dyh;
    int N, D, H, W;
    int xcd_tiles, flags;
};
template <int COB, int NCH>
__global__ __launch_bounds__(BLK, 2) void conv3_bwd_fused_persist_kernel(FusedPArgs a) {
    extern __shared__ __attribute__((aligned(16))) char fusedp_lds[];
    int nw = a.wgx * a.wgy * a.wgz;
    int b = blockIdx.x;
    // The two kinds alternate over the block index: even = weight gradient, odd = input gradient.  Workgroup b is dispatched to
    // XCD b % 8, so this puts ALL weight-gradient workgroups on XCDs 0,2,4,6 and all input-gradient workgroups on XCDs 1,3,5,7 --
    // two of ONE kind per CU, each kind on half of the chip.  Round 4 measured the mapping rounds 2-3 believed they had (the kinds
    // alternating INSIDE every XCD, one of each per CU, same tile sequence for both so that dy is fetched by one L2): 3 us per
    // launch SLOWER (<1,1>: 68.3 vs 65.2 us, <1,2>: 25.5 vs 23.9; profiles/r04_experiments_misc.txt).  The split stays.
    bool is_w = (b & 1) == 0;
    int idx = b >> 1;
    if (is_w) {
        if (idx >= nw) return;
        Bid v{idx % a.wgx, (idx / a.wgx) % a.wgy, idx / (a.wgx * a.wgy), a.wgx, a.wgy, a.wgz};
        conv3_wgrad_body<1, 1, 27>(v, a.wx, a.wxcs, a.wCin, a.wdy, a.wdycs, a.wCout, a.N, a.D, a.H, a.W, a.tZ, a.tY, a.tX, 1, a.slabs, a.wxh,
                                   a.xcd_tiles);
    } else {
        if (idx >= a.pgrid) return;
        Bid v{idx, 0, 0, a.pgrid, 1, 1};
        conv3_mfma_persist_body<COB, NCH, false, true>(v, a.dxin, a.dxcs_in, a.dwp, nullptr, a.dyout, a.dycs_out, a.D, a.H, a.W, a.ptZ, a.ptY,
                                                       a.ptX, a.pnt, nullptr, Halves(), a.dyh, fusedp_lds, a.flags);
    }
}

template <bool BIG, bool SPLITK>
__global__ __launch_bounds__(BLK, 2) void conv3_bwd_fused_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char fused_lds[];       // one block for either body
    int nw = a.wgx * a.wgy * a.wgz, nwp = (nw + 7) & ~7;        // input-gradient workgroups start on XCD 0 again
    int nd = a.dgx * a.dgy * a.dgz;
    int b = blockIdx.x;
    // weight-gradient workgroups FIRST: they are the long pole (measured round 3: input-gradient first +23 us/step, the two
    // kinds interleaved in runs of 8 +57 us/step)
    bool is_w = b < nwp;
    if (!is_w) b -= nwp;
    if (is_w ? b >= nw : b >= nd) return;
    if (is_w) {
        int f = xcd_contig(b, nw), G = a.wgy * a.wgz, yz = f % G;
        Bid v{f / G, yz % a.wgy, yz / a.wgy, a.wgx, a.wgy, a.wgz};
        conv3_wgrad_body<1, 1, 27>(v, a.wx, a.wxcs, a.wCin, a.wdy, a.wdycs, a.wCout, a.N, a.D, a.H, a.W, a.tZ, a.tY, a.tX, 1, a.slabs,
                                   Halves());
    } else {
        if constexpr (BIG) {
            // both output-channel groups of a tile side by side on one XCD, tiles in XCD-contiguous runs
            int f = xcd_contig(b, a.dgx * a.dgy);
            Bid v{f / a.dgy, f % a.dgy, 0, a.dgx, a.dgy, 1, true};
            conv3_mfma_body<4, 8, 1, 16, 2, false, false, true>(v, a.dxin, a.dxcs_in, a.dCin, a.dwp, nullptr, a.dyout, a.dycs_out, a.dCout,
                                                                a.D, a.H, a.W, a.dtZ, a.dtY, a.dtX, nullptr, fused_lds, a.flags);
            return;
        }
        Bid v{b % a.dgx, (b / a.dgx) % a.dgy, b / (a.dgx * a.dgy), a.dgx, a.dgy, a.dgz};
        if constexpr (BIG) {}
        else
            conv3_mfma_body<4, 2, 2, 4, 2, false, SPLITK, true>(v, a.dxin, a.dxcs_in, a.dCin, a.dwp, nullptr, a.dyout, a.dycs_out, a.dCout,
                                                                a.D, a.H, a.W, a.dtZ, a.dtY, a.dtX, a.dpart, fused_lds, a.flags);
    }
}

// ---- first layer (Cin = 1, fp32 input): dW[co][0][tap] = sum_v dy[v][co] * x[v+tap].  N-dimension = the 27 taps
// (two 16-column MFMAs); the B fragment of a lane is 8 consecutive x-positions of ONE tap, read as one aligned
// ds_read_b128 from a tile stored three times, pre-shifted by dx = 0,1,2.
constexpr int C1_ROWS = WIZ * WIY;     // 60 halo rows of 16 positions
__global__ __launch_bounds__(BLK) void conv3_wgrad_c1_kernel(const float* __restrict__ x, const bf16* __restrict__ dy, int dycs,
                                                             int Cout, int N, int D, int H, int W, int tilesZ, int tilesY,
                                                             int tilesX, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) bf16 dys[WNV * 16];
    __shared__ __attribute__((aligned(16))) bf16 xsh[3 * C1_ROWS * 16];
    __shared__ __attribute__((aligned(16))) float red[16 * 64 * 4];
    const char* dysb = reinterpret_cast<const char*>(dys);
    int co0 = blockIdx.y * 16;
    int lane = threadIdx.x & 63;
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, nn = lane & 15;
    int laneA = (((2 * wave + (G >> 1)) * WTX + 8 * (G & 1) + q) * 16 + 4 * p) * 2;
    // B: tap t = nn + 16*h ; element offset of (dx, dz, dy) inside xsh for this lane's row pair / half row
    int offB[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        int t = nn + 16 * h;
        t = t > 26 ? 26 : t;
        int dz = t / 9, dyy = (t / 3) % 3, dx = t % 3;
        offB[h] = ((dx * WIZ + dz) * WIY + (2 * wave + (G >> 1) + dyy)) * 16 + 8 * (G & 1);
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float dbs = 0.f;
    int ntiles = N * tilesZ * tilesY * tilesX;
    // tile-invariant staging maps + register prefetch of the next tile (as conv3_wgrad_body / conv3_c1_fwd_mfma_kernel):
    //   dy  slot it = z-slice it; thread t -> (iy, ix) = ((t >> 1) / 16, (t >> 1) % 16), channel half t & 1
    //   x   slot s of thread t = halo element t + 256 s -> (row, h)
    static_assert(WTY * WTX * 2 == BLK, "one z-slice of dy per staging slot");
    const int hv = threadIdx.x >> 1, hf = threadIdx.x & 1;
    const int a_iy = hv / WTX, a_ix = hv % WTX;
    const int a_rel = (a_iy * W + a_ix) * dycs + hf * 8;
    constexpr int NS = (C1_ROWS * WIX + BLK - 1) / BLK;
    int s_rel[NS], s_pk[NS];
#pragma unroll
    for (int s_ = 0; s_ < NS; s_++) {
        int idx = threadIdx.x + s_ * BLK;
        bool act = idx < C1_ROWS * WIX;
        int h = act ? idx % WIX : 0, row = act ? idx / WIX : 0;
        int iy = row % WIY, iz = row / WIY;
        s_rel[s_] = (iz * H + iy) * W + h;
        s_pk[s_] = iz | (iy << 4) | (h << 8) | (row << 16) | (act ? (1 << 24) : 0);
    }
    bf16x8 va[WTZ];
    float xv[NS];
    auto load_tile = [&](int tile) {
        int t = tile;
        int tx_ = t % tilesX; t /= tilesX;
        int ty_ = t % tilesY; t /= tilesY;
        int tz_ = t % tilesZ; int n = t / tilesZ;
        int z0 = tz_ * WTZ, y0 = ty_ * WTY, x0 = tx_ * WTX;
        const bf16* dyb = dy + ((((int64_t)n * D + z0) * H + y0) * W + x0) * dycs + co0;
        const int64_t a_zs = (int64_t)H * W * dycs;
        bool a_ok = y0 + a_iy < H && x0 + a_ix < W;
#pragma unroll
        for (int it = 0; it < WTZ; it++) {
            va[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (a_ok && z0 + it < D) va[it] = *reinterpret_cast<const bf16x8*>(dyb + it * a_zs + a_rel);
        }
        const float* xb = x + (((int64_t)n * D + (z0 - 1)) * H + (y0 - 1)) * W + (x0 - 1);
#pragma unroll
        for (int s_ = 0; s_ < NS; s_++) {
            int pk = s_pk[s_];
            bool ok = (pk >> 24) && (unsigned)(z0 - 1 + (pk & 15)) < (unsigned)D && (unsigned)(y0 - 1 + ((pk >> 4) & 15)) < (unsigned)H &&
                      (unsigned)(x0 - 1 + ((pk >> 8) & 31)) < (unsigned)W;
            xv[s_] = 0.f;
            if (ok) xv[s_] = xb[s_rel[s_]];
        }
    };
    if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < WTZ; it++) *reinterpret_cast<bf16x8*>(dys + (threadIdx.x + it * BLK) * 8) = va[it];
#pragma unroll
        for (int s_ = 0; s_ < NS; s_++) {
            int pk = s_pk[s_];
            if (pk >> 24) {
                int h = (pk >> 8) & 31, row = (pk >> 16) & 255;
                bf16 vb = (bf16)xv[s_];
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {           // copy dx holds x[.. + xx + dx] at xx = h - dx
                    int xx = h - dx;
                    if (xx >= 0 && xx < 16) xsh[(dx * C1_ROWS + row) * 16 + xx] = vb;
                }
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
#pragma unroll
        for (int z = 0; z < WTZ; z++) {
            bf16x8 A = tr_frag(dysb, laneA + z * (WTY * WTX * 32));
#pragma unroll
            for (int j = 0; j < 8; j++) dbs += (float)A[j];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                bf16x8 B = *reinterpret_cast<const bf16x8*>(xsh + offB[h] + z * (WIY * 16));
                acc[h] = mfma16(A, B, acc[h]);
            }
        }
    }
    int64_t nW = (int64_t)Cout * 27;
    float* slab = slabs + (int64_t)blockIdx.x * (nW + Cout);
    reduce_waves<2>(acc, red, wave, lane, [&](int idx, f32x4 sum) {
        int tap = nn + 16 * idx;
        if (tap < 27) {
#pragma unroll
            for (int r = 0; r < 4; r++) slab[(int64_t)(co0 + 4 * G + r) * 27 + tap] = sum[r];
        }
    });
    __syncthreads();
    float sv = dbs;
    sv += __shfl_xor(sv, 16, 64);
    sv += __shfl_xor(sv, 32, 64);
    if (lane < 16) red[wave * 16 + lane] = sv;
    __syncthreads();
    if (threadIdx.x < 16)
        slab[nW + co0 + threadIdx.x] = (red[threadIdx.x] + red[16 + threadIdx.x]) + (red[32 + threadIdx.x] + red[48 + threadIdx.x]);
}

// ---- first layer forward (Cin = 1, fp32 input): y[v][co] = b[co] + sum_tap w[co][tap] * x[v + tap].
// With one input channel the fp32-FMA kernel is VALU-bound (432 FMAs per voxel = 28 us at 96^3 N=2); as an MFMA
// the work is nothing: K = taps.  K layout: two K-steps; k = 8*kg + 4*h + dx' in step A holds tap ((dz,dy) pair
// 2*kg + h, dx = dx'), dx' = 3 is a zero-weight pad; step B holds the ninth (dz,dy) pair (2,2) in kg = 0, h = 0.
// So a lane's 8 k-values are two runs of 4 CONSECUTIVE x positions -> two ds_read_b64 from a bf16 halo tile that is
// stored four times, pre-shifted by 0..3 elements (copy s holds x[i+s] at i), so the run of voxel x starts 8-byte
// aligned in copy (x & 3).  Persistent workgroups, tile 4 x 8 x 16 (wave = z-slice), BN partial sums like the
// persistent conv kernel.
constexpr int C1F_LD = 24;                         // row pitch (elements) of one shifted copy: 18 halo + pad, 8-B aligned rows
__global__ __launch_bounds__(BLK) void conv3_c1_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                                const float* __restrict__ bias, bf16* __restrict__ y, int ycs,
                                                                int Cout, int N, int D, int H, int W, int tilesZ, int tilesY,
                                                                int tilesX, float* __restrict__ part,
                                                                const float* __restrict__ wscale, int relu) {
    constexpr int ROWS = WIZ * WIY;                 // 60 halo rows
    __shared__ __attribute__((aligned(16))) bf16 xsh[4 * ROWS * C1F_LD];
    __shared__ float red[4][16][2];
    int co0 = blockIdx.y * 16;
    int lane = threadIdx.x & 63;
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int vn = lane & 15, kg = lane >> 4;
    // A fragments (weights), built once: lane (co = vn, kg)
    bf16x8 wa, wb;
    {
        const float* wr = wgt + (int64_t)(co0 + vn) * 27;
        float sc = wscale ? wscale[co0 + vn] : 1.f;           // inference: BatchNorm scale folded into the filter
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int pair = 2 * kg + (j >> 2), dx = j & 3;
            wa[j] = (bf16)((dx < 3) ? wr[pair * 3 + dx] * sc : 0.f);
            wb[j] = (bf16)((kg == 0 && j < 3) ? wr[24 + j] * sc : 0.f);
        }
    }
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) bv[j] = bias ? bias[co0 + kg * 4 + j] : 0.f;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    // B fragment addressing: copy (vn & 3), element 4*(vn >> 2) of the row; rows of the two (dz,dy) pairs of this lane
    int cpy = vn & 3, e0 = (vn >> 2) * 4;
    int rowA0 = ((2 * kg) / 3) * WIY + (2 * kg) % 3, rowA1 = ((2 * kg + 1) / 3) * WIY + (2 * kg + 1) % 3;
    int offA0 = (cpy * ROWS + rowA0) * C1F_LD + e0, offA1 = (cpy * ROWS + rowA1) * C1F_LD + e0;
    int offB = (cpy * ROWS + 2 * WIY + 2) * C1F_LD + e0;
    int ntiles = N * tilesZ * tilesY * tilesX;
    // pad elements (index > 17 - shift) are never written by the staging loop: zero them once, they meet zero weights
    for (int idx = threadIdx.x; idx < 4 * ROWS * C1F_LD; idx += BLK) xsh[idx] = (bf16)0.f;
    // staging map, tile-invariant (round 3: it was re-derived by constant division for every element of every tile, and the
    // loads were consumed where they were issued): slot s of thread t = halo element t + 256 s -> (row, h); the next tile's
    // elements are prefetched into registers under this tile's MFMAs and stores
    constexpr int NS = (ROWS * WIX + BLK - 1) / BLK;
    int s_rel[NS], s_pk[NS];
#pragma unroll
    for (int s_ = 0; s_ < NS; s_++) {
        int idx = threadIdx.x + s_ * BLK;
        bool act = idx < ROWS * WIX;
        int h = act ? idx % WIX : 0, row = act ? idx / WIX : 0;
        int iy = row % WIY, iz = row / WIY;
        s_rel[s_] = (iz * H + iy) * W + h;
        s_pk[s_] = iz | (iy << 4) | (h << 8) | (row << 16) | (act ? (1 << 24) : 0);
    }
    float xv[NS];
    auto load_tile = [&](int tile) {
        int t = tile;
        int tx_ = t % tilesX; t /= tilesX;
        int ty_ = t % tilesY; t /= tilesY;
        int tz_ = t % tilesZ; int n = t / tilesZ;
        int z0 = tz_ * WTZ, y0 = ty_ * WTY, x0 = tx_ * WTX;
        const float* xb = x + (((int64_t)n * D + (z0 - 1)) * H + (y0 - 1)) * W + (x0 - 1);      // halo origin; only in-range lanes load
#pragma unroll
        for (int s_ = 0; s_ < NS; s_++) {
            int pk = s_pk[s_];
            bool ok = (pk >> 24) && (unsigned)(z0 - 1 + (pk & 15)) < (unsigned)D && (unsigned)(y0 - 1 + ((pk >> 4) & 15)) < (unsigned)H &&
                      (unsigned)(x0 - 1 + ((pk >> 8) & 31)) < (unsigned)W;
            xv[s_] = 0.f;
            if (ok) xv[s_] = xb[s_rel[s_]];
        }
    };
    if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        int tx_ = t % tilesX; t /= tilesX;
        int ty_ = t % tilesY; t /= tilesY;
        int tz_ = t % tilesZ; int n = t / tilesZ;
        int z0 = tz_ * WTZ, y0 = ty_ * WTY, x0 = tx_ * WTX;
        __syncthreads();
#pragma unroll
        for (int s_ = 0; s_ < NS; s_++) {
            int pk = s_pk[s_];
            if (pk >> 24) {
                int h = (pk >> 8) & 31, row = (pk >> 16) & 255;
                bf16 vb = (bf16)xv[s_];
#pragma unroll
                for (int sft = 0; sft < 4; sft++) {              // copy sft holds x[i + sft] at i
                    int i = h - sft;
                    if (i >= 0) xsh[(sft * ROWS + row) * C1F_LD + i] = vb;
                }
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
        int gz = z0 + wave, gx = x0 + vn;
        bool okzx = gz < D && gx < W;
        bf16* yrow = y + ((((int64_t)n * D + gz) * H + y0) * W + gx) * ycs + co0 + kg * 4;
        int wbase = wave * WIY * C1F_LD;
#pragma unroll
        for (int r = 0; r < WTY; r++) {
            int rb = wbase + r * C1F_LD;
            bf16x4 a0 = *reinterpret_cast<const bf16x4*>(xsh + offA0 + rb), a1 = *reinterpret_cast<const bf16x4*>(xsh + offA1 + rb);
            bf16x4 b0 = *reinterpret_cast<const bf16x4*>(xsh + offB + rb);
            bf16x8 fa = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            bf16x8 fb = {b0[0], b0[1], b0[2], b0[3], b0[0], b0[1], b0[2], b0[3]};
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = mfma16(wa, fa, acc);
            acc = mfma16(wb, fb, acc);
            bool ok = okzx && (y0 + r) < H;
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = acc[j] + bv[j];
                if (relu) v = fmaxf(v, 0.f);
                o[j] = (bf16)v;
                float q = ok ? (float)o[j] : 0.f;
                s1[j] += q; s2[j] = fmaf(q, q, s2[j]);
            }
            // (round 4: 16-byte stores through row pairs, as in the persistent conv, measured slower here -- 23.5 -> 28.0 us: the kernel is
            // one MFMA pair per row and the pairing serialises two rows)
            if (ok) *reinterpret_cast<bf16x4*>(yrow + (int64_t)r * W * ycs) = o;
        }
    }
    if (part) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float a = s1[j], b = s2[j];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
            if (vn == 0) { red[wave][kg * 4 + j][0] = a; red[wave][kg * 4 + j][1] = b; }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 32; idx += BLK) {
            int k = idx & 1, chn = idx >> 1;
            float v = (red[0][chn][k] + red[1][chn][k]) + (red[2][chn][k] + red[3][chn][k]);
            part[((int64_t)blockIdx.x * 2 + k) * Cout + co0 + chn] = v;
        }
    }
}

// fixed-order parallel slab sum: block = EW elements x 256/EW slab groups (EW = 8 when the slab is small and the
// parallelism has to come from the slab dimension).  MFMA_LAYOUT: slab elements are in the (tap, co-block, ci-block,
// lane, reg) order written above and are un-permuted to torch's (Cout, Cin, 27) here.
template <bool MFMA_LAYOUT, int EW>
__global__ __launch_bounds__(BLK) void slab_reduce2_kernel(const float* __restrict__ slabs, int nslab, int64_t slab_sz,
                                                           int64_t nW, float* __restrict__ dW, float* __restrict__ db,
                                                           int accumulate, int Cin, int Cout) {
    slab_reduce2_body<MFMA_LAYOUT, EW>((int)blockIdx.x, slabs, nslab, slab_sz, nW, dW, db, accumulate, Cin, Cout);
}

__global__ __launch_bounds__(BLK) void slab_reduce_tile_kernel(const float* __restrict__ slabs, int nslab, int64_t slab_sz,
                                                               int64_t nW, float* __restrict__ dW, float* __restrict__ db,
                                                               int accumulate, int Cin, int Cout) {
    slab_reduce_tile_body((int)blockIdx.x, slabs, nslab, slab_sz, nW, dW, db, accumulate, Cin, Cout);
}

// tail of a fused backward launch in ONE launch: blocks [0, nfin) finish the split-K input gradient, the rest sum the
// weight-gradient slabs (every kernel node on the stream is a link of the step's dependent chain)
struct TailArgs {
    const float* part; int ks; int64_t M; int C; bf16* y; int ycs; int nfin;
    const float* slabs; int nslab; int64_t slab_sz, nW; float* dW; float* db; int accumulate, Cin, Cout, layout, ew;
};
__global__ __launch_bounds__(BLK) void bwd_tail_kernel(TailArgs a) {
    int b = blockIdx.x;
    if (b < a.nfin) { splitk_finish_body(b, a.nfin, a.part, a.ks, a.M, a.C, nullptr, a.y, a.ycs); return; }
    b -= a.nfin;
    if (a.layout == 2) slab_reduce_tile_body(b, a.slabs, a.nslab, a.slab_sz, a.nW, a.dW, a.db, a.accumulate, a.Cin, a.Cout);
    else if (a.ew == 8) slab_reduce2_body<true, 8>(b, a.slabs, a.nslab, a.slab_sz, a.nW, a.dW, a.db, a.accumulate, a.Cin, a.Cout);
    else slab_reduce2_body<true, 32>(b, a.slabs, a.nslab, a.slab_sz, a.nW, a.dW, a.db, a.accumulate, a.Cin, a.Cout);
}

struct WgCfg { int cob, cib, nt, tg, nsb; };

inline WgCfg wgrad_cfg(int Cin, int Cout, Geo g, int target = 0) {
    WgCfg c;
    // One co block x one ci block x all 27 taps per workgroup (108 accumulator registers + register prefetch of the
    // next tile = 248 VGPRs, 51 KB LDS -> 2 workgroups per CU).  Measured against fatter register blockings
    // (<2,1,27>: 216 accumulators, 1 workgroup/CU; <2,2,9>): the second resident workgroup hides more than the
    // extra staging of dy / x per (co, ci) pair costs, at every level.
    c.cob = 1;
    c.cib = 1;
    int groups = (Cout / 16) * (Cin / 16);
    int64_t ntiles = (int64_t)g.N * cdiv(g.D, WTZ) * cdiv(g.H, WTY) * cdiv(g.W, WTX);
    c.nt = 27;
    c.tg = 1;
    // persistent: one round of workgroups, 2 per CU
    int64_t want = (target > 0 ? target : 2 * persist_cus()) / (int64_t)groups;
    if (want < 1) want = 1;
    int64_t rounds = cdiv(ntiles, want);               // tiles per workgroup; then the fewest slabs that keep it
    c.nsb = (int)cdiv(ntiles, rounds);
    return c;
}

template <int CO_B, int CI_B, int NT>
int launch_wgrad(const bf16* x, int xcs, int Cin, const bf16* dy, int dycs, int Cout, Geo g, float* slabs, WgCfg c,
                 hipStream_t s, Halves xh) {
    size_t lds = (size_t)(CO_B * WNV + CI_B * WNH) * 32;
    if (lds < 16 * 1024 + 256) lds = 16 * 1024 + 256;
    MI3D_SET_MAX_LDS_ONCE((&conv3_wgrad_mfma_kernel<CO_B, CI_B, NT>), lds);
    dim3 grid((unsigned)(c.nsb * c.tg), (unsigned)(Cout / (16 * CO_B)), (unsigned)(Cin / (16 * CI_B)));
    // full-resolution layers: the XCD-aware tile assignment of the fused launch (the stand-alone kernel must sum the same tiles into
    // the same slabs: both routes produce the same bits)
    const bool full = persist_ok(Cout, Cin, g);
    const int xcd_tiles = (full && !mi3d_routes().no_wgrad_xcd) ? 1 : 0;
    // every other layer: flat, placed grid (see the kernel); same slabs, same tiles per slab, same bits
    int pgx = 0, pgy = 0, pgz = 0;
    if (!full && c.tg == 1 && !mi3d_routes().no_wgrad_xcd) {
        pgx = (int)grid.x; pgy = (int)grid.y; pgz = (int)grid.z;
        grid = dim3((unsigned)(pgx * pgy * pgz));
    }
    hipEvent_t tev0 = nullptr, tev1 = nullptr;
    if (time_hook_take(1, Cin, Cout, tev0, tev1))
        hipExtLaunchKernelGGL((conv3_wgrad_mfma_kernel<CO_B, CI_B, NT>), grid, dim3(BLK), lds, s, tev0, tev1, 0, x, xcs, Cin, dy, dycs, Cout,
                              g.N, g.D, g.H, g.W, cdiv(g.D, WTZ), cdiv(g.H, WTY), cdiv(g.W, WTX), c.tg, slabs, xh, xcd_tiles, pgx, pgy, pgz);
    else
        conv3_wgrad_mfma_kernel<CO_B, CI_B, NT><<<grid, BLK, lds, s>>>(x, xcs, Cin, dy, dycs, Cout, g.N, g.D, g.H, g.W,
                                                                      cdiv(g.D, WTZ), cdiv(g.H, WTY), cdiv(g.W, WTX), c.tg, slabs, xh, xcd_tiles,
                                                                      pgx, pgy, pgz);
    MI3D_LAUNCH_CHECK();
    return 0;
}

inline int c1_nsb(Geo g) {
    int64_t ntiles = (int64_t)g.N * cdiv(g.D, WTZ) * cdiv(g.H, WTY) * cdiv(g.W, WTX);
    return (int)(ntiles < 1024 ? ntiles : 1024);        // round 3: 512 measured neutral
}

}  // namespace

__global__ __launch_bounds__(BLK) void slab_job_kernel(SlabJob q) { slab_job_run(q, (int)blockIdx.x); }
int slab_job_launch(const SlabJob& q, hipStream_t s) {
    if (q.nblocks <= 0) return 0;
    slab_job_kernel<<<q.nblocks, BLK, 0, s>>>(q);
    MI3D_LAUNCH_CHECK();
    return 0;
}

static int wgrad_slab_sum(float* ws, int nsb, int Cin, int Cout, float* dW, float* db, int accumulate, hipStream_t s,
                          SlabJob* pend = nullptr) {
    int64_t nW = (int64_t)Cout * Cin * 27, slab_sz = nW + Cout;
    if (pend) { *pend = slab_job_make((dW && slab_sz >= (800 << 10)) ? 2 : 1, ws, nsb, slab_sz, nW, dW, db, Cin, Cout, accumulate); return 0; }
    if (dW && slab_sz >= (800 << 10))
        slab_reduce_tile_kernel<<<(Cout / 16) * (Cin / 16) * 4, BLK, 0, s>>>(ws, nsb, slab_sz, nW, dW, db, accumulate, Cin, Cout);
    else if (slab_sz < (16 << 10))
        slab_reduce2_kernel<true, 8><<<cdiv(slab_sz, 8), BLK, 0, s>>>(ws, nsb, slab_sz, nW, dW, db, accumulate, Cin, Cout);
    else
        slab_reduce2_kernel<true, 32><<<cdiv(slab_sz, 32), BLK, 0, s>>>(ws, nsb, slab_sz, nW, dW, db, accumulate, Cin, Cout);
    MI3D_LAUNCH_CHECK();
    return 0;
}

size_t conv3_mfma_wgrad_ws_floats(int Cin, int Cout, Geo g) {
    if (Cin == 1) return (size_t)c1_nsb(g) * ((size_t)Cout * 27 + Cout);
    return (size_t)wgrad_cfg(Cin, Cout, g).nsb * ((size_t)Cout * Cin * 27 + Cout);
}

bool conv3_mfma_big_geo(Geo g) { return big_geo(g); }

int conv3_mfma_wgrad(const void* x, int xcs, int Cin, const void* dy, int dycs, int Cout, Geo g, float* dW, float* db,
                     int accumulate, float* ws, size_t ws_floats, hipStream_t s, Halves xh, SlabJob* pend, int wg_target) {
    MI3D_CHECK_ARG(conv3_mfma_supported(Cin, Cout, xcs, dycs) && dycs % 8 == 0, "conv3_mfma_wgrad: unsupported channels");
    WgCfg c = wgrad_cfg(Cin, Cout, g, wg_target);
    int64_t nW = (int64_t)Cout * Cin * 27, slab_sz = nW + Cout;
    MI3D_CHECK_ARG(ws_floats >= (size_t)c.nsb * slab_sz, "conv3_mfma_wgrad: workspace too small");
    const bf16* xp = (const bf16*)x; const bf16* dp = (const bf16*)dy;
    int rc;
    if (c.cob == 1 && c.cib == 1) rc = launch_wgrad<1, 1, 27>(xp, xcs, Cin, dp, dycs, Cout, g, ws, c, s, xh);
    else { MI3D_CHECK_ARG(false, "conv3_mfma_wgrad: no kernel for this block config"); return -1; }
    MI3D_TRY(rc);
    return wgrad_slab_sum(ws, c.nsb, Cin, Cout, dW, db, accumulate, s, pend);
}


extern "C" int mi3d_time_next_conv3_kernel(void* start_event, void* stop_event, int kind, int Cin, int Cout) {
    g_hook = TimeHook();
    if (!start_event || !stop_event) return 0;          // NULL events: disarm
    g_hook.e0 = (hipEvent_t)start_event; g_hook.e1 = (hipEvent_t)stop_event;
    g_hook.kind = kind; g_hook.cin = Cin; g_hook.cout = Cout; g_hook.armed = true;
    return 0;
}
extern "C" int mi3d_time_hook_fired(void) {             // 1: the armed launch happened (events recorded); always disarms
    const int f = g_hook.fired ? 1 : 0;
    g_hook = TimeHook();
    return f;
}

// full-resolution layers: dgrad on the persistent body (Cout -> Cin must be one of its shapes)
bool conv3_mfma_bwd_fused_persist_ok(int Cin, int Cout, int xcs, int dycs, Geo g) {
    return persist_ok(Cout, Cin, g) && conv3_mfma_supported(Cin, Cout, xcs, dycs) && dycs % 8 == 0 && !mi3d_routes().no_fused_bwd &&
           !mi3d_routes().no_fused_bwd_p;
}
int conv3_mfma_bwd_fused_persist(const void* x, int xcs, int Cin, const void* dy, int dycs, int Cout, const void* wp_dgrad, void* dx,
                                 int dxcs, Geo g, float* dW, float* db, int accumulate, float* wgws, size_t wgws_floats,
                                 hipStream_t s, Halves xh, Halves dxh, SlabJob* pend) {
    MI3D_CHECK_ARG(conv3_mfma_bwd_fused_persist_ok(Cin, Cout, xcs, dycs, g) && dx, "conv3_mfma_bwd_fused_persist: unsupported layer");
    int groups = (Cout / 16) * (Cin / 16);
    int64_t ntw = (int64_t)g.N * cdiv(g.D, WTZ) * cdiv(g.H, WTY) * cdiv(g.W, WTX);
    const int wcap = persist_cus(), pcap = persist_cus();   // one workgroup of each kind per CU (512 + 512 = no co-residency: measured equal to unfused)
    int64_t want = wcap / groups; if (want < 1) want = 1;
    int64_t rounds = cdiv(ntw, want);
    int nsb = (int)cdiv(ntw, rounds);
    int64_t nW = (int64_t)Cout * Cin * 27, slab_sz = nW + Cout;
    MI3D_CHECK_ARG(wgws_floats >= (size_t)nsb * slab_sz, "conv3_mfma_bwd_fused_persist: workspace too small");
    FusedPArgs a;
    a.wx = (const bf16*)x; a.wxcs = xcs; a.wCin = Cin; a.wdy = (const bf16*)dy; a.wdycs = dycs; a.wCout = Cout;
    a.tZ = cdiv(g.D, WTZ); a.tY = cdiv(g.H, WTY); a.tX = cdiv(g.W, WTX); a.slabs = wgws;
    a.wgx = nsb; a.wgy = Cout / 16; a.wgz = Cin / 16; a.wxh = xh;
    a.dxin = (const bf16*)dy; a.dxcs_in = dycs; a.dwp = (const bf16*)wp_dgrad; a.dyout = (bf16*)dx; a.dycs_out = dxcs;
    a.ptZ = cdiv(g.D, 4); a.ptY = cdiv(g.H, 8); a.ptX = cdiv(g.W, 16); a.pnt = g.N * a.ptZ * a.ptY * a.ptX;
    a.pgrid = a.pnt < pcap ? a.pnt : pcap; a.dyh = dxh;
    a.N = g.N; a.D = g.D; a.H = g.H; a.W = g.W;
    a.xcd_tiles = mi3d_routes().no_wgrad_xcd ? 0 : 1;
    a.flags = (!(mi3d_routes().no_wide_store & 1) && dxcs % 8 == 0 && ((uintptr_t)dx % 16) == 0 && dxh.delta % 8 == 0) ? 2 : 0;
    int nw = a.wgx * a.wgy * a.wgz;
    int half = nw > a.pgrid ? nw : a.pgrid;
    unsigned nblk = (unsigned)(2 * half);
    // dgrad conv Cout -> Cin: persistent shapes (16,16): <1,1>, (32->16): <1,2>, (16->32): <2,1>
    size_t ldsw = (size_t)(WNV + WNH) * 32;
    // measurement hook (mi3d_time_next_conv3_bwd_kernel): one-shot HIP events tightly around this kernel
    hipEvent_t tev0 = nullptr, tev1 = nullptr;
    time_hook_take(0, Cin, Cout, tev0, tev1);
#define FP(COB_, NCH_)                                                                                                        \
    do {                                                                                                                      \
        size_t ldsp = (size_t)(6 * 10 * 18 * 16 + NCH_ * 14 * COB_ * 512) * 2 + 4 * COB_ * 16 * 2 * 4;                        \
        size_t lds = ldsp > ldsw ? ldsp : ldsw;                                                                               \
        MI3D_SET_MAX_LDS_ONCE((&conv3_bwd_fused_persist_kernel<COB_, NCH_>), lds);                                            \
        if (tev0 && tev1)                                                                                                     \
            hipExtLaunchKernelGGL((conv3_bwd_fused_persist_kernel<COB_, NCH_>), dim3(nblk), dim3(BLK), lds, s, tev0, tev1, 0, a); \
        else                                                                                                                  \
            conv3_bwd_fused_persist_kernel<COB_, NCH_><<<nblk, BLK, lds, s>>>(a);                                               \
    } while (0)
    if (Cout == 16 && Cin == 16) FP(1, 1);
    else if (Cout == 32) FP(1, 2);
    else FP(2, 1);
#undef FP
    MI3D_LAUNCH_CHECK();
    return wgrad_slab_sum(wgws, nsb, Cin, Cout, dW, db, accumulate, s, pend);
}

bool conv3_mfma_bwd_fused_ok(int Cin, int Cout, int xcs, int dycs, int dxcs, Geo g) {
    // generic dgrad tilings with two output blocks (Cin % 32) -- not the persistent full-resolution kernels; both
    // products on the MFMA path
    if (big_geo(g) && (persist_ok(Cout, Cin, g) || mi3d_routes().no_fused_bwd_big)) return false;
    return Cin % 32 == 0 && conv3_mfma_supported(Cin, Cout, xcs, dycs) && dycs % 8 == 0 && dxcs % 8 == 0 &&
           !mi3d_routes().no_fused_bwd;
}

int conv3_mfma_bwd_wg_target(int Cin, int Cout, int xcs, int dycs, int dxcs, Geo g) {
    if (conv3_mfma_bwd_fused_persist_ok(Cin, Cout, xcs, dycs, g)) return persist_cus();
    if (conv3_mfma_bwd_fused_ok(Cin, Cout, xcs, dycs, dxcs, g)) return mi3d_routes().fused_wg_target;
    return 0;
}

int conv3_mfma_bwd_fused(const void* x, int xcs, int Cin, const void* dy, int dycs, int Cout, const void* wp_dgrad, void* dx,
                         int dxcs, Geo g, float* dW, float* db, int accumulate, float* wgws, size_t wgws_floats, float* skws,
                         hipStream_t s, SlabJob* pend, int* ks_deferred) {
    if (ks_deferred) *ks_deferred = 0;
    MI3D_CHECK_ARG(conv3_mfma_bwd_fused_ok(Cin, Cout, xcs, dycs, dxcs, g) && dx && ((uintptr_t)dx % 16) == 0 && skws,
                   "conv3_mfma_bwd_fused: unsupported layer %d->%d", Cin, Cout);
    // the weight-gradient half shares the launch (and the CUs' two workgroup slots) with the input-gradient half: sized for
    // ~288 workgroups instead of the stand-alone kernel's 512 it leaves fewer, fatter slabs (less slab traffic to sum) and lets
    // the input-gradient workgroups start earlier.  Scan at 96^3 (tools/abenv.py, ms/step): 64: 2.71, 128: 2.42, 192: 2.33,
    // 256: 2.32, 288: 2.286, 320: 2.288, 352: 2.296, 384: 2.303, 448: 2.309, 512: 2.313.  MI3D_FUSED_WG_TARGET overrides
    WgCfg c = wgrad_cfg(Cin, Cout, g, mi3d_routes().fused_wg_target);
    int64_t nW = (int64_t)Cout * Cin * 27, slab_sz = nW + Cout;
    MI3D_CHECK_ARG(wgws_floats >= (size_t)c.nsb * slab_sz, "conv3_mfma_bwd_fused: workspace too small");
    bool big = big_geo(g);
    // split-K target of the input-gradient half: 128 workgroups instead of the stand-alone conv's 256 -- it runs beside the
    // weight-gradient workgroups of the same launch (scan at 96^3, ms/step: 32: 2.33, 64: 2.294, 96: 2.282, 128: 2.259,
    // 192: 2.260, 256: 2.284).  MI3D_KS_TARGET_BWD overrides (values above 256 would outgrow the planned split-K scratch)
    int ks = pick_ksplit(Cout, Cin, g, conv3_bwd_ks_target());   // dgrad: input channels = Cout, output channels = Cin (1 if big)
    FusedArgs a;
    a.wx = (const bf16*)x; a.wxcs = xcs; a.wCin = Cin; a.wdy = (const bf16*)dy; a.wdycs = dycs; a.wCout = Cout;
    a.tZ = cdiv(g.D, WTZ); a.tY = cdiv(g.H, WTY); a.tX = cdiv(g.W, WTX); a.slabs = wgws;
    a.wgx = c.nsb; a.wgy = Cout / 16; a.wgz = Cin / 16;
    a.dxin = (const bf16*)dy; a.dxcs_in = dycs; a.dCin = Cout; a.dwp = (const bf16*)wp_dgrad; a.dyout = (bf16*)dx; a.dycs_out = dxcs;
    a.dCout = Cin; a.dtZ = cdiv(g.D, 4); a.dtY = cdiv(g.H, 8); a.dtX = cdiv(g.W, big ? 16 : 8); a.dpart = ks > 1 ? skws : nullptr;
    a.dgx = g.N * a.dtZ * a.dtY * a.dtX; a.dgy = Cin / 32; a.dgz = ks;
    a.N = g.N; a.D = g.D; a.H = g.H; a.W = g.W;
    a.flags = (!(mi3d_routes().no_wide_store & 4) && dxcs % 8 == 0 && ((uintptr_t)dx % 16) == 0) ? 2 : 0;
    size_t lds = (size_t)(WNV + WNH) * 32;
    if (lds < 16 * 1024 + 256) lds = 16 * 1024 + 256;
    MI3D_SET_MAX_LDS_ONCE((&conv3_bwd_fused_kernel<false, true>), lds);
    MI3D_SET_MAX_LDS_ONCE((&conv3_bwd_fused_kernel<false, false>), lds);
    MI3D_SET_MAX_LDS_ONCE((&conv3_bwd_fused_kernel<true, false>), lds);
    unsigned nblk = (unsigned)(((a.wgx * a.wgy * a.wgz + 7) & ~7) + a.dgx * a.dgy * a.dgz);
    if (big) conv3_bwd_fused_kernel<true, false><<<nblk, BLK, lds, s>>>(a);
    else if (ks > 1) conv3_bwd_fused_kernel<false, true><<<nblk, BLK, lds, s>>>(a);
    else conv3_bwd_fused_kernel<false, false><<<nblk, BLK, lds, s>>>(a);
    MI3D_LAUNCH_CHECK();
    if (ks > 1 && ks_deferred) {
        // the consumer of dx (the BatchNorm-backward reduction of the layer below) sums the split-K partials itself
        *ks_deferred = ks;
        return wgrad_slab_sum(wgws, c.nsb, Cin, Cout, dW, db, accumulate, s, pend);
    }
    if (ks > 1 && !mi3d_routes().no_bwd_tail) {
        int64_t tot = g.M() * (Cin / 8);
        TailArgs t;
        t.part = skws; t.ks = ks; t.M = g.M(); t.C = Cin; t.y = (bf16*)dx; t.ycs = dxcs;
        t.nfin = (int)(cdiv(tot, BLK) > 2048 ? 2048 : cdiv(tot, BLK));
        t.slabs = wgws; t.nslab = c.nsb; t.slab_sz = slab_sz; t.nW = nW; t.dW = dW; t.db = db; t.accumulate = accumulate;
        t.Cin = Cin; t.Cout = Cout;
        t.layout = (dW && slab_sz >= (800 << 10)) ? 2 : 1;
        t.ew = slab_sz < (16 << 10) ? 8 : 32;
        int nsl = t.layout == 2 ? (Cout / 16) * (Cin / 16) * 4 : (int)cdiv(slab_sz, (int64_t)t.ew);
        bwd_tail_kernel<<<t.nfin + nsl, BLK, 0, s>>>(t);
        MI3D_LAUNCH_CHECK();
        return 0;
    }
    if (ks > 1) {
        int64_t tot = g.M() * (Cin / 8);
        splitk_finish_kernel<<<cdiv(tot, BLK) > 2048 ? 2048 : cdiv(tot, BLK), BLK, 0, s>>>(skws, ks, g.M(), Cin, nullptr, (bf16*)dx, dxcs, 0);
        MI3D_LAUNCH_CHECK();
    }
    return wgrad_slab_sum(wgws, c.nsb, Cin, Cout, dW, db, accumulate, s, pend);
}

// first layer: x fp32 single channel (N,D,H,W), dy bf16 channels-last, Cout % 16 == 0
int conv3_mfma_wgrad_c1(const float* x, const void* dy, int dycs, int Cout, Geo g, float* dW, float* db, int accumulate,
                        float* ws, size_t ws_floats, hipStream_t s, SlabJob* pend) {
    MI3D_CHECK_ARG(Cout % 16 == 0 && dycs % 8 == 0, "conv3_mfma_wgrad_c1: unsupported channels");
    int nsb = c1_nsb(g);
    int64_t nW = (int64_t)Cout * 27, slab_sz = nW + Cout;
    MI3D_CHECK_ARG(ws_floats >= (size_t)nsb * slab_sz, "conv3_mfma_wgrad_c1: workspace too small");
    dim3 grid((unsigned)nsb, (unsigned)(Cout / 16));
    conv3_wgrad_c1_kernel<<<grid, BLK, 0, s>>>(x, (const bf16*)dy, dycs, Cout, g.N, g.D, g.H, g.W, cdiv(g.D, WTZ), cdiv(g.H, WTY),
                                               cdiv(g.W, WTX), ws);
    MI3D_LAUNCH_CHECK();
    if (pend) { *pend = slab_job_make(0, ws, nsb, slab_sz, nW, dW, db, 1, Cout, accumulate); return 0; }
    slab_reduce2_kernel<false, 8><<<cdiv(slab_sz, 8), BLK, 0, s>>>(ws, nsb, slab_sz, nW, dW, db, accumulate, 1, Cout);
    MI3D_LAUNCH_CHECK();
    return 0;
}

// first layer forward: x fp32 single channel (N,D,H,W), y bf16 channels-last, Cout % 16 == 0; part != NULL -> BN partial
// sums [conv3_c1_fwd_stat_blocks][2][Cout] of the stored (rounded) values
int conv3_c1_fwd_stat_blocks(Geo g) {
    int64_t ntiles = (int64_t)g.N * cdiv(g.D, WTZ) * cdiv(g.H, WTY) * cdiv(g.W, WTX);
    return (int)(ntiles < 1024 ? ntiles : 1024);        // round 3 (ms/step): 512: +10 us; 768: =; 2048: +3 us kernel; 3456: +8 us kernel
}
int conv3_c1_fwd_mfma(const float* x, const float* w, const float* bias, void* y, int ycs, int Cout, Geo g, float* part,
                      hipStream_t s, const float* wscale, int relu) {
    MI3D_CHECK_ARG(Cout % 16 == 0 && ycs % 4 == 0 && ((uintptr_t)y % 8) == 0, "conv3_c1_fwd_mfma: unsupported channels");
    dim3 grid((unsigned)conv3_c1_fwd_stat_blocks(g), (unsigned)(Cout / 16));
    conv3_c1_fwd_mfma_kernel<<<grid, BLK, 0, s>>>(x, w, bias, (bf16*)y, ycs, Cout, g.N, g.D, g.H, g.W, cdiv(g.D, WTZ),
                                                  cdiv(g.H, WTY), cdiv(g.W, WTX), part, wscale, relu);
    MI3D_LAUNCH_CHECK();
    return 0;
}
