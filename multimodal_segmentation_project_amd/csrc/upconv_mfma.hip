// upconv_mfma.hip — ConvTranspose3d(k=2,s=2) on the matrix cores (bf16 in / fp32 acc), forward, input-gradient
// and weight-gradient.  Reference: nn.ConvTranspose3d(2f,f,2,stride=2), models/unet.py:56-58,79.
//
// The op is a per-voxel GEMM  out[v][tap][co] = sum_ci x[v][ci] W[ci][co][tap]  + a 2x2x2 pixel shuffle:
//   fwd      : D[co][voxel] = A(W^T: 16 co x 32 ci) * B(x: 32 ci x 16 voxels); the B fragment of a lane is 16 B of
//              one voxel's channels straight from global memory (no LDS); each lane stores 4 channels (8 B) of one
//              output voxel directly into the concat buffer slice.
//   bwd data : D[ci][voxel] = A(W: 16 ci x 32 (tap,co)) * B(g gathered: 32 (tap,co) x 16 voxels), same structure.
//   bwd wgt  : K = voxels is the strided index -> x and g tiles in LDS, fragments via ds_read_b64_tr_b16,
//              persistent workgroups, cross-wave LDS reduce, deterministic slabs (same scheme as conv3 wgrad).
#include "ops.h"

namespace {
constexpr int BLK = 256;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x8 tr_frag(const char* base, int byteoff) {
    auto* p0 = (lds_bf16x4*)(base + byteoff);
    auto* p1 = (lds_bf16x4*)(base + byteoff + 128);
    bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p0);
    bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p1);
    return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// wf[(((tap*COBN + cob)*KS + ks)*64 + lane)*8 + j] = W[32ks + 8G + j][cob*16 + (lane&15)][tap]
// wb[((cib*S + s)*64 + lane)*8 + j]                = W[cib*16 + (lane&15)][co0 + j][tap],  (tap,co0) from kk0 = 32s + 8G
__global__ void upconv_pack_mfma_kernel(const float* __restrict__ w, int Cin, int Cout, bf16* __restrict__ wf,
                                        bf16* __restrict__ wb) {
    int64_t n = (int64_t)Cin * Cout * 8;
    int KS = Cin / 32, COBN = Cout / 16, S = Cout / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += (int64_t)gridDim.x * blockDim.x) {
        bool bw = i >= n;
        int64_t k = bw ? i - n : i;
        int j = k & 7, lane = (k >> 3) & 63, G = lane >> 4;
        int64_t r = k >> 9;
        if (!bw) {
            int ks = r % KS; r /= KS; int cob = r % COBN; int tap = r / COBN;
            int ci = 32 * ks + 8 * G + j, co = cob * 16 + (lane & 15);
            wf[k] = (bf16)w[((int64_t)ci * Cout + co) * 8 + tap];
        } else {
            int s = r % S; int cib = r / S;
            int kk0 = 32 * s + 8 * G, tap = kk0 / Cout, co = kk0 % Cout + j, ci = cib * 16 + (lane & 15);
            wb[k] = (bf16)w[((int64_t)ci * Cout + co) * 8 + tap];
        }
    }
}

// ---------------------------------------------------------------------------------------------- forward
template <int KS>
__global__ __launch_bounds__(BLK) void upconv_mfma_fwd_kernel(const bf16* __restrict__ x, int xcs, const bf16* __restrict__ wf,
                                                              const float* __restrict__ bias, bf16* __restrict__ y, int ycs,
                                                              int Cout, int N, int D, int H, int W, int wide) {
    int64_t M = (int64_t)N * D * H * W;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, vn = lane & 15, G = lane >> 4;
    int COBN = Cout / 16;
    int64_t ngroups = (M + 15) / 16;
    // KS == 1 and one 16-channel output block (the full-resolution upconv, 32 -> 16): the 8 tap fragments are loop
    // invariant -> keep them in registers instead of re-loading 8 KB per 16 voxels
    bool hoist = KS == 1 && COBN == 1;
    bf16x8 wh[8];
    if (hoist) {
#pragma unroll
        for (int tap = 0; tap < 8; tap++) wh[tap] = *reinterpret_cast<const bf16x8*>(wf + (int64_t)tap * 512 + lane * 8);
    }
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
        int64_t v = grp * 16 + vn;
        bool ok = v < M;
        int64_t vc = ok ? v : M - 1;
        bf16x8 Bf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) Bf[ks] = *reinterpret_cast<const bf16x8*>(x + vc * xcs + 32 * ks + 8 * G);
        // 32-bit index math (M < 2^31 checked by the launcher): 64-bit div/mod cost ~80 instructions each
        unsigned vu = (unsigned)vc; int w_ = (int)(vu % (unsigned)W); unsigned r = vu / (unsigned)W; int h_ = (int)(r % (unsigned)H); r /= (unsigned)H; int d_ = (int)(r % (unsigned)D); int n = (int)(r / (unsigned)D);
        if (wide) {
            // wide stores (round 4; all 8 taps in this workgroup, ycs % 8 == 0): the taps c = 0 and c = 1 of one (a, b) are the two
            // x-neighbours 2w and 2w + 1 of the output.  Their two results trade halves through v_permlane16_swap; afterwards lane
            // (vn, G) holds channels (G >> 1) * 8 .. + 7 of voxel 2w + (G & 1): one 16-B store per lane, 64 contiguous bytes per input
            // voxel and 1 KB per wave instead of 8-B pieces with a 64-B stride
            typedef unsigned __attribute__((ext_vector_type(2))) u32x2;
            typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
#pragma unroll
            for (int ab = 0; ab < 4; ab++) {
                int a = ab >> 1, b = ab & 1;
                bf16* yq = y + ((((int64_t)n * 2 * D + 2 * d_ + a) * 2 * H + 2 * h_ + b) * 2 * W + 2 * w_ + (G & 1)) * ycs + (G >> 1) * 8;
                for (int cob = blockIdx.y; cob < COBN; cob += gridDim.y) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    const bf16* wp0 = wf + (((int64_t)(2 * ab) * COBN + cob) * KS) * 512 + lane * 8;
                    const bf16* wp1 = wf + (((int64_t)(2 * ab + 1) * COBN + cob) * KS) * 512 + lane * 8;
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) {
                        acc0 = mfma16(hoist ? wh[2 * ab] : *reinterpret_cast<const bf16x8*>(wp0 + ks * 512), Bf[ks], acc0);
                        acc1 = mfma16(hoist ? wh[2 * ab + 1] : *reinterpret_cast<const bf16x8*>(wp1 + ks * 512), Bf[ks], acc1);
                    }
                    bf16x4 o0, o1;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float bj = bias ? bias[cob * 16 + 4 * G + j] : 0.f;
                        o0[j] = (bf16)(acc0[j] + bj); o1[j] = (bf16)(acc1[j] + bj);
                    }
                    u32x2 u0 = __builtin_bit_cast(u32x2, o0), u1 = __builtin_bit_cast(u32x2, o1);
                    u32x2 p0 = __builtin_amdgcn_permlane16_swap(u0[0], u1[0], false, false);
                    u32x2 p1 = __builtin_amdgcn_permlane16_swap(u0[1], u1[1], false, false);
                    u32x4 wv = {p0[0], p1[0], p0[1], p1[1]};
                    if (ok) *reinterpret_cast<u32x4*>(yq + cob * 16) = wv;
                }
            }
            continue;
        }
#pragma unroll
        for (int tap = 0; tap < 8; tap++) {
            // few voxels (deep levels): the launcher spreads the 8 taps over blockIdx.z -> 8x shorter dependent chain
            if (gridDim.z > 1 && tap != (int)blockIdx.z) continue;
            int a = tap >> 2, b = (tap >> 1) & 1, c = tap & 1;
            bf16* yp = y + ((((int64_t)n * 2 * D + 2 * d_ + a) * 2 * H + 2 * h_ + b) * 2 * W + 2 * w_ + c) * ycs + 4 * G;
            for (int cob = blockIdx.y; cob < COBN; cob += gridDim.y) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const bf16* wp = wf + (((int64_t)tap * COBN + cob) * KS) * 512 + lane * 8;
#pragma unroll
                for (int ks = 0; ks < KS; ks++) acc = mfma16(hoist ? wh[tap] : *reinterpret_cast<const bf16x8*>(wp + ks * 512), Bf[ks], acc);
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] = (bf16)(acc[j] + (bias ? bias[cob * 16 + 4 * G + j] : 0.f));
                if (ok) *reinterpret_cast<bf16x4*>(yp + cob * 16) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward data
struct UBid { int x, y, z, gx, gy; };        // virtual block index / grid (bodies shared with the fused launch below)
template <int S>   // S = 8*Cout/32 K-steps
__device__ __forceinline__ void upconv_mfma_bwd_data_body(UBid bid_, const bf16* __restrict__ g, int gcs, int Cout,
                                                          const bf16* __restrict__ wb, bf16* __restrict__ dx, int dxcs,
                                                          int Cin, int N, int D, int H, int W) {
    int64_t M = (int64_t)N * D * H * W;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, vn = lane & 15, G = lane >> 4;
    int CIBN = Cin / 16;
    int64_t ngroups = (M + 15) / 16;
    for (int64_t grp = (int64_t)bid_.x * 4 + wave; grp < ngroups; grp += (int64_t)bid_.gx * 4) {
        int64_t v = grp * 16 + vn;
        bool ok = v < M;
        int64_t vc = ok ? v : M - 1;
        // 32-bit index math (M < 2^31 checked by the launcher): 64-bit div/mod cost ~80 instructions each
        unsigned vu = (unsigned)vc; int w_ = (int)(vu % (unsigned)W); unsigned r = vu / (unsigned)W; int h_ = (int)(r % (unsigned)H); r /= (unsigned)H; int d_ = (int)(r % (unsigned)D); int n = (int)(r / (unsigned)D);
        bf16x8 Bf[S];
#pragma unroll
        for (int s = 0; s < S; s++) {
            int kk0 = 32 * s + 8 * G, tap = kk0 / Cout, co0 = kk0 % Cout;
            int a = tap >> 2, b = (tap >> 1) & 1, c = tap & 1;
            Bf[s] = *reinterpret_cast<const bf16x8*>(g + ((((int64_t)n * 2 * D + 2 * d_ + a) * 2 * H + 2 * h_ + b) * 2 * W + 2 * w_ + c) * gcs + co0);
        }
        for (int cib = bid_.y; cib < CIBN; cib += bid_.gy) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const bf16* wp = wb + ((int64_t)cib * S) * 512 + lane * 8;
#pragma unroll
            for (int s = 0; s < S; s++) acc = mfma16(*reinterpret_cast<const bf16x8*>(wp + s * 512), Bf[s], acc);
            bf16x4 o = {(bf16)acc[0], (bf16)acc[1], (bf16)acc[2], (bf16)acc[3]};
            if (ok) *reinterpret_cast<bf16x4*>(dx + v * dxcs + cib * 16 + 4 * G) = o;
        }
    }
}

template <int S>
__global__ __launch_bounds__(BLK) void upconv_mfma_bwd_data_kernel(const bf16* __restrict__ g, int gcs, int Cout,
                                                                   const bf16* __restrict__ wb, bf16* __restrict__ dx, int dxcs,
                                                                   int Cin, int N, int D, int H, int W) {
    upconv_mfma_bwd_data_body<S>(UBid{(int)blockIdx.x, (int)blockIdx.y, 0, (int)gridDim.x, (int)gridDim.y}, g, gcs, Cout, wb, dx, dxcs,
                                 Cin, N, D, H, W);
}

// Few voxels, many K-steps (deep levels): one workgroup per 16-voxel group and channel block; the S K-steps are split
// over the 4 waves (each loads S/4 gathered fragments + S/4 weight fragments) and summed through LDS -> the dependent
// load -> MFMA chain is 4x shorter and there are 4x more workgroups.
template <int S>
__device__ __forceinline__ void upconv_mfma_bwd_data_ks_body(UBid bid_, float (*red)[64][4], const bf16* __restrict__ g, int gcs, int Cout,
                                                             const bf16* __restrict__ wb, bf16* __restrict__ dx, int dxcs,
                                                             int Cin, int N, int D, int H, int W) {
    static_assert(S % 4 == 0, "K-steps split over 4 waves");
    constexpr int SW = S / 4;
    int64_t M = (int64_t)N * D * H * W;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, vn = lane & 15, G = lane >> 4;
    int64_t v = (int64_t)bid_.x * 16 + vn;
    bool ok = v < M;
    int64_t vc = ok ? v : M - 1;
    unsigned vu = (unsigned)vc; int w_ = (int)(vu % (unsigned)W); unsigned r = vu / (unsigned)W; int h_ = (int)(r % (unsigned)H); r /= (unsigned)H; int d_ = (int)(r % (unsigned)D); int n = (int)(r / (unsigned)D);
    bf16x8 Bf[SW];
#pragma unroll
    for (int i = 0; i < SW; i++) {
        int s = wave * SW + i;
        int kk0 = 32 * s + 8 * G, tap = kk0 / Cout, co0 = kk0 % Cout;
        int a = tap >> 2, b = (tap >> 1) & 1, c = tap & 1;
        Bf[i] = *reinterpret_cast<const bf16x8*>(g + ((((int64_t)n * 2 * D + 2 * d_ + a) * 2 * H + 2 * h_ + b) * 2 * W + 2 * w_ + c) * gcs + co0);
    }
    int cib = bid_.y;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bf16* wp = wb + ((int64_t)cib * S + wave * SW) * 512 + lane * 8;
#pragma unroll
    for (int i = 0; i < SW; i++) acc = mfma16(*reinterpret_cast<const bf16x8*>(wp + i * 512), Bf[i], acc);
    *reinterpret_cast<f32x4*>(&red[wave][lane][0]) = acc;
    __syncthreads();
    if (wave == 0) {
        f32x4 t = acc;
#pragma unroll
        for (int wv = 1; wv < 4; wv++) {
            f32x4 u = *reinterpret_cast<const f32x4*>(&red[wv][lane][0]);
            t[0] += u[0]; t[1] += u[1]; t[2] += u[2]; t[3] += u[3];
        }
        bf16x4 o = {(bf16)t[0], (bf16)t[1], (bf16)t[2], (bf16)t[3]};
        if (ok) *reinterpret_cast<bf16x4*>(dx + v * dxcs + cib * 16 + 4 * G) = o;
    }
}

template <int S>
__global__ __launch_bounds__(BLK) void upconv_mfma_bwd_data_ks_kernel(const bf16* __restrict__ g, int gcs, int Cout,
                                                                      const bf16* __restrict__ wb, bf16* __restrict__ dx, int dxcs,
                                                                      int Cin, int N, int D, int H, int W) {
    __shared__ float red[4][64][4];
    upconv_mfma_bwd_data_ks_body<S>(UBid{(int)blockIdx.x, (int)blockIdx.y, 0, (int)gridDim.x, (int)gridDim.y}, red, g, gcs, Cout, wb, dx,
                                    dxcs, Cin, N, D, H, W);
}

// ---------------------------------------------------------------------------------------- backward weight
constexpr int UV = 128;    // input voxels per LDS tile = 4 K-steps, one per wave
template <typename Emit>
__device__ __forceinline__ void reduce_waves32(f32x4 (&acc)[32], float* red, int wave, int lane, Emit emit) {
#pragma unroll
    for (int base = 0; base < 32; base += 4) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<f32x4*>(red + ((j * 4 + wave) * 64 + lane) * 4) = acc[base + j];
        __syncthreads();
        f32x4 s = *reinterpret_cast<f32x4*>(red + ((wave * 4 + 0) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; w++) {
            f32x4 t = *reinterpret_cast<f32x4*>(red + ((wave * 4 + w) * 64 + lane) * 4);
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        emit(base + wave, s);
    }
}

// workgroup = 2 ci-blocks x 2 co-blocks x 8 taps = 32 accumulator tiles; x tile [2][UV][16], g tile [2][8][UV][16]
__device__ __forceinline__ void upconv_mfma_bwd_weight_body(UBid bid_, const bf16* __restrict__ x, int xcs, int Cin,
                                                            const bf16* __restrict__ g, int gcs, int Cout, int N,
                                                            int D, int H, int W, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    bf16* xs = reinterpret_cast<bf16*>(lds_raw);            // [2][UV][16]
    bf16* gs = xs + 2 * UV * 16;                            // [2][8][UV][16]
    const char* xsb = reinterpret_cast<const char*>(xs);
    const char* gsb = reinterpret_cast<const char*>(gs);
    int ci0 = bid_.y * 32, co0 = bid_.z * 32;
    int cbn = (Cout - co0) >= 32 ? 2 : 1;                   // Cout = 16 -> one co block (second plane zero)
    int lane = threadIdx.x & 63;
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    int laneK = ((wave * 32 + 8 * G + q) * 16 + 4 * p) * 2;
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbs[2] = {0.f, 0.f};
    int64_t M = (int64_t)N * D * H * W;
    int64_t ntile = (M + UV - 1) / UV;
    constexpr int NA = (2 * UV * 2) / BLK, NB = (2 * 8 * UV * 2) / BLK;
    static_assert(UV * 2 == BLK && NA == 2 && NB == 16, "staging map: slot it of thread t = (block/tap it, voxel t >> 1, half t & 1)");
    bf16x8 va[NA], vb[NB];
    // Every staging slot of a thread belongs to the SAME voxel (t >> 1): slot it of x is ci-block it, slot it of g is
    // (co-block it / 8, tap it % 8).  One voxel decomposition per tile (round 3: there was one per slot -- three runtime divisions
    // x 16 slots = ~1000 vector instructions per tile against 32 MFMAs), the 16 tap / block offsets are scalars.
    const int hv = threadIdx.x >> 1, hf = threadIdx.x & 1;
    auto load_tile = [&](int64_t tile) {
        int64_t v = tile * UV + hv;
        bool ok = v < M;
        unsigned vu = (unsigned)(ok ? v : 0);
        int w_ = (int)(vu % (unsigned)W); unsigned r = vu / (unsigned)W; int h_ = (int)(r % (unsigned)H); r /= (unsigned)H;
        int d_ = (int)(r % (unsigned)D); int n = (int)(r / (unsigned)D);
        const bf16* xp = x + v * xcs + ci0 + hf * 8;
        const bf16* gp = g + ((((int64_t)n * 2 * D + 2 * d_) * 2 * H + 2 * h_) * 2 * W + 2 * w_) * gcs + co0 + hf * 8;
#pragma unroll
        for (int it = 0; it < NA; it++) {
            va[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) va[it] = *reinterpret_cast<const bf16x8*>(xp + it * 16);
        }
#pragma unroll
        for (int it = 0; it < NB; it++) {
            const int tap = it % 8, cb = it / 8;
            const int ta = tap >> 2, tb = (tap >> 1) & 1, tc = tap & 1;
            vb[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (ok && cb < cbn) vb[it] = *reinterpret_cast<const bf16x8*>(gp + (((int64_t)ta * 2 * H + tb) * 2 * W + tc) * gcs + cb * 16);
        }
    };
    // the next tile's global loads are issued before this tile's MFMAs (register prefetch)
    if ((int64_t)bid_.x < ntile) load_tile(bid_.x);
    for (int64_t tile = bid_.x; tile < ntile; tile += bid_.gx) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NA; it++) *reinterpret_cast<bf16x8*>(xs + (threadIdx.x + it * BLK) * 8) = va[it];
#pragma unroll
        for (int it = 0; it < NB; it++) *reinterpret_cast<bf16x8*>(gs + (threadIdx.x + it * BLK) * 8) = vb[it];
        __syncthreads();
        if (tile + bid_.gx < ntile) load_tile(tile + bid_.gx);
        bf16x8 A[2];
#pragma unroll
        for (int a = 0; a < 2; a++) A[a] = tr_frag(xsb, laneK + a * (UV * 32));
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int tap = 0; tap < 8; tap++) {
                bf16x8 B = tr_frag(gsb, laneK + (cb * 8 + tap) * (UV * 32));
                if (bid_.y == 0) {
#pragma unroll
                    for (int j = 0; j < 8; j++) dbs[cb] += (float)B[j];
                }
#pragma unroll
                for (int a = 0; a < 2; a++) acc[(a * 2 + cb) * 8 + tap] = mfma16(A[a], B, acc[(a * 2 + cb) * 8 + tap]);
            }
    }
    int64_t nW = (int64_t)Cin * Cout * 8;
    float* slab = slabs + (int64_t)bid_.x * (nW + Cout);
    float* red = reinterpret_cast<float*>(lds_raw);
    reduce_waves32(acc, red, wave, lane, [&](int idx, f32x4 sum) {
        int tap = idx % 8, cb = (idx / 8) % 2, a = idx / 16;
        if (cb < cbn) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int ci = ci0 + a * 16 + 4 * G + r, co = co0 + cb * 16 + (lane & 15);
                slab[((int64_t)ci * Cout + co) * 8 + tap] = sum[r];
            }
        }
    });
    if (bid_.y == 0) {
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < 2; cb++) {
            float sv = dbs[cb];
            sv += __shfl_xor(sv, 16, 64);
            sv += __shfl_xor(sv, 32, 64);
            if (lane < 16) red[(cb * 4 + wave) * 16 + lane] = sv;
        }
        __syncthreads();
        if (threadIdx.x < cbn * 16) {
            int cb = threadIdx.x / 16, c = threadIdx.x % 16;
            slab[nW + co0 + threadIdx.x] = (red[(cb * 4 + 0) * 16 + c] + red[(cb * 4 + 1) * 16 + c]) +
                                           (red[(cb * 4 + 2) * 16 + c] + red[(cb * 4 + 3) * 16 + c]);
        }
    }
}

__global__ __launch_bounds__(BLK) void upconv_mfma_bwd_weight_kernel(const bf16* __restrict__ x, int xcs, int Cin,
                                                                     const bf16* __restrict__ g, int gcs, int Cout, int N,
                                                                     int D, int H, int W, float* __restrict__ slabs) {
    upconv_mfma_bwd_weight_body(UBid{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y}, x, xcs, Cin, g,
                                gcs, Cout, N, D, H, W, slabs);
}

// Input gradient and weight gradient of a transposed conv are independent: one launch, weight-gradient workgroups first
// (they are the longer ones), data-gradient workgroups behind them on a flat grid.
struct UFusedArgs {
    const bf16* x; int xcs, Cin; const bf16* g; int gcs, Cout; const bf16* wb; bf16* dx; int dxcs; float* slabs;
    int N, D, H, W; int wgx, wgy, wgz, dgx, dgy;
    int map;
};
template <int S, bool KSPLIT>
__global__ __launch_bounds__(BLK) void upconv_mfma_bwd_fused_kernel(UFusedArgs a) {
    int nw = a.wgx * a.wgy * a.wgz, nd = a.dgx * a.dgy;
    // interleave the two kinds (even slots: weight gradient, odd: data gradient) while both last, so that a CU holds
    // one of each; the surplus of the longer list follows
    int b = blockIdx.x, m = nw < nd ? nw : nd;
    bool is_w; int idx;
    const int m16 = (2 * m) & ~15;          // whole groups of 16 blocks: 8 XCDs x (one weight-gradient + one data-gradient block)
    if (a.map && b < m16) {
        // round 4: workgroup b runs on XCD b % 8, so "even = weight gradient, odd = data gradient" gave each XCD ONE kind; now the
        // kinds alternate inside an XCD (its k-th block, k = b / 8, is a weight-gradient block iff k is even)
        int xcd = b & 7, k = b >> 3;
        is_w = (k & 1) == 0;
        idx = (k >> 1) * 8 + xcd;
    } else if (b < 2 * m) { is_w = (b & 1) == 0; idx = b >> 1; }
    else { is_w = nw > nd; idx = m + (b - 2 * m); }
    if (is_w) {
        b = idx;
        upconv_mfma_bwd_weight_body(UBid{b % a.wgx, (b / a.wgx) % a.wgy, b / (a.wgx * a.wgy), a.wgx, a.wgy}, a.x, a.xcs, a.Cin, a.g,
                                    a.gcs, a.Cout, a.N, a.D, a.H, a.W, a.slabs);
    } else {
        b = idx;
        UBid v{b % a.dgx, b / a.dgx, 0, a.dgx, a.dgy};
        if constexpr (KSPLIT) {
            extern __shared__ __attribute__((aligned(16))) char uf_lds[];
            upconv_mfma_bwd_data_ks_body<S>(v, reinterpret_cast<float (*)[64][4]>(uf_lds), a.g, a.gcs, a.Cout, a.wb, a.dx, a.dxcs, a.Cin,
                                            a.N, a.D, a.H, a.W);
        } else {
            upconv_mfma_bwd_data_body<S>(v, a.g, a.gcs, a.Cout, a.wb, a.dx, a.dxcs, a.Cin, a.N, a.D, a.H, a.W);
        }
    }
}

template <int EW>
__global__ __launch_bounds__(BLK) void slab_reduce3_kernel(const float* __restrict__ slabs, int nslab, int64_t slab_sz,
                                                           int64_t nW, float* __restrict__ dW, float* __restrict__ db,
                                                           int accumulate) {
    constexpr int SG = BLK / EW;                  // EW = 8 for small slabs: parallelism from the slab dimension
    __shared__ float red[SG][EW];
    int e = threadIdx.x % EW, sg = threadIdx.x / EW;
    int64_t i = (int64_t)blockIdx.x * EW + e;
    float s = 0.f;
    if (i < slab_sz)
        for (int b = sg; b < nslab; b += SG) s += slabs[(int64_t)b * slab_sz + i];
    red[sg][e] = s;
    __syncthreads();
    if (sg == 0 && i < slab_sz) {
        float tsum = 0.f;
#pragma unroll
        for (int k = 0; k < SG; k += 8)
            tsum += ((red[k][e] + red[k + 1][e]) + (red[k + 2][e] + red[k + 3][e])) +
                    ((red[k + 4][e] + red[k + 5][e]) + (red[k + 6][e] + red[k + 7][e]));
        if (i < nW) { if (dW) dW[i] = accumulate ? dW[i] + tsum : tsum; }
        else if (db) { db[i - nW] = accumulate ? db[i - nW] + tsum : tsum; }
    }
}

inline int upw_nsb(int Cin, int Cout, Geo g) {
    int64_t ntile = (g.M() + UV - 1) / UV;
    int groups = (Cin / 32) * cdiv(Cout, 32);
    int64_t want = cdiv(512, groups);
    return (int)(ntile < want ? ntile : want);
}
inline int wave_grid(int64_t M) {
    int64_t w = (M + 63) / 64;
    return (int)(w < 1 ? 1 : (w > 4096 ? 4096 : w));
}
}  // namespace

bool upconv2_mfma_supported(int Cin, int Cout, int xcs, int ycs) {
    int ks = Cin / 32, s = Cout / 4;
    return Cin % 32 == 0 && Cout % 16 == 0 && xcs % 8 == 0 && ycs % 8 == 0 && (ks == 1 || ks == 2 || ks == 4 || ks == 8) &&
           (s == 4 || s == 8 || s == 16 || s == 32);
}
size_t upconv2_mfma_pack_elems(int Cin, int Cout) { return 2 * (size_t)Cin * Cout * 8; }     // fwd + bwd images

int upconv2_mfma_pack(const float* w, int Cin, int Cout, void* wp, hipStream_t s) {
    int64_t n = (int64_t)upconv2_mfma_pack_elems(Cin, Cout);
    bf16* wf = (bf16*)wp;
    upconv_pack_mfma_kernel<<<cdiv(n, 256) > 2048 ? 2048 : cdiv(n, 256), 256, 0, s>>>(w, Cin, Cout, wf, wf + n / 2);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int upconv2_mfma_fwd(const void* x, int xcs, int Cin, const void* wp, const float* bias, void* y, int ycs, int Cout, Geo g,
                     hipStream_t s) {
    MI3D_CHECK_ARG(upconv2_mfma_supported(Cin, Cout, xcs, ycs), "upconv2_mfma_fwd: unsupported channels %d->%d", Cin, Cout);
    MI3D_CHECK_ARG(g.M() < (1ll << 31), "upconv2_mfma_fwd: more than 2^31 input voxels");
    const bf16* xp = (const bf16*)x; const bf16* wf = (const bf16*)wp; bf16* yp = (bf16*)y;
    int gx = wave_grid(g.M());
    int gy = 1;
    while (gx * gy < 512 && gy < Cout / 16) gy *= 2;      // few voxels (deep levels): parallelise over channel blocks
    int gz = (gx * gy < 512 && !(Cin == 32 && Cout == 16)) ? 8 : 1;     // ... and over the 8 taps
    dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)gz);
    const int wide = (gz == 1 && ycs % 8 == 0 && ((uintptr_t)yp % 16) == 0 && !(mi3d_routes().no_wide_store & 8)) ? 1 : 0;
#define UF(K) upconv_mfma_fwd_kernel<K><<<grid, BLK, 0, s>>>(xp, xcs, wf, bias, yp, ycs, Cout, g.N, g.D, g.H, g.W, wide)
    switch (Cin / 32) { case 1: UF(1); break; case 2: UF(2); break; case 4: UF(4); break; default: UF(8); break; }
#undef UF
    MI3D_LAUNCH_CHECK();
    return 0;
}

size_t upconv2_mfma_bwd_ws_floats(int Cin, int Cout, Geo g) {
    return (size_t)upw_nsb(Cin, Cout, g) * ((size_t)Cin * Cout * 8 + Cout);
}

int upconv2_mfma_bwd(const void* x, int xcs, int Cin, const void* gy, int gycs, int Cout, const void* wp, void* dx, int dxcs,
                     float* dW, float* db, int accumulate, float* ws, size_t ws_floats, Geo g, hipStream_t s, SlabJob* pend) {
    MI3D_CHECK_ARG(upconv2_mfma_supported(Cin, Cout, xcs, gycs), "upconv2_mfma_bwd: unsupported channels");
    MI3D_CHECK_ARG(g.M() < (1ll << 31), "upconv2_mfma_bwd: more than 2^31 input voxels");
    const bf16* xp = (const bf16*)x; const bf16* gp = (const bf16*)gy;
    const bf16* wb = (const bf16*)wp + (size_t)Cin * Cout * 8;
    if (dx && (dW || db) && !mi3d_routes().no_fused_upbwd) {
        int gx = wave_grid(g.M());
        int gy = 1;
        while (gx * gy < 512 && gy < Cin / 16) gy *= 2;
        int64_t gkx = cdiv(g.M(), 16);
        bool ksp = gx * gy < 512 && Cout / 4 >= 8 && gkx * (Cin / 16) <= 8192;
        // about one workgroup of each kind per CU; fewer weight-gradient workgroups where a slab is big (round 3 scan of the cap
        // 128 / 192 / 256 / 384, kernel us: 32->16 35.9 / 33.7 / 29.7 / 39.1, 64->32 25.5 / 24.9 / 27.8 / 30.1, 128->64 20.2 / 21.1 / 22.6 / 22.5;
        // data-gradient cap 128 / 512: +6 ... +14 us)
        const int64_t nWs = (int64_t)Cin * Cout * 8;
        const int wcap = nWs >= 65536 ? 128 : nWs >= 16384 ? 192 : 256, dcap = 256;
        int groups = (Cin / 32) * (int)cdiv(Cout, 32);
        int64_t ntile = (g.M() + UV - 1) / UV;
        int64_t want = cdiv((int64_t)wcap, (int64_t)groups);
        int nsb = (int)(ntile < want ? ntile : want);
        int64_t nW = (int64_t)Cin * Cout * 8, slab_sz = nW + Cout;
        MI3D_CHECK_ARG(ws_floats >= (size_t)nsb * slab_sz, "upconv2_mfma_bwd: workspace too small");
        if (!ksp && gx * gy > dcap) gx = dcap / gy < 1 ? 1 : dcap / gy;          // persistent data-gradient workgroups
        UFusedArgs a{xp, xcs, Cin, gp, gycs, Cout, wb, (bf16*)dx, dxcs, ws, g.N, g.D, g.H, g.W,
                     nsb, Cin / 32, (int)cdiv(Cout, 32), ksp ? (int)gkx : gx, ksp ? Cin / 16 : gy,
                     mi3d_routes().no_upbwd_xcd_mix ? 0 : 1};
        size_t lds = (size_t)(2 * UV + 16 * UV) * 32;
        unsigned nblk = (unsigned)(a.wgx * a.wgy * a.wgz + a.dgx * a.dgy);
#define UFL(SS, KS_)                                                                                                          \
        do {                                                                                                                  \
            MI3D_SET_MAX_LDS_ONCE((&upconv_mfma_bwd_fused_kernel<SS, KS_>), lds);                                             \
            upconv_mfma_bwd_fused_kernel<SS, KS_><<<nblk, BLK, lds, s>>>(a);                                                    \
        } while (0)
        switch (Cout / 4) {
            case 4: UFL(4, false); break;
            case 8: if (ksp) UFL(8, true); else UFL(8, false); break;
            case 16: if (ksp) UFL(16, true); else UFL(16, false); break;
            default: if (ksp) UFL(32, true); else UFL(32, false); break;
        }
#undef UFL
        MI3D_LAUNCH_CHECK();
        if (pend) { *pend = slab_job_make(0, ws, nsb, slab_sz, nW, dW, db, Cin, Cout, accumulate); return 0; }
        if (slab_sz < (16 << 10)) slab_reduce3_kernel<8><<<cdiv(slab_sz, 8), BLK, 0, s>>>(ws, nsb, slab_sz, nW, dW, db, accumulate);
        else slab_reduce3_kernel<32><<<cdiv(slab_sz, 32), BLK, 0, s>>>(ws, nsb, slab_sz, nW, dW, db, accumulate);
        MI3D_LAUNCH_CHECK();
        return 0;
    }
    if (dx) {
        int gx = wave_grid(g.M());
        int gy = 1;
        while (gx * gy < 512 && gy < Cin / 16) gy *= 2;
        dim3 grid((unsigned)gx, (unsigned)gy);
        bf16* dp = (bf16*)dx;
#define UB(SS) upconv_mfma_bwd_data_kernel<SS><<<grid, BLK, 0, s>>>(gp, gycs, Cout, wb, dp, dxcs, Cin, g.N, g.D, g.H, g.W)
#define UBK(SS) upconv_mfma_bwd_data_ks_kernel<SS><<<gridk, BLK, 0, s>>>(gp, gycs, Cout, wb, dp, dxcs, Cin, g.N, g.D, g.H, g.W)
        dim3 gridk((unsigned)cdiv(g.M(), 16), (unsigned)(Cin / 16));
        bool ksp = gx * gy < 512 && Cout / 4 >= 8 && (int64_t)gridk.x * gridk.y <= 8192;     // deep levels
        switch (Cout / 4) {
            case 4: UB(4); break;
            case 8: if (ksp) UBK(8); else UB(8); break;
            case 16: if (ksp) UBK(16); else UB(16); break;
            default: if (ksp) UBK(32); else UB(32); break;
        }
#undef UBK
#undef UB
        MI3D_LAUNCH_CHECK();
    }
    if (dW || db) {
        int nsb = upw_nsb(Cin, Cout, g);
        int64_t nW = (int64_t)Cin * Cout * 8, slab_sz = nW + Cout;
        MI3D_CHECK_ARG(ws_floats >= (size_t)nsb * slab_sz, "upconv2_mfma_bwd: workspace too small");
        size_t lds = (size_t)(2 * UV + 16 * UV) * 32;
        MI3D_SET_MAX_LDS_ONCE(&upconv_mfma_bwd_weight_kernel, lds);
        dim3 grid((unsigned)nsb, (unsigned)(Cin / 32), (unsigned)cdiv(Cout, 32));
        upconv_mfma_bwd_weight_kernel<<<grid, BLK, lds, s>>>(xp, xcs, Cin, gp, gycs, Cout, g.N, g.D, g.H, g.W, ws);
        MI3D_LAUNCH_CHECK();
        if (pend) { *pend = slab_job_make(0, ws, nsb, slab_sz, nW, dW, db, Cin, Cout, accumulate); return 0; }
        if (slab_sz < (16 << 10)) slab_reduce3_kernel<8><<<cdiv(slab_sz, 8), BLK, 0, s>>>(ws, nsb, slab_sz, nW, dW, db, accumulate);
        else slab_reduce3_kernel<32><<<cdiv(slab_sz, 32), BLK, 0, s>>>(ws, nsb, slab_sz, nW, dW, db, accumulate);
        MI3D_LAUNCH_CHECK();
    }
    return 0;
}
