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
constexpr int LOSS_MAXBLK = 512;

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

// logits[n][co][v] = bias[co] + sum_ci z[n,v,ci] * w[co][ci]       (Cout <= NCO per blockIdx.z group)
// grid = (blocks over voxel groups, N, cout groups); a thread owns VV consecutive voxels of one sample so the fp32
// class planes are written as 16-byte stores and no per-voxel 64-bit division is needed.
template <typename T, bool VEC, int NCO, int VV>
__global__ __launch_bounds__(BLK) void conv1_fwd_kernel(const T* __restrict__ z, int zcs, int Cin, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ logits, int Cout,
                                                        int64_t V) {
    int n = blockIdx.y;
    int co0 = blockIdx.z * NCO;
    int nco = min(NCO, Cout - co0);
    const T* zn = z + (int64_t)n * V * zcs;
    float* ln = logits + ((int64_t)n * Cout + co0) * V;
    int64_t ngrp = V / VV;
    for (int64_t grp = (int64_t)blockIdx.x * BLK + threadIdx.x; grp < ngrp; grp += (int64_t)gridDim.x * BLK) {
        int64_t v0 = grp * VV;
        float acc[NCO][VV];
#pragma unroll
        for (int j = 0; j < NCO; j++)
#pragma unroll
            for (int k = 0; k < VV; k++) acc[j][k] = (bias && j < nco) ? bias[co0 + j] : 0.f;
        for (int c0 = 0; c0 < Cin; c0 += CINB) {
            int nci = min(CINB, Cin - c0);
#pragma unroll
            for (int k = 0; k < VV; k++) {
                float zv[CINB];
                load_cin_block<T, VEC>(zn + (v0 + k) * zcs + c0, nci, zv);
#pragma unroll
                for (int j = 0; j < NCO; j++) {
                    if (j < nco) {
#pragma unroll
                        for (int i = 0; i < CINB; i++)
                            if (VEC || i < nci) acc[j][k] = fmaf(zv[i], w[(int64_t)(co0 + j) * Cin + c0 + i], acc[j][k]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NCO; j++) {
            if (j < nco) {
                if constexpr (VV == 4) *reinterpret_cast<float4*>(ln + (int64_t)j * V + v0) = float4{acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
                else ln[(int64_t)j * V + v0] = acc[j][0];
            }
        }
    }
}

// per block (blockIdx.z = 16-channel input block): dz[v][ci] = sum_co dl[co][v] w[co][ci];
// slab[blockIdx.y * gridDim.x + blockIdx.x] = { dW[co][ci] partial, db[co] partial }
template <typename T, bool VEC, int NCO, int VV>
__global__ __launch_bounds__(BLK) void conv1_bwd_kernel(const T* __restrict__ z, int zcs, int Cin, const float* __restrict__ w,
                                                        const float* __restrict__ dl, int Cout, T* __restrict__ dz, int dzcs,
                                                        int64_t V, float* __restrict__ slabs) {
    __shared__ float red[4][NCO * CINB + NCO];
    int n = blockIdx.y;
    int c0 = blockIdx.z * CINB;
    int nci = min(CINB, Cin - c0);
    const T* zn = z + (int64_t)n * V * zcs + c0;
    T* dzn = dz ? dz + (int64_t)n * V * dzcs + c0 : nullptr;
    const float* dln = dl + (int64_t)n * Cout * V;
    float aw[NCO][CINB], ab[NCO], wr[NCO][CINB];
#pragma unroll
    for (int j = 0; j < NCO; j++) {
        ab[j] = 0.f;
#pragma unroll
        for (int i = 0; i < CINB; i++) { aw[j][i] = 0.f; wr[j][i] = (j < Cout && i < nci) ? w[(int64_t)j * Cin + c0 + i] : 0.f; }
    }
    int64_t ngrp = V / VV;
    for (int64_t grp = (int64_t)blockIdx.x * BLK + threadIdx.x; grp < ngrp; grp += (int64_t)gridDim.x * BLK) {
        int64_t v0 = grp * VV;
        float g[NCO][VV];
#pragma unroll
        for (int j = 0; j < NCO; j++) {
            if (j < Cout) {
                if constexpr (VV == 4) {
                    float4 t = *reinterpret_cast<const float4*>(dln + (int64_t)j * V + v0);
                    g[j][0] = t.x; g[j][1] = t.y; g[j][2] = t.z; g[j][3] = t.w;
                } else g[j][0] = dln[(int64_t)j * V + v0];
            } else {
#pragma unroll
                for (int k = 0; k < VV; k++) g[j][k] = 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < VV; k++) {
            float zv[CINB], o[CINB];
            load_cin_block<T, VEC>(zn + (v0 + k) * zcs, nci, zv);
#pragma unroll
            for (int i = 0; i < CINB; i++) o[i] = 0.f;
#pragma unroll
            for (int j = 0; j < NCO; j++) {
                ab[j] += g[j][k];
#pragma unroll
                for (int i = 0; i < CINB; i++) {
                    aw[j][i] = fmaf(g[j][k], zv[i], aw[j][i]);
                    o[i] = fmaf(g[j][k], wr[j][i], o[i]);
                }
            }
            if (dzn) {
                if constexpr (VEC) {
                    float lo[8], hi[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) { lo[i] = o[i]; hi[i] = o[8 + i]; }
                    st8<T>(dzn + (v0 + k) * dzcs, lo);
                    st8<T>(dzn + (v0 + k) * dzcs + 8, hi);
                } else {
#pragma unroll
                    for (int i = 0; i < CINB; i++) if (i < nci) dzn[(v0 + k) * dzcs + i] = from_f<T>(o[i]);
                }
            }
        }
    }
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < NCO; j++) {
#pragma unroll
        for (int i = 0; i < CINB; i++) {
            float sv = wave_sum(aw[j][i]);
            if (lane == 0) red[wave][j * CINB + i] = sv;
        }
        float sb = wave_sum(ab[j]);
        if (lane == 0) red[wave][NCO * CINB + j] = sb;
    }
    __syncthreads();
    int64_t nW = (int64_t)Cout * Cin;
    float* slab = slabs + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (nW + Cout);
    for (int idx = threadIdx.x; idx < NCO * CINB + NCO; idx += BLK) {
        float sv = red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx];
        if (idx < NCO * CINB) {
            int j = idx / CINB, i = idx - j * CINB;
            if (j < Cout && i < nci) slab[(int64_t)j * Cin + c0 + i] = sv;
        } else {
            int j = idx - NCO * CINB;
            if (j < Cout && blockIdx.z == 0) slab[nW + j] = sv;
        }
    }
}

// ---- bf16 fast path of the backward (Cout <= 8, 16-channel input blocks): both contractions on the matrix cores.
// The scalar kernel above spends 128 FMAs + conversions per voxel (VALU-bound: 40 us at 96^3 N=2 for 18 us of traffic).
//   dz[v][ci] = sum_co dl[co][v] w[co][ci]   : A = w^T (16 ci x 32 k), class j sits at k = 8*(j&3) + (j>>2), rest zero;
//                                               B = dl: lane (voxel n, k-group kg) supplies dl[kg (+4)][v_n]  -> one
//                                               scalar load per lane and class slot, no gather
//   dW[co][ci] = sum_v dl[co][v] z[v][ci]    : K = 32 voxels; A = z^T via the transposing LDS read of the wave's own
//                                               32 x 16 tile, B = dl: lane (co = n, kg) supplies 8 consecutive voxels
//                                               of class n (two float4 loads); result rows = ci, columns = co
// A wave owns 32 consecutive voxels per iteration; workgroup = 4 waves; slab layout as the scalar kernel.
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_h;
__device__ __forceinline__ bf16x8 tr_frag_h(const char* base, int byteoff) {
    auto* p0 = (lds_bf16x4_h*)(base + byteoff);
    auto* p1 = (lds_bf16x4_h*)(base + byteoff + 128);
    bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p0);
    bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p1);
    return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
constexpr int C1W = 8;          // waves per workgroup: one slab per workgroup, so more waves = same parallelism with fewer slabs
__global__ __launch_bounds__(C1W * 64) void conv1_bwd_mfma_kernel(const bf16* __restrict__ z, int zcs, int Cin, const float* __restrict__ w,
                                                             const float* __restrict__ dl, int Cout, bf16* __restrict__ dz, int dzcs,
                                                             int64_t V, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) bf16 zt[C1W][32 * 16];
    __shared__ float red[C1W][64][4];
    __shared__ float redb[C1W][16];
    int n = blockIdx.y, c0 = blockIdx.z * 16;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int vn = lane & 15, kg = lane >> 4;
    const bf16* zn = z + (int64_t)n * V * zcs + c0;
    bf16* dzn = dz ? dz + (int64_t)n * V * dzcs + c0 : nullptr;
    const float* dln = dl + (int64_t)n * Cout * V;
    // A fragment of the dz product: lane (m = ci = vn, kg): slot s holds w[kg + 4 s][ci]
    bf16x8 aw;
#pragma unroll
    for (int j = 0; j < 8; j++) aw[j] = (bf16)0.f;
    aw[0] = (bf16)(kg < Cout ? w[(int64_t)kg * Cin + c0 + vn] : 0.f);
    aw[1] = (bf16)(kg + 4 < Cout ? w[(int64_t)(kg + 4) * Cin + c0 + vn] : 0.f);
    f32x4 accW = {0.f, 0.f, 0.f, 0.f};
    float dbs = 0.f;
    const char* ztb = reinterpret_cast<const char*>(zt[wave]);
    int laneK = ((8 * kg + ((lane & 15) >> 2)) * 16 + 4 * (lane & 3)) * 2;
    for (int64_t v0 = ((int64_t)blockIdx.x * C1W + wave) * 32; v0 < V; v0 += (int64_t)gridDim.x * (C1W * 32)) {
        // stage this wave's 32 x 16 z tile (zero rows beyond V)
        {
            int64_t v = v0 + (lane >> 1);
            bf16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
            if (v < V) t = *reinterpret_cast<const bf16x8*>(zn + v * zcs + (lane & 1) * 8);
            *reinterpret_cast<bf16x8*>(zt[wave] + (lane >> 1) * 16 + (lane & 1) * 8) = t;
        }
        // dW: B = dl, 8 consecutive voxels of class vn
        bf16x8 bd = {0, 0, 0, 0, 0, 0, 0, 0};
        if (vn < Cout) {
            int64_t vb = v0 + 8 * kg;
            float t8[8];
            if (vb + 8 <= V && ((V & 3) == 0)) {
                float4 p = *reinterpret_cast<const float4*>(dln + (int64_t)vn * V + vb), q = *reinterpret_cast<const float4*>(dln + (int64_t)vn * V + vb + 4);
                t8[0] = p.x; t8[1] = p.y; t8[2] = p.z; t8[3] = p.w; t8[4] = q.x; t8[5] = q.y; t8[6] = q.z; t8[7] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) t8[j] = (vb + j < V) ? dln[(int64_t)vn * V + vb + j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) { bd[j] = (bf16)t8[j]; dbs += t8[j]; }
        }
        // dz: two 16-voxel blocks
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
            int64_t v = v0 + nb * 16 + vn;
            bf16x8 bl = {0, 0, 0, 0, 0, 0, 0, 0};
            if (v < V) {
                if (kg < Cout) bl[0] = (bf16)dln[(int64_t)kg * V + v];
                if (kg + 4 < Cout) bl[1] = (bf16)dln[(int64_t)(kg + 4) * V + v];
            }
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw, bl, o, 0, 0, 0);
            if (dzn && v < V) {
                bf16x4 ob = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
                *reinterpret_cast<bf16x4*>(dzn + v * dzcs + 4 * kg) = ob;
            }
        }
        // the tile is private to this wave: LDS writes above are complete for the wave before the transposing read
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        bf16x8 az = tr_frag_h(ztb, laneK);
        accW = __builtin_amdgcn_mfma_f32_16x16x32_bf16(az, bd, accW, 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
    }
    // accW: rows ci = 4*kg + r, column co = vn.  Cross-wave sum, then the slab of this block
    *reinterpret_cast<f32x4*>(&red[wave][lane][0]) = accW;
    float sb = dbs;
    sb += __shfl_xor(sb, 16, 64);
    sb += __shfl_xor(sb, 32, 64);
    if (lane < 16) redb[wave][lane] = sb;
    __syncthreads();
    int64_t nW = (int64_t)Cout * Cin;
    float* slab = slabs + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (nW + Cout);
    if (threadIdx.x < 64) {
        int l = threadIdx.x, co = l & 15, g4 = l >> 4;
        if (co < Cout) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float sv = 0.f;
#pragma unroll
                for (int wv = 0; wv < C1W; wv++) sv += red[wv][l][r];
                slab[(int64_t)co * Cin + c0 + 4 * g4 + r] = sv;
            }
        }
    } else if (threadIdx.x < 64 + 16) {
        int co = threadIdx.x - 64;
        if (co < Cout && blockIdx.z == 0) {
            float sv = 0.f;
#pragma unroll
            for (int wv = 0; wv < C1W; wv++) sv += redb[wv][co];
            slab[nW + co] = sv;
        }
    }
}

// ---------------------------------------------------------------------------------------- seg loss
// per-voxel softmax helpers (C <= NC, fully unrolled with predicates)
template <int NC>
__device__ __forceinline__ void softmax_c(const float (&z)[NC], int C, float inv_t, float (&p)[NC], float& lse) {
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NC; c++) if (c < C) mx = fmaxf(mx, z[c] * inv_t);
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < NC; c++) { p[c] = c < C ? __expf(z[c] * inv_t - mx) : 0.f; se += p[c]; }
    float r = __builtin_amdgcn_rcpf(se);          // 1 ulp; the IEEE division costs ~10 VALU slots per voxel in VALU-bound passes
#pragma unroll
    for (int c = 0; c < NC; c++) p[c] *= r;
    lse = mx + __logf(se);
}

// VV consecutive voxels of one sample: class planes as 16-byte loads (VV = 4) -- grid = (blocks, N), no per-voxel
// 64-bit division
template <int NC, int VV>
__device__ __forceinline__ void load_planes(const float* __restrict__ base, int C, int64_t V, int64_t v0, float (&z)[VV][NC]) {
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (c < C) {
            if constexpr (VV == 4) {
                float4 t = *reinterpret_cast<const float4*>(base + (int64_t)c * V + v0);
                z[0][c] = t.x; z[1][c] = t.y; z[2][c] = t.z; z[3][c] = t.w;
            } else z[0][c] = base[(int64_t)c * V + v0];
        } else {
#pragma unroll
            for (int k = 0; k < VV; k++) z[k][c] = 0.f;
        }
    }
}
template <int VV>
__device__ __forceinline__ void load_labels(const int64_t* __restrict__ lb, int (&t)[VV]) {
    if constexpr (VV == 4) {
        longlong2 a = *reinterpret_cast<const longlong2*>(lb), b = *reinterpret_cast<const longlong2*>(lb + 2);
        t[0] = (int)a.x; t[1] = (int)a.y; t[2] = (int)b.x; t[3] = (int)b.y;
    } else t[0] = (int)lb[0];
}

constexpr int NQ = 2 + 3 * MAXC;   // ce, kl, I[c], P[c], T[c]

// METRICS: the same pass also produces the argmax / label count partials of seg_metrics_kernel (the training step needs
// both on the same logits: one read of logits + labels instead of two).
// The pass is VALU-bound (4 classes: softmax + per-class sums + argmax per voxel), so what can leave the vector unit does:
// EXACT (C == NC) drops every `c < C` predicate, TEACH compiles the distillation term in or out, and the integer counts
// (label histogram, argmax histogram, intersections) are wave ballots + scalar popcounts — the vector unit only does the
// compares.  T[c] = sum [t == c] is taken from the exact label histogram instead of a float accumulator.
template <int NC, int VV, bool METRICS, bool EXACT, bool TEACH>
__global__ __launch_bounds__(BLK) void seg_loss_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                           const float* __restrict__ teacher, int C_, int64_t V,
                                                           float inv_t, double* __restrict__ part,
                                                           unsigned long long* __restrict__ counts) {
    const int C = EXACT ? NC : C_;
    constexpr int NQL = 2 + 3 * NC;
    __shared__ float red[4][NQL];
    __shared__ unsigned redc[4][3 * NC + 1];
    unsigned ni[NC], np[NC], nt[NC];                 // wave-uniform (scalar registers)
#pragma unroll
    for (int c = 0; c < NC; c++) ni[c] = np[c] = nt[c] = 0;
    int n = blockIdx.y;
    const float* lg = logits + (int64_t)n * C * V;
    const float* tg = TEACH ? teacher + (int64_t)n * C * V : nullptr;
    const int64_t* lb = labels + (int64_t)n * V;
    float q[2 + 2 * NC];                             // ce, kl, I[c], P[c]
#pragma unroll
    for (int i = 0; i < 2 + 2 * NC; i++) q[i] = 0.f;
    int64_t ngrp = V / VV;
    for (int64_t grp = (int64_t)blockIdx.x * BLK + threadIdx.x; grp < ngrp; grp += (int64_t)gridDim.x * BLK) {
        int64_t v0 = grp * VV;
        float z[VV][NC];
        int t[VV];
        load_planes<NC, VV>(lg, C, V, v0, z);
        load_labels<VV>(lb + v0, t);
        float zt_[VV][NC];
        if constexpr (TEACH) load_planes<NC, VV>(tg, C, V, v0, zt_);
#pragma unroll
        for (int k = 0; k < VV; k++) {
            float p[NC], lse;
            softmax_c<NC>(z[k], C, 1.f, p, lse);
            float zt = 0.f;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (EXACT || c < C) {
                    bool is = (c == t[k]);
                    zt = is ? z[k][c] : zt;
                    q[2 + c] += is ? p[c] : 0.f;
                    q[2 + NC + c] += p[c];
                }
            }
            q[0] += lse - zt;
            {
                float bvv = z[k][0];
                int best = 0;
#pragma unroll
                for (int c = 1; c < NC; c++)
                    if ((EXACT || c < C) && z[k][c] > bvv) { bvv = z[k][c]; best = c; }
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    unsigned long long mt = __ballot(t[k] == c);
                    nt[c] += (unsigned)__popcll(mt);
                    if constexpr (METRICS) {
                        unsigned long long mb = __ballot(best == c);
                        np[c] += (unsigned)__popcll(mb);
                        ni[c] += (unsigned)__popcll(mb & mt);
                    }
                }
            }
            if constexpr (TEACH) {
                float ps[NC], pt[NC], ls, lt;
                softmax_c<NC>(z[k], C, inv_t, ps, ls);
                softmax_c<NC>(zt_[k], C, inv_t, pt, lt);
                float kl = 0.f;
#pragma unroll
                for (int c = 0; c < NC; c++)
                    if ((EXACT || c < C) && pt[c] > 0.f) kl += pt[c] * ((zt_[k][c] * inv_t - lt) - (z[k][c] * inv_t - ls));
                q[1] += kl;
            }
        }
    }
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 2 + 2 * NC; i++) {
        float sv = wave_sum(q[i]);
        if (lane == 0) red[wave][i] = sv;
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            red[wave][2 + 2 * NC + c] = (float)nt[c];                     // exact: a wave sees < 2^24 voxels per block
            redc[wave][c] = ni[c]; redc[wave][NC + c] = np[c]; redc[wave][2 * NC + c] = nt[c];
        }
    }
    __syncthreads();
    // partial rows are COMPACT: 2 + 3C doubles (ce, kl, I[c], P[c], T[c]) and 3C + 1 counts — the single finalize
    // workgroup is bound by pulling the rows through one CU
    const int nqc = 2 + 3 * C;
    if ((int)threadIdx.x < nqc) {
        int i = threadIdx.x, src = i;
        if (i >= 2) { int k = (i - 2) / C, c = (i - 2) % C; src = 2 + k * NC + c; }
        double v = (double)red[0][src] + (double)red[1][src] + (double)red[2][src] + (double)red[3][src];
        part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * nqc + i] = v;
    }
    if constexpr (METRICS) {
        if ((int)threadIdx.x < 3 * C + 1) {
            int i = threadIdx.x;
            unsigned long long v = 0;
            if (i < 3 * C) {
                int src = (i / C) * NC + (i % C);
                v = (unsigned long long)redc[0][src] + redc[1][src] + redc[2][src] + redc[3][src];
            } else {                                            // n_correct = sum_c n_inter[c]
#pragma unroll
                for (int c = 0; c < NC; c++) v += (unsigned long long)redc[0][c] + redc[1][c] + redc[2][c] + redc[3][c];
            }
            counts[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (3 * C + 1) + i] = v;
        }
    }
}

__device__ __forceinline__ void loss_row_sum(const double* __restrict__ part, int nblk, int C, int q, int lane, double* sums);
__device__ __forceinline__ void loss_scalars(const double* sums, int N, int C, int64_t V, LossCfg cfg, float* loss_out, float* coef);
// one 1024-thread block: wave w sums quantities w, w+16 over the block partials (lane-strided doubles + wave tree:
// fixed order), then thread 0 computes the loss and the gradient coefficients
__device__ __forceinline__ void seg_loss_finalize_body(const double* __restrict__ part, int nblk, int N, int C,
                                                       int64_t V, LossCfg cfg, float* loss_out, float* coef) {
    __shared__ double sums[NQ];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nqc = 2 + 3 * C;
    if (threadIdx.x < NQ) sums[threadIdx.x] = 0.0;
    __syncthreads();
    for (int q = wave; q < nqc; q += 16) loss_row_sum(part, nblk, C, q, lane, sums);
    __syncthreads();
    if (threadIdx.x == 0) loss_scalars(sums, N, C, V, cfg, loss_out, coef);
}

// quantity q of the compact partial rows, summed by one wave (lane-strided doubles + wave tree: fixed order) into the MAXC layout
__device__ __forceinline__ void loss_row_sum(const double* __restrict__ part, int nblk, int C, int q, int lane, double* sums) {
    const int nqc = 2 + 3 * C;
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += part[(int64_t)b * nqc + q];
    s = wave_sum_d(s);
    if (lane == 0) sums[q < 2 ? q : 2 + ((q - 2) / C) * MAXC + (q - 2) % C] = s;
}

__device__ __forceinline__ void loss_scalars(const double* sums, int N, int C, int64_t V, LossCfg cfg, float* loss_out, float* coef) {
    {
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

__global__ __launch_bounds__(1024) void seg_loss_finalize_kernel(const double* __restrict__ part, int nblk, int N, int C,
                                                                  int64_t V, LossCfg cfg, float* loss_out, float* coef) {
    seg_loss_finalize_body(part, nblk, N, C, V, cfg, loss_out, coef);
}

// dL/dz of one voxel (header of this file); shared by the plain backward pass and the head-fused one so that both
// produce the same bits from the same logits
template <int NC>
__device__ __forceinline__ void dlogits_voxel(const float (&z)[NC], int t, int C, const float (&A)[NC], const float (&B)[NC],
                                              float ce_s, float go, const float (&kd)[NC], float (&o)[NC]) {
    // no implicit contraction in here: which multiply fuses with which add would otherwise depend on the kernel this is inlined
    // into (measured: one ulp between the two callers at C = 3); the fused multiply-adds are written out
#pragma clang fp contract(off)
    float p[NC], g[NC], lse;
    softmax_c<NC>(z, C, 1.f, p, lse);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        g[c] = (c == t ? A[c] : 0.f) + B[c];
        dot = fmaf(g[c], p[c], dot);
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        float u = p[c] * (g[c] - dot) + kd[c];
        o[c] = go * fmaf(ce_s, p[c] - (c == t ? 1.f : 0.f), u);
    }
}

template <int NC, int VV, bool EXACT, bool TEACH>
__global__ __launch_bounds__(BLK) void seg_loss_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                           const float* __restrict__ teacher, int C_, int64_t V,
                                                           float inv_t, const float* __restrict__ coef,
                                                           const float* __restrict__ grad_out, float* __restrict__ dlogits) {
    const int C = EXACT ? NC : C_;
    int n = blockIdx.y;
    const float* lg = logits + (int64_t)n * C * V;
    const float* tg = TEACH ? teacher + (int64_t)n * C * V : nullptr;
    const int64_t* lb = labels + (int64_t)n * V;
    float* dg = dlogits + (int64_t)n * C * V;
    float go = grad_out ? grad_out[0] : 1.f;
    float A[NC], B[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) { A[c] = coef[c]; B[c] = coef[MAXC + c]; }
    float ce_s = coef[2 * MAXC], kd_s = coef[2 * MAXC + 1];
    int64_t ngrp = V / VV;
    for (int64_t grp = (int64_t)blockIdx.x * BLK + threadIdx.x; grp < ngrp; grp += (int64_t)gridDim.x * BLK) {
        int64_t v0 = grp * VV;
        float z[VV][NC], o[VV][NC];
        int t[VV];
        load_planes<NC, VV>(lg, C, V, v0, z);
        load_labels<VV>(lb + v0, t);
        float zt_[VV][NC];
        if constexpr (TEACH) load_planes<NC, VV>(tg, C, V, v0, zt_);
#pragma unroll
        for (int k = 0; k < VV; k++) {
            float kd[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) kd[c] = 0.f;
            if constexpr (TEACH) {
                float ps[NC], pt[NC], ls, lt;
                softmax_c<NC>(z[k], C, inv_t, ps, ls);
                softmax_c<NC>(zt_[k], C, inv_t, pt, lt);
#pragma unroll
                for (int c = 0; c < NC; c++) kd[c] = kd_s * (ps[c] - pt[c]);
            }
            dlogits_voxel<NC>(z[k], t[k], C, A, B, ce_s, go, kd, o[k]);
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (EXACT || c < C) {
                if constexpr (VV == 4) *reinterpret_cast<float4*>(dg + (int64_t)c * V + v0) = float4{o[0][c], o[1][c], o[2][c], o[3][c]};
                else dg[(int64_t)c * V + v0] = o[0][c];
            }
        }
    }
}

// ------------------------------------------------------------------------------- head + loss in one pass (training step)
// The training step needs the logits only inside the loss: logits = head(z) feed seg_loss_fwd (+ metrics) and, recomputed from
// the same z, seg_loss_bwd -> conv1_bwd.  Unfused that is logits written once and read twice and dlogits written and read once
// (5 x 28 MB at 96^3 N=2) and two more launches; fused, neither tensor exists.  The head is evaluated with conv1_fwd_kernel's
// exact expression (bias, then fmaf over ci in order), so every logit has the bits of the unfused path; the forward sums differ
// from seg_loss_fwd_kernel's only in the order the voxels are added (one voxel per thread here: consecutive lanes read
// consecutive 32-byte channel rows), the backward is bit-identical to seg_loss_bwd + conv1_bwd_mfma given the same `coef`.
template <int NC>
__device__ __forceinline__ void head_logits16(const bf16* __restrict__ zrow, int Cin, const float* __restrict__ w,
                                              const float* __restrict__ bias, int C, float (&z)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; c++) z[c] = (bias && c < C) ? bias[c] : 0.f;
    for (int c0 = 0; c0 < Cin; c0 += CINB) {
        float zv[CINB];
        load_cin_block<bf16, true>(zrow + c0, CINB, zv);
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (c < C) {
#pragma unroll
                for (int i = 0; i < CINB; i++) z[c] = fmaf(zv[i], w[(int64_t)c * Cin + c0 + i], z[c]);
            }
        }
    }
}

// HLW waves per workgroup: the partial rows (one per workgroup) are capped at LOSS_MAXBLK for the single-workgroup finalize, so the
// waves a streaming pass needs to cover the memory latency come from fat workgroups (1024 threads: 32 waves per CU at 2 per CU)
constexpr int HLW = 16;
template <int NC, bool METRICS, bool EXACT, bool TEACH>
__global__ __launch_bounds__(HLW * 64) void head_loss_fwd_kernel(const bf16* __restrict__ zin, int zcs, int Cin, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const int64_t* __restrict__ labels,
                                                            const float* __restrict__ teacher, float inv_t,
                                                            int C_, int64_t V, double* __restrict__ part,
                                                            unsigned long long* __restrict__ counts, float* __restrict__ logits_opt) {
    const int C = EXACT ? NC : C_;
    constexpr int NQL = 2 + 3 * NC;
    __shared__ float red[HLW][NQL];
    __shared__ unsigned redc[HLW][3 * NC + 1];
    unsigned ni[NC], np[NC], nt[NC];                 // wave-uniform (scalar registers)
#pragma unroll
    for (int c = 0; c < NC; c++) ni[c] = np[c] = nt[c] = 0;
    int n = blockIdx.y;
    const bf16* zn = zin + (int64_t)n * V * zcs;
    const int64_t* lb = labels + (int64_t)n * V;
    const float* tg = TEACH ? teacher + (int64_t)n * C * V : nullptr;
    float q[2 + 2 * NC];                             // ce, kl, I[c], P[c]
#pragma unroll
    for (int i = 0; i < 2 + 2 * NC; i++) q[i] = 0.f;
    // every wave runs the same number of iterations (the ballots below are wave-wide): out-of-range lanes carry t = -1.
    // Two 64-voxel groups per iteration, loads of both issued first (a wave has only ~3 iterations: latency, not issue, bounds it)
    constexpr int U = 2;
    const int64_t gstride = (int64_t)gridDim.x * (HLW * 64);
    int64_t base = (int64_t)blockIdx.x * (HLW * 64) + (threadIdx.x & ~63);
    for (; base < V; base += U * gstride) {
        bf16x8 zr[U][2];
        int tt[U];
        bool lv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int64_t v = base + u * gstride + (threadIdx.x & 63);
            lv[u] = v < V;
            tt[u] = -1;
            zr[u][0] = zr[u][1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (lv[u]) {
                static_assert(CINB == 16, "one 32-byte row");
                zr[u][0] = *reinterpret_cast<const bf16x8*>(zn + v * zcs);
                zr[u][1] = *reinterpret_cast<const bf16x8*>(zn + v * zcs + 8);
                tt[u] = (int)lb[v];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (base + u * gstride >= V) break;                       // wave-uniform
            int64_t v = base + u * gstride + (threadIdx.x & 63);
            bool live = lv[u];
            float z[NC];
            int t = tt[u];
            if (Cin == CINB) {
                __attribute__((aligned(16))) bf16 row[16];
                *reinterpret_cast<bf16x8*>(row) = zr[u][0];
                *reinterpret_cast<bf16x8*>(row + 8) = zr[u][1];
                head_logits16<NC>(row, CINB, w, bias, C, z);
            } else if (live) head_logits16<NC>(zn + v * zcs, Cin, w, bias, C, z);
            if (live) {
                if (logits_opt) {
#pragma unroll
                    for (int c = 0; c < NC; c++)
                        if (EXACT || c < C) logits_opt[((int64_t)n * C + c) * V + v] = z[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < NC; c++) z[c] = 0.f;
            }
            float p[NC], lse;
            softmax_c<NC>(z, C, 1.f, p, lse);
            float zt = 0.f;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (EXACT || c < C) {
                    bool is = (c == t);
                    zt = is ? z[c] : zt;
                    q[2 + c] += is ? p[c] : 0.f;
                    q[2 + NC + c] += live ? p[c] : 0.f;
                }
            }
            q[0] += live ? lse - zt : 0.f;
            if constexpr (TEACH) {                       // the distillation term of seg_loss_fwd_kernel, teacher logits from memory
                float ztv[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) ztv[c] = ((EXACT || c < C) && live) ? tg[(int64_t)c * V + v] : 0.f;
                float ps[NC], pt[NC], ls, lt;
                softmax_c<NC>(z, C, inv_t, ps, ls);
                softmax_c<NC>(ztv, C, inv_t, pt, lt);
                float kl = 0.f;
#pragma unroll
                for (int c = 0; c < NC; c++)
                    if ((EXACT || c < C) && pt[c] > 0.f) kl += pt[c] * ((ztv[c] * inv_t - lt) - (z[c] * inv_t - ls));
                q[1] += live ? kl : 0.f;
            }
            float bvv = z[0];
            int best = 0;
#pragma unroll
            for (int c = 1; c < NC; c++)
                if ((EXACT || c < C) && z[c] > bvv) { bvv = z[c]; best = c; }
            if (!live) best = -1;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                unsigned long long mt = __ballot(t == c);
                nt[c] += (unsigned)__popcll(mt);
                if constexpr (METRICS) {
                    unsigned long long mb = __ballot(best == c);
                    np[c] += (unsigned)__popcll(mb);
                    ni[c] += (unsigned)__popcll(mb & mt);
                }
            }
        }
    }
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 2 + 2 * NC; i++) {
        float sv = wave_sum(q[i]);
        if (lane == 0) red[wave][i] = sv;
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            red[wave][2 + 2 * NC + c] = (float)nt[c];
            redc[wave][c] = ni[c]; redc[wave][NC + c] = np[c]; redc[wave][2 * NC + c] = nt[c];
        }
    }
    __syncthreads();
    const int nqc = 2 + 3 * C;                        // the compact rows of seg_loss_fwd_kernel: same finalize kernels
    if ((int)threadIdx.x < nqc) {
        int i = threadIdx.x, src = i;
        if (i >= 2) { int k = (i - 2) / C, c = (i - 2) % C; src = 2 + k * NC + c; }
        double v = 0.0;
#pragma unroll
        for (int wv = 0; wv < HLW; wv++) v += (double)red[wv][src];
        part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * nqc + i] = v;
    }
    if constexpr (METRICS) {
        if ((int)threadIdx.x < 3 * C + 1) {
            int i = threadIdx.x;
            unsigned long long v = 0;
            if (i < 3 * C) {
                int src = (i / C) * NC + (i % C);
                for (int wv = 0; wv < HLW; wv++) v += redc[wv][src];
            } else {
                for (int wv = 0; wv < HLW; wv++)
#pragma unroll
                    for (int c = 0; c < NC; c++) v += redc[wv][c];
            }
            counts[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (3 * C + 1) + i] = v;
        }
    }
}

// conv1_bwd_mfma_kernel with dlogits made on the spot: the wave stages its 32 x 16 z tile as before, lanes l and l + 32
// re-derive voxel l's logits from that tile (head_logits16: the forward's bits), turn them into dL/dz with dlogits_voxel and
// leave them in a wave-private [class][32 voxels] LDS tile, from which both MFMA operands are read where the plain kernel
// reads the dlogits planes.  Cin = 16 (one input block), Cout <= 4.
template <bool TEACH>
__global__ __launch_bounds__(C1W * 64) void head_loss_bwd_mfma_kernel(const bf16* __restrict__ z, int zcs, const float* __restrict__ w,
                                                                      const float* __restrict__ bias, const int64_t* __restrict__ labels,
                                                                      const float* __restrict__ teacher, float inv_t,
                                                                      int Cout, const float* __restrict__ coef,
                                                                      const float* __restrict__ grad_out, bf16* __restrict__ dz, int dzcs,
                                                                      int64_t V, float* __restrict__ slabs) {
    constexpr int NC = 4, Cin = 16;
    __shared__ __attribute__((aligned(16))) bf16 zt[C1W][2][32 * 16];
    __shared__ __attribute__((aligned(16))) float dlt[C1W][2][NC * 32];
    __shared__ float red[C1W][64][4];
    __shared__ float redb[C1W][16];
    int n = blockIdx.y;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int vn = lane & 15, kg = lane >> 4;
    const bf16* zn = z + (int64_t)n * V * zcs;
    bf16* dzn = dz ? dz + (int64_t)n * V * dzcs : nullptr;
    const bool wide_dz = dzn && (dzcs & 7) == 0 && (reinterpret_cast<uintptr_t>(dzn) & 15) == 0;       // uniform: 16-byte dz stores
    const int64_t* lb = labels + (int64_t)n * V;
    const float* tg = TEACH ? teacher + (int64_t)n * Cout * V : nullptr;
    float go = grad_out ? grad_out[0] : 1.f;
    float A[NC], B[NC], kd0[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) { A[c] = coef[c]; B[c] = coef[MAXC + c]; kd0[c] = 0.f; }
    float ce_s = coef[2 * MAXC], kd_s = coef[2 * MAXC + 1];
    bf16x8 aw;
#pragma unroll
    for (int j = 0; j < 8; j++) aw[j] = (bf16)0.f;
    aw[0] = (bf16)(kg < Cout ? w[(int64_t)kg * Cin + vn] : 0.f);
    f32x4 accW = {0.f, 0.f, 0.f, 0.f};
    float dbs = 0.f;
    int laneK = ((8 * kg + ((lane & 15) >> 2)) * 16 + 4 * (lane & 3)) * 2;
    // the wave's 32-voxel chunks are those of conv1_bwd_mfma_kernel, in its order, but taken two at a time: lanes 0-31 make the
    // dlogits of chunk k, lanes 32-63 those of chunk k + 1 (64 distinct voxels per pass of the per-voxel arithmetic); a chunk
    // beyond V is all zeros and adds nothing
    const int64_t stride = (int64_t)gridDim.x * (C1W * 32);
    for (int64_t v0 = ((int64_t)blockIdx.x * C1W + wave) * 32; v0 < V; v0 += 2 * stride) {
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            int64_t v = v0 + hh * stride + (lane >> 1);
            bf16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
            if (v < V) t = *reinterpret_cast<const bf16x8*>(zn + v * zcs + (lane & 1) * 8);
            *reinterpret_cast<bf16x8*>(zt[wave][hh] + (lane >> 1) * 16 + (lane & 1) * 8) = t;
        }
        const int hl = lane >> 5;
        int64_t vl = v0 + hl * stride + (lane & 31);
        int tl = vl < V ? (int)lb[vl] : 0;
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the tiles are private to this wave
        __builtin_amdgcn_wave_barrier();
        {
            float zl[NC], o[NC];
            head_logits16<NC>(zt[wave][hl] + (lane & 31) * 16, Cin, w, bias, Cout, zl);
            if constexpr (TEACH) {                       // the distillation term of seg_loss_bwd_kernel
                float ztv[NC], ps[NC], pt[NC], ls, lt;
#pragma unroll
                for (int c = 0; c < NC; c++) ztv[c] = (c < Cout && vl < V) ? tg[(int64_t)c * V + vl] : 0.f;
                softmax_c<NC>(zl, Cout, inv_t, ps, ls);
                softmax_c<NC>(ztv, Cout, inv_t, pt, lt);
#pragma unroll
                for (int c = 0; c < NC; c++) kd0[c] = kd_s * (ps[c] - pt[c]);
            }
            dlogits_voxel<NC>(zl, tl, Cout, A, B, ce_s, go, kd0, o);
#pragma unroll
            for (int c = 0; c < NC; c++) dlt[wave][hl][c * 32 + (lane & 31)] = (c < Cout && vl < V) ? o[c] : 0.f;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int64_t vc = v0 + hh * stride;
            // dW: B = dl, 8 consecutive voxels of class vn
            bf16x8 bd = {0, 0, 0, 0, 0, 0, 0, 0};
            if (vn < Cout) {
                float4 p4 = *reinterpret_cast<const float4*>(&dlt[wave][hh][vn * 32 + 8 * kg]), q4 = *reinterpret_cast<const float4*>(&dlt[wave][hh][vn * 32 + 8 * kg + 4]);
                float t8[8] = {p4.x, p4.y, p4.z, p4.w, q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                for (int j = 0; j < 8; j++) { bd[j] = (bf16)t8[j]; dbs += t8[j]; }
            }
            bf16x4 obn[2];
#pragma unroll
            for (int nb = 0; nb < 2; nb++) {
                int64_t v = vc + nb * 16 + vn;
                bf16x8 bl = {0, 0, 0, 0, 0, 0, 0, 0};
                if (v < V && kg < Cout) bl[0] = (bf16)dlt[wave][hh][kg * 32 + nb * 16 + vn];
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw, bl, o, 0, 0, 0);
                obn[nb] = bf16x4{(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
                if (!wide_dz && dzn && v < V) *reinterpret_cast<bf16x4*>(dzn + v * dzcs + 4 * kg) = obn[nb];
            }
            if (wide_dz) {
                // 16-byte stores (round 4): the two 16-voxel groups trade halves (v_permlane16_swap); lane (vn, kg) then holds channels
                // (kg >> 1) * 8 .. + 7 of voxel vc + (kg & 1) * 16 + vn
                typedef unsigned __attribute__((ext_vector_type(2))) u32x2;
                typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
                u32x2 u0 = __builtin_bit_cast(u32x2, obn[0]), u1 = __builtin_bit_cast(u32x2, obn[1]);
                u32x2 p0 = __builtin_amdgcn_permlane16_swap(u0[0], u1[0], false, false);
                u32x2 p1 = __builtin_amdgcn_permlane16_swap(u0[1], u1[1], false, false);
                u32x4 wv = {p0[0], p1[0], p0[1], p1[1]};
                const int64_t vw = vc + (kg & 1) * 16 + vn;
                if (vw < V) *reinterpret_cast<u32x4*>(dzn + vw * dzcs + (kg >> 1) * 8) = wv;
            }
            bf16x8 az = tr_frag_h(reinterpret_cast<const char*>(zt[wave][hh]), laneK);
            accW = __builtin_amdgcn_mfma_f32_16x16x32_bf16(az, bd, accW, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    *reinterpret_cast<f32x4*>(&red[wave][lane][0]) = accW;
    float sb = dbs;
    sb += __shfl_xor(sb, 16, 64);
    sb += __shfl_xor(sb, 32, 64);
    if (lane < 16) redb[wave][lane] = sb;
    __syncthreads();
    int64_t nW = (int64_t)Cout * Cin;
    float* slab = slabs + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (nW + Cout);
    if (threadIdx.x < 64) {
        int l = threadIdx.x, co = l & 15, g4 = l >> 4;
        if (co < Cout) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float sv = 0.f;
#pragma unroll
                for (int wv = 0; wv < C1W; wv++) sv += red[wv][l][r];
                slab[(int64_t)co * Cin + 4 * g4 + r] = sv;
            }
        }
    } else if (threadIdx.x < 64 + 16) {
        int co = threadIdx.x - 64;
        if (co < Cout) {
            float sv = 0.f;
#pragma unroll
            for (int wv = 0; wv < C1W; wv++) sv += redb[wv][co];
            slab[nW + co] = sv;
        }
    }
}

// ------------------------------------------------------------------------------------------ metrics
template <int NC, int VV>
__global__ __launch_bounds__(BLK) void seg_metrics_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                          int C, int64_t V, unsigned long long* __restrict__ counts) {
    // counts: [0..C) n_inter, [MAXC..) n_pred, [2*MAXC..) n_tgt, [3*MAXC] n_correct
    int n = blockIdx.y;
    const float* lg = logits + (int64_t)n * C * V;
    const int64_t* lb = labels + (int64_t)n * V;
    unsigned ni[NC], np[NC], nt[NC], nc = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) ni[c] = np[c] = nt[c] = 0;
    int64_t ngrp = V / VV;
    for (int64_t grp = (int64_t)blockIdx.x * BLK + threadIdx.x; grp < ngrp; grp += (int64_t)gridDim.x * BLK) {
        int64_t v0 = grp * VV;
        float z[VV][NC];
        int t[VV];
        load_planes<NC, VV>(lg, C, V, v0, z);
        load_labels<VV>(lb + v0, t);
#pragma unroll
        for (int k = 0; k < VV; k++) {
            float bv = z[k][0];
            int best = 0;
#pragma unroll
            for (int c = 1; c < NC; c++)
                if (c < C && z[k][c] > bv) { bv = z[k][c]; best = c; }       // first maximum wins, as torch.argmax
#pragma unroll
            for (int c = 0; c < NC; c++) {
                ni[c] += (best == c && t[k] == c) ? 1u : 0u;
                np[c] += (best == c) ? 1u : 0u;
                nt[c] += (t[k] == c) ? 1u : 0u;
            }
            nc += (best == t[k]) ? 1u : 0u;
        }
    }
    // per-thread counts -> wave sums -> LDS -> one partial row per block (exact integers, no atomics)
    __shared__ unsigned red[4][3 * NC + 1];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    auto wsum = [&](unsigned v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; };
#pragma unroll
    for (int c = 0; c < NC; c++) {
        unsigned a = wsum(ni[c]), b = wsum(np[c]), d = wsum(nt[c]);
        if (lane == 0) { red[wave][c] = a; red[wave][NC + c] = b; red[wave][2 * NC + c] = d; }
    }
    unsigned e = wsum(nc);
    if (lane == 0) red[wave][3 * NC] = e;
    __syncthreads();
    if ((int)threadIdx.x < 3 * C + 1) {                       // compact row: 3C + 1 counts
        int i = threadIdx.x;
        int src = i == 3 * C ? 3 * NC : (i / C) * NC + (i % C);
        unsigned long long v = (unsigned long long)red[0][src] + red[1][src] + red[2][src] + red[3][src];
        counts[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (3 * C + 1) + i] = v;
    }
}

__device__ __forceinline__ void count_row_sum(const unsigned long long* part, int nblk, int C, int q, int lane, unsigned long long* counts);
__device__ __forceinline__ void metric_scalars(const unsigned long long* counts, int N, int C, int D, int64_t V, float* out);
// Q1: the reference's class loop is range(1, pred.size(1)) AFTER argmax -> bound = first spatial dim D
__device__ __forceinline__ void seg_metrics_finalize_body(const unsigned long long* part, int nblk, int N, int C,
                                                          int D, int64_t V, float* out) {
    __shared__ unsigned long long counts[3 * MAXC + 1];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ncc = 3 * C + 1;
    if (threadIdx.x < 3 * MAXC + 1) counts[threadIdx.x] = 0;
    __syncthreads();
    for (int q = wave; q < ncc; q += 16) count_row_sum(part, nblk, C, q, lane, counts);     // exact integer sums, wave-parallel
    __syncthreads();
    if (threadIdx.x != 0) return;
    metric_scalars(counts, N, C, D, V, out);
}

__device__ __forceinline__ void count_row_sum(const unsigned long long* part, int nblk, int C, int q, int lane, unsigned long long* counts) {
    const int ncc = 3 * C + 1;
    unsigned long long s = 0;
    for (int b = lane; b < nblk; b += 64) s += part[(int64_t)b * ncc + q];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) counts[q == 3 * C ? 3 * MAXC : (q / C) * MAXC + q % C] = s;
}

__device__ __forceinline__ void metric_scalars(const unsigned long long* counts, int N, int C, int D, int64_t V, float* out) {
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

__global__ __launch_bounds__(1024) void seg_metrics_finalize_kernel(const unsigned long long* part, int nblk, int N, int C,
                                                                    int D, int64_t V, float* out) {
    seg_metrics_finalize_body(part, nblk, N, C, D, V, out);
}
// both finalizes of the fused loss + metrics pass in one launch (one chain link less)
__global__ __launch_bounds__(1024) void seg_loss_metrics_finalize_kernel(const double* __restrict__ part, const unsigned long long* cnt,
                                                                         int nblk, int N, int C, int D, int64_t V, LossCfg cfg,
                                                                         float* loss_out, float* coef, float* met_out) {
    // all 2 + 3C + 3C + 1 row sums in ONE phase (two rounds of the 16 waves, every load issued before the first wait), then the
    // two scalar tails on two waves at once: same sums in the same order as the two bodies one after the other, one
    // memory round trip and one serial tail less on the step's dependent chain
    __shared__ double sums[NQ];
    __shared__ unsigned long long counts[3 * MAXC + 1];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nqc = 2 + 3 * C, ncc = 3 * C + 1;
    if (threadIdx.x < NQ) sums[threadIdx.x] = 0.0;
    if (threadIdx.x < 3 * MAXC + 1) counts[threadIdx.x] = 0;
    __syncthreads();
    for (int q = wave; q < nqc + ncc; q += 16) {
        if (q < nqc) loss_row_sum(part, nblk, C, q, lane, sums);
        else count_row_sum(cnt, nblk, C, q - nqc, lane, counts);
    }
    __syncthreads();
    if (threadIdx.x == 0) loss_scalars(sums, N, C, V, cfg, loss_out, coef);
    if (threadIdx.x == 64) metric_scalars(counts, N, C, D, V, met_out);
}

// raw exact counts for the per-class evaluation metrics (test_model.py:242-285): out[0..C) n_inter, [C..2C) n_pred,
// [2C..3C) n_tgt, [3C] n_correct  (int64)
__global__ __launch_bounds__(1024) void seg_counts_finalize_kernel(const unsigned long long* part, int nblk, int C, long long* out) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int q = wave; q < 3 * C + 1; q += 16) {            // rows and output share the compact layout
        unsigned long long s = 0;
        for (int b = lane; b < nblk; b += 64) s += part[(int64_t)b * (3 * C + 1) + q];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) out[q] = (long long)s;
    }
}

constexpr int METRIC_BLOCKS = 512;
inline int sgrid(int64_t total, int cap) {
    int64_t w = (total + BLK - 1) / BLK;
    return (int)(w < 1 ? 1 : (w > cap ? cap : w));
}
inline bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }
constexpr int CONV1_NBLK = 512;
}  // namespace

// blocks per sample for the (blocks, N) grids: `cap` blocks in total, at least one per sample
inline int per_sample_blocks(int64_t V, int vv, int N, int cap) {
    int64_t w = (V / vv + BLK - 1) / BLK, c = cap / N;
    if (c < 1) c = 1;
    return (int)(w < 1 ? 1 : (w > c ? c : w));
}
inline bool vv4(int64_t V, const void* a, const void* b = nullptr, const void* c = nullptr) {
    return V % 4 == 0 && al16(a) && al16(b) && al16(c);
}

int conv1_fwd(int dtype, const void* z, int zcs, int Cin, const float* w, const float* bias, float* logits, int Cout,
              int N, int64_t V, hipStream_t s) {
    MI3D_CHECK_ARG(Cin >= 1 && Cout >= 1, "conv1_fwd: bad channels");
    DISPATCH_T(dtype, T, {
        bool vec = Cin % CINB == 0 && zcs % 8 == 0 && al16(z);
        // one voxel per thread: consecutive lanes read consecutive 32-byte channel rows (a thread owning 4 voxels would
        // spread every 16-byte load instruction of a wave over 64 cache lines -- measured 1.7x slower)
        if (vec && Cout <= 4) {
            dim3 grid((unsigned)per_sample_blocks(V, 1, N, 4096), (unsigned)N, 1);
            conv1_fwd_kernel<T, true, 4, 1><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, bias, logits, Cout, V);
        } else {
            dim3 grid((unsigned)per_sample_blocks(V, 1, N, 4096), (unsigned)N, (unsigned)cdiv(Cout, MAXC));
            if (vec) conv1_fwd_kernel<T, true, MAXC, 1><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, bias, logits, Cout, V);
            else conv1_fwd_kernel<T, false, MAXC, 1><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, bias, logits, Cout, V);
        }
        MI3D_LAUNCH_CHECK();
    });
    return 0;
}

size_t conv1_bwd_ws_floats(int Cin, int Cout) { return (size_t)8192 * ((size_t)Cin * Cout + Cout); }

int conv1_bwd(int dtype, const void* z, int zcs, int Cin, const float* w, const float* dlogits, int Cout, void* dz,
              int dzcs, float* dW, float* db, int accumulate, float* ws, int N, int64_t V, hipStream_t s, SlabJob* pend) {
    MI3D_CHECK_ARG(Cout <= MAXC, "conv1_bwd: out_channels %d > %d unsupported", Cout, MAXC);
    MI3D_CHECK_ARG(N <= CONV1_NBLK, "conv1_bwd: batch %d > %d unsupported", N, CONV1_NBLK);
    int64_t nW = (int64_t)Cin * Cout;
    int nslab = 0;
    // every slab element is written by exactly one (blockIdx.x, blockIdx.y, blockIdx.z) block
    if (dtype == MI3D_BF16 && Cin % 16 == 0 && zcs % 8 == 0 && al16(z) && (!dz || (dzcs % 4 == 0 && ((uintptr_t)dz % 8) == 0)) &&
        !mi3d_routes().no_conv1_mfma) {
        const int capb = 1024;             // 8-wave workgroups, one slab each (conv1_bwd_ws_floats covers 8192)
        int64_t want = (V + C1W * 32 - 1) / (C1W * 32);       // voxels per workgroup iteration
        int bx = capb / N < 1 ? 1 : capb / N;
        if (bx > want) bx = (int)want;
        dim3 grid((unsigned)bx, (unsigned)N, (unsigned)(Cin / 16));
        conv1_bwd_mfma_kernel<<<grid, C1W * 64, 0, s>>>((const bf16*)z, zcs, Cin, w, dlogits, Cout, (bf16*)dz, dzcs, V, ws);
        MI3D_LAUNCH_CHECK();
        if (pend) { *pend = slab_job_make(0, ws, bx * N, nW + Cout, nW, dW, db, Cin, Cout, accumulate); return 0; }
        return slab_reduce(ws, bx * N, nW + Cout, nW, dW, db, accumulate, s);
    }
    DISPATCH_T(dtype, T, {
        bool vec = Cin % CINB == 0 && zcs % 8 == 0 && al16(z) && (!dz || (dzcs % 8 == 0 && al16(dz)));
        bool v4 = false;        // see conv1_fwd: one voxel per thread keeps the channel-row loads coalesced
        int bx = per_sample_blocks(V, v4 ? 4 : 1, N, CONV1_NBLK);
        nslab = bx * N;
        dim3 grid((unsigned)bx, (unsigned)N, (unsigned)cdiv(Cin, CINB));
        if (v4) conv1_bwd_kernel<T, true, 4, 4><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, dlogits, Cout, (T*)dz, dzcs, V, ws);
        else if (vec && Cout <= 4) conv1_bwd_kernel<T, true, 4, 1><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, dlogits, Cout, (T*)dz, dzcs, V, ws);
        else if (vec) conv1_bwd_kernel<T, true, MAXC, 1><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, dlogits, Cout, (T*)dz, dzcs, V, ws);
        else if (Cout <= 4) conv1_bwd_kernel<T, false, 4, 1><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, dlogits, Cout, (T*)dz, dzcs, V, ws);
        else conv1_bwd_kernel<T, false, MAXC, 1><<<grid, BLK, 0, s>>>((const T*)z, zcs, Cin, w, dlogits, Cout, (T*)dz, dzcs, V, ws);
        MI3D_LAUNCH_CHECK();
    });
    return slab_reduce(ws, nslab, nW + Cout, nW, dW, db, accumulate, s);
}

size_t seg_loss_ws_bytes(int C) { return (size_t)LOSS_MAXBLK * NQ * sizeof(double); }

int seg_loss_fwd(const float* logits, const int64_t* labels, const float* teacher, int N, int C, int64_t V, LossCfg cfg,
                 float* loss_out, float* coef, void* ws, hipStream_t s, int D, float* metrics_out, void* metrics_ws) {
    MI3D_CHECK_ARG(C >= 1 && C <= MAXC, "seg_loss: %d classes unsupported (max %d)", C, MAXC);
    MI3D_CHECK_ARG(cfg.w_kd == 0.f || teacher, "seg_loss: distillation weight without teacher logits");
    MI3D_CHECK_ARG(N <= LOSS_MAXBLK, "seg_loss: batch %d > %d unsupported", N, LOSS_MAXBLK);
    const float* tch = cfg.w_kd != 0.f ? teacher : nullptr;
    bool v4 = vv4(V, logits, labels, tch);
    int bx = per_sample_blocks(V, v4 ? 4 : 1, N, LOSS_MAXBLK);
    dim3 grid((unsigned)bx, (unsigned)N);
    float it = 1.f / cfg.temp;
    bool met = metrics_out && metrics_ws;
    unsigned long long* cw = (unsigned long long*)metrics_ws;
#define SLF3(NC_, VV_, EX_, TE_)                                                                                          \
    do {                                                                                                                   \
        if (met) seg_loss_fwd_kernel<NC_, VV_, true, EX_, TE_><<<grid, BLK, 0, s>>>(logits, labels, tch, C, V, it, (double*)ws, cw);   \
        else seg_loss_fwd_kernel<NC_, VV_, false, EX_, TE_><<<grid, BLK, 0, s>>>(logits, labels, tch, C, V, it, (double*)ws, nullptr); \
    } while (0)
#define SLF(NC_, VV_)                                                                                                      \
    do {                                                                                                                   \
        if (C == NC_ && tch) SLF3(NC_, VV_, true, true);                                                                  \
        else if (C == NC_) SLF3(NC_, VV_, true, false);                                                                   \
        else if (tch) SLF3(NC_, VV_, false, true);                                                                        \
        else SLF3(NC_, VV_, false, false);                                                                                \
    } while (0)
    if (C <= 4 && v4) SLF(4, 4);
    else if (C <= 4) SLF(4, 1);
    else if (v4) SLF(MAXC, 4);
    else SLF(MAXC, 1);
#undef SLF3
#undef SLF
    MI3D_LAUNCH_CHECK();
    if (met) seg_loss_metrics_finalize_kernel<<<1, 1024, 0, s>>>((const double*)ws, cw, bx * N, N, C, D, V, cfg, loss_out, coef, metrics_out);
    else seg_loss_finalize_kernel<<<1, 1024, 0, s>>>((const double*)ws, bx * N, N, C, V, cfg, loss_out, coef);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int seg_loss_bwd(const float* logits, const int64_t* labels, const float* teacher, int N, int C, int64_t V, LossCfg cfg,
                 const float* coef, const float* grad_out, float* dlogits, hipStream_t s) {
    MI3D_CHECK_ARG(C >= 1 && C <= MAXC, "seg_loss_bwd: %d classes unsupported", C);
    const float* tch = cfg.w_kd != 0.f ? teacher : nullptr;
    bool v4 = vv4(V, logits, labels, tch) && al16(dlogits);
    dim3 grid((unsigned)per_sample_blocks(V, v4 ? 4 : 1, N, 4096), (unsigned)N);
    float it = 1.f / cfg.temp;
#define SLB3(NC_, VV_, EX_, TE_) seg_loss_bwd_kernel<NC_, VV_, EX_, TE_><<<grid, BLK, 0, s>>>(logits, labels, tch, C, V, it, coef, grad_out, dlogits)
#define SLB(NC_, VV_)                                                         \
    do {                                                                      \
        if (C == NC_ && tch) SLB3(NC_, VV_, true, true);                      \
        else if (C == NC_) SLB3(NC_, VV_, true, false);                       \
        else if (tch) SLB3(NC_, VV_, false, true);                            \
        else SLB3(NC_, VV_, false, false);                                    \
    } while (0)
    if (C <= 4 && v4) SLB(4, 4);
    else if (C <= 4) SLB(4, 1);
    else if (v4) SLB(MAXC, 4);
    else SLB(MAXC, 1);
#undef SLB
#undef SLB3
    MI3D_LAUNCH_CHECK();
    return 0;
}

size_t seg_metrics_ws_bytes(int C) { return (size_t)METRIC_BLOCKS * (3 * MAXC + 1) * sizeof(unsigned long long); }

int seg_metrics(const float* logits, const int64_t* labels, int N, int C, int D, int64_t V, float* out, void* ws,
                hipStream_t s, int64_t* counts_out) {
    MI3D_CHECK_ARG(C >= 1 && C <= MAXC, "seg_metrics: %d classes unsupported", C);
    MI3D_CHECK_ARG(N <= METRIC_BLOCKS, "seg_metrics: batch %d > %d unsupported", N, METRIC_BLOCKS);
    bool v4 = vv4(V, logits, labels);
    int bx = per_sample_blocks(V, v4 ? 4 : 1, N, METRIC_BLOCKS);
    dim3 grid((unsigned)bx, (unsigned)N);
    unsigned long long* cw = (unsigned long long*)ws;
    if (C <= 4 && v4) seg_metrics_kernel<4, 4><<<grid, BLK, 0, s>>>(logits, labels, C, V, cw);
    else if (C <= 4) seg_metrics_kernel<4, 1><<<grid, BLK, 0, s>>>(logits, labels, C, V, cw);
    else if (v4) seg_metrics_kernel<MAXC, 4><<<grid, BLK, 0, s>>>(logits, labels, C, V, cw);
    else seg_metrics_kernel<MAXC, 1><<<grid, BLK, 0, s>>>(logits, labels, C, V, cw);
    MI3D_LAUNCH_CHECK();
    if (out) seg_metrics_finalize_kernel<<<1, 1024, 0, s>>>((const unsigned long long*)ws, bx * N, N, C, D, V, out);
    if (counts_out) seg_counts_finalize_kernel<<<1, 1024, 0, s>>>((const unsigned long long*)ws, bx * N, C, (long long*)counts_out);
    MI3D_LAUNCH_CHECK();
    return 0;
}

// ---- head + loss fused (training step)
bool head_loss_ok(int dtype, const void* z, int zcs, int Cin, int C, LossCfg cfg) {
    return dtype == MI3D_BF16 && Cin % CINB == 0 && zcs % 8 == 0 && al16(z) && C >= 1 && C <= 4 && !mi3d_routes().no_head_loss;
}
bool head_loss_bwd_ok(int dtype, const void* z, int zcs, int Cin, int C, LossCfg cfg, const void* dz, int dzcs) {
    return head_loss_ok(dtype, z, zcs, Cin, C, cfg) && Cin == 16 && (!dz || (dzcs % 4 == 0 && ((uintptr_t)dz % 8) == 0)) &&
           !mi3d_routes().no_conv1_mfma;
}

int head_loss_fwd(const void* z, int zcs, int Cin, const float* w, const float* bias, const int64_t* labels, const float* teacher,
                  int N, int C, int64_t V, LossCfg cfg, float* loss_out, float* coef, void* ws, hipStream_t s, int D, float* metrics_out,
                  void* metrics_ws, float* logits_opt) {
    MI3D_CHECK_ARG(head_loss_ok(MI3D_BF16, z, zcs, Cin, C, cfg), "head_loss_fwd: unsupported shape Cin=%d C=%d", Cin, C);
    MI3D_CHECK_ARG(cfg.w_kd == 0.f || teacher, "head_loss_fwd: distillation weight without teacher logits");
    const float* tch = cfg.w_kd != 0.f ? teacher : nullptr;
    const float it = 1.f / cfg.temp;
    MI3D_CHECK_ARG(N <= LOSS_MAXBLK, "head_loss_fwd: batch %d > %d unsupported", N, LOSS_MAXBLK);
    int64_t wantb = (V + HLW * 64 - 1) / (HLW * 64), capb = LOSS_MAXBLK / N < 1 ? 1 : LOSS_MAXBLK / N;
    int bx = (int)(wantb < capb ? wantb : capb);
    dim3 grid((unsigned)bx, (unsigned)N);
    bool met = metrics_out && metrics_ws;
    unsigned long long* cw = (unsigned long long*)metrics_ws;
    const bf16* zp = (const bf16*)z;
#define HLF3(ME_, EX_, TE_) head_loss_fwd_kernel<4, ME_, EX_, TE_><<<grid, HLW * 64, 0, s>>>(zp, zcs, Cin, w, bias, labels, tch, it, C, V, (double*)ws, cw, logits_opt)
#define HLF(ME_, EX_) do { if (tch) HLF3(ME_, EX_, true); else HLF3(ME_, EX_, false); } while (0)
    if (met && C == 4) HLF(true, true);
    else if (met) HLF(true, false);
    else if (C == 4) HLF(false, true);
    else HLF(false, false);
#undef HLF
#undef HLF3
    MI3D_LAUNCH_CHECK();
    if (met) seg_loss_metrics_finalize_kernel<<<1, 1024, 0, s>>>((const double*)ws, cw, bx * N, N, C, D, V, cfg, loss_out, coef, metrics_out);
    else seg_loss_finalize_kernel<<<1, 1024, 0, s>>>((const double*)ws, bx * N, N, C, V, cfg, loss_out, coef);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int head_loss_bwd(const void* z, int zcs, int Cin, const float* w, const float* bias, const int64_t* labels, const float* teacher, int C,
                  LossCfg cfg, const float* coef, const float* grad_out, void* dz, int dzcs, float* dW, float* db, int accumulate,
                  float* ws, int N, int64_t V, hipStream_t s, SlabJob* pend) {
    MI3D_CHECK_ARG(head_loss_bwd_ok(MI3D_BF16, z, zcs, Cin, C, cfg, dz, dzcs), "head_loss_bwd: unsupported shape Cin=%d C=%d", Cin, C);
    MI3D_CHECK_ARG(cfg.w_kd == 0.f || teacher, "head_loss_bwd: distillation weight without teacher logits");
    const float* tch = cfg.w_kd != 0.f ? teacher : nullptr;
    const float it = 1.f / cfg.temp;
    MI3D_CHECK_ARG(N <= CONV1_NBLK, "head_loss_bwd: batch %d > %d unsupported", N, CONV1_NBLK);
    int64_t nW = (int64_t)Cin * C;
    const int capb = 1024;             // as conv1_bwd's matrix-core route: same workgroup -> voxel map, same slabs
    int64_t want = (V + C1W * 32 - 1) / (C1W * 32);
    int bx = capb / N < 1 ? 1 : capb / N;
    if (bx > want) bx = (int)want;
    dim3 grid((unsigned)bx, (unsigned)N, 1);
    if (tch) head_loss_bwd_mfma_kernel<true><<<grid, C1W * 64, 0, s>>>((const bf16*)z, zcs, w, bias, labels, tch, it, C, coef, grad_out, (bf16*)dz, dzcs, V, ws);
    else head_loss_bwd_mfma_kernel<false><<<grid, C1W * 64, 0, s>>>((const bf16*)z, zcs, w, bias, labels, nullptr, it, C, coef, grad_out, (bf16*)dz, dzcs, V, ws);
    MI3D_LAUNCH_CHECK();
    if (pend) { *pend = slab_job_make(0, ws, bx * N, nW + C, nW, dW, db, Cin, C, accumulate); return 0; }
    return slab_reduce(ws, bx * N, nW + C, nW, dW, db, accumulate, s);
}
