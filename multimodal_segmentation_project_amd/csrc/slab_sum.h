// slab_sum.h -- fixed-order sums of per-workgroup weight-gradient slabs as __device__ bodies, so that a slab sum can ride
// in another kernel's launch (every kernel node on the stream is a link of the step's dependent chain, ~2-5 us each).
#pragma once
#include "common.h"

// one deferred slab sum.  layout: 0 = slab element i -> output i, 1 = MFMA-native conv slab (un-permuted to
// (Cout,Cin,27)), 2 = the same through an LDS transpose (MB-sized weight tensors); ew = elements per block
struct SlabJob {
    const float* slabs = nullptr; float* dW = nullptr; float* db = nullptr;
    int64_t slab_sz = 0, nW = 0;
    int nslab = 0, Cin = 0, Cout = 0, layout = 0, ew = 32, accumulate = 0, nblocks = 0;
};

inline SlabJob slab_job_make(int layout, const float* slabs, int nslab, int64_t slab_sz, int64_t nW, float* dW, float* db,
                             int Cin, int Cout, int accumulate) {
    SlabJob q;
    q.slabs = slabs; q.dW = dW; q.db = db; q.slab_sz = slab_sz; q.nW = nW; q.nslab = nslab; q.Cin = Cin; q.Cout = Cout;
    q.layout = layout; q.accumulate = accumulate;
    if (layout == 2) { q.ew = 0; q.nblocks = (Cout / 16) * (Cin / 16) * 4; }
    else {
        if (layout == 1) q.ew = slab_sz < (16 << 10) ? 8 : 32;
        else q.ew = slab_sz < 128 ? 1 : slab_sz < 1024 ? 4 : slab_sz < (16 << 10) ? 8 : 32;
        q.nblocks = (int)((slab_sz + q.ew - 1) / q.ew);
    }
    return q;
}

#ifdef __HIPCC__
constexpr int SLAB_BLK = 256;

// block = EW elements x 256/EW slab groups (EW small when the slab is small and the parallelism has to come from the
// slab dimension).  MFMA_LAYOUT: slab elements are in (tap, co-block, ci-block, lane, reg) order.
template <bool MFMA_LAYOUT, int EW>
__device__ __forceinline__ void slab_reduce2_body(int blk, const float* __restrict__ slabs, int nslab, int64_t slab_sz,
                                                  int64_t nW, float* __restrict__ dW, float* __restrict__ db,
                                                  int accumulate, int Cin, int Cout) {
    constexpr int SG = SLAB_BLK / EW;
    __shared__ float red[SG][EW];
    int e = threadIdx.x % EW, sg = threadIdx.x / EW;
    int64_t i = (int64_t)blk * EW + e;
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
        if (i < nW) {
            int64_t o = i;
            if (MFMA_LAYOUT) {
                int r = i & 3, lane = (i >> 2) & 63; int64_t t = i >> 8;
                int CIBN = Cin / 16, COBN = Cout / 16;
                int cib = t % CIBN; t /= CIBN; int cob = t % COBN; int tap = t / COBN;
                int co = cob * 16 + 4 * (lane >> 4) + r, ci = cib * 16 + (lane & 15);
                o = ((int64_t)co * Cin + ci) * 27 + tap;
            }
            if (dW) dW[o] = accumulate ? dW[o] + tsum : tsum;
        } else if (db) { db[i - nW] = accumulate ? db[i - nW] + tsum : tsum; }
    }
}

// Large weight tensors (few slabs, MBs of output): one block per (co-block, ci-block, 4-row group G) sums the 27 tap
// tiles' 64-float row groups (256 B contiguous reads), transposes through LDS and writes the four (co) rows'
// contiguous 16 ci x 27 tap runs -- scattered 4-byte stores cost 4x the time there.
__device__ __forceinline__ void slab_reduce_tile_body(int blk, const float* __restrict__ slabs, int nslab, int64_t slab_sz,
                                                      int64_t nW, float* __restrict__ dW, float* __restrict__ db,
                                                      int accumulate, int Cin, int Cout) {
    __shared__ float out[4 * 16 * 27];
    int CIBN = Cin / 16, COBN = Cout / 16;
    int t = blk;
    int G = t & 3; t >>= 2;
    int ib = t % CIBN, cb = t / CIBN;
    constexpr int NE = (27 * 64 + SLAB_BLK - 1) / SLAB_BLK;
    float acc[NE];
    const float* p[NE];
#pragma unroll
    for (int k = 0; k < NE; k++) {
        int e = threadIdx.x + k * SLAB_BLK;
        int tap = e < 27 * 64 ? (e >> 6) : 26, j = e & 63;
        p[k] = slabs + (((int64_t)tap * COBN + cb) * CIBN + ib) * 256 + G * 64 + j;
        acc[k] = 0.f;
    }
    for (int b = 0; b < nslab; b++) {              // NE independent 256 B-coalesced loads in flight per slab
#pragma unroll
        for (int k = 0; k < NE; k++) acc[k] += p[k][(int64_t)b * slab_sz];
    }
#pragma unroll
    for (int k = 0; k < NE; k++) {
        int e = threadIdx.x + k * SLAB_BLK;
        if (e < 27 * 64) { int tap = e >> 6, j = e & 63; out[((j & 3) * 16 + (j >> 2)) * 27 + tap] = acc[k]; }
    }
    __syncthreads();
    for (int m = threadIdx.x; m < 4 * 432; m += SLAB_BLK) {
        int r = m / 432, k = m - r * 432;
        int64_t o = ((int64_t)(cb * 16 + 4 * G + r) * Cin + ib * 16) * 27 + k;
        float v = out[m];
        dW[o] = accumulate ? dW[o] + v : v;
    }
    if (db && blk == 0) {
        for (int c = threadIdx.x; c < Cout; c += SLAB_BLK) {
            float s = 0.f;
            for (int b = 0; b < nslab; b++) s += slabs[(int64_t)b * slab_sz + nW + c];
            db[c] = accumulate ? db[c] + s : s;
        }
    }
}

// block `blk` (0 <= blk < q.nblocks) of job q; 256 threads
__device__ __forceinline__ void slab_job_run(const SlabJob& q, int blk) {
    if (q.layout == 2) slab_reduce_tile_body(blk, q.slabs, q.nslab, q.slab_sz, q.nW, q.dW, q.db, q.accumulate, q.Cin, q.Cout);
    else if (q.layout == 1) {
        if (q.ew == 8) slab_reduce2_body<true, 8>(blk, q.slabs, q.nslab, q.slab_sz, q.nW, q.dW, q.db, q.accumulate, q.Cin, q.Cout);
        else slab_reduce2_body<true, 32>(blk, q.slabs, q.nslab, q.slab_sz, q.nW, q.dW, q.db, q.accumulate, q.Cin, q.Cout);
    } else {
        if (q.ew == 1) slab_reduce2_body<false, 1>(blk, q.slabs, q.nslab, q.slab_sz, q.nW, q.dW, q.db, q.accumulate, q.Cin, q.Cout);
        else if (q.ew == 4) slab_reduce2_body<false, 4>(blk, q.slabs, q.nslab, q.slab_sz, q.nW, q.dW, q.db, q.accumulate, q.Cin, q.Cout);
        else if (q.ew == 8) slab_reduce2_body<false, 8>(blk, q.slabs, q.nslab, q.slab_sz, q.nW, q.dW, q.db, q.accumulate, q.Cin, q.Cout);
        else slab_reduce2_body<false, 32>(blk, q.slabs, q.nslab, q.slab_sz, q.nW, q.dW, q.db, q.accumulate, q.Cin, q.Cout);
    }
}
#endif  // __HIPCC__
