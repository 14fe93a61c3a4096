// preproc.hip — the input pipeline's per-volume arithmetic on the GPU: what CombinedDataset.__getitem__
// (utils/dataloader.py:148-200) does in numpy on two DataLoader workers, as HBM-bound kernels on the device.
//   preprocess_ct   :111-117  clip to the abdominal window [-160, 240] HU, scale to [0, 1]
//   preprocess_mri  :128-144  z-score (population std), np.percentile([1, 99]) clip (linear interpolation), min-max to [0, 1]
//   label remaps    :162-181  AMOS table {0:0, 1:1, 2:3, 3:3, 6:2, else 0}; CHAOS value ranges
// The percentile needs EXACT order statistics: 4-pass 8-bit radix select on order-preserving keys of the z-scored values
// (integer histograms via integer atomics: deterministic), for the four ranks floor/ceil of q*(n-1) at once.
#include "ops.h"

namespace {
constexpr int BLK = 256;
constexpr int NPART = 1024;
constexpr int NSEL = 4;

struct MriWs {                     // layout of the caller-owned workspace
    double part[NPART];
    double sum[2];                 // sum, sum of squared deviations
    float stat[4];                 // mean, std + 1e-8 (float32 like numpy), low, high
    unsigned hist[NSEL][256];
    unsigned prefix[NSEL];         // key bits decided so far
    unsigned long long rank[NSEL]; // rank still to find inside the prefix bucket
    float sel[NSEL];
};

__device__ __forceinline__ unsigned okey(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ikey(unsigned k) {
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

__global__ __launch_bounds__(BLK) void ct_window_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, float lo, float hi) {
    float inv = hi - lo;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        float v = in[i];
        v = v < lo ? lo : (v > hi ? hi : v);
        out[i] = (v - lo) / inv;
    }
}

// pass 0: sum(x); pass 1: sum((x - mean)^2)   -> part[blockIdx.x]
__global__ __launch_bounds__(BLK) void mri_sum_kernel(const float* __restrict__ in, int64_t n, int pass, MriWs* ws) {
    __shared__ double red[BLK / 64];
    double mean = pass ? (double)ws->stat[0] : 0.0;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        double v = (double)in[i];
        if (pass) { v -= mean; v *= v; }
        s += v;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ws->part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(BLK) void mri_sum_finalize_kernel(int nblk, int64_t n, int pass, double q_lo, double q_hi, MriWs* ws) {
    __shared__ double red[BLK / 64];
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += BLK) s += ws->part[b];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = (red[0] + red[1]) + (red[2] + red[3]);
        if (pass == 0) ws->stat[0] = (float)(t / (double)n);                       // np.mean -> float32
        else {
            ws->stat[1] = (float)sqrt(t / (double)n) + 1e-8f;                       // np.std (population) + 1e-8 in float32
            // ranks of np.percentile(method='linear'): virtual index q/100*(n-1) -> floor and floor+1
            double v0 = q_lo / 100.0 * (double)(n - 1), v1 = q_hi / 100.0 * (double)(n - 1);
            unsigned long long k0 = (unsigned long long)floor(v0), k1 = (unsigned long long)floor(v1);
            ws->rank[0] = k0; ws->rank[1] = k0 + 1 < (unsigned long long)n ? k0 + 1 : k0;
            ws->rank[2] = k1; ws->rank[3] = k1 + 1 < (unsigned long long)n ? k1 + 1 : k1;
            for (int j = 0; j < NSEL; j++) ws->prefix[j] = 0u;
        }
    }
    for (int i = threadIdx.x; i < NSEL * 256; i += BLK) ws->hist[i / 256][i % 256] = 0u;
}

// histogram of key byte `pass` (most significant first) over the elements whose higher bytes equal the selection's prefix.
// Selections that still share their prefix (all four in pass 0; each percentile's floor / ceil pair almost always) share
// ONE histogram — that of the first selection with that prefix (its "leader") — so an element costs one LDS atomic per
// distinct prefix, not four on the same address.
__device__ __forceinline__ void sel_leaders(const unsigned* pf, int* lead) {
#pragma unroll
    for (int j = 0; j < NSEL; j++) {
        lead[j] = j;
#pragma unroll
        for (int i = NSEL - 1; i >= 0; i--)
            if (i < j && pf[i] == pf[j]) lead[j] = i;
    }
}
__global__ __launch_bounds__(BLK) void mri_hist_kernel(const float* __restrict__ in, int64_t n, int pass, MriWs* ws) {
    __shared__ unsigned h[NSEL][256];
    for (int i = threadIdx.x; i < NSEL * 256; i += BLK) h[i / 256][i % 256] = 0u;
    __syncthreads();
    float mean = ws->stat[0], sd = ws->stat[1];
    int shift = 24 - 8 * pass;
    unsigned mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    unsigned pf[NSEL];
    int lead[NSEL];
#pragma unroll
    for (int j = 0; j < NSEL; j++) pf[j] = ws->prefix[j];
    sel_leaders(pf, lead);
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        unsigned k = okey((in[i] - mean) / sd);
        unsigned byte = (k >> shift) & 255u;
#pragma unroll
        for (int j = 0; j < NSEL; j++)
            if (lead[j] == j && (k & mask) == pf[j]) atomicAdd(&h[j][byte], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NSEL * 256; i += BLK)
        if (h[i / 256][i % 256]) atomicAdd(&ws->hist[i / 256][i % 256], h[i / 256][i % 256]);
}
// pick the bucket holding the rank, descend; after the last pass the prefix IS the key of the order statistic.
// One wave per selection: lane l owns buckets 4l..4l+3, a wave prefix sum finds the bucket.  blockDim = 256.
__global__ __launch_bounds__(BLK) void mri_pick_kernel(int pass, double q_lo, double q_hi, int64_t n, MriWs* ws) {
    const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned pf[NSEL];
    int lead[NSEL];
#pragma unroll
    for (int q = 0; q < NSEL; q++) pf[q] = ws->prefix[q];
    sel_leaders(pf, lead);
    const unsigned long long r = ws->rank[j];
    const uint4 c = *(const uint4*)&ws->hist[lead[j]][4 * lane];
    const unsigned long long own = (unsigned long long)c.x + c.y + c.z + c.w;
    unsigned long long incl = own;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned long long up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const unsigned long long excl = incl - own;
    const unsigned long long total = __shfl(incl, 63);
    const bool mine = total > r ? (excl <= r && r < incl) : lane == 63;     // (total <= r cannot happen; stay in range)
    __syncthreads();                                                        // every read of hist / prefix / rank is done
    if (mine) {
        unsigned long long acc = excl;
        int b = 4 * lane;
        const unsigned cc[4] = {c.x, c.y, c.z, c.w};
        int t = 0;
        for (; t < 3; t++) {
            if (acc + cc[t] > r) break;
            acc += cc[t];
        }
        b += t;
        ws->rank[j] = r - acc;
        const unsigned np = pf[j] | ((unsigned)b << (24 - 8 * pass));
        ws->prefix[j] = np;
        if (pass == 3) ws->sel[j] = ikey(np);
    }
    for (int i = threadIdx.x; i < NSEL * 256; i += blockDim.x) ws->hist[i / 256][i % 256] = 0u;
    if (pass == 3) {
        __syncthreads();
        if (threadIdx.x == 0) {
            double v0 = q_lo / 100.0 * (double)(n - 1), v1 = q_hi / 100.0 * (double)(n - 1);
            double g0 = v0 - floor(v0), g1 = v1 - floor(v1);
            double a0 = (double)ikey(ws->prefix[0]), b0 = (double)ikey(ws->prefix[1]);
            double a1 = (double)ikey(ws->prefix[2]), b1 = (double)ikey(ws->prefix[3]);
            ws->stat[2] = (float)(a0 + (b0 - a0) * g0);          // np.percentile lerp
            ws->stat[3] = (float)(a1 + (b1 - a1) * g1);
        }
    }
}
__global__ __launch_bounds__(BLK) void mri_apply_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, const MriWs* ws) {
    float mean = ws->stat[0], sd = ws->stat[1], lo = ws->stat[2], hi = ws->stat[3];
    float den = hi - lo + 1e-8f;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        float z = (in[i] - mean) / sd;
        z = z < lo ? lo : (z > hi ? hi : z);
        out[i] = (z - lo) / den;
    }
}

__global__ __launch_bounds__(BLK) void remap_labels_kernel(const int64_t* __restrict__ in, int64_t* __restrict__ out, int64_t n, int kind) {
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        int64_t l = in[i], o = l;
        if (kind == 1) {            // AMOS (dataloader.py:43-49,163-167): spleen 1, kidneys 2|3 -> 3, liver 6 -> 2, everything else 0
            o = l == 1 ? 1 : (l == 2 || l == 3) ? 3 : l == 6 ? 2 : 0;
        } else if (kind == 2) {     // CHAOS (dataloader.py:52-58,168-181): grey-value ranges
            o = (l >= 55 && l <= 70) ? 2 : (l >= 110 && l <= 135) ? 3 : (l >= 175 && l <= 200) ? 3 : (l >= 240 && l <= 255) ? 1 : 0;
        }
        out[i] = o;
    }
}

inline int sgrid(int64_t n, int cap = 2048) {
    int64_t w = (n + BLK - 1) / BLK;
    return (int)(w < 1 ? 1 : (w > cap ? cap : w));
}
}  // namespace

extern "C" {

int mi3d_preprocess_ct(const float* in, float* out, int64_t n, float window_min, float window_max, void* stream) {
    MI3D_CHECK_ARG(in && out && n >= 1 && window_max > window_min, "mi3d_preprocess_ct: bad arguments");
    ct_window_kernel<<<sgrid(n), BLK, 0, (hipStream_t)stream>>>(in, out, n, window_min, window_max);
    MI3D_LAUNCH_CHECK();
    return 0;
}

size_t mi3d_preprocess_mri_workspace_bytes(void) { return sizeof(MriWs); }

int mi3d_preprocess_mri(const float* in, float* out, int64_t n, float p_low, float p_high, void* workspace, void* stream) {
    MI3D_CHECK_ARG(in && out && workspace && n >= 2 && p_low >= 0.f && p_high <= 100.f && p_low < p_high,
                   "mi3d_preprocess_mri: bad arguments");
    MI3D_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "mi3d_preprocess_mri: workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    MriWs* ws = (MriWs*)workspace;
    int nb = sgrid(n, NPART);
    for (int pass = 0; pass < 2; pass++) {
        mri_sum_kernel<<<nb, BLK, 0, s>>>(in, n, pass, ws);
        MI3D_LAUNCH_CHECK();
        mri_sum_finalize_kernel<<<1, BLK, 0, s>>>(nb, n, pass, (double)p_low, (double)p_high, ws);
        MI3D_LAUNCH_CHECK();
    }
    for (int pass = 0; pass < 4; pass++) {
        mri_hist_kernel<<<sgrid(n, 1024), BLK, 0, s>>>(in, n, pass, ws);
        MI3D_LAUNCH_CHECK();
        mri_pick_kernel<<<1, BLK, 0, s>>>(pass, (double)p_low, (double)p_high, n, ws);
        MI3D_LAUNCH_CHECK();
    }
    mri_apply_kernel<<<sgrid(n), BLK, 0, s>>>(in, out, n, ws);
    MI3D_LAUNCH_CHECK();
    return 0;
}

int mi3d_remap_labels(const int64_t* in, int64_t* out, int64_t n, int kind, void* stream) {
    MI3D_CHECK_ARG(in && out && n >= 1 && kind >= 0 && kind <= 2, "mi3d_remap_labels: bad arguments");
    remap_labels_kernel<<<sgrid(n), BLK, 0, (hipStream_t)stream>>>(in, out, n, kind);
    MI3D_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
