// Micro-benchmark: cost of a software grid barrier among co-resident workgroups on MI355X (8 XCDs) vs the cost of a
// dependent kernel boundary.  Each phase: every workgroup writes a value, barrier, reads its neighbour's value (checks
// cross-XCD visibility).   hipcc --offload-arch=gfx950 -O3 -o gridbar grid_barrier_bench.hip && ./gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* cnt, unsigned target, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE);          // agent scope by default for global atomics in HIP
        unsigned spins = 0;
        while (__atomic_load_n(cnt, __ATOMIC_ACQUIRE) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { *err = 1; break; }       // exit condition every wave reaches
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void barrier_kernel(unsigned* cnt, float* data, int phases, int* err, float* out) {
    int nb = gridDim.x, b = blockIdx.x;
    float acc = 0.f;
    for (int p = 0; p < phases; p++) {
        if (threadIdx.x == 0) data[(p & 1) * nb + b] = (float)(p + b);
        grid_barrier(cnt, (unsigned)(p + 1) * nb, err);
        float v = __builtin_nontemporal_load(&data[(p & 1) * nb + (b + 37) % nb]);
        if (threadIdx.x == 0) { if (v != (float)(p + (b + 37) % nb)) *err = 2; acc += v; }
    }
    if (threadIdx.x == 0) out[b] = acc;
}

__global__ void tiny_kernel(float* data, int n, int p) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] = data[(i + 37) % n] + p;
}

int main() {
    unsigned* cnt; float* data; int* err; float* out;
    CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&data, 2 * 1024 * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&out, 1024 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nb : {64, 128, 256, 512}) {
        for (int phases : {1, 101}) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipMemset(cnt, 0, 4)); CK(hipMemset(err, 0, 4));
                CK(hipEventRecord(e0));
                barrier_kernel<<<nb, 256>>>(cnt, data, phases, err, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            printf("grid %3d  phases %3d  %.2f us total  err %d\n", nb, phases, best * 1e3, herr);
        }
    }
    // dependent tiny kernels on one stream and inside a graph
    hipStream_t s; CK(hipStreamCreate(&s));
    int K = 100;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, s));
        for (int p = 0; p < K; p++) tiny_kernel<<<256, 256, 0, s>>>(data, 1024, p);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream: %d dependent tiny kernels %.2f us each\n", K, ms * 1e3 / K);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < K; p++) tiny_kernel<<<256, 256, 0, s>>>(data, 1024, p);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph: %d dependent tiny kernels %.2f us each\n", K, ms * 1e3 / K);
    }
    return 0;
}
