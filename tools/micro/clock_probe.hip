// clock_probe: what does s_memtime count?  A wave spins on dependent v_fma for a fixed instruction count while stamping
// s_memtime and s_memrealtime (100 MHz); the host times the kernel with events.  Also runs an MFMA-saturating variant on all
// CUs to read the core clock under matrix load.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void spin(unsigned long long* out, int iters, int mfma) {
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    float a = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    bf16x8 x = {1, 1, 1, 1, 1, 1, 1, 1};
    if (mfma) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 16; j++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, acc, 0, 0, 0);
        }
    } else {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 16; j++) a = fmaf(a, 1.0001f, 0.5f);
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (a == 12345.f || acc[0] == 12345.f) out[2] = 1;
}
int main() {
    unsigned long long* d; unsigned long long h[3];
    hipMalloc(&d, 24);
    for (int mode = 0; mode < 3; mode++) {
        int mfma = mode > 0, blocks = mode == 2 ? 2048 : 1, iters = 20000;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        spin<<<blocks, 256>>>(d, 100, mfma); hipDeviceSynchronize();
        hipEventRecord(a); spin<<<blocks, 256>>>(d, iters, mfma); hipEventRecord(b); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        double n = (double)iters * 16;
        printf("mode %d (%s, %d blocks): memtime %llu ticks, memrealtime %llu ticks (100 MHz -> %.1f us), event %.1f us; memtime rate %.1f MHz; %s/instr: %.2f memtime ticks\n",
               mode, mfma ? "mfma" : "v_fma", blocks, h[0], h[1], h[1] / 100.0, ms * 1e3, h[0] / (h[1] / 100.0), mfma ? "mfma" : "fma", h[0] / n);
    }
    return 0;
}
