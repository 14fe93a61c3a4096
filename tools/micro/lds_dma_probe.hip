#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__global__ void k(const float4* __restrict__ g, float4* out, unsigned bytes) {
    __shared__ __attribute__((aligned(16))) float4 buf[256];
    int wave = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t r = make_rsrc(g, bytes);
    unsigned voff = (threadIdx.x & 1) ? threadIdx.x * 16 : 0xffffff00u;   // odd lanes in bounds, even lanes out of bounds -> 0
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(buf + wave * 64), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = buf[threadIdx.x];
}
int main() {
    float4 *g, *o; hipMalloc(&g, 256 * 16); hipMalloc(&o, 256 * 16);
    float h[1024]; for (int i = 0; i < 1024; i++) h[i] = i + 1;
    hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
    k<<<1, 256>>>(g, o, 4096); 
    float r[1024]; hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; t++) for (int j = 0; j < 4; j++) { float want = (t & 1) ? h[t * 4 + j] : 0.f; if (r[t * 4 + j] != want) bad++; }
    printf("bad %d  r[4..7] = %g %g %g %g  r[0] = %g\n", bad, r[4], r[5], r[6], r[7], r[0]);
    return bad != 0;
}
