// cu_mask_probe.hip -- which CUs does a stream created with hipExtStreamCreateWithCUMask use?  (round 4: CU partition for the
// aux-stream weight gradients).  Every workgroup records (XCC_ID, SE_ID, SH_ID, CU_ID) from the hardware registers; the host
// prints the set of (xcc, se, cu) per mask.   hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_probe tools/micro/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
#include <tuple>

__global__ void probe(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    // hold the slot a little so that the workgroups spread over every CU the queue may use
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {}
}

static void run(const char* name, hipStream_t s, unsigned* d, int nb) {
    hipMemsetAsync(d, 0xff, nb * 8, s);
    probe<<<nb, 64, 0, s>>>(d, 200000);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::set<std::tuple<int, int, int, int>> cus;
    int per_xcc[8] = {0};
    for (int b = 0; b < nb; b++) {
        unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        if (cus.insert({(int)xcc, se, sh, cu}).second) per_xcc[xcc & 7]++;
    }
    printf("%-28s distinct CUs %3zu  per XCC:", name, cus.size());
    for (int i = 0; i < 8; i++) printf(" %2d", per_xcc[i]);
    printf("\n");
}

int main() {
    unsigned* d;
    const int nb = 4096;
    hipMalloc(&d, nb * 8);
    hipStream_t s0;
    hipStreamCreate(&s0);
    run("plain stream", s0, d, nb);
    struct { const char* name; uint32_t m[8]; } cases[] = {
        {"bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 0..63", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0}},
        {"bits 0..127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}},
        {"bits 128..255", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}},
        {"every 8th bit (0,8,..)", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
        {"bits with (i%8)<2", {0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u}},
        {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
    };
    for (auto& c : cases) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, c.m);
        if (e != hipSuccess) { printf("%-28s create failed: %s\n", c.name, hipGetErrorString(e)); continue; }
        run(c.name, s, d, nb);
        hipStreamDestroy(s);
    }
    return 0;
}
