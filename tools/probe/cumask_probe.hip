// Probe: does hipExtStreamCreateWithCUMask confine workgroups, and how do mask bits map to XCDs / CUs on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <set>
__global__ void where(unsigned* out, int spin) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
    long t0 = clock64();
    while (clock64() - t0 < spin) {}
}
static void run(const char* label, hipStream_t s, unsigned* dev, int nblk) {
    hipMemsetAsync(dev, 0xff, nblk * 8, s);
    where<<<nblk, 256, 0, s>>>(dev, 200000);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * nblk);
    hipMemcpy(h.data(), dev, nblk * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int b = 0; b < nblk; ++b) {
        unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        per_xcc[xcc].insert(se * 32 + sh * 16 + cu);
    }
    printf("%s:", label);
    for (auto& kv : per_xcc) printf(" xcc%u:%zu", kv.first, kv.second.size());
    printf("\n");
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    unsigned* dev; hipMalloc(&dev, 4096 * 8);
    hipStream_t s0; hipStreamCreate(&s0);
    run("unmasked 2048 blocks", s0, dev, 2048);
    struct { const char* name; std::vector<uint32_t> mask; } cases[] = {
        {"bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 32..63", {0, 0xffffffffu, 0, 0, 0, 0, 0, 0}},
        {"bits i%8==0", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
        {"bits i%8==3", {0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u}},
        {"bits 0..223", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0}},
    };
    for (auto& c : cases) {
        hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)c.mask.size(), c.mask.data());
        if (e != hipSuccess) { printf("%s: create failed %d %s\n", c.name, (int)e, hipGetErrorString(e)); continue; }
        run(c.name, s, dev, 2048);
        hipStreamDestroy(s);
    }
    return 0;
}
