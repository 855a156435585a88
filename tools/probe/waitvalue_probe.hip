// Does hipStreamWaitValue32 work on plain device memory here, and how long after the producing kernel's store does the
// waiting stream's next kernel start?  Stream A: a kernel that stores 1 to *flag at its first instruction and then
// spins for ~200 us.  Stream B: hipStreamWaitValue32(flag >= 1), then a kernel that records the 100 MHz clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void producer(unsigned* flag, long long* t) {
    t[0] = wall_clock64();
    __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    while (wall_clock64() - t[0] < 20000) __builtin_amdgcn_s_sleep(10);
    t[1] = wall_clock64();
}
__global__ void consumer(long long* t) { t[2] = wall_clock64(); }
int main() {
    unsigned* flag; long long* t;
    hipStream_t a, b;
    hipStreamCreate(&a); hipStreamCreate(&b);
    hipHostMalloc(&t, 64);
    for (int mode = 0; mode < 2; ++mode) {
        hipError_t e = mode == 0 ? hipMalloc(&flag, 64) : hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory);
        printf("mode %d (%s): alloc %s\n", mode, mode ? "signal memory" : "plain hipMalloc", hipGetErrorString(e));
        if (e != hipSuccess) continue;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(flag, 0, 64); t[0] = t[1] = t[2] = 0;
            hipDeviceSynchronize();
            e = hipStreamWaitValue32(b, flag, 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
            consumer<<<1, 64, 0, b>>>(t);
            producer<<<1, 64, 0, a>>>(flag, t);
            hipError_t s = hipDeviceSynchronize();
            printf("  wait %s sync %s: consumer started %.1f us after the producer's store (producer ran %.1f us)\n",
                   hipGetErrorString(e), hipGetErrorString(s), (t[2] - t[0]) / 100.0, (t[1] - t[0]) / 100.0);
        }
    }
    return 0;
}
