// Shader clock actually seen by a latency-bound kernel: s_memtime (shader clock) against the 100 MHz wall clock,
// for grids of 1 .. 1024 workgroups running a dependent FMA chain, an LDS round-trip chain and a barrier chain.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(1024) probe(float* out, long long* t, int iters, int mode) {
    __shared__ float buf[1024];
    const int tid = threadIdx.x;
    buf[tid] = (float)tid;
    __syncthreads();
    float x = out[0];
    long long c0 = clock64(), w0 = wall_clock64();
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) x = fmaf(x, 1.0000001f, 0.5f);
    } else if (mode == 1) {
        int idx = tid;
        for (int i = 0; i < iters; ++i) { idx = (int)buf[idx & 1023]; x += idx; }
    } else {
        for (int i = 0; i < iters; ++i) { buf[tid] = x; __syncthreads(); x += buf[(tid + 64) & 1023]; __syncthreads(); }
    }
    long long c1 = clock64(), w1 = wall_clock64();
    if (tid == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
    if (x == 12345.f) out[1] = x;
}
int main() {
    float* out; long long* t;
    hipMalloc(&out, 64); hipMemset(out, 0, 64);
    hipHostMalloc(&t, 64);
    int wr = 0; hipDeviceGetAttribute(&wr, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock rate %d kHz\n", wr);
    for (int mode = 0; mode < 3; ++mode)
        for (int grid : {1, 6, 256, 2048}) for (int threads : {64, 1024}) {
            const int iters = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                probe<<<grid, threads>>>(out, t, iters, mode);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("mode %d grid %4d threads %4d: %.3f ms, %.1f shader cycles/iter, %.1f ns/iter -> %.0f MHz\n", mode, grid, threads,
                       ms, (double)t[0] / iters, (double)t[1] / iters * 1e6 / wr, (double)t[0] / ((double)t[1] / wr * 1e-3) * 1e-6);
            }
        }
    return 0;
}
