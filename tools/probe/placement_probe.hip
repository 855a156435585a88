// Which resource keeps a one-workgroup kernel waiting for a CU while a chip-filling kernel of small workgroups runs?
// Background: 1024 workgroups x 128 threads x ~48 VGPRs x 24.4 KB LDS spinning for ~600 us (the footprint of the
// Procrustes Jacobi).  Foreground (second stream, 150 us later): one workgroup of T threads, V VGPRs (held live
// by asm), L bytes of LDS; it reports how long after the host-side launch its first instruction ran.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(128) background(float* out, long long ticks) {
    extern __shared__ float lds[];
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = threadIdx.x + i;
    lds[threadIdx.x] = x[3];
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
        for (int i = 0; i < 32; ++i) x[i] = fmaf(x[i], 1.0001f, lds[(threadIdx.x + i) & 127]);
    }
    float s = 0; for (int i = 0; i < 32; ++i) s += x[i];
    if (s == 1234.5f) out[0] = s;
}
template <int V>
__global__ void __launch_bounds__(512) foreground(long long* t, float* out) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) t[1] = wall_clock64();
    float x[V];
#pragma unroll
    for (int i = 0; i < V; ++i) x[i] = out[i & 7] + i;
#pragma unroll
    for (int i = 0; i < V; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[i]));      // V registers live at once
    float s = 0;
#pragma unroll
    for (int i = 0; i < V; ++i) s += x[i];
    if (s == 1234.5f) out[0] = s + lds[0];
}
__global__ void delay(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(20); }
__global__ void stamp(long long* t) { t[0] = wall_clock64(); }
template <int V>
void run(int threads, int lds, hipStream_t a, hipStream_t b, long long* t, float* out, bool with_bg) {
    float gap[4];
    for (int rep = 0; rep < 4; ++rep) {
        hipDeviceSynchronize();
        if (with_bg) background<<<1024, 128, 24400, a>>>(out, 60000);
        delay<<<1, 64, 0, b>>>(15000);                    // 150 us: the background has filled the chip by now
        stamp<<<1, 64, 0, b>>>(t);                        // the foreground stream's clock just before the launch
        foreground<V><<<1, threads, lds, b>>>(t, out);
        hipDeviceSynchronize();
        gap[rep] = (t[1] - t[0]) / 100.0f;
    }
    printf("%s threads %4d VGPRs >= %3d LDS %6d: first instruction %.0f / %.0f / %.0f us after the stamp kernel\n",
           with_bg ? "beside the background" : "alone                ", threads, V, lds, gap[1], gap[2], gap[3]);
}
int main() {
    hipStream_t a, b; hipStreamCreate(&a); hipStreamCreate(&b);
    long long* t; float* out;
    hipHostMalloc(&t, 64); hipMalloc(&out, 4096); hipMemset(out, 0, 4096);
    run<16>(512, 15360, a, b, t, out, false);
    run<16>(512, 15360, a, b, t, out, true);
    run<64>(512, 15360, a, b, t, out, true);
    run<96>(512, 15360, a, b, t, out, true);
    run<120>(512, 15360, a, b, t, out, true);
    run<180>(512, 15360, a, b, t, out, true);
    run<180>(512, 0, a, b, t, out, true);
    run<180>(256, 15360, a, b, t, out, true);
    run<180>(64, 15360, a, b, t, out, true);
    return 0;
}
