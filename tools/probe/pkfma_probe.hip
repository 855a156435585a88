// Issue rate of v_fma_f32 / v_pk_fma_f32 (plain and with op_sel broadcast) per SIMD: one workgroup of W waves, each wave
// runs 8 independent accumulator chains; cycles per instruction per SIMD = cycles * SIMDs / (instructions * waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(1024) probe(float* out, long long* t, int iters) {
    f2 a[8], b = {out[0], out[1]}, c = {out[2], out[3]};
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)i, (float)threadIdx.x};
    __syncthreads();
    long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        }
    }
    long long c1 = clock64();
    __syncthreads();
    if (threadIdx.x == 0) t[0] = c1 - c0;
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    if (s == 12345.f) out[5] = s;
}
int main() {
    float* out; long long* t;
    hipMalloc(&out, 64); hipMemset(out, 0, 64);
    hipHostMalloc(&t, 64);
    const int iters = 4000;
    for (int mode = 0; mode < 5; ++mode)
        for (int waves : {1, 4, 8, 16}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) probe<0><<<1, 64 * waves>>>(out, t, iters);
                if (mode == 1) probe<1><<<1, 64 * waves>>>(out, t, iters);
                if (mode == 2) probe<2><<<1, 64 * waves>>>(out, t, iters);
                if (mode == 3) probe<3><<<1, 64 * waves>>>(out, t, iters);
                if (mode == 4) probe<4><<<1, 64 * waves>>>(out, t, iters);
                hipDeviceSynchronize();
            }
            const double per_wave = (double)t[0] / (iters * 8.0);
            const int per_simd = (waves + 3) / 4;
            printf("mode %d waves %2d: %.2f cycles per instruction per wave, %.2f per SIMD slot\n", mode, waves, per_wave, per_wave / per_simd);
        }
    return 0;
}
