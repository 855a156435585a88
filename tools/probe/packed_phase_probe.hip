// Per-wave phase stamps of tridiag_packed_kernel (compile with -DBASD_TAIL_DBG): slots 0 pass begin, 1 pass end, 2 behind
// barrier A, 3 sums read, 4 w / next pivot row, 5 reflector published (wave 0 only), 6 behind barrier B.
#include "../../vit-inductive-bias-distillation_amd/csrc/tridiag.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 384;
    std::vector<float> h((size_t)n * n), x((size_t)n * n);
    srand(1);
    for (auto& v : x) v = (float)rand() / RAND_MAX - 0.5f;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { float s = 0; for (int k = 0; k < n; ++k) s += x[(size_t)i * n + k] * x[(size_t)j * n + k]; h[(size_t)i * n + j] = s; }
    float *a, *d, *e, *tau, *vh; void* work;
    (void)hipMalloc(&a, sizeof(float) * n * n); (void)hipMalloc(&d, 4 * n); (void)hipMalloc(&e, 4 * n); (void)hipMalloc(&tau, 4 * n);
    (void)hipMalloc(&vh, sizeof(float) * n * n); (void)hipMalloc(&work, basd_tridiag_workspace_bytes(n, 1));
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemcpy(a, h.data(), sizeof(float) * n * n, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        int rc = basd_tridiag(a, (long)n * n, n, 1, d, e, tau, vh, work, 0);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("rc %d  %.3f ms\n", rc, ms);
    }
#ifdef BASD_TAIL_DBG
    std::vector<long long> t(2 * 8 * 2 * 1024);
    (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(basd::g_tail_dbg), sizeof(long long) * 2 * 8 * 2 * 1024);
    for (int jl : {2, 64, 128, 200, 250, 300, 350}) {
        if (jl >= n - 2) continue;
        const long long t0 = t[(jl * 8 + 0) * 8 + 0];
        printf("step %3d  pass begin/end per wave (cycles after wave 0's begin):", jl);
        for (int w = 0; w < 8; ++w) printf("  %lld/%lld", t[(jl * 8 + 0) * 8 + w] - t0, t[(jl * 8 + 1) * 8 + w] - t0);
        printf("\n          wave 0: A %lld sums %lld scalar %lld reflector %lld B %lld | step %lld\n",
               t[(jl * 8 + 2) * 8] - t0, t[(jl * 8 + 3) * 8] - t0, t[(jl * 8 + 4) * 8] - t0, t[(jl * 8 + 5) * 8] - t0,
               t[(jl * 8 + 6) * 8] - t0, t[((jl + 1) * 8) * 8] - t0);
    }
#endif
    return 0;
}
