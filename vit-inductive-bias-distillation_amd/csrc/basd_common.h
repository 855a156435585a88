// Shared device helpers for the BASD loss kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#define BASD_OK 0
#define BASD_EINVAL (-1)
#define BASD_EUNSUPPORTED (-2)

#define BASD_DTYPE_F32 0
#define BASD_DTYPE_BF16 1

#define BASD_CHECK_ARG(cond) \
    do {                     \
        if (!(cond)) return BASD_EINVAL; \
    } while (0)

// Kernel launches never throw; a HIP error is returned as a positive status.
#define BASD_RETURN_LAST()                         \
    do {                                           \
        hipError_t e_ = hipGetLastError();         \
        return e_ == hipSuccess ? BASD_OK : (int)e_; \
    } while (0)

namespace basd {

constexpr int kWave = 64;

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(__hip_bfloat16 x) { return __bfloat162float(x); }
// Pointers read out of a device pointer table are generic, and loads through them are flat_load (they wait on the LDS
// counter as well): kernels that take such a table cast the table entry to global memory and load through these.
#define BASD_GLOBAL_AS __attribute__((address_space(1)))
__device__ __forceinline__ float ldg_f32(const BASD_GLOBAL_AS float* p) { return *p; }
__device__ __forceinline__ float ldg_f32(const BASD_GLOBAL_AS __hip_bfloat16* p) {
    return __uint_as_float((unsigned)*(const BASD_GLOBAL_AS unsigned short*)p << 16);
}

// Reduce over a power-of-two group of `width` adjacent lanes (width <= 64).
template <typename T>
__device__ __forceinline__ T group_sum(T v, int width) {
    for (int m = width >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, width);
    return v;
}

// ---- cross-lane sums without LDS: DPP inside a row of 16 lanes, lane-swaps across rows -------------
// (the __shfl_xor forms above go through ds_bpermute: ~100 cycles per step; these are plain VALU ops)
template <int CTRL>
__device__ __forceinline__ float dpp_get(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// Every lane of an (aligned) row of 16 ends with the row total; all 16 lanes must be active.
__device__ __forceinline__ float row16_allsum(float x) {
    x += dpp_get<0xB1>(x);    // quad_perm [1,0,3,2]
    x += dpp_get<0x4E>(x);    // quad_perm [2,3,0,1]
    x += dpp_get<0x141>(x);   // row_half_mirror
    x += dpp_get<0x140>(x);   // row_mirror
    return x;
}
// Every lane ends with the wave total; ALL 64 lanes must be active (the swaps read the partner's register).
__device__ __forceinline__ float wave64_allsum(float x) {
    x = row16_allsum(x);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

__device__ __forceinline__ float wave_sum(float v) { return wave64_allsum(v); }
__device__ __forceinline__ int wave_sum(int v) { return group_sum(v, kWave); }
__device__ __forceinline__ double wave_sum(double v) { return group_sum(v, kWave); }

__device__ __forceinline__ float wave_max(float v) {
    for (int m = 32; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
    return v;
}

// Block-wide sum through a caller-provided LDS scratch of >= 32 entries.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    T r = T(0);
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}

// Row pointer into a (batch, token, feature) strided view, row index m = b * n_tok + n.
struct TokView {
    const void* ptr;
    long sb, sn, sd;   // strides in elements
    int n_tok;         // tokens per batch element
};

__host__ __device__ __forceinline__ long tok_row_offset(const TokView& v, long m) {
    const long b = m / v.n_tok, n = m - b * v.n_tok;
    return b * v.sb + n * v.sn;
}

// Round-robin (circle method) pairing of n_even players: round r in [0, n_even-1),
// slot t in [0, n_even/2).  Every unordered pair meets exactly once per n_even-1 rounds.
__host__ __device__ __forceinline__ void rr_pair(int n_even, int r, int t, int& p, int& q) {
    const int m = n_even - 1;
    if (t == 0) { p = r; q = m; return; }
    p = (r + t) % m;
    q = (r - t + m) % m;
}

}  // namespace basd
