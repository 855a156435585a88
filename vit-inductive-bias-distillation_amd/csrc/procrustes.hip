// Attention-weighted Procrustes loss (reference src/losses/relational.py:5-50 together with the
// token-count interpolation of src/losses/combined.py:9-14 and the soft teacher-layer mixing of
// src/losses/layer_selector.py:110-112) on MI355X.
//
// The reference forms, per sample, C = S_w^T T_w (D_s x D_t, e.g. 384 x 2048) and takes its
// nuclear norm through a LAPACK SVD.  Here the same number is obtained from n x n matrices,
// n = N_t (teacher tokens, 49 / 196 / 144 at the BASELINE configs):
//
//   T_w = E T_c,  E = diag(sqrt w) I_interp  (N_s x N_t),  T_c = mixed teacher, weighted-centred
//   C   = S_w^T E T_c = A'^T T_c            with A' = E^T S_w   (N_t x D_s)
//   sigma(C) = sigma(L_a^T L_b)             with A'A'^T = L_a L_a^T,  T_c T_c^T = L_b L_b^T
//
// The two Gram matrices are accumulated in fp64 on the f64 MFMA (products of fp32 inputs are
// exact in fp64, so squaring the condition number costs nothing), factored in fp64, and only
// the n x n product M = L_a^T L_b is handed to the fp32 one-sided Jacobi solver (jacobi.hip),
// stacked on top of L_b so that Y = L_b V comes out of the same rotations.  Backward needs
//   d nuc / d A' = K' A',   K' = Y Sigma^+ Y^T   (n x n, symmetric PSD)
// which the finalize kernel emits; the student-token gradient is then a streaming pass.
#include "basd_common.h"
#include "../../include/basd_hip.h"

namespace basd {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// Token weights (relational.py:22-34) from the layer-mixed attention (layer_selector.py:112).
// grid = B, block = 256.
//   attn[l]: (B, H, A, A) with element strides (sb, sh, sq, sk); A = n_a + has_cls
//   atap*  : interpolation n_a -> n_s of the weights (relational.py:29-32), null when n_a == n_s
//   tap*   : interpolation n_t -> n_s of the teacher TOKENS (combined.py:9-14), null when n_t == n_s
//            (inside BASDLoss n_a == n_t; the stand-alone loss may get aligned tokens + raw attention)
//   omega  : (B, n_s) normalised weights on the student grid
//   omega_t: (B, n_t) = I_tok^T omega   (weights folded back onto the teacher token grid)
//   raw_out: (B, n_a) un-normalised attention-grid weights (kept for backward), nullable
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) token_weights_kernel(const void* const* __restrict__ attn_ptrs,
                                                            const float* __restrict__ mix, int L, long sb, long sh,
                                                            long sq, long sk, int H, int A, int has_cls, int n_a,
                                                            int n_t, int n_s, const int* __restrict__ atap0,
                                                            const int* __restrict__ atap1,
                                                            const float* __restrict__ alam,
                                                            const int* __restrict__ tap0,
                                                            const int* __restrict__ tap1,
                                                            const float* __restrict__ lam, float* __restrict__ omega,
                                                            float* __restrict__ omega_t, float* __restrict__ raw_out) {
    extern __shared__ float sm[];
    float* raw = sm;            // n_a
    float* w = sm + n_a;        // n_s
    __shared__ float red[32];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < n_a; j += blockDim.x) {
        float acc = 0.f;
        if (has_cls) {
            for (int h = 0; h < H; ++h) {
                float m = 0.f;
                for (int l = 0; l < L; ++l)
                    m += mix[l] * ldg_f32((const BASD_GLOBAL_AS T*)attn_ptrs[l] + (b * sb + h * sh + (long)(1 + j) * sk));
                acc += m;
            }
            acc /= (float)H;
        } else {
            for (int h = 0; h < H; ++h)
                for (int q = 0; q < A; ++q) {
                    float m = 0.f;
                    for (int l = 0; l < L; ++l)
                        m += mix[l] * ldg_f32((const BASD_GLOBAL_AS T*)attn_ptrs[l] + (b * sb + h * sh + q * sq + (long)j * sk));
                    acc += m;
                }
            acc /= (float)(H * A);
        }
        raw[j] = acc;
        if (raw_out) raw_out[(long)b * n_a + j] = acc;
    }
    __syncthreads();
    float part = 0.f;
    for (int n = tid; n < n_s; n += blockDim.x) {
        float v;
        if (atap0) {
            const float l1 = alam[n];
            v = (1.f - l1) * raw[atap0[n]] + l1 * raw[atap1[n]];
        } else {
            v = raw[n];
        }
        w[n] = v;
        part += v;
    }
    const float total = block_sum(part, red);
    for (int n = tid; n < n_s; n += blockDim.x) {
        const float v = w[n] / total;
        w[n] = v;
        omega[(long)b * n_s + n] = v;
    }
    __syncthreads();
    for (int j = tid; j < n_t; j += blockDim.x) {
        float acc = 0.f;
        if (tap0) {
            for (int n = 0; n < n_s; ++n) {
                const float l1 = lam[n];
                if (tap0[n] == j) acc += (1.f - l1) * w[n];
                if (tap1[n] == j) acc += l1 * w[n];
            }
        } else {
            acc = w[j];
        }
        omega_t[(long)b * n_t + j] = acc;
    }
}

// ---------------------------------------------------------------------------
// Student side: weighted mean, trace, and the down-projection A' = I^T diag(w) (X - 1 mu^T).
// grid = (ceil(D/64), B), block = 256 = 64 features x 4 row groups.   X: (B, N_s, D), feature stride 1.
//   range0/range1: for teacher token j, student rows [range0[j], range1[j]) touch it (null = identity grid)
//   tr_part: (B, nslab) partial traces, one per 64-feature slab (summed in fixed order by the finalize kernel)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) student_project_kernel(const T* __restrict__ X, long sb, long sn, int n_s,
                                                              int n_t, int D, const float* __restrict__ omega,
                                                              const int* __restrict__ tap0,
                                                              const int* __restrict__ tap1,
                                                              const float* __restrict__ lam,
                                                              const int* __restrict__ range0,
                                                              const int* __restrict__ range1,
                                                              float* __restrict__ mu_out, float* __restrict__ tr_part,
                                                              float* __restrict__ Ap) {
    extern __shared__ float sm[];
    float* w = sm;                 // n_s
    __shared__ float red[4][64];
    __shared__ float mu[64];
    __shared__ float scratch[32];
    const int b = blockIdx.y, d0 = blockIdx.x * 64, tid = threadIdx.x;
    const int cl = tid & 63, rg = tid >> 6, d = d0 + cl;
    const bool live = d < D;
    const T* Xb = X + (long)b * sb + d;
    for (int n = tid; n < n_s; n += 256) w[n] = omega[(long)b * n_s + n];
    __syncthreads();
    float acc = 0.f;
    if (live)
        for (int n = rg; n < n_s; n += 4) acc = fmaf(w[n], to_f32(Xb[(long)n * sn]), acc);
    red[rg][cl] = acc;
    __syncthreads();
    if (rg == 0) {
        const float m = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        mu[cl] = m;
        if (live) mu_out[(long)b * D + d] = m;
    }
    __syncthreads();
    const float m = mu[cl];
    // trace: sum_n w_n (x_n - mu)^2 over this slab
    float part = 0.f;
    if (live)
        for (int n = rg; n < n_s; n += 4) {
            const float c = to_f32(Xb[(long)n * sn]) - m;
            part = fmaf(w[n] * c, c, part);
        }
    const float tr = block_sum(part, scratch);
    if (tid == 0) tr_part[(long)b * gridDim.x + blockIdx.x] = tr;
    // A'[j, d] = sum_n I[n, j] w_n (x_n - mu)
    if (!live) return;
    float* Ab = Ap + (long)b * n_t * D + d;
    for (int j = rg; j < n_t; j += 4) {
        const int n0 = range0 ? range0[j] : j, n1 = range1 ? range1[j] : j + 1;
        float a = 0.f;
        for (int n = n0; n < n1; ++n) {
            float coef = 1.f;
            if (tap0) {
                const float l1 = lam[n];
                coef = (tap0[n] == j ? 1.f - l1 : 0.f) + (tap1[n] == j ? l1 : 0.f);
            }
            a = fmaf(coef * w[n], to_f32(Xb[(long)n * sn]) - m, a);
        }
        Ab[(long)j * D] = a;
    }
}

// Same, with the workgroup's (n_s x 64) slab of X staged in LDS by one coalesced pass (16-byte loads): the three
// sweeps (weighted mean, trace, projection) then read LDS instead of pulling 256-byte row segments through L2
// four or five times.  Needs 16-byte aligned rows and (n_s * 65) floats of LDS (n_s <= ~600).
__device__ __forceinline__ float4 ld4f(const float* p) { return *(const float4*)p; }
typedef float native_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned native_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld4f(const BASD_GLOBAL_AS float* p) {
    const native_f32x4 v = *(const BASD_GLOBAL_AS native_f32x4*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 ld4f(const BASD_GLOBAL_AS __hip_bfloat16* p) {
    const native_u32x2 v = *(const BASD_GLOBAL_AS native_u32x2*)p;
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ float4 ld4f(const __hip_bfloat16* p) {
    const uint2 v = *(const uint2*)p;
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
}

template <typename T>
__global__ void __launch_bounds__(256) student_project_lds_kernel(const T* __restrict__ X, long sb, long sn, int n_s,
                                                                  int n_t, int D, const float* __restrict__ omega,
                                                                  const int* __restrict__ tap0,
                                                                  const int* __restrict__ tap1,
                                                                  const float* __restrict__ lam,
                                                                  const int* __restrict__ range0,
                                                                  const int* __restrict__ range1,
                                                                  float* __restrict__ mu_out,
                                                                  float* __restrict__ tr_part, float* __restrict__ Ap) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                      // n_s x 64 slab, row-major
    float* w = sm + (long)n_s * 64;      // n_s
    __shared__ float red[4][64];
    __shared__ float mu[64];
    __shared__ float scratch[32];
    const int b = blockIdx.y, d0 = blockIdx.x * 64, tid = threadIdx.x;
    const int cl = tid & 63, rg = tid >> 6, d = d0 + cl;
    const bool live = d < D;
    const T* Xb = X + (long)b * sb + d0;
    for (int n = tid; n < n_s; n += 256) w[n] = omega[(long)b * n_s + n];
    for (int idx = tid; idx < n_s * 16; idx += 256) {
        const int row = idx >> 4, c4 = (idx & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d0 + c4 < D) v = ld4f(Xb + (long)row * sn + c4);        // D % 4 == 0: the quad is inside or outside
        *(float4*)(xs + row * 64 + c4) = v;
    }
    __syncthreads();
    float acc = 0.f;
    for (int n = rg; n < n_s; n += 4) acc = fmaf(w[n], xs[n * 64 + cl], acc);
    red[rg][cl] = acc;
    __syncthreads();
    if (rg == 0) {
        const float m = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        mu[cl] = m;
        if (live) mu_out[(long)b * D + d] = m;
    }
    __syncthreads();
    const float m = mu[cl];
    float part = 0.f;
    if (live)
        for (int n = rg; n < n_s; n += 4) {
            const float c = xs[n * 64 + cl] - m;
            part = fmaf(w[n] * c, c, part);
        }
    const float tr = block_sum(part, scratch);
    if (tid == 0) tr_part[(long)b * gridDim.x + blockIdx.x] = tr;
    if (!live) return;
    float* Ab = Ap + (long)b * n_t * D + d;
    for (int j = rg; j < n_t; j += 4) {
        const int n0 = range0 ? range0[j] : j, n1 = range1 ? range1[j] : j + 1;
        float a = 0.f;
        for (int n = n0; n < n1; ++n) {
            float coef = 1.f;
            if (tap0) {
                const float l1 = lam[n];
                coef = (tap0[n] == j ? 1.f - l1 : 0.f) + (tap1[n] == j ? l1 : 0.f);
            }
            a = fmaf(coef * w[n], xs[n * 64 + cl] - m, a);
        }
        Ab[(long)j * D] = a;
    }
}

// All E extraction layers in ONE launch, 16-byte LDS accesses.  grid = (ceil(D/32), B, E), block = 256 =
// 8 feature quads x 32 row groups.  The (n_s x 32) slab (128-byte rows, contiguous in LDS) is staged once by
// coalesced 16-byte loads; 16 consecutive lanes cover 256 contiguous LDS bytes, so the ds_write_b128 / ds_read_b128
// are conflict-free by construction.  Thread (fq, rg) owns features 4 fq .. 4 fq + 3 and rows rg, rg + 32, ...: every
// sweep reads float4s (a sixth of the LDS instructions of the 4-byte version above, which ran at 0.9 TB/s of HBM
// traffic, bound by LDS issue), and the interpolation coefficients sit in LDS too (they used to be three dependent
// global loads per row of the projection loop).  27 KB of LDS per workgroup at n_s = 196: five workgroups per CU.
// x_ptrs: device table of E base pointers (same strides); omega + e * omega_e_stride; outputs are (E, B, ...).
constexpr int SP_W = 32;          // slab width in features
template <typename T>
__global__ void __launch_bounds__(256) student_project_v4_kernel(const void* const* __restrict__ x_ptrs, long sb,
                                                                 long sn, int n_s, int n_t, int D,
                                                                 const float* __restrict__ omega, long omega_e_stride,
                                                                 const int* __restrict__ tap0,
                                                                 const int* __restrict__ tap1,
                                                                 const float* __restrict__ lam,
                                                                 const int* __restrict__ range0,
                                                                 const int* __restrict__ range1,
                                                                 float* __restrict__ mu_out,
                                                                 float* __restrict__ tr_part, float* __restrict__ Ap) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                        // n_s x 32 slab, row-major
    float* w = sm + (long)n_s * SP_W;      // n_s: token weights
    float* c0 = w + n_s;                   // n_s: weight of row n towards teacher token t0[n]   ((1 - lam) w)
    float* c1 = c0 + n_s;                  // n_s: ... towards t1[n]                               (lam w)
    int* t0 = (int*)(c1 + n_s);            // n_s
    int* t1 = t0 + n_s;                    // n_s
    __shared__ __attribute__((aligned(16))) float red[32][SP_W];
    __shared__ __attribute__((aligned(16))) float mu[SP_W];
    __shared__ float scratch[32];
    const int b = blockIdx.y, e = blockIdx.z, B = gridDim.y, d0 = blockIdx.x * SP_W, tid = threadIdx.x;
    const int fq = tid & 7, rg = tid >> 3, c4 = 4 * fq;
    const bool live = d0 + c4 < D;         // D % 4 == 0: the quad is inside or outside
    const BASD_GLOBAL_AS T* Xb = (const BASD_GLOBAL_AS T*)x_ptrs[e] + (long)b * sb + d0;
    const float* om = omega + (long)e * omega_e_stride + (long)b * n_s;
    for (int n = tid; n < n_s; n += 256) {
        const float wn = om[n];
        w[n] = wn;
        if (tap0) {
            const float l1 = lam[n];
            c0[n] = (1.f - l1) * wn;
            c1[n] = l1 * wn;
            t0[n] = tap0[n];
            t1[n] = tap1[n];
        }
    }
    for (int n = rg; n < n_s; n += 32) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) v = ld4f(Xb + (long)n * sn + c4);
        *(float4*)(xs + n * SP_W + c4) = v;
    }
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = rg; n < n_s; n += 32) {
        const float4 v = *(const float4*)(xs + n * SP_W + c4);
        const float wn = w[n];
        acc.x = fmaf(wn, v.x, acc.x); acc.y = fmaf(wn, v.y, acc.y);
        acc.z = fmaf(wn, v.z, acc.z); acc.w = fmaf(wn, v.w, acc.w);
    }
    *(float4*)(&red[rg][c4]) = acc;
    __syncthreads();
    if (tid < SP_W) {
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) m += red[q][tid];
        mu[tid] = m;
        if (d0 + tid < D) mu_out[((long)e * B + b) * D + d0 + tid] = m;
    }
    __syncthreads();
    const float4 m4 = *(const float4*)(mu + c4);
    // trace: sum_n w_n (x_n - mu)^2 over this slab
    float part = 0.f;
    if (live)
        for (int n = rg; n < n_s; n += 32) {
            const float4 v = *(const float4*)(xs + n * SP_W + c4);
            const float cx = v.x - m4.x, cy = v.y - m4.y, cz = v.z - m4.z, cw = v.w - m4.w;
            part = fmaf(w[n], fmaf(cx, cx, fmaf(cy, cy, fmaf(cz, cz, cw * cw))), part);
        }
    const float tr = block_sum(part, scratch);
    if (tid == 0) tr_part[((long)e * B + b) * gridDim.x + blockIdx.x] = tr;
    // A'[j, d] = sum_n I[n, j] w_n (x_n - mu)
    if (!live) return;
    float* Ab = Ap + ((long)e * B + b) * n_t * D + d0 + c4;
    for (int j = rg; j < n_t; j += 32) {
        const int n0 = range0 ? range0[j] : j, n1 = range1 ? range1[j] : j + 1;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int n = n0; n < n1; ++n) {
            const float cw = tap0 ? (t0[n] == j ? c0[n] : 0.f) + (t1[n] == j ? c1[n] : 0.f) : w[n];
            const float4 v = *(const float4*)(xs + n * SP_W + c4);
            a.x = fmaf(cw, v.x - m4.x, a.x); a.y = fmaf(cw, v.y - m4.y, a.y);
            a.z = fmaf(cw, v.z - m4.z, a.z); a.w = fmaf(cw, v.w - m4.w, a.w);
        }
        *(float4*)(Ab + (long)j * D) = a;
    }
}

// ---------------------------------------------------------------------------
// Teacher side: mix the L layers, (optionally) resample to the core grid, weighted-centre.
//   Tbar[b, r, :] = sum_l mix_l T_l[b, r, :]                               (layer_selector.py:110-111)
//   That[b, j, :] = (1-lam_j) Tbar[b, g0_j, :] + lam_j Tbar[b, g1_j, :]    (combined.py:9-14; only when the
//                   teacher grid is FINER than the student grid -- otherwise the interpolation is folded
//                   into the student side and That = Tbar)
//   Tc[b, j, :]   = That[b, j, :] - sum_j' omega_t[j'] That[b, j', :]      (relational.py:37,39)
// grid = (ceil(D/64), B), block = 256.  Handles feature-contiguous (sd == 1) and channel-major
// (sn == 1) teachers through an LDS tile of 64 features x n tokens.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) teacher_center_kernel(const void* const* __restrict__ tok_ptrs,
                                                             const float* __restrict__ mix, int L, long sb, long sn,
                                                             long sd, int n, int D, const int* __restrict__ g0,
                                                             const int* __restrict__ g1,
                                                             const float* __restrict__ glam,
                                                             const float* __restrict__ omega_t,
                                                             float* __restrict__ mu_out, float* __restrict__ Tc) {
    extern __shared__ float sm[];
    float* tile = sm;                  // n x 65
    float* wt = sm + (size_t)n * 65;   // n
    float* mu = wt + n;                // 64
    const int b = blockIdx.y, d0 = blockIdx.x * 64, tid = threadIdx.x;
    for (int j = tid; j < n; j += 256) wt[j] = omega_t[(long)b * n + j];
    const int total = n * 64;
    const bool feat_fast = sd == 1;
    for (int idx = tid; idx < total; idx += 256) {
        int j, dd;
        if (feat_fast) { j = idx >> 6; dd = idx & 63; }
        else { dd = idx / n; j = idx - dd * n; }
        const int d = d0 + dd;
        float v = 0.f;
        if (d < D) {
            const long off = b * sb + (long)d * sd;
            if (g0) {
                const long r0 = g0[j], r1 = g1[j];
                float a0 = 0.f, a1 = 0.f;
                for (int l = 0; l < L; ++l) {
                    const BASD_GLOBAL_AS T* p = (const BASD_GLOBAL_AS T*)tok_ptrs[l];
                    a0 = fmaf(mix[l], ldg_f32(p + (off + r0 * sn)), a0);
                    a1 = fmaf(mix[l], ldg_f32(p + (off + r1 * sn)), a1);
                }
                const float l1 = glam[j];
                v = (1.f - l1) * a0 + l1 * a1;
            } else {
                for (int l = 0; l < L; ++l)
                    v = fmaf(mix[l], ldg_f32((const BASD_GLOBAL_AS T*)tok_ptrs[l] + (off + (long)j * sn)), v);
            }
        }
        tile[j * 65 + dd] = v;
    }
    __syncthreads();
    if (tid < 64) {
        float acc = 0.f;
        for (int j = 0; j < n; ++j) acc = fmaf(wt[j], tile[j * 65 + tid], acc);
        mu[tid] = acc;
        if (d0 + tid < D) mu_out[(long)b * D + d0 + tid] = acc;
    }
    __syncthreads();
    float* out = Tc + (long)b * n * D;
    for (int idx = tid; idx < total; idx += 256) {
        const int j = idx >> 6, dd = idx & 63, d = d0 + dd;
        if (d < D) out[(long)j * D + d] = tile[j * 65 + dd] - mu[dd];
    }
}

// The same for G <= 4 groups of mixing weights in ONE pass over the teacher layers (multi-layer teachers: one group per
// extraction layer; the per-group kernel read all L layers once per group -- 4 x 2.5 GB per step at cfg-4).  Slabs of 16
// features so that the G tiles fit LDS together.  omega_t (G, B, n), mu_out (G, B, D), Tc (G, B, n, D).
constexpr int TCM_W = 16, TCM_G = 4;
template <typename T>
__global__ void __launch_bounds__(256) teacher_center_multi_kernel(
    const void* const* __restrict__ tok_ptrs, const float* __restrict__ mix, int L, int G, long sb, long sn, long sd,
    int n, int D, const int* __restrict__ g0, const int* __restrict__ g1, const float* __restrict__ glam,
    const float* __restrict__ omega_t, float* __restrict__ mu_out, float* __restrict__ Tc) {
    extern __shared__ float sm[];
    const int B = gridDim.y, b = blockIdx.y, d0 = blockIdx.x * TCM_W, tid = threadIdx.x;
    float* tile = sm;                                      // G x n x (TCM_W + 1)
    float* wt = sm + (size_t)G * n * (TCM_W + 1);          // G x n
    float* mu = wt + (size_t)G * n;                        // G x TCM_W
    for (int idx = tid; idx < G * n; idx += 256) {
        const int g = idx / n, j = idx - g * n;
        wt[idx] = omega_t[((long)g * B + b) * n + j];
    }
    const int total = n * TCM_W;
    const bool feat_fast = sd == 1;
    for (int idx = tid; idx < total; idx += 256) {
        int j, dd;
        if (feat_fast) { j = idx / TCM_W; dd = idx - j * TCM_W; }
        else { dd = idx / n; j = idx - dd * n; }
        const int d = d0 + dd;
        float v[TCM_G] = {0.f, 0.f, 0.f, 0.f};
        if (d < D) {
            const long off = b * sb + (long)d * sd;
            if (g0) {
                const long r0 = g0[j], r1 = g1[j];
                const float l1 = glam[j], l0 = 1.f - l1;
                for (int l = 0; l < L; ++l) {
                    const BASD_GLOBAL_AS T* p = (const BASD_GLOBAL_AS T*)tok_ptrs[l];
                    const float x = l0 * ldg_f32(p + (off + r0 * sn)) + l1 * ldg_f32(p + (off + r1 * sn));
#pragma unroll
                    for (int g = 0; g < TCM_G; ++g)
                        if (g < G) v[g] = fmaf(mix[g * L + l], x, v[g]);
                }
            } else {
                for (int l = 0; l < L; ++l) {
                    const float x = ldg_f32((const BASD_GLOBAL_AS T*)tok_ptrs[l] + (off + (long)j * sn));
#pragma unroll
                    for (int g = 0; g < TCM_G; ++g)
                        if (g < G) v[g] = fmaf(mix[g * L + l], x, v[g]);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < TCM_G; ++g)
            if (g < G) tile[((size_t)g * n + j) * (TCM_W + 1) + dd] = v[g];
    }
    __syncthreads();
    if (tid < TCM_W * G) {
        const int g = tid / TCM_W, dd = tid - g * TCM_W;
        float acc = 0.f;
        for (int j = 0; j < n; ++j) acc = fmaf(wt[g * n + j], tile[((size_t)g * n + j) * (TCM_W + 1) + dd], acc);
        mu[tid] = acc;
        if (d0 + dd < D) mu_out[((long)g * B + b) * D + d0 + dd] = acc;
    }
    __syncthreads();
    for (int g = 0; g < G; ++g) {
        float* out = Tc + ((long)g * B + b) * n * D;
        for (int idx = tid; idx < total; idx += 256) {
            const int j = idx / TCM_W, dd = idx - j * TCM_W, d = d0 + dd;
            if (d < D) out[(long)j * D + d] = tile[((size_t)g * n + j) * (TCM_W + 1) + dd] - mu[g * TCM_W + dd];
        }
    }
}

// Streaming form of the same (row-major teacher tokens, many layers: cfg-4 mixes 24 ViT-L layers for 4 groups): no LDS
// tile, so full rows are read -- pass 1 writes the mixed tokens uncentred and leaves, per chunk of TCS_ROWS rows, the
// weighted column sums (each thread owns its feature columns: no cross-thread reduction); pass 2 folds the chunk sums
// in a fixed order into the weighted mean and subtracts it in place.  2.5 GB read once + 0.4 GB written + 0.8 GB for
// the centring pass at cfg-4, against 64-byte row segments through a 16-feature LDS tile in the kernel above.
// pass 1: grid = (chunks, B), block = 256.  pass 2: grid = (ceil(D / 256), chunks, G * B), block = 256.
constexpr int TCS_ROWS = 16;
template <typename T>
__global__ void __launch_bounds__(256) teacher_mix_stream_kernel(
    const void* const* __restrict__ tok_ptrs, const float* __restrict__ mix, int L, int G, long sb, long sn, long sd, int n,
    int D, const int* __restrict__ g0, const int* __restrict__ g1, const float* __restrict__ glam,
    const float* __restrict__ omega_t, float* __restrict__ Tc, float* __restrict__ chunk_sum) {
    const int chunk = blockIdx.x, b = blockIdx.y, B = gridDim.y, tid = threadIdx.x;
    const int j_lo = chunk * TCS_ROWS, j_hi = j_lo + TCS_ROWS < n ? j_lo + TCS_ROWS : n;
    for (int d = tid; d < D; d += 256) {
        float musum[TCM_G] = {0.f, 0.f, 0.f, 0.f};
        for (int j = j_lo; j < j_hi; ++j) {
            long o0 = b * sb + (long)j * sn + (long)d * sd, o1 = o0;
            float l1 = 0.f;
            if (g0) {
                o0 = b * sb + (long)g0[j] * sn + (long)d * sd;
                o1 = b * sb + (long)g1[j] * sn + (long)d * sd;
                l1 = glam[j];
            }
            float v[TCM_G] = {0.f, 0.f, 0.f, 0.f};
            for (int l = 0; l < L; ++l) {
                const BASD_GLOBAL_AS T* p = (const BASD_GLOBAL_AS T*)tok_ptrs[l];
                float x = ldg_f32(p + o0);
                if (g0) x = (1.f - l1) * x + l1 * ldg_f32(p + o1);
#pragma unroll
                for (int g = 0; g < TCM_G; ++g)
                    if (g < G) v[g] = fmaf(mix[g * L + l], x, v[g]);
            }
#pragma unroll
            for (int g = 0; g < TCM_G; ++g)
                if (g < G) {
                    Tc[(((long)g * B + b) * n + j) * D + d] = v[g];
                    musum[g] = fmaf(omega_t[((long)g * B + b) * n + j], v[g], musum[g]);
                }
        }
#pragma unroll
        for (int g = 0; g < TCM_G; ++g)
            if (g < G) chunk_sum[(((long)chunk * G + g) * B + b) * D + d] = musum[g];
    }
}

__global__ void __launch_bounds__(256) teacher_center_apply_kernel(const float* __restrict__ chunk_sum, int chunks, int n,
                                                                   int D, float* __restrict__ mu_out,
                                                                   float* __restrict__ Tc) {
    const int d = blockIdx.x * 256 + threadIdx.x, chunk = blockIdx.y;
    const long gb = blockIdx.z, GB = gridDim.z;
    if (d >= D) return;
    float mu = 0.f;
    for (int c = 0; c < chunks; ++c) mu += chunk_sum[((long)c * GB + gb) * D + d];
    if (chunk == 0) mu_out[gb * D + d] = mu;
    const int j_lo = chunk * TCS_ROWS, j_hi = j_lo + TCS_ROWS < n ? j_lo + TCS_ROWS : n;
    float* out = Tc + gb * n * D + d;
    for (int j = j_lo; j < j_hi; ++j) out[(long)j * D] -= mu;
}

// ---------------------------------------------------------------------------
// Batched Gram in fp64 on the f64 MFMA: G[b] = P[b] P[b]^T, P: (n x D) fp32 row-major.
// grid = batch, block = 64 * waves.  Lower tiles (16 x 16) are dealt round-robin to waves.
// v_mfma_f64_16x16x4_f64: lane l holds A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15];
// D: col = l & 15, row = (l >> 4) + 4 * reg.
// ---------------------------------------------------------------------------
constexpr int G64_BK = 32, G64_LD = 36, G64_TPW = 8;

// blockIdx.y = split s of gridDim.y over the contraction index: the split handles D-chunks s, s + S, ... and writes its
// partial Gram to G + s * g_split_stride (gram_f64_reduce_kernel folds them in a fixed order).
__global__ void __launch_bounds__(1024) gram_f64_kernel(const float* __restrict__ P, long p_batch_stride, int n, int D,
                                                        double* __restrict__ G, long g_batch_stride,
                                                        long g_split_stride) {
    extern __shared__ __attribute__((aligned(16))) float tile[];  // (nt*16) x G64_LD
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    G += (long)blockIdx.y * g_split_stride;
    const int nt = (n + 15) / 16, ntiles = nt * (nt + 1) / 2;
    const float* Pb = P + (long)b * p_batch_stride;
    f64x4 acc[G64_TPW];
#pragma unroll
    for (int t = 0; t < G64_TPW; ++t) acc[t] = f64x4{0., 0., 0., 0.};
    // tile list of this wave: linear lower-triangular index -> (ti, tj), ti >= tj
    int ti[G64_TPW], tj[G64_TPW];
#pragma unroll
    for (int t = 0; t < G64_TPW; ++t) {
        const int lin = wave + (t + (int)blockIdx.z * G64_TPW) * nw;      // blockIdx.z: chunk of the tile list (n > 240)
        int r = 0;
        if (lin < ntiles) {
            r = (int)((sqrtf(8.f * lin + 1.f) - 1.f) * 0.5f);
            while ((r + 1) * (r + 2) / 2 <= lin) ++r;
            while (r * (r + 1) / 2 > lin) --r;
            ti[t] = r;
            tj[t] = lin - r * (r + 1) / 2;
        } else {
            ti[t] = -1;
            tj[t] = -1;
        }
    }
    const int rows_pad = nt * 16;
    const int i16 = lane & 15, kq = lane >> 4;
    for (int k0 = blockIdx.y * G64_BK; k0 < D; k0 += G64_BK * gridDim.y) {
        __syncthreads();
        for (int idx = tid; idx < rows_pad * (G64_BK / 4); idx += blockDim.x) {
            const int r = idx / (G64_BK / 4), c4 = (idx - r * (G64_BK / 4)) * 4, k = k0 + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < n) {
                const float* p = Pb + (long)r * D + k;
                if (k + 3 < D) {
                    v = *(const float4*)p;
                } else {
                    if (k < D) v.x = p[0];
                    if (k + 1 < D) v.y = p[1];
                    if (k + 2 < D) v.z = p[2];
                }
            }
            *(float4*)(tile + r * G64_LD + c4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G64_BK / 16; ++g) {
#pragma unroll
            for (int t = 0; t < G64_TPW; ++t) {
                if (ti[t] < 0) continue;
                const float4 a = *(const float4*)(tile + (ti[t] * 16 + i16) * G64_LD + g * 16 + 4 * kq);
                const float4 bb = *(const float4*)(tile + (tj[t] * 16 + i16) * G64_LD + g * 16 + 4 * kq);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a.x, (double)bb.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a.y, (double)bb.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a.z, (double)bb.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a.w, (double)bb.w, acc[t], 0, 0, 0);
            }
        }
    }
    double* Gb = G + (long)b * g_batch_stride;
#pragma unroll
    for (int t = 0; t < G64_TPW; ++t) {
        if (ti[t] < 0) continue;
        const int col = tj[t] * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = ti[t] * 16 + (lane >> 4) + 4 * r;
            if (row < n && col < n) {
                Gb[(long)row * n + col] = acc[t][r];
                Gb[(long)col * n + row] = acc[t][r];
            }
        }
    }
}

// G[i] = sum_s slabs[s][i]  (fixed order: deterministic).  grid = ceil(count/256), block = 256.
__global__ void __launch_bounds__(256) gram_f64_reduce_kernel(const double* __restrict__ slabs, long split_stride,
                                                              int splits, long count, double* __restrict__ G) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double acc = 0.;
    for (int s2 = 0; s2 < splits; ++s2) acc += slabs[(long)s2 * split_stride + i];
    G[i] = acc;
}

// Launch shape of gram_f64_kernel: up to 16 waves x G64_TPW tiles per workgroup; past that (n > 240) the tile list is
// cut into chunks, one workgroup each (every chunk stages all rows of a D-chunk: n <= ~1100 by LDS).
static bool gram_f64_shape(int n, int* waves, int* chunks, size_t* lds) {
    const int nt = (n + 15) / 16, ntiles = nt * (nt + 1) / 2;
    int w = (ntiles + G64_TPW - 1) / G64_TPW;
    if (w < 4) w = ntiles < 4 ? ntiles : 4;
    int c = 1;
    if (w > 16) {
        c = (ntiles + 16 * G64_TPW - 1) / (16 * G64_TPW);
        w = 16;
    }
    *waves = w;
    *chunks = c;
    *lds = sizeof(float) * (size_t)nt * 16 * G64_LD;
    if (*lds > 156 * 1024) return false;
    if (*lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)gram_f64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    return true;
}

// ---------------------------------------------------------------------------
// fp64 Cholesky of symmetric PSD matrices (possibly singular): G = L L^T, packed in LDS.
// grid = batch, block = 256..1024.  A pivot below n*1e-15*max_diag zeroes its column.
// L is written full (n x n row-major, zeros above the diagonal).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int pk(int i, int j, int n) { return j * n - (j * (j - 1)) / 2 + (i - j); }  // i >= j

__global__ void __launch_bounds__(1024) chol_f64_kernel(const double* __restrict__ G, long g_batch_stride, int n,
                                                        double* __restrict__ Lout, long l_batch_stride) {
    extern __shared__ __attribute__((aligned(16))) double a[];  // packed lower, column-major; then n col buffer
    double* col = a + (size_t)n * (n + 1) / 2;
    __shared__ double sh_piv;
    __shared__ double red[32];
    const int m = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    const double* Gm = G + (long)m * g_batch_stride;
    double dmax = 0.;
    for (int j = wave; j < n; j += nw)
        for (int i = j + lane; i < n; i += 64) {
            const double v = Gm[(long)i * n + j];
            a[pk(i, j, n)] = v;
            if (i == j) dmax = fmax(dmax, v);
        }
    // block max of the diagonal
    for (int s = 32; s > 0; s >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, s, 64));
    if (lane == 0) red[wave] = dmax;
    __syncthreads();
    dmax = 0.;
    for (int i = 0; i < nw; ++i) dmax = fmax(dmax, red[i]);
    const double thr = dmax * (double)n * 1e-15;
    for (int j = 0; j < n; ++j) {
        if (tid == 0) {
            const double d = a[pk(j, j, n)];
            sh_piv = d > thr ? sqrt(d) : 0.;
        }
        __syncthreads();
        const double piv = sh_piv;
        const double inv = piv > 0. ? 1. / piv : 0.;
        for (int i = j + tid; i < n; i += nthr) {
            const double v = (i == j) ? piv : a[pk(i, j, n)] * inv;
            a[pk(i, j, n)] = v;
            col[i] = v;
        }
        __syncthreads();
        if (piv > 0.) {
            for (int k = j + 1 + wave; k < n; k += nw) {
                const double ck = col[k];
                for (int i = k + lane; i < n; i += 64) a[pk(i, k, n)] -= col[i] * ck;
            }
        }
        // next iteration's first barrier orders these updates before the pivot read
        __syncthreads();
    }
    double* Lm = Lout + (long)m * l_batch_stride;
    for (int idx = tid; idx < n * n; idx += nthr) {
        const int i = idx / n, j = idx - i * n;
        Lm[idx] = i >= j ? a[pk(i, j, n)] : 0.;
    }
}

// The same factorisation for matrices past LDS (n > ~200, e.g. a ViT teacher at 384 x 384: 576 tokens): left-looking by
// panels of CB_NB columns.  L lives in global memory (Lout, written and re-read by this workgroup only); the current
// panel -- rows j0..n-1 of CB_NB columns -- is accumulated in registers (thread (row group, column): rows rg + 64 r),
// parked in LDS and factored there column by column with the pivot rule of chol_f64_kernel.
// grid = batch, block = 1024; n <= 64 * CB_MAXR.
constexpr int CB_NB = 16, CB_KC = 64, CB_MAXR = 16;
__global__ void __launch_bounds__(1024) chol_f64_blocked_kernel(const double* __restrict__ G, long g_batch_stride, int n,
                                                                double* __restrict__ Lout, long l_batch_stride) {
    extern __shared__ __attribute__((aligned(16))) double sm64[];
    double* panel = sm64;                               // (n - j0) x CB_NB, row-major
    double* lj = panel + (size_t)n * CB_NB;             // CB_NB x (CB_KC + 1): rows j0.. of L, columns k0..
    __shared__ double red[16];
    __shared__ double lrow[CB_NB];
    __shared__ double sh_piv;
    const int m = blockIdx.x, tid = threadIdx.x, c = tid & (CB_NB - 1), rg = tid / CB_NB;
    const double* Gm = G + (long)m * g_batch_stride;
    double* Lm = Lout + (long)m * l_batch_stride;
    double dmax = 0.;
    for (int i = tid; i < n; i += 1024) dmax = fmax(dmax, Gm[(long)i * n + i]);
    for (int s2 = 32; s2 > 0; s2 >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, s2, 64));
    if ((tid & 63) == 0) red[tid >> 6] = dmax;
    __syncthreads();
    dmax = 0.;
    for (int i = 0; i < 16; ++i) dmax = fmax(dmax, red[i]);
    const double thr = dmax * (double)n * 1e-15;
    for (long idx = tid; idx < (long)n * n; idx += 1024) {          // strictly upper triangle
        const int i = (int)(idx / n), j = (int)(idx - (long)i * n);
        if (j > i) Lm[idx] = 0.;
    }
    for (int j0 = 0; j0 < n; j0 += CB_NB) {
        const int nb = n - j0 < CB_NB ? n - j0 : CB_NB, rows = n - j0;
        double acc[CB_MAXR];
#pragma unroll
        for (int r = 0; r < CB_MAXR; ++r) {
            const int i = j0 + rg + 64 * r;
            acc[r] = (i < n && c < nb) ? Gm[(long)i * n + j0 + c] : 0.;
        }
        for (int k0 = 0; k0 < j0; k0 += CB_KC) {
            const int kc = j0 - k0 < CB_KC ? j0 - k0 : CB_KC;
            __syncthreads();
            for (int idx = tid; idx < CB_NB * CB_KC; idx += 1024) {
                const int cc = idx / CB_KC, k = idx - cc * CB_KC;
                lj[cc * (CB_KC + 1) + k] = (cc < nb && k < kc) ? Lm[(long)(j0 + cc) * n + k0 + k] : 0.;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < CB_MAXR; ++r) {
                const int i = j0 + rg + 64 * r;
                if (i < n) {
                    const double* Li = Lm + (long)i * n + k0;
                    double a2 = acc[r];
                    for (int k = 0; k < kc; ++k) a2 = fma(-Li[k], lj[c * (CB_KC + 1) + k], a2);
                    acc[r] = a2;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CB_MAXR; ++r) {
            const int ii = rg + 64 * r;
            if (ii < rows) panel[ii * CB_NB + c] = acc[r];
        }
        __syncthreads();
        for (int jj = 0; jj < nb; ++jj) {
            if (tid == 0) {
                const double d = panel[jj * CB_NB + jj];
                sh_piv = d > thr ? sqrt(d) : 0.;
            }
            __syncthreads();
            const double piv = sh_piv;
            const double inv = piv > 0. ? 1. / piv : 0.;
            for (int ii = jj + tid; ii < rows; ii += 1024) {
                const double v = (ii == jj) ? piv : panel[ii * CB_NB + jj] * inv;
                panel[ii * CB_NB + jj] = v;
                if (ii < CB_NB) lrow[ii] = v;
            }
            __syncthreads();
            if (piv > 0.) {
                for (int idx = tid; idx < (rows - jj - 1) * CB_NB; idx += 1024) {
                    const int ii = jj + 1 + idx / CB_NB, cc = idx & (CB_NB - 1);
                    if (cc > jj && cc < nb && cc <= ii) panel[ii * CB_NB + cc] -= panel[ii * CB_NB + jj] * lrow[cc];
                }
            }
            __syncthreads();
        }
        for (int idx = tid; idx < rows * CB_NB; idx += 1024) {
            const int ii = idx / CB_NB, cc = idx & (CB_NB - 1);
            if (cc < nb) Lm[(long)(j0 + ii) * n + j0 + cc] = ii >= cc ? panel[idx] : 0.;
        }
        __syncthreads();    // the rows just written are read back (by other threads) for the next panel
    }
}

// ---------------------------------------------------------------------------
// W[b] = [ L_a^T L_b ; L_b ]   (2n x n, column-major fp32, leading dimension 2n), fp64 accumulate.
// stack_product_kernel: grid = batch, block = 256, both factors in LDS (n <= 98).
// stack_product_global_kernel: grid = (ceil(n/32), ceil(n/32), batch), tiles of the factors through LDS (larger n).
// ---------------------------------------------------------------------------
// 32 x 32 output tiles, both factors staged through LDS in fp64; the zeros stored above the diagonals make
// k >= 32 max(tile_i, tile_j) the exact start of the sum.  (The first version read the factors element-wise from L2 inside
// the k loop: 5.4 ms per step at cfg-4, 17 ms for 64 cores of 576 tokens.)
__global__ void __launch_bounds__(256) stack_product_global_kernel(const double* __restrict__ La,
                                                                   const double* __restrict__ Lb,
                                                                   long l_batch_stride, int n, float* __restrict__ W,
                                                                   long w_batch_stride, int lb_period) {
    __shared__ double ta[32][33], tb[32][33];     // [k][i], [k][j]
    const int b = blockIdx.z, i0 = blockIdx.x * 32, j0 = blockIdx.y * 32, tid = threadIdx.x;
    const double* A = La + (long)b * l_batch_stride;
    const double* B = Lb + (long)(b % lb_period) * l_batch_stride;
    const int tx = tid & 31, ty = tid >> 5;       // tx: i within the tile, ty + 8 r: j within the tile
    double acc[4] = {0., 0., 0., 0.};
    for (int k0 = i0 > j0 ? i0 : j0; k0 < n; k0 += 32) {
        __syncthreads();
        for (int idx = tid; idx < 1024; idx += 256) {
            const int kk = idx >> 5, c = idx & 31, k = k0 + kk;
            ta[kk][c] = (k < n && i0 + c < n) ? A[(long)k * n + i0 + c] : 0.;
            tb[kk][c] = (k < n && j0 + c < n) ? B[(long)k * n + j0 + c] : 0.;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            const double av = ta[kk][tx];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fma(av, tb[kk][ty + 8 * r], acc[r]);
        }
    }
    float* Wb = W + (long)b * w_batch_stride;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + tx, j = j0 + ty + 8 * r;
        if (i < n && j < n) {
            Wb[(long)j * 2 * n + i] = (float)acc[r];
            Wb[(long)j * 2 * n + n + i] = (float)B[(long)i * n + j];
        }
    }
}

__global__ void __launch_bounds__(256) stack_product_kernel(const double* __restrict__ La,
                                                            const double* __restrict__ Lb, long l_batch_stride,
                                                            int n, float* __restrict__ W, long w_batch_stride,
                                                            int lb_period) {
    // One workgroup per matrix: both factors staged in LDS (the first version read them element-wise from L2 inside
    // the k loop: 0.13 ms at cfg-2 for 0.24 GFLOP).  sa[k * n + i] = L_a[k][i], sb[k * n + j] = L_b[k][j].
    extern __shared__ __attribute__((aligned(16))) double sd[];
    double* sa = sd;
    double* sb = sd + (size_t)n * n;
    const int b = blockIdx.x, tid = threadIdx.x;
    const double* A = La + (long)b * l_batch_stride;
    const double* B = Lb + (long)(b % lb_period) * l_batch_stride;   // one teacher factor for all extraction layers
    for (int idx = tid; idx < n * n; idx += 256) {
        sa[idx] = A[idx];
        sb[idx] = B[idx];
    }
    __syncthreads();
    float* Wb = W + (long)b * w_batch_stride;
    for (int idx = tid; idx < n * n; idx += 256) {
        const int j = idx / n, i = idx - j * n;      // consecutive threads: consecutive i (rows of the output column j)
        double acc = 0.;
        for (int k = (i > j ? i : j); k < n; ++k) acc = fma(sa[k * n + i], sb[k * n + j], acc);
        Wb[(long)j * 2 * n + i] = (float)acc;
        Wb[(long)j * 2 * n + n + i] = (float)sb[i * n + j];
    }
}

// M = L_a^T L_b alone, row-major and compact (n x n at W + b * w_batch_stride): as a column-major matrix that is M^T, and
// one-sided Jacobi on M^T leaves V Sigma in its columns (the right singular vectors of M, which is what Y = L_b V
// needs) -- no riding rows.  grid = (ceil(n/32), ceil(n/32), batch), block = 256.
__global__ void __launch_bounds__(256) stack_product_t_kernel(const double* __restrict__ La, const double* __restrict__ Lb,
                                                              long l_batch_stride, int n, float* __restrict__ W,
                                                              long w_batch_stride, int lb_period) {
    __shared__ double ta[32][33], tb[32][33];     // [k][i], [k][j]
    const int b = blockIdx.z, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32, tid = threadIdx.x;
    const double* A = La + (long)b * l_batch_stride;
    const double* B = Lb + (long)(b % lb_period) * l_batch_stride;
    const int tx = tid & 31, ty = tid >> 5;       // tx: j within the tile, ty + 8 r: i within the tile
    double acc[4] = {0., 0., 0., 0.};
    for (int k0 = i0 > j0 ? i0 : j0; k0 < n; k0 += 32) {
        __syncthreads();
        for (int idx = tid; idx < 1024; idx += 256) {
            const int kk = idx >> 5, c = idx & 31, k = k0 + kk;
            ta[kk][c] = (k < n && i0 + c < n) ? A[(long)k * n + i0 + c] : 0.;
            tb[kk][c] = (k < n && j0 + c < n) ? B[(long)k * n + j0 + c] : 0.;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            const double bv = tb[kk][tx];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fma(ta[kk][ty + 8 * r], bv, acc[r]);
        }
    }
    float* Wb = W + (long)b * w_batch_stride;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 8 * r, j = j0 + tx;
        if (i < n && j < n) Wb[(long)i * n + j] = (float)acc[r];
    }
}

// After the Jacobi on M^T (columns c of X = V_c sigma_c, compact n x n at W + b * w_batch_stride):
//   Z[c][a] = sigma_c^-1.5 sum_{r <= a} L_b[a][r] X[r][c]  = (L_b V)[a][c] sigma_c^-1/2      (0 for truncated sigma)
//   K'[a][b] = sum_c Z[c][a] Z[c][b]                          = Y Sigma^+ Y^T
// 32 x 32 tiles; Z goes to z (batch stride z_batch_stride).  grid = (ceil(n/32), ceil(n/32), batch), block = 256.
__global__ void __launch_bounds__(256) kprime_z_kernel(const float* __restrict__ W, long w_batch_stride,
                                                       const float* __restrict__ sigma, int n,
                                                       const double* __restrict__ Lb, long l_batch_stride, int lb_period,
                                                       float* __restrict__ Z, long z_batch_stride) {
    __shared__ float lt[32][33], wt[32][33];      // lt[a][r], wt[c][r]
    __shared__ float red[4];
    const int b = blockIdx.z, a0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tid = threadIdx.x;
    const float* sg = sigma + (long)b * n;
    float smax = 0.f;
    for (int j = tid; j < n; j += 256) smax = fmaxf(smax, sg[j]);
    smax = wave_max(smax);
    if ((tid & 63) == 0) red[tid >> 6] = smax;
    __syncthreads();
    smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float thr = smax * (float)n * 1.1920929e-7f;
    const float* Wb = W + (long)b * w_batch_stride;
    const double* L = Lb + (long)(b % lb_period) * l_batch_stride;
    const int tx = tid & 31, ty = tid >> 5;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < a0 + 32 && r0 < n; r0 += 32) {
        __syncthreads();
        for (int idx = tid; idx < 1024; idx += 256) {
            const int i = idx >> 5, rr = idx & 31, r = r0 + rr;
            lt[i][rr] = (a0 + i < n && r <= a0 + i) ? (float)L[(long)(a0 + i) * n + r] : 0.f;
            wt[i][rr] = (c0 + i < n && r < n) ? Wb[(long)(c0 + i) * n + r] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int rr = 0; rr < 32; ++rr) {
            const float lv = lt[tx][rr];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(lv, wt[ty + 8 * i][rr], acc[i]);
        }
    }
    float* Zb = Z + (long)b * z_batch_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, a = a0 + tx;
        if (c < n && a < n) {
            const float sv = sg[c];
            Zb[(long)c * n + a] = acc[i] * (sv > thr ? 1.f / (sv * sqrtf(sv)) : 0.f);
        }
    }
}

// The top half U Sigma of the stacked cores, which the backward through the mixing weights reads (teacher_factor: Z =
// L_a U), rebuilt after the TRANSPOSED route -- the riding rows never rode.  X = V Sigma is what the Jacobi on M^T left
// (compact n x n, column c at c * n); M itself (row-major) was stashed before the Jacobi destroyed it, in the unused
// BOTTOM half of the stacked buffer (row i of M at Wst + i * 2n + n):
//   (U Sigma)[i][c] = (M V)[i][c] = sigma_c^-1 sum_r M[i][r] X[r][c]                 (0 for truncated sigma)
// A product with M, not L_a^T (L_b V): the latter loses ||L_a|| ||L_b|| / ||M|| digits in the small-sigma columns (2.7 %
// in the teacher-side gradient at 196 tokens, measured), this one has the absolute accuracy eps sigma_max of columns
// that rode through the rotations.  32 x 32 tiles; grid = (ceil(n/32), ceil(n/32), batch), block = 256.
__global__ void __launch_bounds__(256) m_stash_kernel(const float* __restrict__ Wc, long wc_batch_stride, int n,
                                                      float* __restrict__ Wst, long w_batch_stride) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)n * n) return;
    const int b = blockIdx.y, i = (int)(idx / n), r = (int)(idx - (long)i * n);
    Wst[(long)b * w_batch_stride + (long)i * 2 * n + n + r] = Wc[(long)b * wc_batch_stride + idx];
}
__global__ void __launch_bounds__(256) ustack_from_x_kernel(const float* __restrict__ X, long x_batch_stride,
                                                            const float* __restrict__ sigma, int n,
                                                            const float* __restrict__ Wst, long w_batch_stride,
                                                            float* __restrict__ Out, long out_batch_stride) {
    __shared__ float mt[32][33], xt[32][33];      // mt[i][r], xt[c][r]
    __shared__ float red[4];
    const int b = blockIdx.z, i0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tid = threadIdx.x;
    const float* sg = sigma + (long)b * n;
    float smax = 0.f;
    for (int j = tid; j < n; j += 256) smax = fmaxf(smax, sg[j]);
    smax = wave_max(smax);
    if ((tid & 63) == 0) red[tid >> 6] = smax;
    __syncthreads();
    smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float thr = smax * (float)n * 1.1920929e-7f;
    const float* Xb = X + (long)b * x_batch_stride;
    const float* Wb = Wst + (long)b * w_batch_stride;
    float* Ob = Out + (long)b * out_batch_stride;
    const int tx = tid & 31, ty = tid >> 5;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < n; r0 += 32) {
        __syncthreads();
        for (int idx = tid; idx < 1024; idx += 256) {
            const int q = idx >> 5, rr = idx & 31, r = r0 + rr;
            mt[q][rr] = (i0 + q < n && r < n) ? Wb[(long)(i0 + q) * 2 * n + n + r] : 0.f;
            xt[q][rr] = (c0 + q < n && r < n) ? Xb[(long)(c0 + q) * n + r] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int rr = 0; rr < 32; ++rr) {
            const float mv = mt[tx][rr];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaf(mv, xt[ty + 8 * q][rr], acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = c0 + ty + 8 * q, i = i0 + tx;
        if (c < n && i < n) {
            const float sv = sg[c];
            Ob[(long)c * n + i] = sv > thr ? acc[q] / sv : 0.f;      // compact, column c at c * n
        }
    }
}
// compact n x n columns -> the top half of the stacked layout (column c at c * 2n)
__global__ void __launch_bounds__(256) ustack_place_kernel(const float* __restrict__ Uc, long uc_batch_stride, int n,
                                                           float* __restrict__ Wst, long w_batch_stride) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)n * n) return;
    const int b = blockIdx.y, c = (int)(idx / n), i = (int)(idx - (long)c * n);
    Wst[(long)b * w_batch_stride + (long)c * 2 * n + i] = Uc[(long)b * uc_batch_stride + idx];
}

__global__ void __launch_bounds__(256) kprime_from_z_kernel(const float* __restrict__ Z, long z_batch_stride, int n,
                                                            float* __restrict__ Kp) {
    __shared__ float za[32][33], zb[32][33];      // [c][a], [c][b]
    const int b = blockIdx.z, a0 = blockIdx.y * 32, b0 = blockIdx.x * 32, tid = threadIdx.x;
    const float* Zb = Z + (long)b * z_batch_stride;
    const int tx = tid & 31, ty = tid >> 5;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < n; c0 += 32) {
        __syncthreads();
        for (int idx = tid; idx < 1024; idx += 256) {
            const int cc = idx >> 5, i = idx & 31, c = c0 + cc;
            za[cc][i] = (c < n && a0 + i < n) ? Zb[(long)c * n + a0 + i] : 0.f;
            zb[cc][i] = (c < n && b0 + i < n) ? Zb[(long)c * n + b0 + i] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int cc = 0; cc < 32; ++cc) {
            const float bv = zb[cc][tx];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(za[cc][ty + 8 * i], bv, acc[i]);
        }
    }
    float* K = Kp + (long)b * n * n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int a = a0 + ty + 8 * i, bb = b0 + tx;
        if (a < n && bb < n) K[(long)a * n + bb] = acc[i];
    }
}

// ---------------------------------------------------------------------------
// Finalize: nuclear norm, teacher trace, per-sample loss, and K' = Y Sigma^+ Y^T.
// grid = batch, block = 256.  W is the rotated stack (top: U Sigma, bottom: Y = L_b V).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) procrustes_finalize_kernel(
    const float* __restrict__ W, long w_batch_stride, const float* __restrict__ sigma, int n, int n_s,
    const double* __restrict__ Gb, long g_batch_stride, const float* __restrict__ omega,
    const int* __restrict__ tap0, const int* __restrict__ tap1, const float* __restrict__ lam,
    const float* __restrict__ tr_s, int tr_slabs, float* __restrict__ tr_s_out, float* __restrict__ tr_t_out,
    float* __restrict__ nuc_out, float* __restrict__ loss_out, float* __restrict__ Kp, int t_period) {
    extern __shared__ float sm[];
    float* isig = sm;                      // n : 1/sigma_j or 0 when truncated
    __shared__ float red[32];
    __shared__ double redd[32];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* sg = sigma + (long)b * n;
    float smax = 0.f, ssum = 0.f;
    for (int j = tid; j < n; j += 256) {
        smax = fmaxf(smax, sg[j]);
        ssum += sg[j];
    }
    smax = wave_max(smax);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = smax;
    __syncthreads();
    smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float nuc = block_sum(ssum, red);
    const float thr = smax * (float)n * 1.1920929e-7f;
    for (int j = tid; j < n; j += 256) isig[j] = sg[j] > thr ? 1.f / sg[j] : 0.f;
    // teacher trace on the student grid: sum_n w_n |sum_j I[n,j] tc_j|^2, from the fp64 Gram
    const int bt = b % t_period;          // teacher-side sample (Gram, weights) shared by the extraction layers
    const double* G = Gb + (long)bt * g_batch_stride;
    double part = 0.;
    for (int s = tid; s < n_s; s += 256) {
        const double w = (double)omega[(long)bt * n_s + s];
        if (tap0) {
            const int i0 = tap0[s], i1 = tap1[s];
            const double l1 = (double)lam[s], l0 = 1. - l1;
            part += w * (l0 * l0 * G[(long)i0 * n + i0] + 2. * l0 * l1 * G[(long)i0 * n + i1] + l1 * l1 * G[(long)i1 * n + i1]);
        } else {
            part += w * G[(long)s * n + s];
        }
    }
    const double trt = block_sum(part, redd);
    if (tid == 0) {
        float trs = 0.f;
        for (int k = 0; k < tr_slabs; ++k) trs += tr_s[(long)b * tr_slabs + k];
        tr_s_out[b] = trs;
        tr_t_out[b] = (float)trt;
        nuc_out[b] = nuc;
        loss_out[b] = trs + (float)trt - 2.f * nuc;
    }
    __syncthreads();
    if (Kp) {
        // Yh = Y diag(sigma^-1/2) staged in LDS (column-major like W), K' = Yh Yh^T
        const float* Wb = W + (long)b * w_batch_stride;
        float* Yh = sm + n;
        for (int idx = tid; idx < n * n; idx += 256) {
            const int j = idx / n, r = idx - j * n;
            Yh[idx] = Wb[(long)j * 2 * n + n + r] * sqrtf(isig[j]);
        }
        __syncthreads();
        float* K = Kp + (long)b * n * n;
        for (int idx = tid; idx < n * n; idx += 256) {
            const int r = idx / n, c = idx - r * n;
            float acc = 0.f;
            for (int j = 0; j < n; ++j) acc = fmaf(Yh[j * n + r], Yh[j * n + c], acc);
            K[idx] = acc;
        }
    }
}

// K' = Y diag(sigma^+) Y^T for cores past the LDS-resident finalize kernel (n > 199): 32 x 32 output tiles.
// grid = (ceil(n/32), ceil(n/32), batch), block = 256.  Same truncation rule as procrustes_finalize_kernel.
__global__ void __launch_bounds__(256) kprime_tiled_kernel(const float* __restrict__ W, long w_batch_stride,
                                                           const float* __restrict__ sigma, int n,
                                                           float* __restrict__ Kp) {
    __shared__ float ya[32][33], yb[32][33];      // [j][row]
    __shared__ float red[4];
    const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tid = threadIdx.x;
    const float* sg = sigma + (long)b * n;
    float smax = 0.f;
    for (int j = tid; j < n; j += 256) smax = fmaxf(smax, sg[j]);
    smax = wave_max(smax);
    if ((tid & 63) == 0) red[tid >> 6] = smax;
    __syncthreads();
    smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float thr = smax * (float)n * 1.1920929e-7f;
    const float* Wb = W + (long)b * w_batch_stride;
    const int tx = tid & 31, ty = tid >> 5;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < n; j0 += 32) {
        __syncthreads();
        for (int idx = tid; idx < 1024; idx += 256) {
            const int jj = idx >> 5, rr = idx & 31, j = j0 + jj;
            float sc = 0.f;
            if (j < n && sg[j] > thr) sc = sqrtf(1.f / sg[j]);
            ya[jj][rr] = (j < n && r0 + rr < n) ? Wb[(long)j * 2 * n + n + r0 + rr] * sc : 0.f;
            yb[jj][rr] = (j < n && c0 + rr < n) ? Wb[(long)j * 2 * n + n + c0 + rr] * sc : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int jj = 0; jj < 32; ++jj) {
            const float bv = yb[jj][tx];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(ya[jj][ty + 8 * i], bv, acc[i]);
        }
    }
    float* K = Kp + (long)b * n * n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < n && c < n) K[(long)r * n + c] = acc[i];
    }
}

// ---------------------------------------------------------------------------
// Backward, student tokens:  dX[b,s,:] = coef * w_s * ( (x_s - mu) - interp(H)[s] ),
//   H = K' A' (n_t x D),  coef = *scale_ptr * scale_const.     grid = (n_s, B), block = 128
// Optionally also d loss_b / d omega_s = |x_c|^2 + |t_hat_c|^2 - 2 x_c . interp(H)_s  (un-scaled),
// needed only when the mixing weights carry a gradient (multi-layer teachers).
// ---------------------------------------------------------------------------
// blockIdx.z = extraction layer e when x_ptrs is given (all layers in one launch: X = nullptr, operands are (E, B, ...)
// with omega advanced by omega_e_stride per layer); a single layer otherwise (x_ptrs = nullptr, gridDim.z = 1).
template <typename T>
__global__ void __launch_bounds__(128) student_grad_kernel(const T* __restrict__ X, const void* const* __restrict__ x_ptrs,
                                                           long sb, long sn, int n_s, int n_t,
                                                           int D, const float* __restrict__ omega, long omega_e_stride,
                                                           const float* __restrict__ mu, const float* __restrict__ H,
                                                           const int* __restrict__ tap0, const int* __restrict__ tap1,
                                                           const float* __restrict__ lam,
                                                           const float* __restrict__ scale_ptr, float scale_const,
                                                           float* __restrict__ dX, const float* __restrict__ tnorm2,
                                                           float* __restrict__ gomega) {
    __shared__ float red[32];
    const int s = blockIdx.x, e = blockIdx.z;
    const long B = gridDim.y, b = (long)e * B + blockIdx.y;      // sample index into the (E, B, ...) operands
    if (x_ptrs) X = (const T*)x_ptrs[e];
    const float coef = scale_ptr[e] * scale_const * omega[(long)e * omega_e_stride + (long)blockIdx.y * n_s + s];
    const T* x = X + (long)blockIdx.y * sb + (long)s * sn;
    const float* m = mu + (long)b * D;
    int i0 = s, i1 = s;
    float l1 = 0.f;
    if (tap0) {
        i0 = tap0[s];
        i1 = tap1[s];
        l1 = lam[s];
    }
    const float* h0 = H + ((long)b * n_t + i0) * D;
    const float* h1 = H + ((long)b * n_t + i1) * D;
    float* out = dX + ((long)b * n_s + s) * D;
    float xx = 0.f, xt = 0.f;
    for (int d = threadIdx.x; d < D; d += 128) {
        const float tgt = (1.f - l1) * h0[d] + l1 * h1[d];
        const float xc = to_f32(x[d]) - m[d];
        out[d] = coef * (xc - tgt);
        xx = fmaf(xc, xc, xx);
        xt = fmaf(xc, tgt, xt);
    }
    if (gomega) {
        xx = block_sum(xx, red);
        xt = block_sum(xt, red);
        if (threadIdx.x == 0) gomega[(long)b * n_s + s] = xx + tnorm2[(long)b * n_s + s] - 2.f * xt;
    }
}

// ---------------------------------------------------------------------------
// Backward, student tokens, with H = K' A' formed in the workgroup instead of by a GEMM launch: one workgroup per
// (64-feature slab, sample, layer) stages its A' slab in LDS, wave g computes rows [RP g, RP g + RP) of the H slab with
// K' read through the scalar cache (K' is bit-symmetric, so row k serves as column k: the coefficients of one k are
// contiguous and wave-uniform), parks H in LDS and streams the sample's n_s token rows once.  Removes the H round trip
// (77 MB written + 77 MB read at cfg-2) and a launch whose 49-row GEMMs left two thirds of each MFMA tile empty.
// RP <= n_t <= 4 RP; 16-byte aligned token rows; D % 4 == 0.
// ---------------------------------------------------------------------------
constexpr int SG_W = 64;          // slab width in features
constexpr int SG_LD = SG_W + 4;   // row stride of the H slab in LDS
template <typename T, int RP>
__global__ void __launch_bounds__(256) student_grad_fused_kernel(
    const void* const* __restrict__ x_ptrs, long sb, long sn, int n_s, int n_t, int D, const float* __restrict__ omega,
    long omega_e_stride, const float* __restrict__ mu, const float* __restrict__ Kp, const float* __restrict__ Ap,
    const int* __restrict__ tap0, const int* __restrict__ tap1, const float* __restrict__ lam,
    const float* __restrict__ scale_ptr, float scale_const, float* __restrict__ dX) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* as = sm;                        // n_t x 64: A' slab
    float* hs = sm + (long)n_t * SG_W;     // n_t x 68: H slab
    const int by = blockIdx.y, e = blockIdx.z, d0 = blockIdx.x * SG_W, tid = threadIdx.x;
    const long b = (long)e * gridDim.y + by;
    const int q4 = 4 * (tid & 15);
    const bool live = d0 + q4 < D;
    const float* A = Ap + b * n_t * D + d0;
    for (int k = tid >> 4; k < n_t; k += 16) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) v = *(const float4*)(A + (long)k * D + q4);
        *(float4*)(as + k * SG_W + q4) = v;
    }
    __syncthreads();
    {
        // rows [i0, i0 + RP): the last waves step back so that their RP coefficients stay inside row k of K' (they
        // recompute rows of the wave before them and store the same values)
        const int g = __builtin_amdgcn_readfirstlane(tid >> 6), c = tid & 63;
        const int i0 = RP * g < n_t - RP ? RP * g : n_t - RP;
        const float* K = Kp + b * n_t * n_t + i0;
        float acc[RP];
#pragma unroll
        for (int r = 0; r < RP; ++r) acc[r] = 0.f;
        for (int k = 0; k < n_t; ++k) {
            const float a = as[k * SG_W + c];
            const float* Kk = K + (long)k * n_t;
#pragma unroll
            for (int r = 0; r < RP; ++r) acc[r] = fmaf(Kk[r], a, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < RP; ++r) hs[(i0 + r) * SG_LD + c] = acc[r];
    }
    __syncthreads();
    const BASD_GLOBAL_AS T* Xb = (const BASD_GLOBAL_AS T*)x_ptrs[e] + (long)by * sb + d0 + q4;
    const float* om = omega + (long)e * omega_e_stride + (long)by * n_s;
    float* out = dX + b * n_s * D + d0 + q4;
    const float sc = scale_ptr[e] * scale_const;
    float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) m4 = *(const float4*)(mu + b * D + d0 + q4);
    if (!live) return;
    for (int s0 = tid >> 4; s0 < n_s; s0 += 64) {
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + 16 * u;
            if (s < n_s) x[u] = ld4f(Xb + (long)s * sn);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + 16 * u;
            if (s >= n_s) break;
            int i0 = s, i1 = s;
            float l1 = 0.f;
            if (tap0) {
                i0 = tap0[s];
                i1 = tap1[s];
                l1 = lam[s];
            }
            const float coef = sc * om[s];
            const float4 h0 = *(const float4*)(hs + i0 * SG_LD + q4), h1 = *(const float4*)(hs + i1 * SG_LD + q4);
            float4 o;
            o.x = coef * ((x[u].x - m4.x) - ((1.f - l1) * h0.x + l1 * h1.x));
            o.y = coef * ((x[u].y - m4.y) - ((1.f - l1) * h0.y + l1 * h1.y));
            o.z = coef * ((x[u].z - m4.z) - ((1.f - l1) * h0.z + l1 * h1.z));
            o.w = coef * ((x[u].w - m4.w) - ((1.f - l1) * h0.w + l1 * h1.w));
            *(float4*)(out + (long)s * D) = o;
        }
    }
}

// ---------------------------------------------------------------------------
// Stand-alone token-count interpolation (reference combined.py:9-14), used by the public
// `_align_token_count`; the fused loss path never materialises this tensor.
// grid = (n_out, B), block = 128.   adjoint != 0: scatter-free transpose (out is n_in rows).
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(128) resample_tokens_kernel(const T* __restrict__ X, long sb, long sn, long sd,
                                                              int n_in, int n_out, int D,
                                                              const int* __restrict__ tap0,
                                                              const int* __restrict__ tap1,
                                                              const float* __restrict__ lam, float* __restrict__ out) {
    const int s = blockIdx.x, b = blockIdx.y;
    const int i0 = tap0[s], i1 = tap1[s];
    const float l1 = lam[s], l0 = 1.f - l1;
    const T* x0 = X + (long)b * sb + (long)i0 * sn;
    const T* x1 = X + (long)b * sb + (long)i1 * sn;
    float* o = out + ((long)b * n_out + s) * D;
    for (int d = threadIdx.x; d < D; d += 128) o[d] = l0 * to_f32(x0[(long)d * sd]) + l1 * to_f32(x1[(long)d * sd]);
}

// adjoint: dX[b, j, :] = sum_{s in [range0[j], range1[j])} I[s, j] dY[b, s, :]
__global__ void __launch_bounds__(128) resample_tokens_adjoint_kernel(const float* __restrict__ dY, int n_in,
                                                                      int n_out, int D,
                                                                      const int* __restrict__ tap0,
                                                                      const int* __restrict__ tap1,
                                                                      const float* __restrict__ lam,
                                                                      const int* __restrict__ range0,
                                                                      const int* __restrict__ range1,
                                                                      float* __restrict__ dX) {
    const int j = blockIdx.x, b = blockIdx.y;
    const int s0 = range0[j], s1 = range1[j];
    for (int d = threadIdx.x; d < D; d += 128) {
        float acc = 0.f;
        for (int s = s0; s < s1; ++s) {
            const float l1 = lam[s];
            const float coef = (tap0[s] == j ? 1.f - l1 : 0.f) + (tap1[s] == j ? l1 : 0.f);
            acc = fmaf(coef, dY[((long)b * n_out + s) * D + d], acc);
        }
        dX[((long)b * n_in + j) * D + d] = acc;
    }
}

}  // namespace basd

using namespace basd;

extern "C" {

// relational.py:22-34 on the layer-mixed attention.  attn_ptrs: device array of L pointers.
int basd_token_weights(const void* const* attn_ptrs, int dtype, const float* mix, int L, long sb, long sh, long sq,
                       long sk, int B, int H, int A, int has_cls, int n_a, int n_t, int n_s, const int* atap0,
                       const int* atap1, const float* alam, const int* tap0, const int* tap1, const float* lam,
                       float* omega, float* omega_t, float* raw_out, hipStream_t stream) {
    BASD_CHECK_ARG(attn_ptrs && mix && omega && omega_t && L > 0 && B > 0 && H > 0 && n_a > 0 && n_t > 0 && n_s > 0);
    BASD_CHECK_ARG(A == n_a + (has_cls ? 1 : 0));
    BASD_CHECK_ARG((n_a == n_s) == (atap0 == nullptr));
    BASD_CHECK_ARG((n_t == n_s) == (tap0 == nullptr));
    const size_t lds = sizeof(float) * (size_t)(n_a + n_s);
    if (dtype == BASD_DTYPE_F32)
        token_weights_kernel<float><<<B, 256, lds, stream>>>(attn_ptrs, mix, L, sb, sh, sq, sk, H, A, has_cls, n_a, n_t, n_s, atap0, atap1, alam, tap0, tap1, lam, omega, omega_t, raw_out);
    else if (dtype == BASD_DTYPE_BF16)
        token_weights_kernel<__hip_bfloat16><<<B, 256, lds, stream>>>(attn_ptrs, mix, L, sb, sh, sq, sk, H, A, has_cls, n_a, n_t, n_s, atap0, atap1, alam, tap0, tap1, lam, omega, omega_t, raw_out);
    else
        return BASD_EINVAL;
    BASD_RETURN_LAST();
}

// relational.py:36-45 (student half) + the transpose of combined.py:9-14.
// tr_s: (B, ceil(D/64)) per-slab partial traces (summed by basd_procrustes_finalize).
int basd_student_project(const void* x, int dtype, long sb, long sn, int B, int n_s, int n_t, int D,
                         const float* omega, const int* tap0, const int* tap1, const float* lam, const int* range0,
                         const int* range1, float* mu, float* tr_s, float* a_prime, hipStream_t stream) {
    BASD_CHECK_ARG(x && omega && mu && tr_s && a_prime && B > 0 && n_s > 0 && n_t > 0 && D > 0);
    BASD_CHECK_ARG((n_t == n_s) == (tap0 == nullptr));
    const size_t lds = sizeof(float) * (size_t)n_s;
    const dim3 grid((D + 63) / 64, B);
    // LDS-staged variant: 16-byte aligned row quads and a slab that fits
    const int esz = dtype == BASD_DTYPE_F32 ? 4 : 2;
    const size_t lds_slab = sizeof(float) * (size_t)n_s * 65;
    const bool staged = D % 4 == 0 && ((uintptr_t)x * 1) % 16 == 0 && (sb * esz) % 16 == 0 && (sn * esz) % 16 == 0 &&
                        (esz == 4 || D % 8 == 0) && lds_slab <= 64 * 1024;      // >= 2 workgroups per CU
    if (dtype == BASD_DTYPE_F32) {
        if (staged) {
            (void)hipFuncSetAttribute((const void*)student_project_lds_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_slab);
            student_project_lds_kernel<float><<<grid, 256, lds_slab, stream>>>((const float*)x, sb, sn, n_s, n_t, D, omega, tap0, tap1, lam, range0, range1, mu, tr_s, a_prime);
        } else {
            student_project_kernel<float><<<grid, 256, lds, stream>>>((const float*)x, sb, sn, n_s, n_t, D, omega, tap0, tap1, lam, range0, range1, mu, tr_s, a_prime);
        }
    } else if (dtype == BASD_DTYPE_BF16) {
        if (staged) {
            (void)hipFuncSetAttribute((const void*)student_project_lds_kernel<__hip_bfloat16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_slab);
            student_project_lds_kernel<__hip_bfloat16><<<grid, 256, lds_slab, stream>>>((const __hip_bfloat16*)x, sb, sn, n_s, n_t, D, omega, tap0, tap1, lam, range0, range1, mu, tr_s, a_prime);
        } else {
            student_project_kernel<__hip_bfloat16><<<grid, 256, lds, stream>>>((const __hip_bfloat16*)x, sb, sn, n_s, n_t, D, omega, tap0, tap1, lam, range0, range1, mu, tr_s, a_prime);
        }
    } else {
        return BASD_EINVAL;
    }
    BASD_RETURN_LAST();
}

// layer_selector.py:110-111 (token mixing) + relational.py:37,39 (teacher centring) on the core grid of n tokens.
// g0/g1/glam (nullable): gather taps when the teacher grid (finer than the student's) is resampled first.
int basd_teacher_center(const void* const* tok_ptrs, int dtype, const float* mix, int L, long sb, long sn, long sd,
                        int B, int n, int D, const int* g0, const int* g1, const float* glam, const float* omega_t,
                        float* mu, float* tc, hipStream_t stream) {
    BASD_CHECK_ARG(tok_ptrs && mix && omega_t && mu && tc && L > 0 && B > 0 && n > 0 && D > 0);
    BASD_CHECK_ARG((g0 == nullptr) == (g1 == nullptr) && (g0 == nullptr) == (glam == nullptr));
    const size_t lds = sizeof(float) * ((size_t)n * 65 + n + 64);
    if (lds > 150 * 1024) return BASD_EUNSUPPORTED;
    const dim3 grid((D + 63) / 64, B);
    if (dtype == BASD_DTYPE_F32) {
        (void)hipFuncSetAttribute((const void*)teacher_center_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        teacher_center_kernel<float><<<grid, 256, lds, stream>>>(tok_ptrs, mix, L, sb, sn, sd, n, D, g0, g1, glam, omega_t, mu, tc);
    } else if (dtype == BASD_DTYPE_BF16) {
        (void)hipFuncSetAttribute((const void*)teacher_center_kernel<__hip_bfloat16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        teacher_center_kernel<__hip_bfloat16><<<grid, 256, lds, stream>>>(tok_ptrs, mix, L, sb, sn, sd, n, D, g0, g1, glam, omega_t, mu, tc);
    } else {
        return BASD_EINVAL;
    }
    BASD_RETURN_LAST();
}

// basd_teacher_center for G <= 4 groups of mixing weights (mix: (G, L)) in one pass over the teacher layers.
// omega_t (G, B, n), mu (G, B, D), tc (G, B, n, D).  BASD_EUNSUPPORTED: more groups, or tiles past LDS -- call the
// per-group entry.
int basd_teacher_center_multi(const void* const* tok_ptrs, int dtype, const float* mix, int L, int G, long sb, long sn,
                              long sd, int B, int n, int D, const int* g0, const int* g1, const float* glam,
                              const float* omega_t, float* mu, float* tc, hipStream_t stream) {
    BASD_CHECK_ARG(tok_ptrs && mix && omega_t && mu && tc && L > 0 && G > 0 && B > 0 && n > 0 && D > 0);
    BASD_CHECK_ARG((g0 == nullptr) == (g1 == nullptr) && (g0 == nullptr) == (glam == nullptr));
    BASD_CHECK_ARG(B <= 65535);
    const size_t lds = sizeof(float) * ((size_t)G * n * (TCM_W + 1) + (size_t)G * n + (size_t)G * TCM_W);
    if (G > TCM_G || lds > 150 * 1024) return BASD_EUNSUPPORTED;
    const dim3 grid((D + TCM_W - 1) / TCM_W, B);
    if (dtype == BASD_DTYPE_F32) {
        (void)hipFuncSetAttribute((const void*)teacher_center_multi_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        teacher_center_multi_kernel<float><<<grid, 256, lds, stream>>>(tok_ptrs, mix, L, G, sb, sn, sd, n, D, g0, g1, glam, omega_t, mu, tc);
    } else if (dtype == BASD_DTYPE_BF16) {
        (void)hipFuncSetAttribute((const void*)teacher_center_multi_kernel<__hip_bfloat16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        teacher_center_multi_kernel<__hip_bfloat16><<<grid, 256, lds, stream>>>(tok_ptrs, mix, L, G, sb, sn, sd, n, D, g0, g1, glam, omega_t, mu, tc);
    } else {
        return BASD_EINVAL;
    }
    BASD_RETURN_LAST();
}

// The same for fp32 tokens on the teacher's own grid with 16-byte aligned rows (ViT teachers: cfg-4): a thread owns FOUR
// consecutive features, the layers are read four at a time with all loads in flight before the first use (the scalar form
// issues one 4-byte load per layer and waits for it: 2.5 TB/s over 24 layers of 103 MB).  Falls back to the scalar form
// inside the launch when a layer's base pointer is not 16-byte aligned (wave-uniform).
__global__ void __launch_bounds__(256) teacher_mix_stream_v4_kernel(
    const void* const* __restrict__ tok_ptrs, const float* __restrict__ mix, int L, int G, long sb, long sn, int n, int D,
    const float* __restrict__ omega_t, float* __restrict__ Tc, float* __restrict__ chunk_sum) {
    const int chunk = blockIdx.x, b = blockIdx.y, B = gridDim.y, tid = threadIdx.x;
    const int j_lo = chunk * TCS_ROWS, j_hi = j_lo + TCS_ROWS < n ? j_lo + TCS_ROWS : n;
    bool aligned = true;
    for (int l = 0; l < L; ++l) aligned = aligned && (((uintptr_t)tok_ptrs[l]) & 15) == 0;
    if (!aligned) {       // uniform over the launch
        for (int d = tid; d < D; d += 256) {
            float musum[TCM_G] = {0.f, 0.f, 0.f, 0.f};
            for (int j = j_lo; j < j_hi; ++j) {
                const long o0 = b * sb + (long)j * sn + d;
                float v[TCM_G] = {0.f, 0.f, 0.f, 0.f};
                for (int l = 0; l < L; ++l) {
                    const float x = ldg_f32((const BASD_GLOBAL_AS float*)tok_ptrs[l] + o0);
#pragma unroll
                    for (int g = 0; g < TCM_G; ++g)
                        if (g < G) v[g] = fmaf(mix[g * L + l], x, v[g]);
                }
#pragma unroll
                for (int g = 0; g < TCM_G; ++g)
                    if (g < G) {
                        Tc[(((long)g * B + b) * n + j) * D + d] = v[g];
                        musum[g] = fmaf(omega_t[((long)g * B + b) * n + j], v[g], musum[g]);
                    }
            }
#pragma unroll
            for (int g = 0; g < TCM_G; ++g)
                if (g < G) chunk_sum[(((long)chunk * G + g) * B + b) * D + d] = musum[g];
        }
        return;
    }
    for (int d = 4 * tid; d < D; d += 1024) {
        float4 musum[TCM_G];
#pragma unroll
        for (int g = 0; g < TCM_G; ++g) musum[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = j_lo; j < j_hi; ++j) {
            const long o0 = b * sb + (long)j * sn + d;
            float4 v[TCM_G];
#pragma unroll
            for (int g = 0; g < TCM_G; ++g) v[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            int l = 0;
            for (; l + 4 <= L; l += 4) {
                float4 x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = ld4f((const BASD_GLOBAL_AS float*)tok_ptrs[l + q] + o0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int g = 0; g < TCM_G; ++g)
                        if (g < G) {
                            const float m = mix[g * L + l + q];
                            v[g].x = fmaf(m, x[q].x, v[g].x); v[g].y = fmaf(m, x[q].y, v[g].y);
                            v[g].z = fmaf(m, x[q].z, v[g].z); v[g].w = fmaf(m, x[q].w, v[g].w);
                        }
            }
            for (; l < L; ++l) {
                const float4 x = ld4f((const BASD_GLOBAL_AS float*)tok_ptrs[l] + o0);
#pragma unroll
                for (int g = 0; g < TCM_G; ++g)
                    if (g < G) {
                        const float m = mix[g * L + l];
                        v[g].x = fmaf(m, x.x, v[g].x); v[g].y = fmaf(m, x.y, v[g].y);
                        v[g].z = fmaf(m, x.z, v[g].z); v[g].w = fmaf(m, x.w, v[g].w);
                    }
            }
#pragma unroll
            for (int g = 0; g < TCM_G; ++g)
                if (g < G) {
                    *(float4*)(Tc + (((long)g * B + b) * n + j) * D + d) = v[g];
                    const float om = omega_t[((long)g * B + b) * n + j];
                    musum[g].x = fmaf(om, v[g].x, musum[g].x); musum[g].y = fmaf(om, v[g].y, musum[g].y);
                    musum[g].z = fmaf(om, v[g].z, musum[g].z); musum[g].w = fmaf(om, v[g].w, musum[g].w);
                }
        }
#pragma unroll
        for (int g = 0; g < TCM_G; ++g)
            if (g < G) *(float4*)(chunk_sum + (((long)chunk * G + g) * B + b) * D + d) = musum[g];
    }
}

// Streaming form of basd_teacher_center_multi for row-major teacher tokens (see teacher_mix_stream_kernel): scratch of
// basd_teacher_center_stream_scratch_floats(G, B, n, D) floats.  BASD_EUNSUPPORTED: more than 4 groups, features not
// contiguous.
long basd_teacher_center_stream_scratch_floats(int G, int B, int n, int D) {
    if (G <= 0 || B <= 0 || n <= 0 || D <= 0) return 0;
    return (long)((n + TCS_ROWS - 1) / TCS_ROWS) * G * B * D;
}
int basd_teacher_center_stream(const void* const* tok_ptrs, int dtype, const float* mix, int L, int G, long sb, long sn,
                               long sd, int B, int n, int D, const int* g0, const int* g1, const float* glam,
                               const float* omega_t, float* mu, float* tc, float* scratch, hipStream_t stream) {
    BASD_CHECK_ARG(tok_ptrs && mix && omega_t && mu && tc && scratch && L > 0 && G > 0 && B > 0 && n > 0 && D > 0);
    BASD_CHECK_ARG((g0 == nullptr) == (g1 == nullptr) && (g0 == nullptr) == (glam == nullptr));
    if (G > TCM_G || sd != 1 || B > 65535 || (long)G * B > 65535) return BASD_EUNSUPPORTED;
    const int chunks = (n + TCS_ROWS - 1) / TCS_ROWS;
    if (dtype == BASD_DTYPE_F32 && g0 == nullptr && D % 4 == 0 && sb % 4 == 0 && sn % 4 == 0 && (((uintptr_t)tc) & 15) == 0 &&
        (((uintptr_t)scratch) & 15) == 0)
        teacher_mix_stream_v4_kernel<<<dim3(chunks, B), 256, 0, stream>>>(tok_ptrs, mix, L, G, sb, sn, n, D, omega_t, tc, scratch);
    else if (dtype == BASD_DTYPE_F32)
        teacher_mix_stream_kernel<float><<<dim3(chunks, B), 256, 0, stream>>>(tok_ptrs, mix, L, G, sb, sn, sd, n, D, g0, g1, glam, omega_t, tc, scratch);
    else if (dtype == BASD_DTYPE_BF16)
        teacher_mix_stream_kernel<__hip_bfloat16><<<dim3(chunks, B), 256, 0, stream>>>(tok_ptrs, mix, L, G, sb, sn, sd, n, D, g0, g1, glam, omega_t, tc, scratch);
    else
        return BASD_EINVAL;
    teacher_center_apply_kernel<<<dim3((D + 255) / 256, chunks, G * B), 256, 0, stream>>>(scratch, chunks, n, D, mu, tc);
    BASD_RETURN_LAST();
}

// G[b] = P[b] P[b]^T in fp64 (the bmm of relational.py:47 reduced to the teacher grid).
int basd_gram_f64(const float* p, long p_batch_stride, int n, int D, int batch, double* g, long g_batch_stride,
                  hipStream_t stream) {
    BASD_CHECK_ARG(p && g && n > 0 && D > 0 && batch > 0);
    BASD_CHECK_ARG(((uintptr_t)p & 15) == 0 && D % 4 == 0 && p_batch_stride % 4 == 0);
    int waves, chunks;
    size_t lds;
    if (!gram_f64_shape(n, &waves, &chunks, &lds)) return BASD_EUNSUPPORTED;
    gram_f64_kernel<<<dim3(batch, 1, chunks), 64 * waves, lds, stream>>>(p, p_batch_stride, n, D, g, g_batch_stride, 0);
    BASD_RETURN_LAST();
}

// The same with the contraction split over `splits` workgroups per matrix (few matrices with a long feature axis: the
// teacher side, 256 x (49 x 2048) at cfg-2, left three quarters of the CUs idle).  slabs: splits * batch * n * n doubles
// of scratch; g: contiguous (batch, n, n).
int basd_gram_f64_split(const float* p, long p_batch_stride, int n, int D, int batch, int splits, double* slabs,
                        double* g, hipStream_t stream) {
    BASD_CHECK_ARG(p && g && slabs && n > 0 && D > 0 && batch > 0 && splits >= 1 && splits <= 64);
    BASD_CHECK_ARG(((uintptr_t)p & 15) == 0 && D % 4 == 0 && p_batch_stride % 4 == 0);
    int waves, chunks;
    size_t lds;
    if (!gram_f64_shape(n, &waves, &chunks, &lds)) return BASD_EUNSUPPORTED;
    const long count = (long)batch * n * n;
    gram_f64_kernel<<<dim3(batch, splits, chunks), 64 * waves, lds, stream>>>(p, p_batch_stride, n, D, slabs, (long)n * n, count);
    gram_f64_reduce_kernel<<<(unsigned)((count + 255) / 256), 256, 0, stream>>>(slabs, count, splits, count, g);
    BASD_RETURN_LAST();
}

int basd_chol_f64(const double* g, long g_batch_stride, int n, int batch, double* l, long l_batch_stride,
                  hipStream_t stream) {
    BASD_CHECK_ARG(g && l && n > 0 && batch > 0);
    const size_t lds = sizeof(double) * ((size_t)n * (n + 1) / 2 + n);
    if (lds > 158 * 1024) {
        // past LDS: panels of the factor in LDS, the factor itself in global memory
        if (n > 64 * CB_MAXR) return BASD_EUNSUPPORTED;
        const size_t lds_b = sizeof(double) * ((size_t)n * CB_NB + (size_t)CB_NB * (CB_KC + 1));
        (void)hipFuncSetAttribute((const void*)chol_f64_blocked_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        chol_f64_blocked_kernel<<<batch, 1024, lds_b, stream>>>(g, g_batch_stride, n, l, l_batch_stride);
        BASD_RETURN_LAST();
    }
    (void)hipFuncSetAttribute((const void*)chol_f64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    const int threads = n >= 128 ? 1024 : n >= 64 ? 512 : 256;
    chol_f64_kernel<<<batch, threads, lds, stream>>>(g, g_batch_stride, n, l, l_batch_stride);
    BASD_RETURN_LAST();
}

int basd_stack_product(const double* la, const double* lb, long l_batch_stride, int n, int batch, int lb_period,
                       float* w, long w_batch_stride, hipStream_t stream) {
    BASD_CHECK_ARG(la && lb && w && n > 0 && batch > 0 && lb_period > 0);
    const size_t lds = sizeof(double) * 2 * (size_t)n * n;
    if (lds <= 78 * 1024) {           // two workgroups per CU
        (void)hipFuncSetAttribute((const void*)stack_product_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024);
        stack_product_kernel<<<batch, 256, lds, stream>>>(la, lb, l_batch_stride, n, w, w_batch_stride, lb_period);
    } else {
        BASD_CHECK_ARG(batch <= 65535);
        const int nt = (n + 31) / 32;
        stack_product_global_kernel<<<dim3(nt, nt, batch), 256, 0, stream>>>(la, lb, l_batch_stride, n, w, w_batch_stride, lb_period);
    }
    BASD_RETURN_LAST();
}

// The stacked product without the stack: M = L_a^T L_b row-major and compact at w + b * w_batch_stride (= M^T as a
// column-major matrix: the one-sided Jacobi on it leaves V Sigma, see basd_kprime_from_transposed).
int basd_stack_product_t(const double* la, const double* lb, long l_batch_stride, int n, int batch, int lb_period,
                         float* w, long w_batch_stride, hipStream_t stream) {
    BASD_CHECK_ARG(la && lb && w && n > 0 && batch > 0 && lb_period > 0 && batch <= 65535);
    const int nt = (n + 31) / 32;
    stack_product_t_kernel<<<dim3(nt, nt, batch), 256, 0, stream>>>(la, lb, l_batch_stride, n, w, w_batch_stride, lb_period);
    BASD_RETURN_LAST();
}

// K' = Y Sigma^+ Y^T from the rotated TRANSPOSED core (columns V_c sigma_c, compact n x n at w + b * w_batch_stride),
// sigma and the teacher-side factor L_b: Y = L_b V is formed here instead of riding through the rotations.
// z: scratch, n x n floats per matrix at z + b * z_batch_stride.
int basd_kprime_from_transposed(const float* w, long w_batch_stride, const float* sigma, int n, int batch,
                                const double* lb, long l_batch_stride, int lb_period, float* z, long z_batch_stride,
                                float* k_prime, hipStream_t stream) {
    BASD_CHECK_ARG(w && sigma && lb && z && k_prime && n > 0 && batch > 0 && lb_period > 0 && batch <= 65535);
    const int nt = (n + 31) / 32;
    kprime_z_kernel<<<dim3(nt, nt, batch), 256, 0, stream>>>(w, w_batch_stride, sigma, n, lb, l_batch_stride, lb_period, z, z_batch_stride);
    kprime_from_z_kernel<<<dim3(nt, nt, batch), 256, 0, stream>>>(z, z_batch_stride, n, k_prime);
    BASD_RETURN_LAST();
}

// The transposed route for steps whose backward goes through the mixing weights, in two calls around the Jacobi:
// basd_ustack_stash copies M (row-major, compact at wc + b * wc_batch_stride: what basd_stack_product_t wrote) into the
// unused bottom half of the stacked buffer; basd_ustack_from_transposed then forms U Sigma = M V from it and the rotated
// X = V Sigma (same compact place) into the top half, stacked layout: column c at w_stack + b * w_stack_stride + c * 2n.
int basd_ustack_stash(const float* wc, long wc_batch_stride, int n, int batch, float* w_stack, long w_stack_stride,
                      hipStream_t stream) {
    BASD_CHECK_ARG(wc && w_stack && n > 0 && batch > 0 && batch <= 65535 && w_stack_stride >= 2L * n * n);
    m_stash_kernel<<<dim3((unsigned)(((long)n * n + 255) / 256), batch), 256, 0, stream>>>(wc, wc_batch_stride, n, w_stack, w_stack_stride);
    BASD_RETURN_LAST();
}
int basd_ustack_from_transposed(const float* x, long x_batch_stride, const float* sigma, int n, int batch,
                                float* w_stack, long w_stack_stride, float* scratch, long scratch_batch_stride,
                                float* sigma_u, int max_sweeps, int* jflags, hipStream_t stream) {
    BASD_CHECK_ARG(x && sigma && w_stack && scratch && sigma_u && n > 0 && batch > 0 && batch <= 65535 &&
                   w_stack_stride >= 2L * n * n && scratch_batch_stride >= (long)n * n);
    const int nt = (n + 31) / 32;
    ustack_from_x_kernel<<<dim3(nt, nt, batch), 256, 0, stream>>>(x, x_batch_stride, sigma, n, w_stack, w_stack_stride, scratch, scratch_batch_stride);
    // M V has the columns of U Sigma to ~tol sigma_max, not to tol sigma_c: orthogonality was enforced on V Sigma.  A
    // short one-sided Jacobi on M V itself (it starts almost converged: one or two sweeps) makes them orthogonal relative
    // to their own norms, as columns that rode through the rotations are; sigma_u are their norms.
    int rc = basd_jacobi_onesided(scratch, scratch_batch_stride, n, n, n, batch, nullptr, sigma_u, n, max_sweeps, 0.f,
                                  jflags, nullptr, stream);
    if (rc != BASD_OK) return rc;
    ustack_place_kernel<<<dim3((unsigned)(((long)n * n + 255) / 256), batch), 256, 0, stream>>>(scratch, scratch_batch_stride, n, w_stack, w_stack_stride);
    BASD_RETURN_LAST();
}

// relational.py:45-50 per sample: tr_t, nuclear norm, loss_b = tr_s + tr_t - 2 nuc; K' for backward (nullable).
int basd_procrustes_finalize(const float* w, long w_batch_stride, const float* sigma, int n, int n_s, int batch,
                             int t_period, const double* gb, long g_batch_stride, const float* omega, const int* tap0,
                             const int* tap1, const float* lam, const float* tr_s_part, int tr_slabs, float* tr_s,
                             float* tr_t, float* nuc, float* loss, float* k_prime, hipStream_t stream) {
    BASD_CHECK_ARG(w && sigma && gb && omega && tr_s_part && tr_s && tr_t && nuc && loss && n > 0 && batch > 0 && tr_slabs > 0);
    BASD_CHECK_ARG(t_period > 0);
    size_t lds = sizeof(float) * ((size_t)n + (k_prime ? (size_t)n * n : 0));
    const bool tiled = lds > 156 * 1024;       // K' past LDS: the per-sample terms here, K' by a tiled kernel
    if (tiled) lds = sizeof(float) * (size_t)n;
    if (lds > 156 * 1024) return BASD_EUNSUPPORTED;
    (void)hipFuncSetAttribute((const void*)procrustes_finalize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    procrustes_finalize_kernel<<<batch, 256, lds, stream>>>(w, w_batch_stride, sigma, n, n_s, gb, g_batch_stride, omega, tap0, tap1, lam, tr_s_part, tr_slabs, tr_s, tr_t, nuc, loss, tiled ? nullptr : k_prime, t_period);
    if (tiled) {
        BASD_CHECK_ARG(batch <= 65535);
        const int nt = (n + 31) / 32;
        kprime_tiled_kernel<<<dim3(nt, nt, batch), 256, 0, stream>>>(w, w_batch_stride, sigma, n, k_prime);
    }
    BASD_RETURN_LAST();
}

// Gradient of sum_b coef * loss_b with respect to the student tokens (autograd of relational.py:36-50).
int basd_student_grad(const void* x, int dtype, long sb, long sn, int B, int n_s, int n_t, int D, const float* omega,
                      const float* mu, const float* h, const int* tap0, const int* tap1, const float* lam,
                      const float* scale_ptr, float scale_const, float* dx, const float* tnorm2, float* gomega,
                      hipStream_t stream) {
    BASD_CHECK_ARG(x && omega && mu && h && scale_ptr && dx && B > 0 && n_s > 0 && n_t > 0 && D > 0);
    BASD_CHECK_ARG((gomega == nullptr) || (tnorm2 != nullptr));
    const dim3 grid(n_s, B);
    if (dtype == BASD_DTYPE_F32)
        student_grad_kernel<float><<<grid, 128, 0, stream>>>((const float*)x, nullptr, sb, sn, n_s, n_t, D, omega, 0, mu, h, tap0, tap1, lam, scale_ptr, scale_const, dx, tnorm2, gomega);
    else if (dtype == BASD_DTYPE_BF16)
        student_grad_kernel<__hip_bfloat16><<<grid, 128, 0, stream>>>((const __hip_bfloat16*)x, nullptr, sb, sn, n_s, n_t, D, omega, 0, mu, h, tap0, tap1, lam, scale_ptr, scale_const, dx, tnorm2, gomega);
    else
        return BASD_EINVAL;
    BASD_RETURN_LAST();
}

// The same for all E extraction layers in one launch: x_ptrs device table of E base pointers (common strides),
// omega + e * omega_e_stride (0: one weight vector for all layers), mu / h / dx / tnorm2 / gomega laid out (E, B, ...),
// scale_ptr[e] the upstream gradient of layer e.
int basd_student_grad_multi(const void* const* x_ptrs, int dtype, long sb, long sn, int E, int B, int n_s, int n_t,
                            int D, const float* omega, long omega_e_stride, const float* mu, const float* h,
                            const int* tap0, const int* tap1, const float* lam, const float* scale_ptr,
                            float scale_const, float* dx, const float* tnorm2, float* gomega, hipStream_t stream) {
    BASD_CHECK_ARG(x_ptrs && omega && mu && h && scale_ptr && dx && E > 0 && B > 0 && n_s > 0 && n_t > 0 && D > 0);
    BASD_CHECK_ARG((gomega == nullptr) || (tnorm2 != nullptr));
    BASD_CHECK_ARG(E <= 65535 && B <= 65535);
    const dim3 grid(n_s, B, E);
    if (dtype == BASD_DTYPE_F32)
        student_grad_kernel<float><<<grid, 128, 0, stream>>>(nullptr, x_ptrs, sb, sn, n_s, n_t, D, omega, omega_e_stride, mu, h, tap0, tap1, lam, scale_ptr, scale_const, dx, tnorm2, gomega);
    else if (dtype == BASD_DTYPE_BF16)
        student_grad_kernel<__hip_bfloat16><<<grid, 128, 0, stream>>>(nullptr, x_ptrs, sb, sn, n_s, n_t, D, omega, omega_e_stride, mu, h, tap0, tap1, lam, scale_ptr, scale_const, dx, tnorm2, gomega);
    else
        return BASD_EINVAL;
    BASD_RETURN_LAST();
}

// basd_gemm_tn (H = K' A') + basd_student_grad_multi in one launch, for cores of up to 64 teacher tokens (see the
// kernel).  k_prime (E, B, n_t, n_t) as basd_procrustes_finalize writes it, a_prime (E, B, n_t, D).  Returns
// BASD_EUNSUPPORTED where it does not apply (larger cores, rows not 16-byte aligned): call the two-launch form.
int basd_student_grad_fused(const void* const* x_ptrs, int dtype, long sb, long sn, int E, int B, int n_s, int n_t,
                            int D, int ptrs_16B_aligned, const float* omega, long omega_e_stride, const float* mu,
                            const float* k_prime, const float* a_prime, const int* tap0, const int* tap1,
                            const float* lam, const float* scale_ptr, float scale_const, float* dx,
                            hipStream_t stream) {
    BASD_CHECK_ARG(x_ptrs && omega && mu && k_prime && a_prime && scale_ptr && dx && E > 0 && B > 0 && n_s > 0 && n_t > 0 && D > 0);
    BASD_CHECK_ARG((n_t == n_s) == (tap0 == nullptr));
    BASD_CHECK_ARG(E <= 65535 && B <= 65535);
    const int esz = dtype == BASD_DTYPE_F32 ? 4 : 2;
    const bool ok = ptrs_16B_aligned && n_t <= 64 && D % 4 == 0 && (sb * esz) % 16 == 0 && (sn * esz) % 16 == 0 &&
                    (esz == 4 || D % 8 == 0) && ((uintptr_t)a_prime & 15) == 0 && ((uintptr_t)mu & 15) == 0 &&
                    ((uintptr_t)dx & 15) == 0;
    if (!ok) return BASD_EUNSUPPORTED;
    if (n_t < 4) return BASD_EUNSUPPORTED;
    const int rp = (n_t + 3) / 4;
    const size_t lds = sizeof(float) * (size_t)n_t * (SG_W + SG_LD);
    const dim3 grid((D + SG_W - 1) / SG_W, B, E);
#define LAUNCH_SGF(TY, RP)                                                                                           \
    student_grad_fused_kernel<TY, RP><<<grid, 256, lds, stream>>>(x_ptrs, sb, sn, n_s, n_t, D, omega, omega_e_stride, mu, \
                                                                  k_prime, a_prime, tap0, tap1, lam, scale_ptr,      \
                                                                  scale_const, dx)
#define LAUNCH_SGF_T(TY)                 \
    do {                                 \
        if (rp <= 4) LAUNCH_SGF(TY, 4);  \
        else if (rp <= 8) LAUNCH_SGF(TY, 8);  \
        else if (rp <= 13) LAUNCH_SGF(TY, 13); \
        else LAUNCH_SGF(TY, 16);         \
    } while (0)
    if (dtype == BASD_DTYPE_F32) LAUNCH_SGF_T(float);
    else if (dtype == BASD_DTYPE_BF16) LAUNCH_SGF_T(__hip_bfloat16);
    else return BASD_EINVAL;
#undef LAUNCH_SGF_T
#undef LAUNCH_SGF
    BASD_RETURN_LAST();
}

// basd_student_project for all E extraction layers in one launch (x_ptrs: device table; omega + e * omega_e_stride;
// mu (E, B, D), tr_s (E, B, ceil(D/32)) -- 32-feature slabs here --, a_prime (E, B, n_t, D)).  Returns BASD_EUNSUPPORTED where the vectorised,
// LDS-staged kernel does not apply (rows not 16-byte aligned, slab over 64 KB): call basd_student_project per layer.
int basd_student_project_multi(const void* const* x_ptrs, int dtype, long sb, long sn, int E, int B, int n_s, int n_t,
                               int D, int ptrs_16B_aligned, const float* omega, long omega_e_stride, const int* tap0,
                               const int* tap1, const float* lam, const int* range0, const int* range1, float* mu,
                               float* tr_s, float* a_prime, hipStream_t stream) {
    BASD_CHECK_ARG(x_ptrs && omega && mu && tr_s && a_prime && E > 0 && B > 0 && n_s > 0 && n_t > 0 && D > 0);
    BASD_CHECK_ARG((n_t == n_s) == (tap0 == nullptr));
    BASD_CHECK_ARG(E <= 65535 && B <= 65535);
    const int esz = dtype == BASD_DTYPE_F32 ? 4 : 2;
    const size_t lds = sizeof(float) * ((size_t)n_s * SP_W + 5 * (size_t)n_s);
    const bool ok = ptrs_16B_aligned && D % 4 == 0 && (sb * esz) % 16 == 0 && (sn * esz) % 16 == 0 &&
                    (esz == 4 || D % 8 == 0) && lds <= 96 * 1024;
    if (!ok) return BASD_EUNSUPPORTED;
    const dim3 grid((D + SP_W - 1) / SP_W, B, E);
    if (dtype == BASD_DTYPE_F32) {
        (void)hipFuncSetAttribute((const void*)student_project_v4_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        student_project_v4_kernel<float><<<grid, 256, lds, stream>>>(x_ptrs, sb, sn, n_s, n_t, D, omega, omega_e_stride, tap0, tap1, lam, range0, range1, mu, tr_s, a_prime);
    } else if (dtype == BASD_DTYPE_BF16) {
        (void)hipFuncSetAttribute((const void*)student_project_v4_kernel<__hip_bfloat16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        student_project_v4_kernel<__hip_bfloat16><<<grid, 256, lds, stream>>>(x_ptrs, sb, sn, n_s, n_t, D, omega, omega_e_stride, tap0, tap1, lam, range0, range1, mu, tr_s, a_prime);
    } else {
        return BASD_EINVAL;
    }
    BASD_RETURN_LAST();
}

// combined.py:9-14 as a stand-alone op: out (B, n_out, D) fp32 contiguous.
int basd_resample_tokens(const void* x, int dtype, long sb, long sn, long sd, int B, int n_in, int n_out, int D,
                         const int* tap0, const int* tap1, const float* lam, float* out, hipStream_t stream) {
    BASD_CHECK_ARG(x && tap0 && tap1 && lam && out && B > 0 && n_in > 0 && n_out > 0 && D > 0);
    const dim3 grid(n_out, B);
    if (dtype == BASD_DTYPE_F32)
        resample_tokens_kernel<float><<<grid, 128, 0, stream>>>((const float*)x, sb, sn, sd, n_in, n_out, D, tap0, tap1, lam, out);
    else if (dtype == BASD_DTYPE_BF16)
        resample_tokens_kernel<__hip_bfloat16><<<grid, 128, 0, stream>>>((const __hip_bfloat16*)x, sb, sn, sd, n_in, n_out, D, tap0, tap1, lam, out);
    else
        return BASD_EINVAL;
    BASD_RETURN_LAST();
}

// Adjoint of basd_resample_tokens: dx (B, n_in, D) from dy (B, n_out, D), both fp32 contiguous.
int basd_resample_tokens_adjoint(const float* dy, int B, int n_in, int n_out, int D, const int* tap0,
                                 const int* tap1, const float* lam, const int* range0, const int* range1, float* dx,
                                 hipStream_t stream) {
    BASD_CHECK_ARG(dy && tap0 && tap1 && lam && range0 && range1 && dx && B > 0 && n_in > 0 && n_out > 0 && D > 0);
    resample_tokens_adjoint_kernel<<<dim3(n_in, B), 128, 0, stream>>>(dy, n_in, n_out, D, tap0, tap1, lam, range0, range1, dx);
    BASD_RETURN_LAST();
}

}  // extern "C"
