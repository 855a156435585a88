// One-sided (Hestenes) Jacobi kernels: the solver behind every SVD / symmetric
// eigen-decomposition on the BASD loss path.
//
//   reference call sites replaced (all dispatch to LAPACK through torch):
//     torch.linalg.eigvalsh   src/losses/layer_selector.py:16
//     torch.linalg.svd        src/losses/layer_selector.py:36, :92
//     torch.linalg.svdvals    src/losses/layer_selector.py:99
//     torch.linalg.matrix_norm(ord="nuc")   src/losses/relational.py:48
//
// A matrix W (rows_tot x n, column-major, fp32) is post-multiplied by plane
// rotations until its first `rows_dot` rows have mutually orthogonal columns.
// Rows rows_dot..rows_tot-1 ride along (they carry e.g. a second factor that
// must see the same right rotations).  Column norms are then the singular
// values; for a symmetric PSD input they are the eigenvalues and the
// normalised columns the eigenvectors.
//
// Two execution shapes:
//   * LDS-resident: the whole matrix lives in one CU's LDS, one workgroup per
//     matrix, all sweeps in ONE launch.  Column pairs of a round-robin round
//     are independent; each pair is owned by a group of LPP adjacent lanes and
//     its three dot products are reduced with wave shuffles.
//   * block: matrices that do not fit LDS stay in HBM/L2; one launch per
//     round-robin round over column BLOCKS, one workgroup per block pair,
//     which stages its 2*BW columns in LDS, orthogonalises them (one wave per
//     column pair) and writes them back.  Convergence is tracked in a per-
//     matrix flag so that the launches queued for later sweeps return at once.
#include "basd_common.h"

namespace basd {

struct Rot {
    float c, s;    // fp32 cosine / sine of the plane rotation
    float cl, sl;  // low parts: (c + cl)^2 + (s + sl)^2 = 1 to ~1e-14, so column norms do not drift
    bool apply;
};

// Rotation that makes columns p,q orthogonal given alpha=|p|^2, beta=|q|^2, gamma=p.q
__device__ __forceinline__ Rot make_rotation(float alpha, float beta, float gamma, float tol) {
    Rot r{1.f, 0.f, 0.f, 0.f, false};
    const float lim = tol * sqrtf(alpha) * sqrtf(beta);
    if (!(fabsf(gamma) > lim) || gamma == 0.f) return r;
    const float zeta = (beta - alpha) / (2.f * gamma);
    float t;
    if (fabsf(zeta) > 1e8f) {
        t = 0.5f / zeta;
    } else {
        t = copysignf(1.f, zeta) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
    }
    r.c = 1.f / sqrtf(1.f + t * t);
    r.s = r.c * t;
    // fp32 c, s leave c^2 + s^2 = 1 + O(eps); over the ~n * sweeps rotations a column sees that is a random
    // walk of its norm (measured 6e-6 relative at n = 96).  Fold the defect back in as low-order parts.
    const double defect = 1.0 - (double)r.c * (double)r.c - (double)r.s * (double)r.s;
    const float half = (float)(0.5 * defect);
    r.cl = r.c * half;
    r.sl = r.s * half;
    r.apply = true;
    return r;
}

// Orthogonalise LDS columns p and q (length rows_tot, dots over rows_dot) with a
// group of `width` lanes; `gl` is the lane's index inside its group.
__device__ __forceinline__ bool rotate_pair(float* __restrict__ cp, float* __restrict__ cq, int rows_dot,
                                            int rows_tot, int gl, int width, float tol) {
    float a = 0.f, b = 0.f, g = 0.f;
    for (int r = gl; r < rows_dot; r += width) {
        const float x = cp[r], y = cq[r];
        a = fmaf(x, x, a);
        b = fmaf(y, y, b);
        g = fmaf(x, y, g);
    }
    a = group_sum(a, width);
    b = group_sum(b, width);
    g = group_sum(g, width);
    const Rot rot = make_rotation(a, b, g, tol);
    if (!rot.apply) return false;
    for (int r = gl; r < rows_tot; r += width) {
        const float x = cp[r], y = cq[r];
        cp[r] = fmaf(rot.c, x, fmaf(-rot.s, y, fmaf(rot.cl, x, -rot.sl * y)));
        cq[r] = fmaf(rot.s, x, fmaf(rot.c, y, fmaf(rot.sl, x, rot.cl * y)));
    }
    return true;
}

// ---------------------------------------------------------------------------
// LDS-resident solver.  grid = batch, block = multiple of 64.
// ---------------------------------------------------------------------------
template <int LPP>
__global__ void __launch_bounds__(1024) jacobi_lds_kernel(float* __restrict__ W, long batch_stride, int rows_dot,
                                                           int rows_tot, int n_fixed, const int* __restrict__ n_arr,
                                                           int ld, int max_sweeps, float tol,
                                                           float* __restrict__ colnorm, int colnorm_stride,
                                                           int* __restrict__ sweeps_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int m = blockIdx.x;
    int n = n_arr ? n_arr[m] : n_fixed;
    int rd = rows_dot, rt = rows_tot;
    if (n_arr) {  // square problems of per-matrix order (principal angles): rows follow n
        rd = n;
        rt = n;
    }
    float* Wm = W + (long)m * batch_stride;
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (n <= 0) {
        if (sweeps_out && tid == 0) sweeps_out[m] = 0;
        return;
    }
    // global (ld = rows_tot) -> LDS (ld = `ld`, odd)
    for (int idx = tid; idx < n * rt; idx += nthr) {
        const int c = idx / rt, r = idx - c * rt;
        lds[c * ld + r] = Wm[(long)c * rows_tot + r];
    }
    __syncthreads();

    const int n_even = (n + 1) & ~1;
    const int groups = nthr / LPP, grp = tid / LPP, gl = tid % LPP;
    int sweep = 0;
    for (; sweep < max_sweeps && n > 1; ++sweep) {
        int rotated = 0;
        for (int r = 0; r < n_even - 1; ++r) {
            for (int t = grp; t < n_even / 2; t += groups) {
                int p, q;
                rr_pair(n_even, r, t, p, q);
                if (p >= n || q >= n) continue;  // padding column of an odd-order matrix
                if (p > q) { const int tmp = p; p = q; q = tmp; }
                rotated |= rotate_pair(lds + p * ld, lds + q * ld, rd, rt, gl, LPP, tol) ? 1 : 0;
            }
            __syncthreads();
        }
        if (!__syncthreads_or(rotated)) {
            ++sweep;
            break;
        }
    }
    if (sweeps_out && tid == 0) sweeps_out[m] = sweep;

    // column norms over the dot rows + write back
    for (int c = grp; c < n; c += groups) {
        float a = 0.f;
        for (int r = gl; r < rd; r += LPP) a = fmaf(lds[c * ld + r], lds[c * ld + r], a);
        a = group_sum(a, LPP);
        if (gl == 0) colnorm[(long)m * colnorm_stride + c] = sqrtf(a);
    }
    for (int idx = tid; idx < n * rt; idx += nthr) {
        const int c = idx / rt, r = idx - c * rt;
        Wm[(long)c * rows_tot + r] = lds[c * ld + r];
    }
}

// ---------------------------------------------------------------------------
// Block solver: one launch = one round-robin round over column blocks.
// grid = (nblk/2, batch), block = 64 * BW threads (one wave per column pair).
// flags[m * max_sweeps + s] != 0  <=>  some rotation was applied in sweep s.
// ---------------------------------------------------------------------------
template <int BW>
__global__ void __launch_bounds__(64 * BW) jacobi_block_round_kernel(float* __restrict__ W, long batch_stride,
                                                                     int rows_dot, int rows_tot, int n, int nblk,
                                                                     int round, int sweep, int max_sweeps, int ld,
                                                                     float tol, int* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int m = blockIdx.y;
    if (sweep > 0 && flags[m * max_sweeps + sweep - 1] == 0) return;  // converged in an earlier sweep
    int bi, bj;
    rr_pair(nblk, round, blockIdx.x, bi, bj);
    if (bi > bj) { const int t = bi; bi = bj; bj = t; }
    float* Wm = W + (long)m * batch_stride;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;

    // stage the 2*BW columns (zero-fill columns >= n)
    for (int idx = tid; idx < 2 * BW * rows_tot; idx += nthr) {
        const int lc = idx / rows_tot, r = idx - lc * rows_tot;
        const int gc = (lc < BW ? bi * BW + lc : bj * BW + (lc - BW));
        lds[lc * ld + r] = gc < n ? Wm[(long)gc * rows_tot + r] : 0.f;
    }
    __syncthreads();

    int rotated = 0;
    if (round == 0) {
        // pairs inside block I (waves 0..BW/2-1) and inside block J (waves BW/2..BW-1)
        const int half = wave / (BW / 2), t = wave % (BW / 2), base = half * BW;
        for (int r = 0; r < BW - 1; ++r) {
            int p, q;
            rr_pair(BW, r, t, p, q);
            if (p > q) { const int tmp = p; p = q; q = tmp; }
            const int gp = (half ? bj : bi) * BW + p, gq = (half ? bj : bi) * BW + q;
            if (gp < n && gq < n)
                rotated |= rotate_pair(lds + (base + p) * ld, lds + (base + q) * ld, rows_dot, rows_tot, lane, 64, tol);
            __syncthreads();
        }
    }
    // cross pairs: column `wave` of I with column (wave + r) % BW of J
    for (int r = 0; r < BW; ++r) {
        const int p = wave, q = (wave + r) % BW;
        const int gp = bi * BW + p, gq = bj * BW + q;
        if (gp < n && gq < n)
            rotated |= rotate_pair(lds + p * ld, lds + (BW + q) * ld, rows_dot, rows_tot, lane, 64, tol);
        __syncthreads();
    }
    if (__syncthreads_or(rotated) && tid == 0) atomicOr(&flags[m * max_sweeps + sweep], 1);

    for (int idx = tid; idx < 2 * BW * rows_tot; idx += nthr) {
        const int lc = idx / rows_tot, r = idx - lc * rows_tot;
        const int gc = (lc < BW ? bi * BW + lc : bj * BW + (lc - BW));
        if (gc < n) Wm[(long)gc * rows_tot + r] = lds[lc * ld + r];
    }
}

// column norms of a column-major batch (after the block solver). grid = (ceil(n/4), batch), block 256
__global__ void colnorm_kernel(const float* __restrict__ W, long batch_stride, int rows_dot, int rows_tot, int n,
                               float* __restrict__ colnorm, int colnorm_stride) {
    const int m = blockIdx.y, c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= n) return;
    const float* col = W + (long)m * batch_stride + (long)c * rows_tot;
    float a = 0.f;
    for (int r = lane; r < rows_dot; r += 64) a = fmaf(col[r], col[r], a);
    a = wave_sum(a);
    if (lane == 0) colnorm[(long)m * colnorm_stride + c] = sqrtf(a);
}

// ---------------------------------------------------------------------------
// Sort column norms (descending) and emit the leading normalised columns as ROWS.
// grid = batch, block = 256.  n <= 1024.
//   vals_desc : (batch, n)          sorted norms
//   vecs      : (batch, kmax, rows) row i = i-th column / its norm   (nullable)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sort_extract_kernel(const float* __restrict__ W, long batch_stride, int rows,
                                                            int rows_tot, int n, const float* __restrict__ colnorm,
                                                            int colnorm_stride, float* __restrict__ vals_desc,
                                                            float* __restrict__ vecs, int kmax) {
    __shared__ float key[1024];
    __shared__ int idx[1024];
    const int m = blockIdx.x, tid = threadIdx.x;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += 256) {
        key[i] = i < n ? colnorm[(long)m * colnorm_stride + i] : -1.f;
        idx[i] = i;
    }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const bool desc = (i & k) == 0;
                    const float a = key[i], b = key[l];
                    // ties broken by index so the order is deterministic
                    const bool a_first = a > b || (a == b && idx[i] < idx[l]);
                    if (desc ? !a_first : a_first) {
                        key[i] = b; key[l] = a;
                        const int t = idx[i]; idx[i] = idx[l]; idx[l] = t;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += 256) vals_desc[(long)m * n + i] = key[i];
    if (vecs) {
        const int kk = kmax < n ? kmax : n;
        const float* Wm = W + (long)m * batch_stride;
        for (int e = tid; e < kk * rows; e += 256) {
            const int i = e / rows, r = e - i * rows;
            const float nv = key[i];
            vecs[((long)m * kmax + i) * rows + r] = nv > 0.f ? Wm[(long)idx[i] * rows_tot + r] / nv : 0.f;
        }
    }
}

}  // namespace basd

using namespace basd;

static inline int lds_ld(int rows) { return rows | 1; }

extern "C" {

// Largest LDS the resident solver may use (bytes); leaves room for the runtime.
#define BASD_JACOBI_LDS_LIMIT (156 * 1024)

int basd_jacobi_workspace_ints(int batch, int max_sweeps) { return batch * max_sweeps; }

// One-sided Jacobi on `batch` column-major matrices (rows_tot x n, leading dim rows_tot).
//   n_arr (device, nullable): per-matrix order for square problems (rows = n_arr[m]); the
//   storage still uses rows_tot / batch_stride of the largest problem.
//   colnorm: (batch, colnorm_stride) column norms over the first rows_dot rows.
//   flags: device scratch of basd_jacobi_workspace_ints() ints (block path only, may be null
//   when the LDS path is taken).
int basd_jacobi_onesided(float* W, long batch_stride, int rows_dot, int rows_tot, int n, int batch,
                         const int* n_arr, float* colnorm, int colnorm_stride, int max_sweeps, int* flags,
                         int* sweeps_out, hipStream_t stream) {
    BASD_CHECK_ARG(W && colnorm && rows_dot > 0 && rows_tot >= rows_dot && n > 0 && batch > 0 && max_sweeps > 0);
    const float tol = 1.2e-7f * sqrtf((float)rows_dot);
    const int n_even = (n + 1) & ~1;
    const int ld = lds_ld(rows_tot);
    const size_t lds_bytes = (size_t)n_even * ld * sizeof(float);
    if (lds_bytes <= BASD_JACOBI_LDS_LIMIT) {
        // lanes per pair: enough lanes to cover the column, few enough that a round fits the block
        const int pairs = n_even / 2;
        int lpp = rows_tot >= 256 ? 64 : rows_tot >= 96 ? 32 : rows_tot >= 40 ? 16 : 8;
        int threads = pairs * lpp;
        threads = ((threads + 63) / 64) * 64;
        if (threads > 1024) threads = 1024;
        if (threads < 64) threads = 64;
#define LAUNCH_LDS(L)                                                                                         \
    (void)hipFuncSetAttribute((const void*)jacobi_lds_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                        BASD_JACOBI_LDS_LIMIT);                                                               \
    jacobi_lds_kernel<L><<<batch, threads, lds_bytes, stream>>>(W, batch_stride, rows_dot, rows_tot, n, n_arr, \
                                                                 ld, max_sweeps, tol, colnorm, colnorm_stride, \
                                                                 sweeps_out)
        if (lpp == 64) { LAUNCH_LDS(64); }
        else if (lpp == 32) { LAUNCH_LDS(32); }
        else if (lpp == 16) { LAUNCH_LDS(16); }
        else { LAUNCH_LDS(8); }
#undef LAUNCH_LDS
        BASD_RETURN_LAST();
    }
    BASD_CHECK_ARG(n_arr == nullptr && flags != nullptr);
    constexpr int BW = 16;
    int nblk = (n + BW - 1) / BW;
    nblk = (nblk + 1) & ~1;
    const size_t panel_bytes = (size_t)2 * BW * ld * sizeof(float);
    if (panel_bytes > BASD_JACOBI_LDS_LIMIT) return BASD_EUNSUPPORTED;
    hipError_t e = hipMemsetAsync(flags, 0, sizeof(int) * (size_t)batch * max_sweeps, stream);
    if (e != hipSuccess) return (int)e;
    (void)hipFuncSetAttribute((const void*)jacobi_block_round_kernel<BW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        BASD_JACOBI_LDS_LIMIT);
    for (int s = 0; s < max_sweeps; ++s)
        for (int r = 0; r < nblk - 1; ++r)
            jacobi_block_round_kernel<BW><<<dim3(nblk / 2, batch), 64 * BW, panel_bytes, stream>>>(
                W, batch_stride, rows_dot, rows_tot, n, nblk, r, s, max_sweeps, ld, tol, flags);
    colnorm_kernel<<<dim3((n + 3) / 4, batch), 256, 0, stream>>>(W, batch_stride, rows_dot, rows_tot, n, colnorm,
                                                                 colnorm_stride);
    BASD_RETURN_LAST();
}

// Sort norms descending; optionally emit the top-kmax normalised columns as rows.
int basd_sort_extract(const float* W, long batch_stride, int rows, int rows_tot, int n, int batch,
                      const float* colnorm, int colnorm_stride, float* vals_desc, float* vecs, int kmax,
                      hipStream_t stream) {
    BASD_CHECK_ARG(W && colnorm && vals_desc && n > 0 && n <= 1024 && batch > 0 && rows <= rows_tot);
    sort_extract_kernel<<<batch, 256, 0, stream>>>(W, batch_stride, rows, rows_tot, n, colnorm, colnorm_stride,
                                                   vals_desc, vecs, kmax);
    BASD_RETURN_LAST();
}

}  // extern "C"
