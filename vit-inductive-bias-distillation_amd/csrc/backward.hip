// Backward pieces that only multi-layer (ViT) teachers need: the gradient of the loss with respect to the
// soft layer-mixing weights (reference layer_selector.py:107-112 feeding relational.py) and, through the
// principal angles, with respect to the student tokens ("route (b)" of SURVEY.md section 3.2: autograd of
// layer_selector.py:86-105, i.e. svdvals -> acos -> weighted distance -> eigenvector perturbation of the
// student Gram).  With a single teacher layer these gradients are exactly zero and none of this runs.
#include "basd_common.h"

namespace basd {

// ---------------------------------------------------------------------------
// Teacher-side factor of the Procrustes gradient.  d loss_b / d T_c = 2 (Q - K'') T_c  with
//   Q   = I^T diag(w) I                       (trace term, on the core grid)
//   K'' = Z Sigma^+ Z^T,  Z = L_a U           (nuclear-norm term; U Sigma = rotated top half of W)
// grid = batch, block = 256.  Also emits |t_hat_c[s]|^2 per student token (for d loss / d omega).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) teacher_factor_kernel(
    const float* __restrict__ W, long w_batch_stride, const float* __restrict__ sigma, int n, int n_s,
    const double* __restrict__ La, const double* __restrict__ Gb, long g_batch_stride,
    const float* __restrict__ omega, const int* __restrict__ tap0, const int* __restrict__ tap1,
    const float* __restrict__ lam, const int* __restrict__ range0, const int* __restrict__ range1,
    float* __restrict__ Kt, float* __restrict__ tnorm2) {
    extern __shared__ float sm[];
    float* s15 = sm;          // n : sigma^-1.5 or 0
    float* Zs = sm + n;       // n*n, column-major: Zs[c*n + a] = z_c[a] * sigma_c^-1/2
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* sg = sigma + (long)b * n;
    float smax = 0.f;
    for (int j = tid; j < n; j += 256) smax = fmaxf(smax, sg[j]);
    smax = wave_max(smax);
    if ((tid & 63) == 0) red[tid >> 6] = smax;
    __syncthreads();
    smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float thr = smax * (float)n * 1.1920929e-7f;
    for (int j = tid; j < n; j += 256) {
        const float s = sg[j];
        s15[j] = s > thr ? 1.f / (s * sqrtf(s)) : 0.f;
    }
    __syncthreads();
    const float* Wb = W + (long)b * w_batch_stride;
    const double* L = La + (long)b * g_batch_stride;
    for (int idx = tid; idx < n * n; idx += 256) {
        const int c = idx / n, a = idx - c * n;
        const float* col = Wb + (long)c * 2 * n;
        float acc = 0.f;
        for (int r = 0; r <= a; ++r) acc = fmaf((float)L[(long)a * n + r], col[r], acc);
        Zs[idx] = acc * s15[c];
    }
    __syncthreads();
    const float* w = omega + (long)b * n_s;
    float* K = Kt + (long)b * n * n;
    for (int idx = tid; idx < n * n; idx += 256) {
        const int a = idx / n, bb = idx - a * n;
        float q = 0.f;
        if (tap0) {
            const int s0 = max(range0[a], range0[bb]), s1 = min(range1[a], range1[bb]);
            for (int s = s0; s < s1; ++s) {
                const float l1 = lam[s];
                const float ca = (tap0[s] == a ? 1.f - l1 : 0.f) + (tap1[s] == a ? l1 : 0.f);
                const float cb = (tap0[s] == bb ? 1.f - l1 : 0.f) + (tap1[s] == bb ? l1 : 0.f);
                q = fmaf(w[s] * ca, cb, q);
            }
        } else if (a == bb) {
            q = w[a];
        }
        float acc = 0.f;
        for (int c = 0; c < n; ++c) acc = fmaf(Zs[c * n + a], Zs[c * n + bb], acc);
        K[idx] = q - acc;
    }
    const double* G = Gb + (long)b * g_batch_stride;
    for (int s = tid; s < n_s; s += 256) {
        double v;
        if (tap0) {
            const int i0 = tap0[s], i1 = tap1[s];
            const double l1 = (double)lam[s], l0 = 1. - l1;
            v = l0 * l0 * G[(long)i0 * n + i0] + 2. * l0 * l1 * G[(long)i0 * n + i1] + l1 * l1 * G[(long)i1 * n + i1];
        } else {
            v = G[(long)s * n + s];
        }
        tnorm2[(long)b * n_s + s] = (float)v;
    }
}

// The same for cores past LDS (n > 199): Z = L_a (U Sigma) Sigma^-1.5 goes through a scratch array (batch, n, n) in
// global memory, both contractions are tiled 32 x 32.
//   teacher_z_tiled_kernel:  Zs[c * n + a] = sigma_c^-1.5 sum_{r <= a} L[a][r] W[c][r]      grid = (n/32 (a), n/32 (c), batch)
//   teacher_kt_tiled_kernel: Kt[a][b] = Q[a][b] - sum_c Zs[c][a] Zs[c][b]                   grid = (n/32 (b), n/32 (a), batch)
__global__ void __launch_bounds__(256) teacher_z_tiled_kernel(const float* __restrict__ W, long w_batch_stride,
                                                              const float* __restrict__ sigma, int n,
                                                              const double* __restrict__ La, long g_batch_stride,
                                                              float* __restrict__ Z) {
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
    const double* L = La + (long)b * g_batch_stride;
    const int tx = tid & 31, ty = tid >> 5;        // tx: a within the tile, ty + 8 i: c within the tile
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < a0 + 32 && r0 < n; r0 += 32) {      // L is lower triangular: r <= a
        __syncthreads();
        for (int idx = tid; idx < 1024; idx += 256) {
            const int i = idx >> 5, rr = idx & 31, r = r0 + rr;
            lt[i][rr] = (a0 + i < n && r <= a0 + i) ? (float)L[(long)(a0 + i) * n + r] : 0.f;
            wt[i][rr] = (c0 + i < n && r < n) ? Wb[(long)(c0 + i) * 2 * n + r] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int rr = 0; rr < 32; ++rr) {
            const float lv = lt[tx][rr];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(lv, wt[ty + 8 * i][rr], acc[i]);
        }
    }
    float* Zb = Z + (long)b * n * n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, a = a0 + tx;
        if (c < n && a < n) {
            const float sv = sg[c];
            Zb[(long)c * n + a] = acc[i] * (sv > thr ? 1.f / (sv * sqrtf(sv)) : 0.f);
        }
    }
}

__global__ void __launch_bounds__(256) teacher_kt_tiled_kernel(
    const float* __restrict__ Z, int n, int n_s, const double* __restrict__ Gb, long g_batch_stride,
    const float* __restrict__ omega, const int* __restrict__ tap0, const int* __restrict__ tap1,
    const float* __restrict__ lam, const int* __restrict__ range0, const int* __restrict__ range1,
    float* __restrict__ Kt, float* __restrict__ tnorm2) {
    __shared__ float za[32][33], zb[32][33];      // [c][a], [c][b]
    const int b = blockIdx.z, a0 = blockIdx.y * 32, b0 = blockIdx.x * 32, tid = threadIdx.x;
    const float* Zb = Z + (long)b * n * n;
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
    const float* w = omega + (long)b * n_s;
    float* K = Kt + (long)b * n * n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int a = a0 + ty + 8 * i, bb = b0 + tx;
        if (a >= n || bb >= n) continue;
        float q = 0.f;
        if (tap0) {
            const int s0 = max(range0[a], range0[bb]), s1 = min(range1[a], range1[bb]);
            for (int s = s0; s < s1; ++s) {
                const float l1 = lam[s];
                const float ca = (tap0[s] == a ? 1.f - l1 : 0.f) + (tap1[s] == a ? l1 : 0.f);
                const float cb = (tap0[s] == bb ? 1.f - l1 : 0.f) + (tap1[s] == bb ? l1 : 0.f);
                q = fmaf(w[s] * ca, cb, q);
            }
        } else if (a == bb) {
            q = w[a];
        }
        K[(long)a * n + bb] = q - acc[i];
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        const double* G = Gb + (long)b * g_batch_stride;
        for (int s = tid; s < n_s; s += 256) {
            double v;
            if (tap0) {
                const int i0 = tap0[s], i1 = tap1[s];
                const double l1 = (double)lam[s], l0 = 1. - l1;
                v = l0 * l0 * G[(long)i0 * n + i0] + 2. * l0 * l1 * G[(long)i0 * n + i1] + l1 * l1 * G[(long)i1 * n + i1];
            } else {
                v = G[(long)s * n + s];
            }
            tnorm2[(long)b * n_s + s] = (float)v;
        }
    }
}

// ---------------------------------------------------------------------------
// partial[e][b][l] = < R[e][b] , That_l[b] >   (R = (Q - K'') T_c on the core grid of n tokens;
// That_l = teacher layer l on that grid: gathered with (g0, g1, glam) when the teacher grid is finer).
// grid = (L, B, E), block = 256.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) mix_grad_tokens_kernel(const float* __restrict__ R,
                                                              const void* const* __restrict__ tok_ptrs, long sb,
                                                              long sn, long sd, int n, int D,
                                                              const int* __restrict__ g0, const int* __restrict__ g1,
                                                              const float* __restrict__ glam,
                                                              float* __restrict__ partial) {
    __shared__ float red[32];
    const int l = blockIdx.x, b = blockIdx.y, e = blockIdx.z, B = gridDim.y, L = gridDim.x;
    const float* r = R + ((long)e * B + b) * n * D;
    const T* t = (const T*)tok_ptrs[l] + (long)b * sb;
    float acc = 0.f;
    for (int idx = threadIdx.x; idx < n * D; idx += 256) {
        const int j = idx / D, d = idx - j * D;
        float tv;
        if (g0) {
            const float l1 = glam[j];
            tv = (1.f - l1) * to_f32(t[(long)g0[j] * sn + (long)d * sd]) + l1 * to_f32(t[(long)g1[j] * sn + (long)d * sd]);
        } else {
            tv = to_f32(t[(long)j * sn + (long)d * sd]);
        }
        acc = fmaf(r[idx], tv, acc);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partial[((long)e * B + b) * L + l] = acc;
}

// The same in ONE pass over the teacher layers for all E <= 4 extraction layers (the kernel above reads every layer
// once per extraction layer and R once per teacher layer: 20 GB through L2 per step at cfg-4 for 2.9 GB of operands).
// A workgroup owns MG_ROWS token rows of one sample and walks the layers MG_LT at a time with E x MG_LT accumulators per
// thread; per-chunk sums go to `chunk_part` (chunks, E, B, L) and are folded in a fixed order by mix_grad_fold_kernel.
// grid = (chunks, B), block = 256.
constexpr int MG_ROWS = 16, MG_LT = 8, MG_E = 4;
template <typename T>
__global__ void __launch_bounds__(256) mix_grad_tokens_onepass_kernel(
    const float* __restrict__ R, const void* const* __restrict__ tok_ptrs, long sb, long sn, long sd, int E, int L, int n,
    int D, const int* __restrict__ g0, const int* __restrict__ g1, const float* __restrict__ glam,
    float* __restrict__ chunk_part) {
    __shared__ float red[4][MG_E * MG_LT];
    const int chunk = blockIdx.x, b = blockIdx.y, B = gridDim.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j_lo = chunk * MG_ROWS, j_hi = j_lo + MG_ROWS < n ? j_lo + MG_ROWS : n;
    const int count = (j_hi - j_lo) * D;
    for (int l0 = 0; l0 < L; l0 += MG_LT) {
        float acc[MG_E][MG_LT];
#pragma unroll
        for (int e = 0; e < MG_E; ++e)
#pragma unroll
            for (int u = 0; u < MG_LT; ++u) acc[e][u] = 0.f;
        const BASD_GLOBAL_AS T* tp[MG_LT];
#pragma unroll
        for (int u = 0; u < MG_LT; ++u)
            tp[u] = (const BASD_GLOBAL_AS T*)tok_ptrs[l0 + u < L ? l0 + u : L - 1] + (long)b * sb;
        for (int idx = tid; idx < count; idx += 256) {
            const int jj = idx / D, d = idx - jj * D, j = j_lo + jj;
            float rv[MG_E];
#pragma unroll
            for (int e = 0; e < MG_E; ++e) rv[e] = e < E ? R[(((long)e * B + b) * n + j) * D + d] : 0.f;
            long o0 = (long)j * sn + (long)d * sd, o1 = o0;
            float l1 = 0.f;
            if (g0) {
                o0 = (long)g0[j] * sn + (long)d * sd;
                o1 = (long)g1[j] * sn + (long)d * sd;
                l1 = glam[j];
            }
#pragma unroll
            for (int u = 0; u < MG_LT; ++u) {
                float tv = ldg_f32(tp[u] + o0);
                if (g0) tv = (1.f - l1) * tv + l1 * ldg_f32(tp[u] + o1);
#pragma unroll
                for (int e = 0; e < MG_E; ++e) acc[e][u] = fmaf(rv[e], tv, acc[e][u]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < MG_E; ++e)
#pragma unroll
            for (int u = 0; u < MG_LT; ++u) {
                float v = acc[e][u];
                for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
                if (lane == 0) red[wave][e * MG_LT + u] = v;
            }
        __syncthreads();
        if (tid < MG_E * MG_LT) {
            const int e = tid / MG_LT, u = tid - e * MG_LT;
            if (e < E && l0 + u < L)
                chunk_part[(((long)chunk * E + e) * B + b) * L + l0 + u] =
                    (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        }
    }
}

__global__ void __launch_bounds__(256) mix_grad_fold_kernel(const float* __restrict__ chunk_part, int chunks, long count,
                                                            float* __restrict__ partial) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    float acc = 0.f;
    for (int c = 0; c < chunks; ++c) acc += chunk_part[(long)c * count + i];
    partial[i] = acc;
}

// ---------------------------------------------------------------------------
// Chain d loss / d omega back to the mixing weights through relational.py:22-34 and layer_selector.py:112:
// normalisation, weight interpolation (n_a -> n_s), head / query mean, layer mix.
// partial[e][b][l] = sum_j g_raw[j] * rowmean_l[b][j].     grid = (B, E), block = 256.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) token_weight_bwd_kernel(
    const float* __restrict__ gomega, const float* __restrict__ raw, int n_a, int n_s,
    const int* __restrict__ atap0, const int* __restrict__ atap1, const float* __restrict__ alam,
    const int* __restrict__ arange0, const int* __restrict__ arange1, const void* const* __restrict__ attn_ptrs,
    int L, long sb, long sh, long sq, long sk, int H, int A, int has_cls, float* __restrict__ partial,
    float* __restrict__ graw_out) {
    extern __shared__ float sm[];
    float* gv = sm;             // n_s
    float* graw = sm + n_s;     // n_a
    __shared__ float red[32];
    const int b = blockIdx.x, e = blockIdx.y, B = gridDim.x, tid = threadIdx.x;
    const float* rw = raw + ((long)e * B + b) * n_a;
    const float* go = gomega + ((long)e * B + b) * n_s;
    float tot = 0.f;
    for (int s = tid; s < n_s; s += 256) {
        float v;
        if (atap0) {
            const float l1 = alam[s];
            v = (1.f - l1) * rw[atap0[s]] + l1 * rw[atap1[s]];
        } else {
            v = rw[s];
        }
        gv[s] = v;
        tot += v;
    }
    const float total = block_sum(tot, red);
    float dotp = 0.f;
    for (int s = tid; s < n_s; s += 256) dotp = fmaf(go[s], gv[s] / total, dotp);
    const float c = block_sum(dotp, red);
    for (int s = tid; s < n_s; s += 256) gv[s] = (go[s] - c) / total;      // d / d (interpolated raw weight)
    __syncthreads();
    for (int j = tid; j < n_a; j += 256) {
        float acc = 0.f;
        if (atap0) {
            for (int s = arange0[j]; s < arange1[j]; ++s) {
                const float l1 = alam[s];
                acc = fmaf((atap0[s] == j ? 1.f - l1 : 0.f) + (atap1[s] == j ? l1 : 0.f), gv[s], acc);
            }
        } else {
            acc = gv[j];
        }
        graw[j] = acc;
        if (graw_out) graw_out[((long)e * B + b) * n_a + j] = acc;     // d loss_b / d (raw attention-grid weight)
    }
    __syncthreads();
    for (int l = 0; l < L; ++l) {
        const T* a = (const T*)attn_ptrs[l] + (long)b * sb;
        float acc = 0.f;
        for (int j = tid; j < n_a; j += 256) {
            float m = 0.f;
            if (has_cls) {
                for (int h = 0; h < H; ++h) m += to_f32(a[h * sh + (long)(1 + j) * sk]);
                m /= (float)H;
            } else {
                for (int h = 0; h < H; ++h)
                    for (int q = 0; q < A; ++q) m += to_f32(a[h * sh + q * sq + (long)j * sk]);
                m /= (float)(H * A);
            }
            acc = fmaf(graw[j], m, acc);
        }
        acc = block_sum(acc, red);
        if (tid == 0) partial[((long)e * B + b) * L + l] = acc;
    }
}

// ---------------------------------------------------------------------------
// Stack the (masked) cosine matrix on top of an identity so that the Jacobi solver accumulates the right
// singular vectors:  out[item] is column-major (2 kmax x kmax): column c = [cos[item][c][0..k) , 0.. ; e_c].
// grid = items, block = 256.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) build_angle_stack_kernel(const float* __restrict__ cos, int kmax,
                                                                const int* __restrict__ k_arr,
                                                                float* __restrict__ out) {
    const int m = blockIdx.x, k = k_arr[m];
    const float* cm = cos + (long)m * kmax * kmax;
    float* o = out + (long)m * 2 * kmax * kmax;
    for (int idx = threadIdx.x; idx < 2 * kmax * kmax; idx += 256) {
        const int c = idx / (2 * kmax), r = idx - c * 2 * kmax;
        float v;
        if (r < kmax) v = (c < k && r < k) ? cm[(long)c * kmax + r] : 0.f;
        else v = (r - kmax == c) ? 1.f : 0.f;
        o[idx] = v;
    }
}

// ---------------------------------------------------------------------------
// d (d_grass_sq) / d W for one (student layer, teacher layer) item, W = Vt_s[:k] U_t  (k x k):
//   W = V_A Sigma U_A^T  (the solver saw A = W^T),   dd/dW = V_A diag(f') U_A^T,
//   f'_p = g * sw_p * (-2 theta_p / sqrt(1 - s_p^2)) / sum(sw)   for s_p < 1 - eps, else 0 (the clamp).
// Emits gWt[item][j][i] = dd/dW[i][j] (kmax x kmax, zero outside k x k).   grid = items, block = 256.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) grassmann_distance_bwd_kernel(
    const float* __restrict__ stack, const float* __restrict__ colnorm, int kmax, const int* __restrict__ k_arr,
    const float* __restrict__ sw, int sw_stride, const int* __restrict__ sw_index, const float* __restrict__ gd,
    float* __restrict__ gWt) {
    __shared__ float key[1024];
    __shared__ int idx[1024];
    __shared__ float coef[1024];
    __shared__ float red[32];
    const int m = blockIdx.x, tid = threadIdx.x, k = k_arr[m];
    int np2 = 1;
    while (np2 < kmax) np2 <<= 1;
    for (int i = tid; i < np2; i += 256) {
        key[i] = i < kmax ? colnorm[(long)m * kmax + i] : -1.f;
        idx[i] = i;
        coef[i] = 0.f;
    }
    __syncthreads();
    for (int kk = 2; kk <= np2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const bool desc = (i & kk) == 0;
                    const float a = key[i], b = key[l];
                    const bool a_first = a > b || (a == b && idx[i] < idx[l]);
                    if (desc ? !a_first : a_first) {
                        key[i] = b; key[l] = a;
                        const int t = idx[i]; idx[i] = idx[l]; idx[l] = t;
                    }
                }
            }
            __syncthreads();
        }
    const float* w = sw + (long)sw_index[m] * sw_stride;
    float den = 0.f;
    for (int i = tid; i < k; i += 256) den += w[i];
    den = block_sum(den, red);
    const float g = gd[m];
    for (int p = tid; p < k; p += 256) {
        const float s = key[p];
        const float lim = 1.f - 1.1920929e-7f;
        float f = 0.f;
        if (s < lim && s > 1e-20f) {
            const float th = acosf(s);
            f = g * w[p] * (-2.f * th / sqrtf(1.f - s * s)) / den;
            f /= s;                       // U_A column = rotated top column / sigma
        }
        coef[idx[p]] = f;
    }
    __syncthreads();
    const float* st = stack + (long)m * 2 * kmax * kmax;
    float* out = gWt + (long)m * kmax * kmax;
    for (int e = tid; e < kmax * kmax; e += 256) {
        const int j = e / kmax, i = e - j * kmax;
        float acc = 0.f;
        if (i < k && j < k)
            for (int c = 0; c < kmax; ++c) {
                const float* col = st + (long)c * 2 * kmax;
                acc = fmaf(coef[c] * col[j], col[kmax + i], acc);
            }
        out[e] = acc;
    }
}

// ---------------------------------------------------------------------------
// Eigenvector-perturbation kernel of a symmetric matrix S = V diag(lam) V^T (lam descending):
// for a loss with d/dV = G (only the first kmax columns non-zero) and M = V^T G (D x kmax),
//   d/dS + (d/dS)^T = V K2 V^T,   K2[i][j] = (M[i][j] - M[j][i]) / (lam_j - lam_i),  K2[i][i] = 0.
// grid = (ceil(D*D/256), batch), block = 256.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) eigvec_k2_kernel(const float* __restrict__ M, const float* __restrict__ lam,
                                                        int D, int kmax, float* __restrict__ K2) {
    const int z = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= D * D) return;
    const int i = e / D, j = e - i * D;
    const float* Mz = M + (long)z * D * kmax;
    const float* lz = lam + (long)z * D;
    float v = 0.f;
    if (i != j && (i < kmax || j < kmax)) {
        const float mij = j < kmax ? Mz[(long)i * kmax + j] : 0.f;
        const float mji = i < kmax ? Mz[(long)j * kmax + i] : 0.f;
        const float den = lz[j] - lz[i];
        if (fabsf(den) > 1e-6f * lz[0]) v = (mij - mji) / den;
    }
    K2[(long)z * D * D + e] = v;
}

}  // namespace basd

using namespace basd;

extern "C" {

// Kt = Q - K'' (teacher-side gradient factor, see kernel) and |t_hat_c|^2 per student token.
int basd_teacher_factor(const float* w, long w_batch_stride, const float* sigma, int n, int n_s, int batch,
                        const double* la, const double* gb, long g_batch_stride, const float* omega,
                        const int* tap0, const int* tap1, const float* lam, const int* range0, const int* range1,
                        float* kt, float* tnorm2, hipStream_t stream) {
    BASD_CHECK_ARG(w && sigma && la && gb && omega && kt && tnorm2 && n > 0 && n_s > 0 && batch > 0);
    const size_t lds = sizeof(float) * ((size_t)n + (size_t)n * n);
    if (lds > 156 * 1024) return BASD_EUNSUPPORTED;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)teacher_factor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    teacher_factor_kernel<<<batch, 256, lds, stream>>>(w, w_batch_stride, sigma, n, n_s, la, gb, g_batch_stride, omega,
                                                       tap0, tap1, lam, range0, range1, kt, tnorm2);
    BASD_RETURN_LAST();
}

// basd_teacher_factor for cores past LDS: z_scratch (batch, n, n) floats of device scratch.
int basd_teacher_factor_tiled(const float* w, long w_batch_stride, const float* sigma, int n, int n_s, int batch,
                              const double* la, const double* gb, long g_batch_stride, const float* omega,
                              const int* tap0, const int* tap1, const float* lam, const int* range0, const int* range1,
                              float* kt, float* tnorm2, float* z_scratch, hipStream_t stream) {
    BASD_CHECK_ARG(w && sigma && la && gb && omega && kt && tnorm2 && z_scratch && n > 0 && n_s > 0 && batch > 0);
    BASD_CHECK_ARG(batch <= 65535);
    const int nt = (n + 31) / 32;
    teacher_z_tiled_kernel<<<dim3(nt, nt, batch), 256, 0, stream>>>(w, w_batch_stride, sigma, n, la, g_batch_stride, z_scratch);
    teacher_kt_tiled_kernel<<<dim3(nt, nt, batch), 256, 0, stream>>>(z_scratch, n, n_s, gb, g_batch_stride, omega, tap0,
                                                                     tap1, lam, range0, range1, kt, tnorm2);
    BASD_RETURN_LAST();
}

// partial[e][b][l] = <R[e][b], teacher layer l on the core grid>.   R: (E, B, n, D) fp32.
int basd_mix_grad_tokens(const float* r, const void* const* tok_ptrs, int dtype, int L, long sb, long sn, long sd,
                         int E, int B, int n, int D, const int* g0, const int* g1, const float* glam, float* partial,
                         hipStream_t stream) {
    BASD_CHECK_ARG(r && tok_ptrs && partial && L > 0 && E > 0 && B > 0 && n > 0 && D > 0);
    const dim3 grid(L, B, E);
    if (dtype == BASD_DTYPE_F32)
        mix_grad_tokens_kernel<float><<<grid, 256, 0, stream>>>(r, tok_ptrs, sb, sn, sd, n, D, g0, g1, glam, partial);
    else if (dtype == BASD_DTYPE_BF16)
        mix_grad_tokens_kernel<__hip_bfloat16><<<grid, 256, 0, stream>>>(r, tok_ptrs, sb, sn, sd, n, D, g0, g1, glam, partial);
    else
        return BASD_EINVAL;
    BASD_RETURN_LAST();
}

// The same in one pass over the teacher layers (E <= 4).  scratch: basd_mix_grad_tokens_scratch_floats(E, B, L, n) floats.
long basd_mix_grad_tokens_scratch_floats(int E, int B, int L, int n) {
    if (E <= 0 || B <= 0 || L <= 0 || n <= 0) return 0;
    return (long)((n + MG_ROWS - 1) / MG_ROWS) * E * B * L;
}
int basd_mix_grad_tokens_onepass(const float* r, const void* const* tok_ptrs, int dtype, int L, long sb, long sn, long sd,
                                 int E, int B, int n, int D, const int* g0, const int* g1, const float* glam,
                                 float* partial, float* scratch, hipStream_t stream) {
    BASD_CHECK_ARG(r && tok_ptrs && partial && scratch && L > 0 && E > 0 && B > 0 && n > 0 && D > 0);
    if (E > MG_E || B > 65535) return BASD_EUNSUPPORTED;
    const int chunks = (n + MG_ROWS - 1) / MG_ROWS;
    const dim3 grid(chunks, B);
    if (dtype == BASD_DTYPE_F32)
        mix_grad_tokens_onepass_kernel<float><<<grid, 256, 0, stream>>>(r, tok_ptrs, sb, sn, sd, E, L, n, D, g0, g1, glam, scratch);
    else if (dtype == BASD_DTYPE_BF16)
        mix_grad_tokens_onepass_kernel<__hip_bfloat16><<<grid, 256, 0, stream>>>(r, tok_ptrs, sb, sn, sd, E, L, n, D, g0, g1, glam, scratch);
    else
        return BASD_EINVAL;
    const long count = (long)E * B * L;
    mix_grad_fold_kernel<<<(unsigned)((count + 255) / 256), 256, 0, stream>>>(scratch, chunks, count, partial);
    BASD_RETURN_LAST();
}

// partial[e][b][l] = d loss / d mix_l through the token weights.  gomega: (E, B, n_s); raw: (E, B, n_a).
int basd_token_weight_bwd(const float* gomega, const float* raw, int E, int B, int n_a, int n_s, const int* atap0,
                          const int* atap1, const float* alam, const int* arange0, const int* arange1,
                          const void* const* attn_ptrs, int dtype, int L, long sb, long sh, long sq, long sk, int H,
                          int A, int has_cls, float* partial, float* graw_out, hipStream_t stream) {
    BASD_CHECK_ARG(gomega && raw && attn_ptrs && partial && E > 0 && B > 0 && n_a > 0 && n_s > 0 && L > 0);
    BASD_CHECK_ARG((n_a == n_s) == (atap0 == nullptr));
    const size_t lds = sizeof(float) * (size_t)(n_s + n_a);
    const dim3 grid(B, E);
    if (dtype == BASD_DTYPE_F32)
        token_weight_bwd_kernel<float><<<grid, 256, lds, stream>>>(gomega, raw, n_a, n_s, atap0, atap1, alam, arange0, arange1, attn_ptrs, L, sb, sh, sq, sk, H, A, has_cls, partial, graw_out);
    else if (dtype == BASD_DTYPE_BF16)
        token_weight_bwd_kernel<__hip_bfloat16><<<grid, 256, lds, stream>>>(gomega, raw, n_a, n_s, atap0, atap1, alam, arange0, arange1, attn_ptrs, L, sb, sh, sq, sk, H, A, has_cls, partial, graw_out);
    else
        return BASD_EINVAL;
    BASD_RETURN_LAST();
}

int basd_build_angle_stack(const float* cos, int kmax, const int* k_arr, int items, float* out, hipStream_t stream) {
    BASD_CHECK_ARG(cos && k_arr && out && kmax > 0 && items > 0);
    build_angle_stack_kernel<<<items, 256, 0, stream>>>(cos, kmax, k_arr, out);
    BASD_RETURN_LAST();
}

int basd_grassmann_distance_bwd(const float* stack, const float* colnorm, int kmax, const int* k_arr, const float* sw,
                                int sw_stride, const int* sw_index, const float* gd, int items, float* gwt,
                                hipStream_t stream) {
    BASD_CHECK_ARG(stack && colnorm && k_arr && sw && sw_index && gd && gwt && kmax > 0 && kmax <= 1024 && items > 0);
    grassmann_distance_bwd_kernel<<<items, 256, 0, stream>>>(stack, colnorm, kmax, k_arr, sw, sw_stride, sw_index, gd, gwt);
    BASD_RETURN_LAST();
}

int basd_eigvec_k2(const float* m, const float* lam, int D, int kmax, int batch, float* k2, hipStream_t stream) {
    BASD_CHECK_ARG(m && lam && k2 && D > 0 && kmax > 0 && kmax <= D && batch > 0);
    eigvec_k2_kernel<<<dim3((D * D + 255) / 256, batch), 256, 0, stream>>>(m, lam, D, kmax, k2);
    BASD_RETURN_LAST();
}

}  // extern "C"
