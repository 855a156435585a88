// Grassmannian layer-selector epilogues (reference src/losses/layer_selector.py).
//   mp_rank_kernel          : lower median, Marchenko-Pastur threshold, strict count  (:16-19, :74)
//   grassmann_distance_kernel: sigma -> theta = acos(min(sigma, 1 - eps)) ->
//                              d = sum(sw * theta^2) / sum(sw)                        (:99-105)
#include "basd_common.h"
#include "../../include/basd_hip.h"

namespace basd {

// grid = batch, block = 256.  vals_desc: (batch, n) eigenvalues sorted descending.
// `factor` = (1 + sqrt(D/M))^2 evaluated on the host in float64 exactly as the reference does;
// the threshold is rounded to fp32 before the comparison, as torch does for tensor-vs-scalar `>`.
__global__ void __launch_bounds__(256) mp_rank_kernel(const float* __restrict__ vals_desc, int n, double factor,
                                                      int cap, int* __restrict__ rank_out,
                                                      float* __restrict__ thr_out) {
    __shared__ int red[32];
    const int m = blockIdx.x, tid = threadIdx.x;
    const float* v = vals_desc + (long)m * n;
    const float sigma2 = v[n - 1 - (n - 1) / 2];          // ascending index (n-1)/2 = lower median
    const float lam = (float)((double)sigma2 * factor);
    int cnt = 0;
    for (int i = tid; i < n; i += 256) cnt += v[i] > lam ? 1 : 0;
    cnt = block_sum(cnt, red);
    if (tid == 0) {
        rank_out[m] = cnt < cap ? cnt : cap;
        if (thr_out) thr_out[m] = lam;
    }
}

// grid = items, block = 256.  colnorm: (items, stride) unsorted singular values of the k x k
// cosine matrix; sw: spectral weights of the item's teacher layer (descending), k = k_arr[item].
__global__ void __launch_bounds__(256) grassmann_distance_kernel(const float* __restrict__ colnorm, int stride,
                                                                 const int* __restrict__ k_arr,
                                                                 const float* __restrict__ sw, int sw_stride,
                                                                 const int* __restrict__ sw_index,
                                                                 float* __restrict__ d_out,
                                                                 float* __restrict__ theta_out, int n_valid) {
    __shared__ float key[1024];
    __shared__ float red[32];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int k = k_arr[m];
    // n_valid > 0: the item's matrix was zero-padded to a common order -- sort all n_valid values, use the leading k
    const int nv = n_valid > 0 ? n_valid : k;
    int np2 = 1;
    while (np2 < nv) np2 <<= 1;
    for (int i = tid; i < np2; i += 256) key[i] = i < nv ? colnorm[(long)m * stride + i] : -1.f;
    __syncthreads();
    for (int kk = 2; kk <= np2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const bool desc = (i & kk) == 0;
                    const float a = key[i], b = key[l];
                    if (desc ? a < b : a > b) { key[i] = b; key[l] = a; }
                }
            }
            __syncthreads();
        }
    const float* w = sw + (long)sw_index[m] * sw_stride;
    float num = 0.f, den = 0.f;
    for (int i = tid; i < k; i += 256) {
        const float c = fminf(key[i], 1.f - 1.1920929e-7f);
        const float th = acosf(c);
        if (theta_out) theta_out[(long)m * stride + i] = th;
        num = fmaf(w[i], th * th, num);
        den += w[i];
    }
    num = block_sum(num, red);
    den = block_sum(den, red);
    if (tid == 0) d_out[m] = num / den;   // k == 0 -> 0/0 = NaN, as in the reference
}

// The two Grams of the projected teacher tokens z = t P^T from ONE Gram in the teacher's own space (layer_selector.py:72
// then :13 / :35, without ever forming z): c = P G_c P^T with G_c the centred Gram of t, zbar = P tbar.
//   centred   (z - 1 zbar^T)^T (z - 1 zbar^T) = P G_c P^T                      -> out_c (symmetrised: c is a product
//             of two fp32 GEMMs and only symmetric to round-off)
//   uncentred z^T z / M = (P G_c P^T + M zbar zbar^T) / M                      -> out_u (an addition: no cancellation)
// grid = (ceil(n n / 256), L), block = 256.
__global__ void __launch_bounds__(256) gram_finish_kernel(const float* __restrict__ c, const float* __restrict__ zbar,
                                                          int n, float m_rows, float* __restrict__ out_u,
                                                          float* __restrict__ out_c) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)n * n) return;
    const int l = blockIdx.y, i = (int)(idx / n), j = (int)(idx - (long)i * n);
    const float* cl = c + (long)l * n * n;
    const float sym = 0.5f * (cl[(long)i * n + j] + cl[(long)j * n + i]);
    const float zi = zbar[(long)l * n + i], zj = zbar[(long)l * n + j];
    if (out_c) out_c[(long)l * n * n + idx] = sym;
    if (out_u) out_u[(long)l * n * n + idx] = fmaf(m_rows * zi, zj, sym) / m_rows;
}

// sw[i] = sqrt(max(lambda_i, 0)): singular values of the centred data from Gram eigenvalues.
__global__ void sqrt_clamp_kernel(const float* __restrict__ in, float* __restrict__ out, long count) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = sqrtf(fmaxf(in[i], 0.f));
}

}  // namespace basd

using namespace basd;

extern "C" {

int basd_mp_rank(const float* vals_desc, int n, int batch, double factor, int cap, int* rank_out, float* thr_out,
                 hipStream_t stream) {
    BASD_CHECK_ARG(vals_desc && rank_out && n > 0 && batch > 0);
    mp_rank_kernel<<<batch, 256, 0, stream>>>(vals_desc, n, factor, cap, rank_out, thr_out);
    BASD_RETURN_LAST();
}

int basd_grassmann_distance(const float* colnorm, int stride, const int* k_arr, const float* sw, int sw_stride,
                            const int* sw_index, int items, float* d_out, float* theta_out, hipStream_t stream) {
    BASD_CHECK_ARG(colnorm && k_arr && sw && sw_index && d_out && items > 0 && stride > 0 && stride <= 1024);
    grassmann_distance_kernel<<<items, 256, 0, stream>>>(colnorm, stride, k_arr, sw, sw_stride, sw_index, d_out, theta_out, 0);
    BASD_RETURN_LAST();
}

int basd_grassmann_distance_padded(const float* colnorm, int stride, int n_valid, const int* k_arr, const float* sw,
                                   int sw_stride, const int* sw_index, int items, float* d_out, hipStream_t stream) {
    BASD_CHECK_ARG(colnorm && k_arr && sw && sw_index && d_out && items > 0 && stride > 0 && n_valid > 0 &&
                   n_valid <= stride && stride <= 1024);
    grassmann_distance_kernel<<<items, 256, 0, stream>>>(colnorm, stride, k_arr, sw, sw_stride, sw_index, d_out, nullptr,
                                                         n_valid);
    BASD_RETURN_LAST();
}

int basd_gram_finish(const float* c, const float* zbar, int n, int batch, long m_rows, float* out_u, float* out_c,
                     hipStream_t stream) {
    BASD_CHECK_ARG(c && zbar && (out_u || out_c) && n > 0 && batch > 0 && m_rows > 0);
    gram_finish_kernel<<<dim3((unsigned)(((long)n * n + 255) / 256), batch), 256, 0, stream>>>(c, zbar, n, (float)m_rows,
                                                                                              out_u, out_c);
    BASD_RETURN_LAST();
}

int basd_sqrt_clamp(const float* in, float* out, long count, hipStream_t stream) {
    BASD_CHECK_ARG(in && out && count > 0);
    sqrt_clamp_kernel<<<(unsigned)((count + 255) / 256), 256, 0, stream>>>(in, out, count);
    BASD_RETURN_LAST();
}

}  // extern "C"

namespace basd {
__global__ void repeat_ranks_kernel(const int* __restrict__ ranks, int L, int items, int* __restrict__ k_arr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < items) k_arr[i] = ranks[i % L];
}
}  // namespace basd

extern "C" {

// The part of the selector that follows the rank read-back (layer_selector.py:36-37, :92, :95-105), queued by ONE
// call: leading kmax eigenvectors of the E student and L (centred) teacher Grams from their tridiagonal
// factorisations, S[:k] of the teacher, rotation of the teacher bases by proj_s^T (proj_s folded, :88/:99), the
// E x L cosine matrices Vt_s[:k] U_t, their singular values (one-sided Jacobi, orders ranks[l]) and
// d_grass_sq (E, L).  Host time matters here: these ~14 launches sit between the read-back and the caller's
// backward, and a dozen separate FFI calls cost several times the launches themselves.
//   t_* / s_*: d, e, tau (x n), vh (x n x n), vals (x n) of the L teacher (centred) / E student matrices, n = d_s
//   scratch: z_s, v_s (E kmax n), z_t, u_t, u_rot (L kmax n), sw (L kmax), cos (E L kmax kmax), k_arr (E L),
//            sigma (E L kmax), flags (basd_jacobi_workspace_ints(E L, 20)); sw_index: (E L) = l of item e*L+l
int basd_selector_tail(const float* t_d, const float* t_e, const float* t_tau, const float* t_vh, const float* t_vals,
                       const float* s_d, const float* s_e, const float* s_tau, const float* s_vh, const float* s_vals,
                       int d_s, int E, int L, int kmax, const int* ranks, const float* proj_s_t, float* z_s,
                       float* v_s, float* z_t, float* u_t, float* u_rot, float* sw, float* cos, int* k_arr,
                       const int* sw_index, float* sigma, int* flags, float* d_out, hipStream_t stream) {
    BASD_CHECK_ARG(t_d && s_d && ranks && proj_s_t && z_s && v_s && z_t && u_t && u_rot && sw && cos && k_arr &&
                   sw_index && sigma && flags && d_out && E > 0 && L > 0 && kmax > 0 && kmax <= d_s);
    int st = basd_tridiag_eigenvectors(s_d, s_e, s_tau, s_vh, s_vals, d_s, kmax, E, z_s, v_s, kmax, stream);
    if (st) return st;
    st = basd_tridiag_eigenvectors(t_d, t_e, t_tau, t_vh, t_vals, d_s, kmax, L, z_t, u_t, kmax, stream);
    if (st) return st;
    for (int l = 0; l < L; ++l) {
        st = basd_sqrt_clamp(t_vals + (long)l * d_s, sw + (long)l * kmax, kmax, stream);
        if (st) return st;
    }
    const long kn = (long)kmax * d_s, kk = (long)kmax * kmax;
    st = basd_gemm_nt(u_t, BASD_DTYPE_F32, 0, d_s, 1, 1 << 30, 0, proj_s_t, d_s, 0, L * kmax, d_s, d_s, 1, u_rot, d_s,
                      (long)L * kn, 1.f, nullptr, 0.f, nullptr, nullptr, stream);
    if (st) return st;
    if (L == 1) {
        st = basd_gemm_nt(v_s, BASD_DTYPE_F32, 0, d_s, 1, 1 << 30, kn, u_rot, d_s, 0, kmax, kmax, d_s, E, cos, kmax, kk,
                          1.f, nullptr, 0.f, nullptr, nullptr, stream);
        if (st) return st;
    } else {
        for (int e = 0; e < E; ++e) {
            st = basd_gemm_nt(v_s + e * kn, BASD_DTYPE_F32, 0, d_s, 1, 1 << 30, 0, u_rot, d_s, kn, kmax, kmax, d_s, L,
                              cos + (long)e * L * kk, kmax, kk, 1.f, nullptr, 0.f, nullptr, nullptr, stream);
            if (st) return st;
        }
    }
    const int items = E * L;
    basd::repeat_ranks_kernel<<<(items + 255) / 256, 256, 0, stream>>>(ranks, L, items, k_arr);
    st = basd_jacobi_onesided(cos, kk, kmax, kmax, kmax, items, k_arr, sigma, kmax, 20, 0.f, flags, nullptr, stream);
    if (st) return st;
    return basd_grassmann_distance(sigma, kmax, k_arr, sw, kmax, sw_index, items, d_out, nullptr, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Base criterion of BASDLoss (reference combined.py:56) when it is the stock torch.nn.CrossEntropyLoss (mean reduction,
// no class weights): loss and d loss / d logits in ONE launch instead of the ~15 micro-kernels torch queues for the
// forward and backward of log_softmax / nll / label smoothing -- their host time sits between the rank read-back and
// the end of the step.   row_loss[b] = -sum_c t'_c log softmax(x_b)_c,  t' = (1 - eps) t + eps / C;
// hard labels: t = one-hot(y_b), rows with y_b == ignore_index contribute nothing; soft labels: t = probs[b].
// dlogits[b][c] = (softmax_c - t'_c) / count  (count = rows not ignored; the caller scales by the upstream gradient).
// grid = B, block = 256.  Accumulation in fp32 with the row maximum subtracted, like ATen.
// ---------------------------------------------------------------------------
namespace basd {
template <typename T>
__global__ void __launch_bounds__(256) cross_entropy_kernel(const T* __restrict__ logits, long ld, int C,
                                                            const long* __restrict__ labels,
                                                            const float* __restrict__ probs, long pld, float eps,
                                                            long ignore_index, float* __restrict__ row_loss,
                                                            float* __restrict__ dlogits) {
    __shared__ float red[32];
    const int b = blockIdx.x, tid = threadIdx.x, B = gridDim.x;
    const T* x = logits + (long)b * ld;
    float* dx = dlogits + (long)b * C;
    const long y = labels ? labels[b] : -1;
    const bool ignored = labels && y == ignore_index;
    // mean over the rows that are not ignored (every workgroup counts them itself: B labels from L2)
    float cnt = 0.f;
    if (labels) {
        for (int r = tid; r < B; r += 256) cnt += labels[r] != ignore_index ? 1.f : 0.f;
        cnt = block_sum(cnt, red);
        __syncthreads();
    } else {
        cnt = (float)B;
    }
    // no row counts (every label is ignore_index): torch's mean is 0 / 0 = NaN
    const float inv_count = cnt > 0.f ? 1.f / cnt : __builtin_nanf("");
    if (ignored) {
        for (int c = tid; c < C; c += 256) dx[c] = 0.f;
        if (tid == 0) row_loss[b] = cnt > 0.f ? 0.f : __builtin_nanf("");
        return;
    }
    if (labels && (y < 0 || y >= C)) {
        // a class index outside [0, C) that is not ignore_index: torch device-asserts; here the loss turns NaN (it
        // cannot be missed) and nothing is read out of bounds
        for (int c = tid; c < C; c += 256) dx[c] = 0.f;
        if (tid == 0) row_loss[b] = __builtin_nanf("");
        return;
    }
    float m = -3.4e38f;
    for (int c = tid; c < C; c += 256) m = fmaxf(m, to_f32(x[c]));
    m = wave_max(m);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f, sx = 0.f, tx = 0.f, st = 0.f;   // sum exp, sum x (uniform part), sum t x (target part), sum t
    for (int c = tid; c < C; c += 256) {
        const float v = to_f32(x[c]) - m;
        s += expf(v);
        sx += v;
        if (probs) {
            const float t = probs[(long)b * pld + c];
            tx = fmaf(t, v, tx);
            st += t;
        }
    }
    s = block_sum(s, red);
    sx = block_sum(sx, red);
    if (probs) {
        tx = block_sum(tx, red);
        st = block_sum(st, red);
    } else {
        tx = to_f32(x[y]) - m;
        st = 1.f;
    }
    const float lse = logf(s);
    // -sum_c t'_c (v_c - lse) with t' = (1 - eps) t + eps / C:  lse * sum t' - (1 - eps) tx - eps/C sx, where
    // sum t' = (1 - eps) sum t + eps (soft targets need not be normalised: torch does not assume it either);
    // already divided by the count
    const float sum_tp = fmaf(1.f - eps, st, eps);
    if (tid == 0) row_loss[b] = (lse * sum_tp - (1.f - eps) * tx - eps / (float)C * sx) * inv_count;
    const float inv_s = 1.f / s;
    for (int c = tid; c < C; c += 256) {
        const float p = expf(to_f32(x[c]) - m) * inv_s;
        const float t = probs ? probs[(long)b * pld + c] : (c == y ? 1.f : 0.f);
        dx[c] = (p * sum_tp - ((1.f - eps) * t + eps / (float)C)) * inv_count;
    }
}
}  // namespace basd

extern "C" {
// logits: (B, C) fp32 / bf16 with row stride ld; exactly one of labels (int64, B) / probs (fp32, (B, C), row stride pld).
int basd_cross_entropy(const void* logits, int dtype, long ld, int B, int C, const long* labels, const float* probs,
                       long pld, float label_smoothing, long ignore_index, float* row_loss, float* dlogits,
                       hipStream_t stream) {
    BASD_CHECK_ARG(logits && row_loss && dlogits && B > 0 && C > 0 && ((labels != nullptr) != (probs != nullptr)));
    if (dtype == BASD_DTYPE_F32)
        basd::cross_entropy_kernel<float><<<B, 256, 0, stream>>>((const float*)logits, ld, C, labels, probs, pld, label_smoothing, ignore_index, row_loss, dlogits);
    else if (dtype == BASD_DTYPE_BF16)
        basd::cross_entropy_kernel<__hip_bfloat16><<<B, 256, 0, stream>>>((const __hip_bfloat16*)logits, ld, C, labels, probs, pld, label_smoothing, ignore_index, row_loss, dlogits);
    else
        return BASD_EINVAL;
    BASD_RETURN_LAST();
}
}  // extern "C"
