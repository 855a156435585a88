// Grassmannian layer-selector epilogues (reference src/losses/layer_selector.py).
//   mp_rank_kernel          : lower median, Marchenko-Pastur threshold, strict count  (:16-19, :74)
//   grassmann_distance_kernel: sigma -> theta = acos(min(sigma, 1 - eps)) ->
//                              d = sum(sw * theta^2) / sum(sw)                        (:99-105)
#include "basd_common.h"

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
                                                                 float* __restrict__ theta_out) {
    __shared__ float key[1024];
    __shared__ float red[32];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int k = k_arr[m];
    int np2 = 1;
    while (np2 < k) np2 <<= 1;
    for (int i = tid; i < np2; i += 256) key[i] = i < k ? colnorm[(long)m * stride + i] : -1.f;
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
    grassmann_distance_kernel<<<items, 256, 0, stream>>>(colnorm, stride, k_arr, sw, sw_stride, sw_index, d_out, theta_out);
    BASD_RETURN_LAST();
}

int basd_sqrt_clamp(const float* in, float* out, long count, hipStream_t stream) {
    BASD_CHECK_ARG(in && out && count > 0);
    sqrt_clamp_kernel<<<(unsigned)((count + 255) / 256), 256, 0, stream>>>(in, out, count);
    BASD_RETURN_LAST();
}

}  // extern "C"
