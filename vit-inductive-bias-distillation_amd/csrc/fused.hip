// One library call = one stage of the loss step.  The per-kernel entry points stay (tests and the general paths use
// them); these functions queue the same launches in the same order from C, because a dozen separate FFI calls from
// Python cost several times the launches themselves and that host time sits in front of the kernels of the caller's
// stream (round-1 timeline: the first Procrustes kernel started 1.05 ms into a 3.5 ms step).
#include "basd_common.h"
#include "../../include/basd_hip.h"

namespace basd {

// combined.py:76-85 on the device: per-layer means of loss_b (E, B), their mean, the UW-SO weights of (ce, geo) from the
// DETACHED values, the total.  One workgroup; means accumulated in fp64 in a fixed order.
__global__ void __launch_bounds__(256) uwso_combine_kernel(const float* __restrict__ ce, const float* __restrict__ loss_b,
                                                           int E, int B, float* __restrict__ out) {
    constexpr int CH = 32;                  // layers per pass: one barrier per pass, not two per layer
    __shared__ double red[CH][4];
    const int tid = threadIdx.x;
    double geo_sum = 0.;                    // thread 0 only
    for (int e0 = 0; e0 < E; e0 += CH) {
        const int ne = E - e0 < CH ? E - e0 : CH;
        for (int e = 0; e < ne; ++e) {
            double acc = 0.;
            for (int b = tid; b < B; b += 256) acc += (double)loss_b[(long)(e0 + e) * B + b];
            for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
            if ((tid & 63) == 0) red[e][tid >> 6] = acc;
        }
        __syncthreads();
        if (tid == 0)
            for (int e = 0; e < ne; ++e) {
                const float layer = (float)(((red[e][0] + red[e][1]) + (red[e][2] + red[e][3])) / (double)B);
                out[4 + E + e0 + e] = layer;
                geo_sum += (double)layer;
            }
        __syncthreads();
    }
    if (tid == 0) {
        const float c = ce[0], geo = (float)(geo_sum / (double)E);
        const float eps = 1.1920929e-7f;                       // torch.finfo(float32).eps
        const float i0 = 1.f / fmaxf(c, eps), i1 = 1.f / fmaxf(geo, eps);
        const float w0 = i0 / (i0 + i1), w1 = i1 / (i0 + i1);
        out[0] = w0;
        out[1] = w1;
        out[2] = w0 * c + w1 * geo;
        out[3] = geo;
        for (int e = 0; e < E; ++e) out[4 + e] = w1 / (float)E;
    }
}

__global__ void __launch_bounds__(256) scale_unless_one_kernel(float* __restrict__ x, long count4, long count,
                                                               const float* __restrict__ num,
                                                               const float* __restrict__ den) {
    const float f = den ? num[0] / den[0] : num[0];
    if (f == 1.f) return;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < count4) {
        float4 v = ((float4*)x)[i];
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        ((float4*)x)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(count - 4 * count4)) x[4 * count4 + threadIdx.x] *= f;
}

}  // namespace basd

extern "C" {

int basd_scale_unless_one(float* x, long count, const float* num, const float* den, hipStream_t stream) {
    BASD_CHECK_ARG(x && num && count > 0 && ((uintptr_t)x & 15) == 0);
    const long count4 = count / 4;
    basd::scale_unless_one_kernel<<<(unsigned)((count4 + 255) / 256 + (count4 == 0)), 256, 0, stream>>>(x, count4, count, num, den);
    BASD_RETURN_LAST();
}

// combined.py:76-85 and the student gradients: the end of basd_procrustes_forward_fused, shared by its two SVD routes.
static int procrustes_tail(const BasdProcrustesArgs* a, const float* grad_layers, int E, int B, int n_s, int n, int d_s,
                           int EB, long nn, long om_stride, hipStream_t st) {
    int rc;
#define BASD_TRY(call)            \
    do {                          \
        rc = (call);              \
        if (rc != BASD_OK) return rc; \
    } while (0)
    // combined.py:76-85: the UW-SO weights and the total, on the device (the student gradients below are then final)
    if (a->uw_ce) {
        BASD_CHECK_ARG(a->uw_out != nullptr);
        basd::uwso_combine_kernel<<<1, 256, 0, st>>>(a->uw_ce, a->loss_b, E, B, a->uw_out);
        grad_layers = a->uw_out + 4;
    }
    // student gradients for the upstream gradients grad_layers (E floats on the device): H = K' A', then one pass
    if (a->dx) {
        BASD_CHECK_ARG(a->k_prime && a->h && grad_layers);
        rc = basd_student_grad_fused(a->student_ptrs, (int)a->s_dtype, a->s_sb, a->s_sn, E, B, n_s, n, d_s,
                                     (int)a->s_aligned, a->omega, om_stride, a->mu_s, a->k_prime, a->a_prime, a->tap0,
                                     a->tap1, a->lam, grad_layers, 2.0f / (float)B, a->dx, st);
        if (rc != BASD_EUNSUPPORTED) return rc;
        BASD_TRY(basd_gemm_tn(a->k_prime, a->a_prime, BASD_DTYPE_F32, 0, n, 1, nn, 0, d_s, 1, (long)n * d_s, 1 << 30, n,
                              n, d_s, EB, nullptr, nullptr, 1, nullptr, a->h, d_s, (long)n * d_s, 1.f, st));
        BASD_TRY(basd_student_grad_multi(a->student_ptrs, (int)a->s_dtype, a->s_sb, a->s_sn, E, B, n_s, n, d_s, a->omega,
                                         om_stride, a->mu_s, a->h, a->tap0, a->tap1, a->lam, grad_layers,
                                         2.0f / (float)B, a->dx, nullptr, nullptr, st));
    }
#undef BASD_TRY
    return BASD_OK;
}

// Test hook: 0 keeps the stacked cores [M; L_b] (riding rows) for every shape, 2 drops the batch threshold.
static int g_transposed_cores = 1;
int basd_procrustes_tuning(int transposed_cores) {
    if (transposed_cores >= 0) g_transposed_cores = transposed_cores > 2 ? 2 : transposed_cores;
    return BASD_OK;
}

// relational.py:22-50 for all extraction layers against the (mixed) teacher, forward and -- optionally -- the student
// gradients for given upstream gradients; see BasdProcrustesArgs in include/basd_hip.h.
int basd_procrustes_forward_fused(const BasdProcrustesArgs* a, hipStream_t st) {
    BASD_CHECK_ARG(a && a->student_ptrs && a->tok_ptrs && a->attn_ptrs && a->mix);
    const int E = (int)a->E, L = (int)a->L, G = (int)a->G, B = (int)a->B, n_s = (int)a->n_s, n_t = (int)a->n_t;
    const int d_s = (int)a->d_s, d_t = (int)a->d_t, H = (int)a->H, A = (int)a->A, n_a = (int)a->n_a, n = (int)a->n;
    const float* grad_layers_in = a->grad_layers;
    BASD_CHECK_ARG(E > 0 && L > 0 && (G == 1 || G == E) && B > 0 && n == (n_s < n_t ? n_s : n_t));
    int rc;
#define BASD_TRY(call)            \
    do {                          \
        rc = (call);              \
        if (rc != BASD_OK) return rc; \
    } while (0)
    // teacher side: token weights + mixed, centred teacher, once per group (G = 1: shared by all layers)
    int centred = BASD_EUNSUPPORTED;
    for (int g = 0; g < G; ++g) {
        BASD_TRY(basd_token_weights(a->attn_ptrs, (int)a->a_dtype, a->mix + (long)g * L, L, a->a_sb, a->a_sh, a->a_sq,
                                    a->a_sk, B, H, A, (int)a->has_cls, n_a, n, n_s, a->atap0, a->atap1, a->alam,
                                    a->tap0, a->tap1, a->lam, a->omega + (long)g * B * n_s,
                                    a->omega_t + (long)g * B * n, a->raw ? a->raw + (long)g * B * n_a : nullptr, st));
    }
    // mixed, centred teacher: all groups in one pass over the teacher layers where that applies
    if (G > 1 && L >= 4 && basd_teacher_center_stream_scratch_floats(G, B, n, d_t) <= (long)E * B * 2 * n * n) {
        // many layers: the streaming form; its per-chunk column sums borrow W, which is not written before the stacked
        // product
        centred = basd_teacher_center_stream(a->tok_ptrs, (int)a->t_dtype, a->mix, L, G, a->t_sb, a->t_sn, a->t_sd, B, n,
                                             d_t, a->g0, a->g1, a->glam, a->omega_t, a->mu_t, a->tc, a->W, st);
        if (centred != BASD_OK && centred != BASD_EUNSUPPORTED) return centred;
    }
    if (G > 1 && centred != BASD_OK) {
        centred = basd_teacher_center_multi(a->tok_ptrs, (int)a->t_dtype, a->mix, L, G, a->t_sb, a->t_sn, a->t_sd, B, n,
                                            d_t, a->g0, a->g1, a->glam, a->omega_t, a->mu_t, a->tc, st);
        if (centred != BASD_OK && centred != BASD_EUNSUPPORTED) return centred;
    }
    for (int g = 0; g < G && centred != BASD_OK; ++g) {
        BASD_TRY(basd_teacher_center(a->tok_ptrs, (int)a->t_dtype, a->mix + (long)g * L, L, a->t_sb, a->t_sn, a->t_sd,
                                     B, n, d_t, a->g0, a->g1, a->glam, a->omega_t + (long)g * B * n,
                                     a->mu_t + (long)g * B * d_t, a->tc + (long)g * B * n * d_t, st));
    }
    // student side: all layers in one launch where the vectorised kernel applies
    const long om_stride = G == 1 ? 0 : (long)B * n_s;
    int slabs = (d_s + 31) / 32;          // partial traces per sample: 32-feature slabs (multi) or 64 (per layer)
    rc = basd_student_project_multi(a->student_ptrs, (int)a->s_dtype, a->s_sb, a->s_sn, E, B, n_s, n, d_s,
                                    (int)a->s_aligned, a->omega, om_stride, a->tap0, a->tap1, a->lam, a->range0,
                                    a->range1, a->mu_s, a->tr_part, a->a_prime, st);
    if (rc == BASD_EUNSUPPORTED) {
        slabs = (d_s + 63) / 64;
        for (int e = 0; e < E; ++e)
            BASD_TRY(basd_student_project(a->student_host_ptrs[e], (int)a->s_dtype, a->s_sb, a->s_sn, B, n_s, n, d_s,
                                          a->omega + e * om_stride, a->tap0, a->tap1, a->lam, a->range0, a->range1,
                                          a->mu_s + (long)e * B * d_s, a->tr_part + (long)e * B * slabs,
                                          a->a_prime + (long)e * B * n * d_s, st));
    } else if (rc != BASD_OK) {
        return rc;
    }
    // fp64 Grams on the core grid, Cholesky factors, stacked product, SVD, per-sample terms
    const long nn = (long)n * n;
    const int EB = E * B, GB = G * B;
    BASD_TRY(basd_gram_f64(a->a_prime, (long)n * d_s, n, d_s, EB, a->g_all, nn, st));
    if (a->g_slabs && a->g_splits > 1)
        BASD_TRY(basd_gram_f64_split(a->tc, (long)n * d_t, n, d_t, GB, (int)a->g_splits, a->g_slabs,
                                     a->g_all + (long)EB * nn, st));
    else
        BASD_TRY(basd_gram_f64(a->tc, (long)n * d_t, n, d_t, GB, a->g_all + (long)EB * nn, nn, st));
    BASD_TRY(basd_chol_f64(a->g_all, nn, n, EB + GB, a->l_all, nn, st));
    // Cores that the plain 4-lane LDS solver takes, when nothing needs U Sigma afterwards (no gradient through the
    // mixing weights): the Jacobi runs on M^T alone -- its columns come out as V Sigma -- and Y = L_b V is formed when K'
    // is (half the rows per pair-step, no two-pass / block solver at cfg-5's 144 tokens).
    // With gradients through the mixing weights (a->raw set) the backward reads U Sigma, the top half of the stacked
    // cores: it is rebuilt from the transposed route's Z afterwards (a->w_stack), so those steps take this route too.
    if ((a->raw == nullptr || a->w_stack != nullptr) && EB <= 65535 && basd_jacobi_plain4_fits(n) &&
        (g_transposed_cores == 2 || (g_transposed_cores == 1 && EB >= 128))) {
        BASD_TRY(basd_stack_product_t(a->l_all, a->l_all + (long)EB * nn, nn, n, EB, GB, a->W, 2 * nn, st));
        if (a->raw != nullptr) BASD_TRY(basd_ustack_stash(a->W, 2 * nn, n, EB, a->w_stack, 2 * nn, st));
        BASD_TRY(basd_jacobi_onesided(a->W, 2 * nn, n, n, n, EB, nullptr, a->sigma, n, (int)a->max_sweeps, 0.f,
                                      a->jflags, a->sweeps, st));
        BASD_TRY(basd_procrustes_finalize(a->W, 2 * nn, a->sigma, n, n_s, EB, GB, a->g_all + (long)EB * nn, nn, a->omega,
                                          a->tap0, a->tap1, a->lam, a->tr_part, slabs, a->tr_s, a->tr_t, a->nuc,
                                          a->loss_b, nullptr, st));
        if (a->k_prime)
            BASD_TRY(basd_kprime_from_transposed(a->W, 2 * nn, a->sigma, n, EB, a->l_all + (long)EB * nn, nn, GB,
                                                 a->W + nn, 2 * nn, a->k_prime, st));
        if (a->raw != nullptr) {
            BASD_CHECK_ARG(a->sigma_u != nullptr);
            // (scratch: the z region behind X, free again once K' has been formed)
            BASD_TRY(basd_ustack_from_transposed(a->W, 2 * nn, a->sigma, n, EB, a->w_stack, 2 * nn, a->W + nn, 2 * nn,
                                                 a->sigma_u, (int)a->max_sweeps, a->jflags, st));
        }
        return procrustes_tail(a, grad_layers_in, E, B, n_s, n, d_s, EB, nn, om_stride, st);
    }
    BASD_TRY(basd_stack_product(a->l_all, a->l_all + (long)EB * nn, nn, n, EB, GB, a->W, 2 * nn, st));
    rc = BASD_EUNSUPPORTED;
    if (a->jac_ws)
        rc = basd_jacobi_stacked_twopass(a->W, 2 * nn, n, EB, a->sigma, n, (int)a->max_sweeps, 0.f, a->jac_ws, a->sweeps,
                                         st);
    if (rc == BASD_EUNSUPPORTED)
        rc = basd_jacobi_onesided(a->W, 2 * nn, n, 2 * n, n, EB, nullptr, a->sigma, n, (int)a->max_sweeps, 0.f,
                                  a->jflags, a->sweeps, st);
    if (rc != BASD_OK) return rc;
    BASD_TRY(basd_procrustes_finalize(a->W, 2 * nn, a->sigma, n, n_s, EB, GB, a->g_all + (long)EB * nn, nn, a->omega,
                                      a->tap0, a->tap1, a->lam, a->tr_part, slabs, a->tr_s, a->tr_t, a->nuc,
                                      a->loss_b, a->k_prime, st));
    return procrustes_tail(a, grad_layers_in, E, B, n_s, n, d_s, EB, nn, om_stride, st);
#undef BASD_TRY
}


// Events for ordering two streams from inside a library call (see basd_tridiag_ranked's mid_event).
int basd_event_create(void** out) {
    BASD_CHECK_ARG(out);
    hipError_t e = hipEventCreateWithFlags((hipEvent_t*)out, hipEventDisableTiming);
    return e == hipSuccess ? BASD_OK : (int)e;
}
int basd_event_destroy(void* ev) {
    hipError_t e = hipEventDestroy((hipEvent_t)ev);
    return e == hipSuccess ? BASD_OK : (int)e;
}
int basd_stream_wait_event(hipStream_t stream, void* ev) {
    BASD_CHECK_ARG(ev);
    hipError_t e = hipStreamWaitEvent(stream, (hipEvent_t)ev, 0);
    return e == hipSuccess ? BASD_OK : (int)e;
}

}  // extern "C"
