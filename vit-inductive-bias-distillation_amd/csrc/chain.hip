// The Grassmannian layer selector of one loss step (reference layer_selector.py:69-74, :86-105, :131-138) queued by ONE
// library call over three streams: projections and Gram matrices (fp32 MFMA), the Householder tridiagonalisation of
// all 2L + E matrices with the Marchenko-Pastur ranks written to pinned host memory by the kernel that finishes the
// factorisation, and -- without waiting for the host -- the rest of the selector with the ranks read ON THE DEVICE:
// spectra, leading eigenvectors, principal angles, d_grass_sq.
//
// Why one call: round 2 queued this from Python (~40 torch.empty, ~15 FFI calls, event objects): the host needed 0.7 ms
// per step and the first chain kernel started 0.28 ms after the loss was entered; every one of those microseconds sits
// in front of the chain the host then waits for.  Why device-side ranks: the part behind the rank read-back used to be
// sized by the host (kmax = max rank) and was therefore queued one step late; here it is sized by a HINT (the previous
// step's largest rank) and every kernel takes the actual rank from device memory -- correct for any hint >= the
// largest rank; the caller re-queues the tail (basd_selector_chain_tail) in the rare step where the rank grew past it.
#include "basd_common.h"
#include "../../include/basd_hip.h"
#include <stdlib.h>

namespace basd {

// sw[l][i] = sqrt(max(lambda_i, 0)), i < kmax: singular values S[:k] of the centred projected teacher tokens from the
// eigenvalues of their Gram (layer_selector.py:36-37).  grid = L, block = 256.
__global__ void __launch_bounds__(256) chain_sw_kernel(const float* __restrict__ vals, int n, int kmax,
                                                       float* __restrict__ sw) {
    const int l = blockIdx.x;
    for (int i = threadIdx.x; i < kmax; i += 256) sw[(long)l * kmax + i] = sqrtf(fmaxf(vals[(long)l * n + i], 0.f));
}

// k_arr[e * L + l] = ranks[l] (clamped to [1, kmax]: a rank-0 layer raises on the host, layer_selector.py:105 would
// divide 0 by 0; a rank past the hint is re-queued by the caller -- neither result is ever observed).
__global__ void chain_k_arr_kernel(const int* __restrict__ ranks, int L, int items, int kmax, int* __restrict__ k_arr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < items) {
        int k = ranks[i % L];
        k = k < 1 ? 1 : k;
        k_arr[i] = k > kmax ? kmax : k;
    }
}

// Rank certificate (BasdSelectorChain.cert_mirror, basd_rank_certificate): flag = 1 iff for every one of the L symmetric
// n x n matrices G (uncentred teacher Grams / M = A + zbar zbar^T, A = the centred Gram / M, PSD)
//     max(|zbar|^2, ||G||_F^2 / tr G)  >  1.5 factor min(tr G / c, (tr G - |zbar|^2) / (c - 1)),    c = n - (n - 1) / 2.
// Left: lower bounds of the largest eigenvalue (Rayleigh quotient of zbar; sum l^2 <= l_1 sum l).  Right: upper bounds of
// the lower median l_c -- c eigenvalues are >= it and all are >= 0, so c l_c <= tr G; and l_c(G) <= l_{c-1}(A) (rank-one
// interlacing) <= tr A / (c - 1).  The 1.5 covers what fp32 does to the computed spectrum (errors ~ n eps l_1 = 2e-5 l_1:
// 1.001 would do).  Then l_1 > fp32(l_c factor): no Marchenko-Pastur rank is 0.  Sums in fp64.
// Grid (CERT_BLOCKS, L) x 256 threads: partial sums go to `scratch` (4 doubles per matrix: ||G||_F^2, tr G, spare, spare;
// then one int ticket) by fp64 atomics; the block that draws the last ticket evaluates the condition for every matrix,
// writes the flag and leaves the scratch zeroed for the next launch.  Small workgroups on purpose: a 1024-thread block
// waited ~80 us for a CU with 16 free wave slots inside a saturated step.
constexpr int CERT_BLOCKS = 32;
__global__ void __launch_bounds__(256) rank_certificate_kernel(const float* __restrict__ grams, const float* __restrict__ zbar,
                                                               int n, int L, double factor, double* __restrict__ scratch,
                                                               int* __restrict__ host_flag) {
    __shared__ double part[2][4];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l = blockIdx.y;
    const long nn = (long)n * n;
    const float* g = grams + l * nn;
    double s2 = 0.0, tr = 0.0;
    for (long idx = (long)blockIdx.x * 256 + tid; idx < nn; idx += (long)CERT_BLOCKS * 256) {
        const double v = (double)g[idx];
        s2 = fma(v, v, s2);
    }
    if (blockIdx.x == 0)
        for (int i = tid; i < n; i += 256) tr += (double)g[(long)i * n + i];
    for (int off = 32; off > 0; off >>= 1) {
        s2 += __shfl_down(s2, off, 64);
        tr += __shfl_down(tr, off, 64);
    }
    if (lane == 0) { part[0][wave] = s2; part[1][wave] = tr; }
    __syncthreads();
    int* ticket = (int*)(scratch + 4L * L);
    if (tid == 0) {
        atomicAdd(scratch + 4 * l + 0, part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        if (blockIdx.x == 0) atomicAdd(scratch + 4 * l + 1, part[1][0] + part[1][1] + part[1][2] + part[1][3]);
        __threadfence();
        s_last = atomicAdd(ticket, 1) == CERT_BLOCKS * L - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // the last block: one wave per matrix in turn (|zbar|^2, then the test)
    __shared__ int s_ok;
    if (tid == 0) s_ok = 1;
    __syncthreads();
    for (int m = wave; m < L; m += 4) {
        double zz = 0.0;
        if (zbar)
            for (int i = lane; i < n; i += 64) {
                const double z = (double)zbar[(long)m * n + i];
                zz = fma(z, z, zz);
            }
        for (int off = 32; off > 0; off >>= 1) zz += __shfl_down(zz, off, 64);
        if (lane == 0) {
            const double a = __hip_atomic_load(scratch + 4 * m + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double b = __hip_atomic_load(scratch + 4 * m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double c = (double)(n - (n - 1) / 2);
            bool good = false;
            if (b > 0.0 && c > 1.0) {
                const double lo = fmax(zz, a / b);
                const double tr_a = fmax(b - zz, 1e-4 * b);      // the subtraction cancels when the mean dominates: floor it
                const double hi = fmin(b / c, tr_a / (c - 1.0));
                good = lo > 1.5 * factor * hi;
            }
            if (!good) atomicAnd(&s_ok, 0);       // also catches NaN
            scratch[4 * m + 0] = 0.0;
            scratch[4 * m + 1] = 0.0;
        }
    }
    __syncthreads();
    if (tid == 0) {
        *ticket = 0;
        *host_flag = s_ok;
        __threadfence_system();
    }
}

// Principal-angle matrices of per-item order k = k_arr[item] inside a common kmax x kmax storage: everything outside the
// leading k x k block is zeroed, so that a fixed-order solver may take them (the singular values are those of the block
// plus kmax - k exact zeros).  grid = items, block = 256.
__global__ void __launch_bounds__(256) chain_mask_cos_kernel(float* __restrict__ cos, int kmax, const int* __restrict__ k_arr) {
    const int k = k_arr[blockIdx.x];
    float* c = cos + (long)blockIdx.x * kmax * kmax;
    for (int idx = threadIdx.x; idx < kmax * kmax; idx += 256) {
        const int i = idx / kmax, j = idx - i * kmax;
        if (i >= k || j >= k) c[idx] = 0.f;
    }
}

__global__ void __launch_bounds__(256) debug_fill_lds_kernel(unsigned pattern, int words) {
    extern __shared__ unsigned fill_words[];
    for (int i = threadIdx.x; i < words; i += 256) fill_words[i] = pattern;
    __syncthreads();
    if (fill_words[(threadIdx.x * 97) % words] != pattern) __builtin_trap();      // keeps the stores
}

// A fixed delay (no memory polling: it cannot deadlock, whatever runs or does not run beside it): one wave asleep for
// ~3.4 us x `rounds` at 2.4 GHz.  Queued at the head of the student side (mode 3), which the teacher's Grams release: the whole-CU factorisation workgroups of the teacher side, released by the same Grams on
// another stream, get their CUs BEFORE the student side's throughput launches refill every free slot.
__global__ void chain_delay_kernel(int rounds) {
    for (int i = 0; i < rounds; ++i) __builtin_amdgcn_s_sleep(127);
}

}  // namespace basd

#define BASD_TRY(call)                \
    do {                              \
        int rc_ = (call);             \
        if (rc_ != BASD_OK) return rc_; \
    } while (0)
#define BASD_HIP(call)                          \
    do {                                        \
        hipError_t e_ = (call);                 \
        if (e_ != hipSuccess) return (int)e_;   \
    } while (0)
// optional measurement event (nullable)
#define BASD_MARK(ev, stream)                                         \
    do {                                                              \
        if (ev) BASD_HIP(hipEventRecord((hipEvent_t)(ev), (stream))); \
    } while (0)

extern "C" {

int basd_event_record(void* ev, hipStream_t stream) {
    BASD_CHECK_ARG(ev);
    BASD_HIP(hipEventRecord((hipEvent_t)ev, stream));
    return BASD_OK;
}
int basd_event_synchronize(void* ev) {
    BASD_CHECK_ARG(ev);
    BASD_HIP(hipEventSynchronize((hipEvent_t)ev));
    return BASD_OK;
}
// A stream of the given priority relative to the device's range: -1 = highest, 0 = default, +1 = lowest.  (torch offers
// high and default only; the selector's student side -- throughput launches nothing waits for -- wants the LOWEST, so that
// the dispatcher serves the chains the step waits for first.)
int basd_stream_create_priority(void** out, int level) {
    BASD_CHECK_ARG(out && level >= -1 && level <= 1);
    int least = 0, greatest = 0;           // numerically: greatest priority <= least priority
    BASD_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int prio = level < 0 ? greatest : (level > 0 ? least : 0);
    BASD_HIP(hipStreamCreateWithPriority((hipStream_t*)out, hipStreamNonBlocking, prio));
    return BASD_OK;
}

// diagnostics (tools/step_clock.py): events that carry a time stamp, and the time between two of them
int basd_event_create_timed(void** out) {
    BASD_CHECK_ARG(out);
    BASD_HIP(hipEventCreateWithFlags((hipEvent_t*)out, hipEventDefault));
    return BASD_OK;
}
int basd_event_elapsed_ms(void* from, void* to, float* ms_out) {
    BASD_CHECK_ARG(from && to && ms_out);
    BASD_HIP(hipEventElapsedTime(ms_out, (hipEvent_t)from, (hipEvent_t)to));
    return BASD_OK;
}
// 1 when everything recorded before the event has completed, 0 when not yet
int basd_event_query(void* ev) {
    if (!ev) return BASD_EINVAL;
    const hipError_t e = hipEventQuery((hipEvent_t)ev);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return BASD_EINVAL;
}

// layer_selector.py:36-37, :92, :95-105 for the matrices of one chain: spectra of the L centred teacher and E student
// Grams, their leading kmax eigenvectors, S[:k], the rotation of the teacher bases by proj_s^T (proj_s is orthogonal:
// folded from :88 into :99), the E x L cosine matrices, their singular values of order ranks[l] and d_grass_sq (E, L).
// Queued on a->tail_stream behind ev_ranks (and ev_student in the split modes); records ev_tail.
int basd_selector_chain_tail(const BasdSelectorChain* a, int kmax, int exact_k) {
    BASD_CHECK_ARG(a && a->d && a->vals && a->zv && a->vecs && a->u_rot && a->sw && a->cos && a->sigma && a->d_out &&
                   a->k_arr && a->sw_index && a->jflags && a->ranks && a->proj_s_t);
    const int E = (int)a->E, L = (int)a->L, n = (int)a->d_s;
    BASD_CHECK_ARG(E > 0 && L > 0 && kmax > 0 && kmax <= n && kmax <= (int)a->kmax_cap);
    if (!exact_k && !basd_jacobi_lds_square_fits(kmax) && !(kmax >= 96 && basd_jacobi_plain4_fits(kmax))) return BASD_EUNSUPPORTED;
    hipStream_t st = a->tail_stream;
    BASD_HIP(hipStreamWaitEvent(st, (hipEvent_t)a->ev_ranks, 0));
    if (a->mode != 0 && a->mode != 4) BASD_HIP(hipStreamWaitEvent(st, (hipEvent_t)a->ev_student, 0));
    const long nn = (long)n * n, kn = (long)kmax * n, kk = (long)kmax * kmax;
    // matrices [L, 2L + E): centred teacher Grams, then the student Grams
    const float* dz = a->d + (long)L * n;
    const float* ez = a->e + (long)L * n;
    const float* tz = a->tau + (long)L * n;
    const float* vz = a->vh + (long)L * nn;
    float* lam = a->vals + (long)L * n;
    BASD_TRY(basd_tridiag_eigenvalues(dz, ez, n, L + E, lam, st));
    BASD_MARK(a->tm_spec, st);
    // rows of vecs: (L + E, kmax, n) -- teacher bases first, then Vt_s[:kmax] of every student layer
    BASD_TRY(basd_tridiag_eigenvectors(dz, ez, tz, vz, lam, n, kmax, L + E, a->zv, a->vecs, kmax, st));
    basd::chain_sw_kernel<<<L, 256, 0, st>>>(lam, n, kmax, a->sw);
    const float* u_t = a->vecs;
    const float* v_s = a->vecs + (long)L * kn;
    BASD_TRY(basd_gemm_nt(u_t, BASD_DTYPE_F32, 0, n, 1, 1 << 30, 0, a->proj_s_t, n, 0, L * kmax, n, n, 1, a->u_rot, n,
                          (long)L * kn, 1.f, nullptr, 0.f, nullptr, nullptr, st));
    if (L == 1) {
        BASD_TRY(basd_gemm_nt(v_s, BASD_DTYPE_F32, 0, n, 1, 1 << 30, kn, a->u_rot, n, 0, kmax, kmax, n, E, a->cos, kmax,
                              kk, 1.f, nullptr, 0.f, nullptr, nullptr, st));
    } else {
        for (int e = 0; e < E; ++e)
            BASD_TRY(basd_gemm_nt(v_s + e * kn, BASD_DTYPE_F32, 0, n, 1, 1 << 30, 0, a->u_rot, n, kn, kmax, kmax, n, L,
                                  a->cos + (long)e * L * kk, kmax, kk, 1.f, nullptr, 0.f, nullptr, nullptr, st));
    }
    const int items = E * L;
    basd::chain_k_arr_kernel<<<(items + 255) / 256, 256, 0, st>>>(a->ranks, L, items, kmax, a->k_arr);
    if (!exact_k && kmax >= 96 && basd_jacobi_plain4_fits(kmax)) {
        // large speculative orders (teachers of rank ~100-190): the per-matrix-order solver runs a pair per DPP row (1.4 us
        // per round at 168 x 168, four workgroups in all); zero-padded to the common order the matrices take the
        // register-resident odd-even solver (4 lanes per pair), and the distance kernel sorts all kmax values
        basd::chain_mask_cos_kernel<<<items, 256, 0, st>>>(a->cos, kmax, a->k_arr);
        BASD_TRY(basd_jacobi_onesided(a->cos, kk, kmax, kmax, kmax, items, nullptr, a->sigma, kmax, 20, 0.f, a->jflags,
                                      nullptr, st));
        BASD_TRY(basd_grassmann_distance_padded(a->sigma, kmax, kmax, a->k_arr, a->sw, kmax, a->sw_index, items, a->d_out,
                                                st));
    } else {
        BASD_TRY(basd_jacobi_onesided(a->cos, kk, kmax, kmax, kmax, items, exact_k ? nullptr : a->k_arr, a->sigma, kmax, 20,
                                      0.f, a->jflags, nullptr, st));
        BASD_TRY(basd_grassmann_distance(a->sigma, kmax, a->k_arr, a->sw, kmax, a->sw_index, items, a->d_out, nullptr, st));
    }
    BASD_HIP(hipEventRecord((hipEvent_t)a->ev_tail, st));
    return BASD_OK;
}

// Test hook: every CU's LDS filled with `pattern` (e.g. a NaN): kernels must not depend on what the previous tenant of
// their CU left in LDS (tests/test_gpu_kernels.py runs the solvers behind it and demands bit-identical results).
int basd_debug_fill_lds(unsigned pattern, hipStream_t stream) {
    const int bytes = 80 * 1024;
    (void)hipFuncSetAttribute((const void*)basd::debug_fill_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    basd::debug_fill_lds_kernel<<<4096, 256, bytes, stream>>>(pattern, bytes / 4);
    BASD_RETURN_LAST();
}

long basd_rank_certificate_scratch_bytes(int batch) { return batch > 0 ? 32L * batch + 8 : 0; }

int basd_rank_certificate(const float* grams, const float* zbar, int n, int batch, double factor, double* scratch,
                          int* flag, hipStream_t stream) {
    BASD_CHECK_ARG(grams && flag && scratch && n > 2 && batch > 0 && batch <= 65535 && factor > 0.0);
    basd::rank_certificate_kernel<<<dim3(basd::CERT_BLOCKS, batch), 256, 0, stream>>>(grams, zbar, n, batch, factor, scratch,
                                                                                     flag);
    BASD_RETURN_LAST();
}

int basd_selector_chain(const BasdSelectorChain* a) {
    BASD_CHECK_ARG(a && a->teacher_host_ptrs && a->student_ptrs && a->proj_t && a->z && a->z_sums && a->z_ptrs &&
                   a->z_means && a->t_slabs && a->s_partial && a->s_means && a->s_slabs && a->grams && a->d && a->e &&
                   a->tau && a->vh && a->tri_work && a->ranks && a->ev_fork && a->ev_student && a->ev_ranks && a->ev_tail);
    const int E = (int)a->E, L = (int)a->L, B = (int)a->B, n_s = (int)a->n_s, n_t = (int)a->n_t;
    const int n = (int)a->d_s, d_t = (int)a->d_t, mode = (int)a->mode;
    BASD_CHECK_ARG(E > 0 && L > 0 && B > 0 && n > 1 && d_t > 0 && mode >= 0 && mode <= 4);
    const long M_t = (long)B * n_t, M_s = (long)B * n_s;
    // the uncentred Gram is formed on the feature side (layer_selector.py:12-13); the token-side form (:14-15, fewer
    // rows than features) has a different order: not covered here, the caller takes the per-kernel entry points
    if (M_t < n || M_t > 0x7fffffff || M_s > 0x7fffffff) return BASD_EUNSUPPORTED;
    const long nn = (long)n * n;
    const int tiles = (int)((M_t + 127) / 128);
    hipStream_t cs = a->chain_stream, ss = a->student_stream;

    // ---- order against the caller: inputs are ready on main_stream; this slot's buffers are free once the tail that
    // read them last has finished
    // (main_stream NULL: the caller has recorded ev_fork itself, at the point of ITS stream the chain may start behind)
    if (a->main_stream) BASD_HIP(hipEventRecord((hipEvent_t)a->ev_fork, a->main_stream));
    BASD_HIP(hipStreamWaitEvent(cs, (hipEvent_t)a->ev_fork, 0));
    if (ss != cs) BASD_HIP(hipStreamWaitEvent(ss, (hipEvent_t)a->ev_fork, 0));
    if (a->ev_slot_free) {
        BASD_HIP(hipStreamWaitEvent(cs, (hipEvent_t)a->ev_slot_free, 0));
        if (ss != cs) BASD_HIP(hipStreamWaitEvent(ss, (hipEvent_t)a->ev_slot_free, 0));
    }

    // ---- mode 3 with an early launch: the teacher's factorisation takes its CUs NOW and waits for the Grams' word
    const bool early = mode == 3 && a->go_flag != nullptr && a->fact_stream != nullptr && a->go_value != 0;
    if (early) {
        hipStream_t fs = a->fact_stream;
        if (a->ev_slot_free) BASD_HIP(hipStreamWaitEvent(fs, (hipEvent_t)a->ev_slot_free, 0));
        int rc_early = basd_tridiag_ranked_gated(a->grams, nn, n, 2 * L, a->d, a->e, a->tau, a->vh, a->tri_work, L,
                                                 a->mp_factor, (int)a->rank_cap, a->ranks, a->host_mirror, a->go_flag,
                                                 (unsigned)a->go_value, (int)a->go_budget, fs);
        if (rc_early != BASD_OK) return rc_early;
        BASD_HIP(hipEventRecord((hipEvent_t)a->ev_ranks, fs));
    }
    // ---- teacher: z_l = tokens_l proj_t^T (+ column sums of every 128-row tile and the column means folded from them),
    // the centred Gram of every layer in one symmetric launch (layer_selector.py:35) and the uncentred one / M (:13)
    // from it: z^T z / M = (G_c + M zbar zbar^T) / M -- an addition (no cancellation), half the MFMA work of two Grams
    for (int l = 0; l < L; ++l)
        BASD_TRY(basd_gemm_nt(a->teacher_host_ptrs[l], (int)a->t_dtype, a->t_sb, a->t_sn, a->t_sd, n_t, 0, a->proj_t, d_t,
                              0, (int)M_t, n, d_t, 1, a->z + (long)l * M_t * n, n, M_t * n, 1.f, nullptr, 0.f,
                              a->z_sums + (long)l * tiles * n, a->z_means + (long)l * n, cs));
    BASD_MARK(a->tm_proj, cs);
    if (mode == 4) {       // the student side starts behind the teacher's projection (both are full-chip MFMA launches)
        BASD_CHECK_ARG(a->ev_tgram != nullptr && ss != cs);
        BASD_HIP(hipEventRecord((hipEvent_t)a->ev_tgram, cs));
        BASD_HIP(hipStreamWaitEvent(ss, (hipEvent_t)a->ev_tgram, 0));
    }
    BASD_TRY(basd_syrk_multi(a->z_ptrs, BASD_DTYPE_F32, 0, n, 1, 1 << 30, (int)M_t, n, L, a->z_means, nullptr,
                             (int)a->t_splits, a->t_slabs, a->grams + (long)L * nn, nn, 1, nullptr, 0, 0, cs));
    BASD_TRY(basd_gram_finish(a->grams + (long)L * nn, a->z_means, n, L, M_t, a->grams, nullptr, cs));
    BASD_MARK(a->tm_tgram, cs);
    if (mode == 3) BASD_CHECK_ARG(a->ev_tg0 != nullptr);
    if (a->ev_tg0) BASD_HIP(hipEventRecord((hipEvent_t)a->ev_tg0, cs));
    if (a->cert_mirror) {
        // "every rank >= 1", proven from the uncentred Grams alone where the spectrum allows it (see the header): the host
        // then need not wait for the factorisation to know that the reference would not have raised
        BASD_CHECK_ARG(a->cert_stream && a->ev_cert && a->ev_tg0 && a->cert_scratch);
        if (a->cert_stream != cs) BASD_HIP(hipStreamWaitEvent(a->cert_stream, (hipEvent_t)a->ev_tg0, 0));
        BASD_TRY(basd_rank_certificate(a->grams, a->z_means, n, L, a->mp_factor, a->cert_scratch, a->cert_mirror, a->cert_stream));
        BASD_HIP(hipEventRecord((hipEvent_t)a->ev_cert, a->cert_stream));
    }

    auto student_grams = [&](hipStream_t st) -> int {
        // centred Grams of the E student layers (:88-91; proj_s folded into the principal angles)
        BASD_MARK(a->tm_scol0, st);
        BASD_TRY(basd_colmean_multi(a->student_ptrs, (int)a->s_dtype, a->s_sb, a->s_sn, a->s_sd, n_s, (int)M_s, n, E,
                                    (int)a->s_parts, a->s_partial, a->s_means, (int)a->s_vec_ok, st));
        BASD_MARK(a->tm_scol1, st);
        BASD_TRY(basd_syrk_multi(a->student_ptrs, (int)a->s_dtype, a->s_sb, a->s_sn, a->s_sd, n_s, (int)M_s, n, E,
                                 a->s_means, nullptr, (int)a->s_splits, a->s_slabs, a->grams + 2L * L * nn, nn,
                                 (int)a->s_vec_ok, nullptr, 0, 0, st));
        BASD_MARK(a->tm_sgram, st);
        return BASD_OK;
    };

    if (mode == 0 || mode == 4) {
        // ONE factorisation launch over all 2L + E matrices: the student Grams join the teacher's before it
        BASD_TRY(student_grams(ss));
        if (ss != cs) {
            BASD_HIP(hipEventRecord((hipEvent_t)a->ev_student, ss));
            BASD_HIP(hipStreamWaitEvent(cs, (hipEvent_t)a->ev_student, 0));
        }
        BASD_MARK(a->tm_tri0, cs);
        BASD_TRY(basd_tridiag_ranked(a->grams, nn, n, 2 * L + E, a->d, a->e, a->tau, a->vh, a->tri_work, L, a->mp_factor,
                                     (int)a->rank_cap, a->ranks, a->host_mirror, a->tm_mid, cs));
        BASD_HIP(hipEventRecord((hipEvent_t)a->ev_ranks, cs));
    } else {
        // teacher matrices first (the host waits for their ranks); the student side on its own stream, in mode 1 held
        // back until the ranks are out (its Gram launch is the largest MFMA launch of the step)
        BASD_MARK(a->tm_tri0, cs);
        if (early) {
            BASD_TRY(basd_flag_set(a->go_flag, (unsigned)a->go_value, cs));       // behind the Grams: the factorisation may read
        } else {
            BASD_TRY(basd_tridiag_ranked(a->grams, nn, n, 2 * L, a->d, a->e, a->tau, a->vh, a->tri_work, L, a->mp_factor,
                                         (int)a->rank_cap, a->ranks, a->host_mirror, a->tm_mid, cs));
            BASD_HIP(hipEventRecord((hipEvent_t)a->ev_ranks, cs));
        }
        BASD_CHECK_ARG(a->tri_work_s != nullptr && ss != cs);
        if (mode == 1) BASD_HIP(hipStreamWaitEvent(ss, (hipEvent_t)a->ev_ranks, 0));
        if (mode == 3) {
            // released by the teacher's Grams like the teacher's factorisation (queued above on `cs`), but a short fixed
            // delay later: that factorisation's whole-CU workgroups are placed first
            BASD_HIP(hipStreamWaitEvent(ss, (hipEvent_t)a->ev_tg0, 0));
            if (a->release_delay > 0) basd::chain_delay_kernel<<<1, 64, 0, ss>>>((int)a->release_delay);
        }
        BASD_TRY(student_grams(ss));
        BASD_TRY(basd_tridiag(a->grams + 2L * L * nn, nn, n, E, a->d + 2L * L * n, a->e + 2L * L * n,
                              a->tau + 2L * L * n, a->vh + 2L * L * nn, a->tri_work_s, ss));
        if (a->student_status_mirror) {
            // status words of the student factorisation (last 32 bytes of its workspace): read by the host a step later
            const char* err = (const char*)a->tri_work_s + basd_tridiag_workspace_bytes(n, E) - 32;
            BASD_HIP(hipMemcpyAsync(a->student_status_mirror, err, 32, hipMemcpyDeviceToHost, ss));
        }
        BASD_HIP(hipEventRecord((hipEvent_t)a->ev_student, ss));
    }
    if (a->kmax > 0) return basd_selector_chain_tail(a, (int)a->kmax, 0);
    return BASD_OK;
}

}  // extern "C"
