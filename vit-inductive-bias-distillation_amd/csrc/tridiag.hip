// Symmetric eigen-solver for the selector's D_s x D_s Gram matrices when only the eigenvalues and/or the
// leading k eigenvectors are needed (reference call sites: torch.linalg.eigvalsh layer_selector.py:16;
// the Vt[:k] / S[:k] part of torch.linalg.svd at :36 and :92):
//
//   Householder tridiagonalisation  ->  Sturm-sequence bisection (all eigenvalues)
//        ->  inverse iteration on T for the top k  ->  back-transformation with the reflectors.
//
// Why not the Jacobi solver here: one-sided Jacobi on a 384 x 384 matrix is a chain of ~14 sweeps x 383
// dependent pair-steps (about 5000 launches-worth of latency); tridiagonalisation is 382 dependent steps.
// Jacobi stays in use where ALL eigenvectors are needed (backward of multi-layer teachers) and for the
// small LDS-resident SVDs.
//
// One workgroup per matrix; the matrix stays in L2 and is streamed row-wise (coalesced), the Householder
// vector lives in LDS.  Everything is fp32, like LAPACK's ssytd2 / sstebz / sstein it restates.
#include "basd_common.h"
#include <stdlib.h>
#include <atomic>

namespace basd {

// ---------------------------------------------------------------------------
// A = Q T Q^T, lower variant: Q = H_0 H_1 ... H_{n-2}, H_j = I - tau_j v_j v_j^T.
// grid = batch, block = 1024.   A (n x n row-major) is destroyed; d (n), e (n-1), tau (n-1) out;
// Vh (n x n row-major): row j = v_j (zeros up to j, 1 at j+1), for the back-transformation.
// ---------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only: global stores (reflector rows, updated matrix rows) stay in
// flight across it.  That is sufficient here because every matrix row is always read and written by the SAME
// wave (static row ownership below), whose own accesses to an address are served in issue order.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One barrier: every call site owns its scratch slots, and between two uses of a slot array all threads pass
// at least one other workgroup barrier (so nobody can still be reading the previous round's partials).
__device__ __forceinline__ float block_sum_lds(float v, float* scratch, int nw) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    lds_barrier();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}

// ---- hand-off of per-row results between the workgroups that share one matrix -----------------
// One 16-byte granule per matrix row and step parity: (p_r, captured column entry, tag, 0), written and polled
// with sc1 accesses (served by L2 / fabric, never by a CU's L1; coherent across XCDs).  The tag travels with
// the data, so there is no separate flag, fence or atomic on the dependent chain: a consumer simply re-reads
// the granule until it carries the tag of the current step.  Two parities suffice: a workgroup can only be one
// step ahead of the slowest one (it needs that one's granules of the step in between).
// tag = (launch nonce << 12) | step: granules left behind by an earlier launch in a recycled buffer never match,
// so the buffer needs no clearing (and no fill kernel's cached lines can land on top of a published granule).
typedef unsigned tri_u32x4 __attribute__((ext_vector_type(4)));
typedef float tri_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void granule_store(uint4* p, float a, float b, unsigned tag) {
    const tri_u32x4 x = {__float_as_uint(a), __float_as_uint(b), tag, __float_as_uint(a) ^ __float_as_uint(b) ^ tag};
    // the trailing s_nop covers the >8-byte store's data-register hazard: hipcc pads nothing after inline asm,
    // and its next instruction may otherwise overwrite x before the store has read it
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ tri_u32x4 granule_load(const uint4* p) {
    tri_u32x4 x;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p) : "memory");
    return x;
}

// Eight row chunks in flight, then one wait: hipcc's schedulers otherwise sink each load next to its first
// use and serialise the eight L2 round trips (observed: load, s_waitcnt vmcnt(0), arithmetic, next load ...).
__device__ __forceinline__ void load8_rows(float4 (&a)[8], const float* p0, const float* p1, const float* p2,
                                           const float* p3, const float* p4, const float* p5, const float* p6,
                                           const float* p7) {
    tri_f32x4 x0, x1, x2, x3, x4, x5, x6, x7;
    asm volatile(
        "global_load_dwordx4 %0, %8, off\n\t"
        "global_load_dwordx4 %1, %9, off\n\t"
        "global_load_dwordx4 %2, %10, off\n\t"
        "global_load_dwordx4 %3, %11, off\n\t"
        "global_load_dwordx4 %4, %12, off\n\t"
        "global_load_dwordx4 %5, %13, off\n\t"
        "global_load_dwordx4 %6, %14, off\n\t"
        "global_load_dwordx4 %7, %15, off\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7)
        : "memory");
    a[0] = make_float4(x0.x, x0.y, x0.z, x0.w); a[1] = make_float4(x1.x, x1.y, x1.z, x1.w);
    a[2] = make_float4(x2.x, x2.y, x2.z, x2.w); a[3] = make_float4(x3.x, x3.y, x3.z, x3.w);
    a[4] = make_float4(x4.x, x4.y, x4.z, x4.w); a[5] = make_float4(x5.x, x5.y, x5.z, x5.w);
    a[6] = make_float4(x6.x, x6.y, x6.z, x6.w); a[7] = make_float4(x7.x, x7.y, x7.z, x7.w);
}

#ifdef BASD_TAIL_DBG
// diagnostic builds: s_memtime / 100 MHz stamps of the last factorisations (tools/tail_stamps_in_step.py); two sets:
// launches that also deliver the ranks (the chain the host waits for) and the others
__device__ long long g_tail_dbg[2 * 8 * 2 * 1024];
#endif
constexpr int TRI_RB = 8;      // rows per group: RB independent load streams keep L2 latency covered
constexpr int TRI_BLK = 32;    // rows per ownership block (4 groups)

// grid = P * batch_pad workgroups of 1024 threads; workgroup id = p * batch_pad + z works on matrix z as
// member p of P (ids that differ by a multiple of 8 are dealt to one XCD: the P members share an L2).
// Row blocks of 32 are owned block-cyclically (block b -> member b % P) for the whole factorisation: a
// member reads and writes only its own rows of A.  Per step every member forms the reflector redundantly
// (same inputs, same arithmetic, bit-identical), runs the fused update + matrix-vector pass over its rows,
// publishes (p_r, next column entry) of those rows and collects the other members' -- the only exchange.
// P == 1 degenerates to the single-workgroup factorisation (no exchange at all).
template <bool VEC, bool FULL>
__global__ void __launch_bounds__(1024) tridiag_kernel(float* __restrict__ A, long a_batch_stride, int n, int batch,
                                                       int batch_pad, int P, float* __restrict__ d,
                                                       float* __restrict__ e, float* __restrict__ tau_out,
                                                       float* __restrict__ Vh, uint4* __restrict__ xg,
                                                       int* __restrict__ err, unsigned tag_base, int lag_member,
                                                       int j_stop, float* __restrict__ pend) {
    // ONE pass over the trailing block per step: the rank-2 update of step j-1 is applied lazily while the
    // rows are read for the matrix-vector product of step j (a' = a - v_r w_c - w_r v_c ; p_r += a' u_c), and
    // the column the next reflector is built from is captured on the way.  All vectors are indexed by ABSOLUTE
    // row/column (zero below the active block), so the pass runs over 16-byte aligned column chunks.
    // Latency-critical: a chain of dependent steps on few waves, usually sharing its SIMDs with throughput
    // kernels of other streams (Gram MFMA loops, the Procrustes Jacobi).  Top wave priority makes the issue
    // arbiter serve these waves first; the background kernels lose next to nothing.
    __builtin_amdgcn_s_setprio(3);
#ifdef BASD_TAIL_DBG
    // 100 MHz clock at the begin / end of the shared stage (workgroup 0; set 0: the two-matrix teacher launch)
    if (blockIdx.x == 0 && threadIdx.x == 0) g_tail_dbg[(batch == 2 ? 0 : 8 * 2 * 1024) + 8 * 2 * 1024 - 8] = wall_clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* v = sm;             // previous reflector
    float* w = sm + n;         // its w = p + gamma v
    float* u = sm + 2 * n;     // current reflector
    float* pw = sm + 3 * n;    // current p, then current w
    float* col = sm + 4 * n;   // column j of the up-to-date matrix (rows >= j)
    float* cap4 = sm + 5 * n;  // 4 n: per row, the aligned 4-column chunk holding column r0 as captured by the pass
    __shared__ float red[32];
    const int z = blockIdx.x % batch_pad, p = blockIdx.x / batch_pad;
    if (z >= batch) return;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, nw = nthr >> 6;
    // wave index as a scalar: everything derived from it (row bases, per-row reflector entries) then lives in
    // SGPRs, and the rows are addressed as scalar base + one shared per-lane column offset
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* Az = A + (long)z * a_batch_stride;
    float* dz = d + (long)z * n;
    float* ez = e + (long)z * n;
    float* tz = tau_out + (long)z * n;
    float* Vz = Vh + (long)z * n * n;
    uint4* xz = xg + (long)z * 2 * n;
    constexpr int RB = TRI_RB;
    const int nblk = (n + TRI_BLK - 1) / TRI_BLK;
    int budget = 1 << 22;      // polls before a member gives up on its partners (never reached when all are resident)
    if (blockIdx.x == 0 && threadIdx.x < 8) __hip_atomic_store(err + threadIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int r = tid; r < n; r += nthr) {
        v[r] = 0.f;
        w[r] = 0.f;
        col[r] = Az[(long)r * n];
    }
    __syncthreads();
    // my last block: once the active rows have passed it I hold no live row, nobody reads my granules any more
    // and nobody would wait for me either (the lock-step below is among members with live rows) -- so leave.
    const int my_last_blk = ((nblk - 1 - p) / P) * P + p;
    const int last_owner = (nblk - 1) % P;           // owns row n-1: alive to the end
    // Reflector of step j from column j of the current matrix (col[], rows >= j): every thread forms tau / beta
    // itself (same inputs everywhere: no broadcast); the member that owns row j+1 -- it has live rows at step j,
    // hence is in step -- writes d_j, e_j, tau_j and the reflector row.  `xn2` = sum_{r > j+1} col[r]^2.
    float tau = 0.f;
    auto form_reflector = [&](int j, float xn2) {
        const int r0 = j + 1;
        const float alpha = col[r0];
        float beta = alpha;
        tau = 0.f;
        if (xn2 > 0.f) {
            beta = -copysignf(sqrtf(fmaf(alpha, alpha, xn2)), alpha);
            tau = (beta - alpha) / beta;
        }
        const bool writes_v = p == (r0 / TRI_BLK) % P;
        if (writes_v && tid == 0) {
            dz[j] = col[j];
            ez[j] = beta;
            tz[j] = tau;
        }
        const float scal = tau != 0.f ? 1.f / (alpha - beta) : 0.f;
        for (int r = tid; r < n; r += nthr) {
            const float ur = r < r0 ? 0.f : (r == r0 ? 1.f : col[r] * scal);
            u[r] = ur;
            if (writes_v) Vz[(long)j * n + r] = ur;
        }
        lds_barrier();
    };
    {
        float part = 0.f;
        for (int r = 2 + tid; r < n; r += nthr) part = fmaf(col[r], col[r], part);
        form_reflector(0, block_sum_lds(part, red, nw));
    }
    // j_stop < n - 1: only steps [0, j_stop) are done here; the trailing block of order n - j_stop goes on in
    // tridiag_tail_kernel (registers of one CU, no exchange), which gets the matrix rows as stored by the last pass
    // plus the still pending rank-2 update (v, w) of step j_stop - 1 through `pend`.
    for (int j = 0; j < n - 1 && j < j_stop; ++j) {
        const int r0 = j + 1;
        if (r0 / TRI_BLK > my_last_blk) break;       // uniform over the workgroup
        if (p == lag_member) __builtin_amdgcn_s_sleep(127);      // test hook: one member falls behind every step
        const int c_begin = VEC ? (r0 & ~3) : r0;
        // my live row groups: local group lg -> block (lg / 4) * P + p, group lg % 4 of it; wave lg % nw owns it
        const int b_first = r0 / TRI_BLK;
        const int lb0 = b_first > p ? (b_first - p + P - 1) / P : 0;
        for (int lg = lb0 * 4 + ((wave - (lb0 * 4) % nw + nw) % nw);; lg += nw) {
            const int b = (lg >> 2) * P + p;
            if (b >= nblk) break;
            const int rb = (b * 4 + (lg & 3)) * RB;
            if (rb >= n) break;
            if (rb + RB <= r0) continue;
            // FULL (n a multiple of the group size, vector path): no predicates at all.  Rows of the first live
            // group that lie above r0 are already reduced; they are carried along (their pending update is a
            // finished row's business, their p and captured entries land in slots nobody reads), so the whole
            // chunk iteration is one basic block: eight loads in flight, then arithmetic, then eight stores.
            float acc[RB], vr[RB], wr[RB];
            bool live[RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                live[q] = FULL || (rb + q >= r0 && rb + q < n);
                acc[q] = 0.f;
                vr[q] = live[q] ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v[rb + q]))) : 0.f;
                wr[q] = live[q] ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w[rb + q]))) : 0.f;
            }
            if (VEC) {
                for (int c = c_begin + 4 * lane; c < n; c += 256) {
                    const float4 vv = *(const float4*)(v + c), ww = *(const float4*)(w + c);
                    const float4 uu = *(const float4*)(u + c);
                    float4 a[RB];
                    if (FULL) {
                        const float* rp = Az + (long)rb * n + c;
                        load8_rows(a, rp, rp + n, rp + 2 * n, rp + 3 * n, rp + 4 * n, rp + 5 * n, rp + 6 * n, rp + 7 * n);
                    } else {
#pragma unroll
                        for (int q = 0; q < RB; ++q)
                            if (live[q]) a[q] = *(const float4*)(Az + (long)(rb + q) * n + c);
                    }
#pragma unroll
                    for (int q = 0; q < RB; ++q) {
                        if (live[q]) {
                            a[q].x -= fmaf(vr[q], ww.x, wr[q] * vv.x);
                            a[q].y -= fmaf(vr[q], ww.y, wr[q] * vv.y);
                            a[q].z -= fmaf(vr[q], ww.z, wr[q] * vv.z);
                            a[q].w -= fmaf(vr[q], ww.w, wr[q] * vv.w);
                            acc[q] = fmaf(a[q].x, uu.x, fmaf(a[q].y, uu.y, fmaf(a[q].z, uu.z, fmaf(a[q].w, uu.w, acc[q]))));
                        }
                    }
#pragma unroll
                    for (int q = 0; q < RB; ++q)
                        if (live[q]) *(float4*)(Az + (long)(rb + q) * n + c) = a[q];
                    if (c == c_begin) {      // the chunk holding column r0: capture it for the next reflector
#pragma unroll
                        for (int q = 0; q < RB; ++q)
                            if (live[q]) *(float4*)(cap4 + 4 * (rb + q)) = a[q];
                    }
                }
            } else {
                for (int c = c_begin + lane; c < n; c += 64) {
                    const float vc = v[c], wc = w[c], uc = u[c];
#pragma unroll
                    for (int q = 0; q < RB; ++q) {
                        if (live[q]) {
                            const float an = Az[(long)(rb + q) * n + c] - fmaf(vr[q], wc, wr[q] * vc);
                            Az[(long)(rb + q) * n + c] = an;
                            acc[q] = fmaf(an, uc, acc[q]);
                            if (c == r0) cap4[4 * (rb + q)] = an;
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const float sum = wave_sum(acc[q]);
                if (lane == 0 && live[q]) pw[rb + q] = tau * sum;
            }
        }
        lds_barrier();
        const int coff = r0 - c_begin;      // position of column r0 inside the captured chunk
        if (P > 1) {
            uint4* xp = xz + (long)(j & 1) * n;
            const unsigned tag = tag_base | (unsigned)j;
            // publish my live blocks: wave (local block % nw), lanes 0..31 -> 512 contiguous bytes per store
            if (lane < TRI_BLK) {
                for (int lb = lb0 + wave; lb * P + p < nblk; lb += nw) {
                    const int r = (lb * P + p) * TRI_BLK + lane;
                    if (r < n) granule_store(xp + r, pw[r], cap4[4 * r + coff], tag);
                }
            }
            // collect the other members' rows
            for (int r = b_first * TRI_BLK + tid; r < n; r += nthr) {
                if ((r / TRI_BLK) % P == p) continue;
                tri_u32x4 g = granule_load(xp + r);
                while ((g.z != tag || g.w != (g.x ^ g.y ^ g.z)) && budget > 0) {
                    __builtin_amdgcn_s_sleep(1);
                    --budget;
                    g = granule_load(xp + r);
                }
                if (g.z != tag && err[1] == 0) {      // first give-up of the launch: leave a trace for the host
                    err[1] = j + 1;
                    err[2] = r;
                    err[3] = p | (z << 8);
                    err[4] = (int)g.z;
                    err[5] = (int)tag;
                }
                pw[r] = __uint_as_float(g.x);
                cap4[4 * r + coff] = __uint_as_float(g.y);
            }
            lds_barrier();
        }
        float gp = 0.f;
        for (int r = r0 + tid; r < n; r += nthr) gp = fmaf(pw[r], u[r], gp);
        const float pw0 = pw[r0];                        // read before anything below rewrites LDS
        const float gamma = -0.5f * tau * block_sum_lds(gp, red + 16, nw);
        const float w0 = fmaf(gamma, 1.f, pw0);          // u[r0] = 1
        // w_j = p + gamma u; column r0 of the matrix with update j applied -- the column the NEXT reflector is
        // built from, so its norm is accumulated here and the reflector of step j+1 is formed right away (one
        // block reduction and one loop per step less than forming it at the top of the step); (u, w_j) become the
        // pending update
        float part = 0.f;
        for (int r = tid; r < n; r += nthr) {
            const float ur = u[r];
            const float wn = r >= r0 ? fmaf(gamma, ur, pw[r]) : 0.f;
            if (r >= r0) {
                const float cn = cap4[4 * r + coff] - fmaf(ur, w0, wn);
                col[r] = cn;
                if (r > r0 + 1) part = fmaf(cn, cn, part);
            }
            v[r] = ur;
            w[r] = wn;
        }
        const float xn2 = block_sum_lds(part, red, nw);   // its barrier also publishes col[] / v[] / w[]
        if (j + 1 < n - 1 && j + 1 < j_stop) form_reflector(j + 1, xn2);
    }
    __syncthreads();
#ifdef BASD_TAIL_DBG
    if (p == last_owner && z == 0 && threadIdx.x == 0) g_tail_dbg[(batch == 2 ? 0 : 8 * 2 * 1024) + 8 * 2 * 1024 - 7] = wall_clock64();
#endif
    if (budget <= 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (j_stop < n - 1) {
        // hand-over: the owner of row j_stop was in step to the end, its (v, w) are those of step j_stop - 1
        if (p == (j_stop / TRI_BLK) % P) {
            float* pz = pend + (long)z * 2 * n;
            for (int r = tid; r < n; r += nthr) {
                pz[r] = v[r];
                pz[n + r] = w[r];
            }
        }
        return;
    }
    if (p == last_owner) {
        if (tid == 0) {
            dz[n - 1] = col[n - 1];
            ez[n - 1] = 0.f;
            tz[n - 1] = 0.f;
        }
        for (int c = tid; c < n; c += nthr) Vz[(long)(n - 1) * n + c] = 0.f;
    }
}

// ---------------------------------------------------------------------------
// The same factorisation for a trailing block that fits ONE CU's registers: steps [j0, n-1) on the block of order
// m = n - j0 <= 64 * CPL, one workgroup of 64 * WAVES threads per matrix, no exchange with anybody.
// Thread (wave w, lane l) keeps M[w + WAVES * i][l + 64 * k] (i < RPW, k < CPL) in registers for the whole
// factorisation; a step is one pass over the registers (pending rank-2 update of the previous step applied lazily,
// product with the current reflector accumulated on the way) and four workgroup barriers:
//   pass      a' = a - v_r w_c - w_r v_c ;  acc_c += a' u_r           (u_r, v_r, w_r wave-uniform: LDS broadcasts)
//   A         per-wave column partials -> LDS;  threads c < m: p_c = tau * sum over waves
//   B         gamma = -tau/2 p.u ;  w = p + gamma u ;  next pivot row (captured by the pass) updated analytically
//   C         its norm -> reflector of the next step (d, e, tau written)
//   D         u, v, w published;  reflector row stored to Vh
// It is the algorithm of tridiag_kernel run on the transpose (rows take the part of columns: the matrix is
// symmetric), with the row-wise sums replaced by per-lane column sums so that no cross-lane reduction sits in the
// pass.  Rows that are already reduced are skipped (wave-uniform), so the pass shrinks with the trailing block.
// n <= 64 * CPL: the whole factorisation (j0 = 0, pend = nullptr) -- nothing spins anywhere.
// n >  64 * CPL: tridiag_kernel does steps [0, j0) and hands over the pending update (pend: v, w of step j0 - 1).
// ---------------------------------------------------------------------------
// Marchenko-Pastur rank of the tridiagonal (dz, ez) by the whole workgroup (defined below, next to the Sturm count).
struct MpRankOut {
    int* rank_out;        // (count) device
    int* host_mirror;     // nullable: pinned host memory, count + 8 ints
    const int* status;    // nullable: 8 status words of the factorisation, copied behind the ranks
    double factor;
    int cap, count;
};
__device__ __forceinline__ void mp_rank_block(const float* dz, const float* ez, int n, int z, const MpRankOut& o,
                                              float* thr_out);

template <int WAVES, int RPW, int CPL>
__global__ void __launch_bounds__(64 * WAVES) tridiag_tail_kernel(float* __restrict__ A, long a_batch_stride, int n,
                                                                  int j0, const float* __restrict__ pend,
                                                                  float* __restrict__ d, float* __restrict__ e,
                                                                  float* __restrict__ tau_out,
                                                                  float* __restrict__ Vh, MpRankOut rk) {
    constexpr int MMAX = 64 * CPL;
    static_assert(WAVES * RPW >= MMAX, "every row needs an owner");
    static_assert(WAVES * 64 >= MMAX, "one thread per column in the vector phases");
    __builtin_amdgcn_s_setprio(3);
    __shared__ float u[MMAX], v[MMAX], w[MMAX], cap[MMAX], pw[MMAX], col[MMAX];
    __shared__ float dloc[MMAX], eloc[MMAX];      // d, e of the local steps (for the rank at the end)
    __shared__ float part[WAVES][MMAX];
    __shared__ float red[2][16];
    const int z = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = n - j0;
    float* Az = A + (long)z * a_batch_stride;
    float* dz = d + (long)z * n;
    float* ez = e + (long)z * n;
    float* tz = tau_out + (long)z * n;
    float* Vz = Vh + (long)z * n * n;
    const float* pz = pend ? pend + (long)z * 2 * n : nullptr;

    // ---- load the trailing block (pending update applied), zero padded to MMAX
    float a[RPW][CPL];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wave + WAVES * i;
        const float vr = (pz && r < m) ? pz[j0 + r] : 0.f, wr = (pz && r < m) ? pz[n + j0 + r] : 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c = lane + 64 * k;
            float x = 0.f;
            if (r < m && c < m) {
                x = Az[(long)(j0 + r) * n + j0 + c];
                if (pz) x -= fmaf(vr, pz[n + j0 + c], wr * pz[j0 + c]);
            }
            a[i][k] = x;
        }
    }
    for (int c = tid; c < MMAX; c += 64 * WAVES) {
        u[c] = 0.f;
        v[c] = 0.f;
        w[c] = 0.f;
        cap[c] = 0.f;
    }
    // row 0 of the block: what the first reflector is built from
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) col[lane + 64 * k] = a[0][k];
    }
    __syncthreads();

    // reflector of local step jl from col[] (row jl of the current block, entries >= jl), as tridiag_kernel's
    // form_reflector; threads c < MMAX write u; d / e / tau by one thread.  xn2 = sum_{c > jl + 1} col[c]^2.
    float tau = 0.f;
    auto form_reflector = [&](int jl, float xn2) {
        const int r0 = jl + 1;
        const float alpha = col[r0];
        float beta = alpha;
        tau = 0.f;
        if (xn2 > 0.f) {
            // 1-ulp hardware sqrt / reciprocal: IEEE sqrtf and division are ~30-instruction software sequences, three
            // of them in a row on the dependent chain of every step
            beta = -copysignf(__builtin_amdgcn_sqrtf(fmaf(alpha, alpha, xn2)), alpha);
            tau = (beta - alpha) * __builtin_amdgcn_rcpf(beta);
        }
        if (tid == 0) {
            dz[j0 + jl] = dloc[jl] = col[jl];
            ez[j0 + jl] = eloc[jl] = beta;
            tz[j0 + jl] = tau;
        }
        const float scal = tau != 0.f ? __builtin_amdgcn_rcpf(alpha - beta) : 0.f;
        if (tid < MMAX) u[tid] = tid < r0 ? 0.f : (tid == r0 ? 1.f : col[tid] * scal);
    };
    auto block_sum4 = [&](float x, float* slot) {      // sum over the threads < MMAX (whole waves), one barrier
        x = wave_sum(x);
        if (lane == 0 && tid < MMAX) slot[wave] = x;
        __syncthreads();
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) r += slot[i];
        return r;
    };
    {
        const float cv = tid < MMAX ? col[tid] : 0.f;
        const float xn2 = block_sum4((tid > 1 && tid < MMAX) ? cv * cv : 0.f, red[0]);
        form_reflector(0, xn2);
        __syncthreads();
    }
    // the reflector row of step j in Vh: zeros up to column j, then u
    auto store_reflector = [&](int jl) {
        float* vrow = Vz + (long)(j0 + jl) * n;
        for (int c = tid; c < n; c += 64 * WAVES) vrow[c] = c < j0 ? 0.f : u[c - j0];
    };
    store_reflector(0);

    for (int jl = 0; jl < m - 1; ++jl) {
        const int r0 = jl + 1;
        // ---- pass over the registers
        float vc[CPL], wc[CPL], acc[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            vc[k] = v[lane + 64 * k];
            wc[k] = w[lane + 64 * k];
            acc[k] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int r = wave + WAVES * i;
            if (r >= r0 && r < m) {                      // wave-uniform
                const float ur = u[r], vr = v[r], wr = w[r];
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    a[i][k] -= fmaf(vr, wc[k], wr * vc[k]);
                    acc[k] = fmaf(a[i][k], ur, acc[k]);
                }
                if (r == r0) {                           // the row the next reflector is built from
#pragma unroll
                    for (int k = 0; k < CPL; ++k) cap[lane + 64 * k] = a[i][k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < CPL; ++k) part[wave][lane + 64 * k] = acc[k];
        __syncthreads();                                                                   // A
        float pc = 0.f, uc = 0.f;
        if (tid < MMAX) {
            float sacc = 0.f;
#pragma unroll
            for (int q = 0; q < WAVES; ++q) sacc += part[q][tid];
            pc = tau * sacc;
            uc = u[tid];
            pw[tid] = pc;
        }
        const float gamma = -0.5f * tau * block_sum4(tid < MMAX ? pc * uc : 0.f, red[1]);   // B (publishes pw)
        float wn = 0.f, part2 = 0.f;
        if (tid < MMAX) {
            const float w0 = pw[r0] + gamma;             // u[r0] = 1
            if (tid >= r0) {
                wn = fmaf(gamma, uc, pc);
                const float cn = cap[tid] - fmaf(uc, w0, wn);
                col[tid] = cn;
                if (tid > r0 + 1) part2 = cn * cn;
            }
        }
        const float xn2 = block_sum4(part2, red[0]);                                        // C (publishes col)
        if (tid < MMAX) {
            v[tid] = uc;
            w[tid] = wn;
        }
        if (jl + 1 < m - 1) form_reflector(jl + 1, xn2);
        __syncthreads();                                                                    // D
        if (jl + 1 < m - 1) store_reflector(jl + 1);
    }
    if (tid == 0) {
        dz[n - 1] = dloc[m - 1] = col[m - 1];
        ez[n - 1] = eloc[m - 1] = 0.f;
        tz[n - 1] = 0.f;
    }
    for (int c = tid; c < n; c += 64 * WAVES) Vz[(long)(n - 1) * n + c] = 0.f;
    // The Marchenko-Pastur rank of the first `count` matrices of the batch right here: the host is waiting for it, and
    // this workgroup already owns a CU -- a separate launch would queue for wave slots behind the throughput kernels
    // of the other streams (rocprofv3, cfg-2: 0.17-0.49 ms inside a step for 0.05 ms of work).  The tridiagonal is
    // staged in LDS (the partials buffer is free now); entries of the first stage come from global memory.
    if (rk.rank_out && z < rk.count && 2 * n <= 8 * MMAX) {
        __syncthreads();
        float* dl = &part[0][0];
        float* el = dl + n;
        for (int c = tid; c < n; c += 64 * WAVES) {
            dl[c] = c < j0 ? dz[c] : dloc[c - j0];
            el[c] = c < j0 ? ez[c] : eloc[c - j0];
        }
        __syncthreads();
        mp_rank_block(dl, el, n, z, rk, nullptr);
    }
}

// ---------------------------------------------------------------------------
// tridiag_tail_kernel, second form (8 waves).  What limits a step of the first form is not the barriers (~100 cycles
// each) but INSTRUCTION ISSUE and the LDS return path: a SIMD issues one wave64 VALU instruction per 4 cycles, the waves
// of a SIMD share it, and s_memtime stamps (tools/probe/tail_phase_probe.hip) put the pass at 1500-2300 cycles and the
// scalar part of the step -- which the threads < m of four waves ran between four barriers -- at ~3000.  Per matrix row
// the first form issues 8 packed FMAs, 10 v_mov that splat (u_r, v_r, w_r) into register pairs, and a uniform branch.
// Here
//   * rows are consumed in pairs (r_a = wave + 16 P, r_b = r_a + 8); per wave and pair LDS holds (v_a, w_a, v_b, w_b)
//     and (u_a, u_b), fetched one pair ahead with one b128 + one b64 read, and a wave-uniform operand of a packed FMA is
//     taken from either half of a register pair with op_sel / op_sel_hi: a pair of rows is 12 v_pk_fma_f32 and nothing
//     else -- no splat by the VALU, no splat by LDS (that form was bound by the LDS return path);
//   * the pass starts at its first live pair (rows below r0 are finished: the live pairs are a suffix): one straight-line
//     instance per entry pair behind a jump table, and the pivot row is copied out behind a second one: two indirect
//     jumps instead of ~70 uniform branches;
//   * lane l of a wave owns the columns c_k = (l & 7) + 32 (l >> 3) + 8 k, k < 4:
//     exactly the rows of pairs 2 (l >> 3), 2 (l >> 3) + 1 of wave l & 7, so the column form of u, v, w IS the row
//     operand table -- wave 0 publishes a step with three ds_write_b128 per lane;
//   * the scalar part of the step (partials -> p, gamma, w, the next pivot row, its norm, the next reflector) is ONE
//     wave's dependent chain: wave 0 keeps u in column form in registers, uses wave-level sums only and branch-free
//     selects, and is alone on its SIMD meanwhile (the others wait at barrier B) -- its instructions cost latency, not
//     the issue slots of every wave; d, e, tau collect in LDS and are stored once at the end; the reflector row is
//     stored by the LAST wave behind barrier B;
//   * two workgroup barriers per step: A publishes the partials and the captured row, B the operands of the next pass.
// Same arithmetic as the first form (same update order, same reflector formulas).
// ---------------------------------------------------------------------------
#ifdef BASD_TAIL_DBG
// s_memtime stamps at the phase boundaries (tools/probe/tail_phase_probe.hip)
#define TAIL_STAMP(slot) do { if (z == 0 && lane == 0 && (wave == 0 || wave == WAVES - 1)) g_tail_dbg[(rk.rank_out ? 0 : 8 * 2 * 1024) + (jl * 8 + (slot)) * 2 + (wave != 0)] = clock64(); } while (0)
// shader clock against the constant 100 MHz clock over the whole step loop (last four words of the set)
#define TAIL_CLOCKS(which) do { if (z == 0 && tid == 0) { long long* q_ = g_tail_dbg + (rk.rank_out ? 0 : 8 * 2 * 1024) + 8 * 2 * 1024 - 4 + 2 * (which); q_[0] = clock64(); q_[1] = wall_clock64(); } } while (0)
#else
#define TAIL_STAMP(slot) do { } while (0)
#define TAIL_CLOCKS(which) do { } while (0)
#endif
// test builds (EXTRA=-DBASD_TAIL_JITTER): waves fall asleep at the phase boundaries in a wave- and step-dependent pattern;
// results must not change by a bit (tools/probe/tail_phase_probe.hip checks against the four-barrier kernel)
#ifdef BASD_TAIL_JITTER
#define TAIL_JITTER(slot) do { if (((wave * 7 + jl * 3 + (slot)) % 5) == 0) __builtin_amdgcn_s_sleep(60); } while (0)
#else
#define TAIL_JITTER(slot) do { } while (0)
#endif
typedef float tri_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float lane_bcast(float x, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}
// x[q] of four wave-uniform values by a wave-uniform q in 0..3: scalar selects (hipcc turns the C form into branches)
__device__ __forceinline__ float uniform_select4(float t0, float t1, float t2, float t3, int q) {
    int r;
    asm("s_cmp_eq_u32 %5, 1\n\t"
        "s_cselect_b32 %0, %2, %1\n\t"
        "s_cmp_eq_u32 %5, 2\n\t"
        "s_cselect_b32 %0, %3, %0\n\t"
        "s_cmp_eq_u32 %5, 3\n\t"
        "s_cselect_b32 %0, %4, %0"
        : "=&s"(r)
        : "s"(__float_as_int(t0)), "s"(__float_as_int(t1)), "s"(__float_as_int(t2)), "s"(__float_as_int(t3)), "s"(q)
        : "scc");
    return __int_as_float(r);
}
// eight b128 rows, 1 KB apart, in flight together (hipcc serialises them: read, wait, add, read ...)
__device__ __forceinline__ void lds_read8_b128(unsigned addr, tri_f32x4 (&x)[8]) {
    asm volatile(
        "ds_read_b128 %0, %8\n\t"
        "ds_read_b128 %1, %8 offset:1024\n\t"
        "ds_read_b128 %2, %8 offset:2048\n\t"
        "ds_read_b128 %3, %8 offset:3072\n\t"
        "ds_read_b128 %4, %8 offset:4096\n\t"
        "ds_read_b128 %5, %8 offset:5120\n\t"
        "ds_read_b128 %6, %8 offset:6144\n\t"
        "ds_read_b128 %7, %8 offset:7168\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])
        : "v"(addr)
        : "memory");
}
__device__ __forceinline__ void pkfma_lo(tri_f2& a, tri_f2 p, tri_f2 b) {      // a += p.x * b
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a) : "v"(p), "v"(b));
}
__device__ __forceinline__ void pkfma_hi(tri_f2& a, tri_f2 p, tri_f2 b) {      // a += p.y * b
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a) : "v"(p), "v"(b));
}
// rows r_a, r_b of one pair:  a' = a - w_r v_c - v_r w_c ;  acc_c += a' u_r   (nvc = -v_c, nwc = -w_c)
__device__ __forceinline__ void tail_pair_update(tri_f2 (&ra)[2], tri_f2 (&rb)[2], tri_f32x4 vw, tri_f2 u2,
                                                 const tri_f2 (&nvc)[2], const tri_f2 (&nwc)[2], tri_f2 (&acc)[2]) {
    const tri_f2 pa = {vw.x, vw.y}, pb = {vw.z, vw.w};         // (v_a, w_a), (v_b, w_b)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        pkfma_hi(ra[q], pa, nvc[q]);
        pkfma_hi(rb[q], pb, nvc[q]);
        pkfma_lo(ra[q], pa, nwc[q]);
        pkfma_lo(rb[q], pb, nwc[q]);
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[q]) : "v"(ra[q]), "v"(u2));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[q]) : "v"(rb[q]), "v"(u2));
    }
}

// One pair of the pass.  The operands of pair P were requested while pair P - 1 was computed (plain LDS loads: the
// compiler places the waits, also across the case labels through which the pass is entered at its first live pair);
// hand-counted s_waitcnt around asm reads would leave stale registers in front of every control-flow join.
#define TAIL_PAIR(P)                                                                             \
    case P: {                                                                                    \
        const tri_f32x4 cvw = nvw;                                                               \
        const tri_f2 cu = nu;                                                                    \
        if ((P) + 1 < NP) {                                                                      \
            nvw = opvw[(P) + 1 < NP ? (P) + 1 : (P)];                                            \
            nu = opu[(P) + 1 < NP ? (P) + 1 : (P)];                                              \
        }                                                                                        \
        tail_pair_update(a[2 * (P)], a[2 * (P) + 1], cvw, cu, nvc, nwc, acc);                    \
    }                                                                                            \
    [[fallthrough]];
#define TAIL_CAP(I)                                                                                               \
    case I:                                                                                                       \
        cap4[lane] = tri_f32x4{a[I][0].x, a[I][0].y, a[I][1].x, a[I][1].y};                                       \
        break;

__global__ void __launch_bounds__(512) tridiag_tail2_kernel(float* __restrict__ A, long a_batch_stride, int n, int j0,
                                                            const float* __restrict__ pend, float* __restrict__ d,
                                                            float* __restrict__ e, float* __restrict__ tau_out,
                                                            float* __restrict__ Vh, MpRankOut rk) {
    // 8 waves: two per SIMD (a 16-wave variant of this form was no faster and needs more than its 128 VGPRs)
    constexpr int WAVES = 8, LW = 3, CPL = 4, MMAX = 64 * CPL, RPW = MMAX / WAVES, NP = RPW / 2;   // row r = wave + WAVES i
    __builtin_amdgcn_s_setprio(3);
#ifdef BASD_TAIL_DBG
    if (blockIdx.x == 0 && threadIdx.x == 0) g_tail_dbg[(rk.rank_out ? 0 : 8 * 2 * 1024) + 8 * 2 * 1024 - 6] = wall_clock64();
#endif
    __shared__ tri_f32x4 op_vw[WAVES][NP];        // (v_a, w_a, v_b, w_b) of the wave's pair P: rows wave + 2 WAVES P, + WAVES
    __shared__ __attribute__((aligned(16))) tri_f2 op_u[WAVES][NP];   // (u_a, u_b)
    __shared__ tri_f32x4 part4[WAVES][64];        // per-wave column partials of the lane's four columns
    __shared__ tri_f32x4 cap4[64];                // the captured pivot row
    __shared__ float dloc[MMAX], eloc[MMAX], tloc[MMAX];
    const int z = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = n - j0;
    float* Az = A + (long)z * a_batch_stride;
    float* dz = d + (long)z * n;
    float* ez = e + (long)z * n;
    float* tz = tau_out + (long)z * n;
    float* Vz = Vh + (long)z * n * n;
    const float* pz = pend ? pend + (long)z * 2 * n : nullptr;
    const int cbase = (lane & (WAVES - 1)) + 4 * WAVES * (lane >> LW);    // column of k = 0; k -> + WAVES k
    auto col_of = [&](int k) { return cbase + WAVES * k; };

    // ---- the trailing block (pending update of the shared stage applied), zero padded to MMAX
    tri_f2 a[RPW][2];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wave + WAVES * i;
        const float vr = (pz && r < m) ? pz[j0 + r] : 0.f, wr = (pz && r < m) ? pz[n + j0 + r] : 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c = col_of(k);
            float x = 0.f;
            if (r < m && c < m) {
                x = Az[(long)(j0 + r) * n + j0 + c];
                if (pz) x -= fmaf(vr, pz[n + j0 + c], wr * pz[j0 + c]);
            }
            a[i][k >> 1][k & 1] = x;
        }
    }
    // the part of the reflector rows left of the block is zero for every local step: written once, here
    for (int idx = tid; idx < m * j0; idx += 64 * WAVES) {
        const int r = idx / j0, c = idx - r * j0;
        Vz[(long)(j0 + r) * n + c] = 0.f;
    }
    // where this lane's columns sit in the operand tables (as rows): wave lane % WAVES, pairs 2 (lane / WAVES) and + 1
    tri_f32x4* const my_vw = &op_vw[lane & (WAVES - 1)][2 * (lane >> LW)];
    tri_f32x4* const my_u = (tri_f32x4*)&op_u[lane & (WAVES - 1)][2 * (lane >> LW)];

    // column-form state of wave 0
    float uc[CPL], cn[CPL], wn[CPL];
    float tau = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) uc[k] = cn[k] = wn[k] = 0.f;
    // column-form entry c, broadcast (c wave-uniform); branch-free: four readlanes and scalar selects
    auto pick = [&](const float (&x)[CPL], int c) {
        const int l = ((c & (WAVES - 1)) + WAVES * (c >> (LW + 2))) & 63, q = (c >> LW) & 3;
        const float t0 = lane_bcast(x[0], l), t1 = lane_bcast(x[1], l), t2 = lane_bcast(x[2], l), t3 = lane_bcast(x[3], l);
        return uniform_select4(t0, t1, t2, t3, q);
    };
    // wave 0: reflector of local step jl from cn[] (row jl of the current block); publishes u, v (= the old u), w
    auto next_reflector = [&](int jl, bool last) {
        const int r0 = jl + 1;
        float part2 = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k)        // column c_k = cbase + WAVES k against r0: the uniform side carries the k
            part2 = cbase > r0 - WAVES * k ? fmaf(cn[k], cn[k], part2) : part2;
        const float xn2 = wave_sum(part2);
        const float dnew = pick(cn, jl), alpha = pick(cn, r0);
        const bool live = !last && xn2 > 0.f;
        const float beta = live ? -copysignf(__builtin_amdgcn_sqrtf(fmaf(alpha, alpha, xn2)), alpha) : (last ? 0.f : alpha);
        tau = live ? (beta - alpha) * __builtin_amdgcn_rcpf(beta) : 0.f;
        const float scal = live ? __builtin_amdgcn_rcpf(alpha - beta) : 0.f;
        float vo[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int rk = r0 - WAVES * k;
            vo[k] = uc[k];
            uc[k] = last ? 0.f : (cbase < rk ? 0.f : (cbase == rk ? 1.f : cn[k] * scal));
        }
        my_vw[0] = tri_f32x4{vo[0], wn[0], vo[1], wn[1]};
        my_vw[1] = tri_f32x4{vo[2], wn[2], vo[3], wn[3]};
        my_u[0] = tri_f32x4{uc[0], uc[1], uc[2], uc[3]};
        if (lane == 0) {
            dloc[jl] = dnew;
            eloc[jl] = beta;
            tloc[jl] = tau;
        }
    };
    // the LAST wave, behind barrier B: reflector row of local step jl from the published u
    auto store_reflector = [&](int jl) {
        float* vrow = Vz + (long)(j0 + jl) * n + j0;
        const tri_f32x4 u4 = my_u[0];
        if (cbase + 0 < m) vrow[cbase + 0] = u4.x;
        if (cbase + WAVES < m) vrow[cbase + WAVES] = u4.y;
        if (cbase + 2 * WAVES < m) vrow[cbase + 2 * WAVES] = u4.z;
        if (cbase + 3 * WAVES < m) vrow[cbase + 3 * WAVES] = u4.w;
    };
    TAIL_CLOCKS(0);
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) cn[k] = a[0][k >> 1][k & 1];
        next_reflector(0, m <= 1);
    }
    lds_barrier();
    if (wave == WAVES - 1) store_reflector(0);

    for (int jl = 0; jl < m - 1; ++jl) {
        const int r0 = jl + 1;
        // ---- pass over the registers
        TAIL_STAMP(0);
        TAIL_JITTER(0);
        tri_f2 nvc[2], nwc[2], acc[2];
        {
            const tri_f32x4 o0 = my_vw[0], o1 = my_vw[1];
            nvc[0] = tri_f2{-o0.x, -o0.z}; nvc[1] = tri_f2{-o1.x, -o1.z};
            nwc[0] = tri_f2{-o0.y, -o0.w}; nwc[1] = tri_f2{-o1.y, -o1.w};
            acc[0] = acc[1] = tri_f2{0.f, 0.f};
        }
        {
            // first pair with a live row: rows wave + 2 WAVES P and + WAVES against r0
            const int p0 = r0 > wave + WAVES ? (r0 - wave - WAVES + 2 * WAVES - 1) >> (LW + 1) : 0;
            const tri_f32x4* const opvw = &op_vw[wave][0];
            const tri_f2* const opu = &op_u[wave][0];
            tri_f32x4 nvw = opvw[p0 < NP ? p0 : 0];
            tri_f2 nu = opu[p0 < NP ? p0 : 0];
            switch (p0) {
                TAIL_PAIR(0)
                TAIL_PAIR(1)
                TAIL_PAIR(2)
                TAIL_PAIR(3)
                TAIL_PAIR(4)
                TAIL_PAIR(5)
                TAIL_PAIR(6)
                TAIL_PAIR(7)
                TAIL_PAIR(8)
                TAIL_PAIR(9)
                TAIL_PAIR(10)
                TAIL_PAIR(11)
                TAIL_PAIR(12)
                TAIL_PAIR(13)
                TAIL_PAIR(14)
                TAIL_PAIR(15)
                default: break;
            }
        }
        if (wave == (r0 & (WAVES - 1))) {
            switch (r0 >> LW) {
                TAIL_CAP(0)
                TAIL_CAP(1)
                TAIL_CAP(2)
                TAIL_CAP(3)
                TAIL_CAP(4)
                TAIL_CAP(5)
                TAIL_CAP(6)
                TAIL_CAP(7)
                TAIL_CAP(8)
                TAIL_CAP(9)
                TAIL_CAP(10)
                TAIL_CAP(11)
                TAIL_CAP(12)
                TAIL_CAP(13)
                TAIL_CAP(14)
                TAIL_CAP(15)
                TAIL_CAP(16)
                TAIL_CAP(17)
                TAIL_CAP(18)
                TAIL_CAP(19)
                TAIL_CAP(20)
                TAIL_CAP(21)
                TAIL_CAP(22)
                TAIL_CAP(23)
                TAIL_CAP(24)
                TAIL_CAP(25)
                TAIL_CAP(26)
                TAIL_CAP(27)
                TAIL_CAP(28)
                TAIL_CAP(29)
                TAIL_CAP(30)
                TAIL_CAP(31)
                default: break;
            }
        }
        part4[wave][lane] = tri_f32x4{acc[0].x, acc[0].y, acc[1].x, acc[1].y};
        TAIL_STAMP(1);
        TAIL_JITTER(1);
        lds_barrier();                                                                     // A
        TAIL_STAMP(2);
        TAIL_JITTER(2);
        if (wave == 0) {
            // ---- the scalar part of the step: one wave, wave-level sums only
            tri_f32x4 x8[8];
            lds_read8_b128((unsigned)(uintptr_t)&part4[0][lane], x8);
            const tri_f32x4 s4 = ((x8[0] + x8[1]) + (x8[2] + x8[3])) + ((x8[4] + x8[5]) + (x8[6] + x8[7]));
            float pc[CPL], capc[CPL];
            const tri_f32x4 c4 = cap4[lane];
            pc[0] = tau * s4.x; pc[1] = tau * s4.y; pc[2] = tau * s4.z; pc[3] = tau * s4.w;
            capc[0] = c4.x; capc[1] = c4.y; capc[2] = c4.z; capc[3] = c4.w;
            TAIL_STAMP(3);
            TAIL_JITTER(3);
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k) dot = fmaf(pc[k], uc[k], dot);
            const float gamma = -0.5f * tau * wave_sum(dot);
            const float w0 = pick(pc, r0) + gamma;             // u[r0] = 1
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const bool on = cbase >= r0 - WAVES * k;         // column c_k >= r0
                wn[k] = on ? fmaf(gamma, uc[k], pc[k]) : 0.f;
                cn[k] = on ? capc[k] - fmaf(uc[k], w0, wn[k]) : 0.f;
            }
            TAIL_STAMP(4);
            TAIL_JITTER(4);
            next_reflector(jl + 1, jl + 1 >= m - 1);
            TAIL_STAMP(5);
            TAIL_JITTER(5);
        }
        lds_barrier();                                                                     // B
        TAIL_STAMP(6);
        TAIL_JITTER(6);
        if (wave == WAVES - 1) store_reflector(jl + 1);
    }
    TAIL_CLOCKS(1);
    lds_barrier();
    for (int c = tid; c < m; c += 64 * WAVES) {
        dz[j0 + c] = dloc[c];
        ez[j0 + c] = eloc[c];
        tz[j0 + c] = tloc[c];
    }
    if (rk.rank_out && z < rk.count && 2 * n <= WAVES * 256) {
        float* dl = (float*)&part4[0][0];
        float* el = dl + n;
        for (int c = tid; c < n; c += 64 * WAVES) {
            dl[c] = c < j0 ? dz[c] : dloc[c - j0];
            el[c] = c < j0 ? ez[c] : eloc[c - j0];
        }
        __syncthreads();
        mp_rank_block(dl, el, n, z, rk, nullptr);
    }
#ifdef BASD_TAIL_DBG
    if (blockIdx.x == 0 && threadIdx.x == 0) g_tail_dbg[(rk.rank_out ? 0 : 8 * 2 * 1024) + 8 * 2 * 1024 - 5] = wall_clock64();
#endif
}
#undef TAIL_PAIR
#undef TAIL_CAP
// ---------------------------------------------------------------------------
// tridiag_packed_kernel: the WHOLE factorisation of one matrix of order 256 < n <= 384 in the registers of ONE CU -- no
// shared stage, no members, nothing spins, nothing of the matrix is re-read from L2 -- by keeping only the UPPER
// TRIANGLE (295 KB of the CU's 512 KB of VGPRs; the full 384^2 matrix is 590 KB).
//
// Why: the two-stage layout above (steps [0, n - 256) shared by up to 16 workgroups through L2 granules, then
// tridiag_tail2_kernel) takes 0.97 ms alone at n = 384 but 1.3-1.4 ms inside a training step: its 16-96 spinning
// workgroups share their CUs' issue slots, LDS and the L2 with the Procrustes kernels of the caller's stream, and cost
// THOSE 0.4 ms as well (round-3 step clock).  A workgroup that owns its CU's register file shares nothing.
//
// Layout (8 waves x 64 lanes): wave w owns the rows r = w + 8 i (i < 48), lane l the columns c = l + 64 k (k < 6); an
// element a[r][c] with c >= r lives in (w, l).  Rows are kept in PAIRS P (rows i = 2P, 2P + 1: r_a = w + 16 P,
// r_b = r_a + 8) as packed registers (a[r_a][c], a[r_b][c]); row pair P needs the column chunks k >= P >> 2 (both rows
// of a pair start in the same chunk): 4 (6 + 5 + 4 + 3 + 2 + 1) = 84 packed registers = 168 VGPRs per lane.  In a
// pair's first ("boundary") chunk the lanes left of the diagonal are padding (kept 0) and the DIAGONAL element is
// stored HALVED: every stored element then contributes to both y_c += a u_r and y_r += a u_c, and the diagonal's two
// half contributions add up to a[r][r] u_r (halving and doubling are exact).
//
// A step (same algorithm and update order as tridiag_tail2_kernel, run on the upper triangle):
//   pass      per row pair and chunk: a -= m (v_r w_c + w_r v_c)  [pending update of the previous step; m = the
//             boundary mask 1 / 0.5 / 0 folded into the ROW operands], acc_c += a u_r (per-lane, per column: no
//             cross-lane sum), t_r += a u_c (per row: summed across the 64 lanes of the wave afterwards).  All four
//             products are v_pk_fma_f32 over the row pair; a wave-uniform operand pair (v_a, v_b) is used as is, a
//             per-lane column operand is broadcast from one half of a register pair with op_sel.
//   row sums  the 8 partials t of a batch of 4 row pairs are reduce-scattered across the wave: permlane32_swap +
//             packed add, permlane16_swap + packed add, then 4 DPP steps inside a row of 16 lanes -- 17 instructions per
//             8 matrix rows instead of 8 x 6.
//   capture   the wave that owns row r0 = j + 1 copies it out (column form): the next reflector is built from it.
//   A         barrier; wave 0: p = tau (sum over waves of acc + t), gamma, w, next pivot row, its norm, next reflector
//             (one wave's scalar chain, wave-level sums only), publishes the operands of the next pass;   B  barrier.
// The pass starts at the first batch of 4 row pairs that still has a live row (rows < r0 are finished: their u_r is
// 0, so whatever the pass does to them is never read).
// ---------------------------------------------------------------------------
constexpr int PK_WAVES = 8, PK_NMAX = 384, PK_RPW = PK_NMAX / PK_WAVES, PK_PAIRS = PK_RPW / 2, PK_CH = PK_NMAX / 64;
__host__ __device__ constexpr int pk_boff(int kb) { return 4 * (kb * PK_CH - kb * (kb - 1) / 2); }   // 0, 24, 44, 60, 72, 80
__host__ __device__ constexpr int pk_idx(int P, int k) { return pk_boff(P >> 2) + (P & 3) * (PK_CH - (P >> 2)) + (k - (P >> 2)); }
static_assert(pk_boff(6) == 84, "84 packed registers");

template <int H>
__device__ __forceinline__ void pk_fma_bc(tri_f2& acc, tri_f2 a, tri_f2 b) {      // acc += a * b[H] (b's half broadcast)
    if constexpr (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(b));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(a), "v"(b));
}
template <int H>
__device__ __forceinline__ void pk_fnma_bc(tri_f2& acc, tri_f2 a, tri_f2 b) {     // acc -= a * b[H] (neg modifiers: free)
    if constexpr (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "+v"(acc) : "v"(a), "v"(b));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pk_fma(tri_f2& acc, tri_f2 a, tri_f2 b) {          // acc += a * b
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ tri_f2 pk_mul(tri_f2 a, tri_f2 b) {
    tri_f2 r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ tri_f2 pk_add(tri_f2 a, tri_f2 b) {
    tri_f2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (x, y) partial sums of two values each: x's lower / y's lower halves... one level of a reduce-scatter across the wave.
// After the call lanes [0, 32) hold lo's sum over {l, l + 32}, lanes [32, 64) hi's -- for both components.
__device__ __forceinline__ tri_f2 pk_fold32(tri_f2 lo, tri_f2 hi) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo.x), __float_as_uint(hi.x), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo.y), __float_as_uint(hi.y), false, false);
    return pk_add(tri_f2{__uint_as_float(a[0]), __uint_as_float(b[0])}, tri_f2{__uint_as_float(a[1]), __uint_as_float(b[1])});
}
// the same one level down: even rows of 16 lanes end with lo's sum over {l, l + 16}, odd rows with hi's
__device__ __forceinline__ tri_f2 pk_fold16(tri_f2 lo, tri_f2 hi) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo.x), __float_as_uint(hi.x), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo.y), __float_as_uint(hi.y), false, false);
    return pk_add(tri_f2{__uint_as_float(a[0]), __uint_as_float(b[0])}, tri_f2{__uint_as_float(a[1]), __uint_as_float(b[1])});
}

#ifdef BASD_TAIL_DBG
#define PK_STAMP(slot) do { if (z == 0 && lane == 0 && jl < 384) g_tail_dbg[(jl * 8 + (slot)) * 8 + wave] = clock64(); } while (0)
#else
#define PK_STAMP(slot) do { } while (0)
#endif
// x[q] of six wave-uniform values by a wave-uniform q in 0..5: scalar selects (no branches, no VALU)
__device__ __forceinline__ float uniform_select6(float t0, float t1, float t2, float t3, float t4, float t5, int q) {
    int r;
    asm("s_cmp_eq_u32 %7, 1\n\t"
        "s_cselect_b32 %0, %2, %1\n\t"
        "s_cmp_eq_u32 %7, 2\n\t"
        "s_cselect_b32 %0, %3, %0\n\t"
        "s_cmp_eq_u32 %7, 3\n\t"
        "s_cselect_b32 %0, %4, %0\n\t"
        "s_cmp_eq_u32 %7, 4\n\t"
        "s_cselect_b32 %0, %5, %0\n\t"
        "s_cmp_eq_u32 %7, 5\n\t"
        "s_cselect_b32 %0, %6, %0"
        : "=&s"(r)
        : "s"(__float_as_int(t0)), "s"(__float_as_int(t1)), "s"(__float_as_int(t2)), "s"(__float_as_int(t3)),
          "s"(__float_as_int(t4)), "s"(__float_as_int(t5)), "s"(q)
        : "scc");
    return __int_as_float(r);
}

// Lane <-> column map inside a chunk of 64 columns: lane l holds column 64 k + pi(l), pi(l) = 8 (l & 7) + (l >> 3) (an
// involution).  As a ROW, that column belongs to wave l >> 3, row index i = 8 k + (l & 7): batch k, slot l & 7 -- so the
// eight lanes l & 7 = 0..7 of one 8-lane group hold, chunk by chunk, exactly the eight rows of one batch of one wave, and
// wave 0 publishes a step's row operands with 18 ds_write_b32 at immediate offsets from ONE per-lane base address.
__device__ __forceinline__ int pk_pi(int l) { return 8 * (l & 7) + (l >> 3); }

// one row pair of the pass (chunks K >= P >> 2); row operands of the pair: VAB = (v_a, v_b), WAB = (w_a, w_b), U2 = (u_a, u_b)
#define PK_CHUNK(P, K, VAB, WAB, U2)                                               \
    if constexpr ((K) >= ((P) >> 2) && (K) < PK_CH) {                              \
        tri_f2 a_ = A2[pk_idx(P, K)];                                              \
        pk_fnma_bc<(K) & 1>(a_, VAB, NW[(K) >> 1]);                                \
        pk_fnma_bc<(K) & 1>(a_, WAB, NV[(K) >> 1]);                                \
        A2[pk_idx(P, K)] = a_;                                                     \
        pk_fma(acc2[K], a_, U2);                                                   \
        pk_fma_bc<(K) & 1>(t2[(P) & 3], a_, UC[(K) >> 1]);                         \
    }
#define PK_PAIR(P, VAB, WAB, U2)                                                   \
    {                                                                              \
        const tri_f2 vm_ = pk_mul(VAB, MK2[(P) & 3]), wm_ = pk_mul(WAB, MK2[(P) & 3]); \
        constexpr int KB_ = (P) >> 2;                                              \
        PK_CHUNK(P, KB_, vm_, wm_, U2)                                             \
        PK_CHUNK(P, KB_ + 1, VAB, WAB, U2)                                         \
        PK_CHUNK(P, KB_ + 2, VAB, WAB, U2)                                         \
        PK_CHUNK(P, KB_ + 3, VAB, WAB, U2)                                         \
        PK_CHUNK(P, KB_ + 4, VAB, WAB, U2)                                         \
        PK_CHUNK(P, KB_ + 5, VAB, WAB, U2)                                         \
    }
// one batch of 4 row pairs (rows i in [8 KB, 8 KB + 8) of every wave) and the cross-lane sums of their 8 partials;
// the batch's row operands are 24 consecutive floats of the table: v x 8, w x 8, u x 8 (slot = i & 7)
#define PK_BATCH(KB)                                                               \
    case KB: {                                                                     \
        const tri_f32x4* ob_ = (const tri_f32x4*)(optab_w + 24 * (KB));            \
        tri_f2 t2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};           \
        {                                                                          \
            const tri_f32x4 v0_ = ob_[0], w0_ = ob_[2], u0_ = ob_[4];              \
            PK_PAIR(4 * (KB) + 0, (tri_f2{v0_.x, v0_.y}), (tri_f2{w0_.x, w0_.y}), (tri_f2{u0_.x, u0_.y})) \
            PK_PAIR(4 * (KB) + 1, (tri_f2{v0_.z, v0_.w}), (tri_f2{w0_.z, w0_.w}), (tri_f2{u0_.z, u0_.w})) \
        }                                                                          \
        {                                                                          \
            const tri_f32x4 v1_ = ob_[1], w1_ = ob_[3], u1_ = ob_[5];              \
            PK_PAIR(4 * (KB) + 2, (tri_f2{v1_.x, v1_.y}), (tri_f2{w1_.x, w1_.y}), (tri_f2{u1_.x, u1_.y})) \
            PK_PAIR(4 * (KB) + 3, (tri_f2{v1_.z, v1_.w}), (tri_f2{w1_.z, w1_.w}), (tri_f2{u1_.z, u1_.w})) \
        }                                                                          \
        tri_f2 s_ = pk_fold16(pk_fold32(t2[0], t2[1]), pk_fold32(t2[2], t2[3]));   \
        s_.x = row16_allsum(s_.x);                                                 \
        s_.y = row16_allsum(s_.y);                                                 \
        /* row of 16 lanes 0: pair 0, 1: pair 2, 2: pair 1, 3: pair 3 */          \
        if ((lane & 15) == 0) *(tri_f2*)(yrow_q + 8 * (KB)) = s_;                  \
    }                                                                              \
    [[fallthrough]];
#define PK_CAP(P)                                                                  \
    case P: {                                                                      \
        _Pragma("unroll") for (int k = (P) >> 2; k < PK_CH; ++k) {                 \
            const tri_f2 a_ = A2[pk_idx(P, k)];                                    \
            float x_ = half ? a_.y : a_.x;                                         \
            if (k == ((P) >> 2)) x_ *= capfix;                                     \
            cap[k * 64 + lane] = x_;                                               \
        }                                                                          \
    } break;

__global__ void __launch_bounds__(64 * PK_WAVES) tridiag_packed_kernel(float* __restrict__ A, long a_batch_stride, int n,
                                                                       float* __restrict__ d, float* __restrict__ e,
                                                                       float* __restrict__ tau_out,
                                                                       float* __restrict__ Vh, MpRankOut rk,
                                                                       int clocks, const unsigned* go_flag,
                                                                       unsigned go_value, int go_budget) {
    constexpr int WAVES = PK_WAVES, CH = PK_CH;
    __builtin_amdgcn_s_setprio(3);
    if (go_flag) {
        // EARLY LAUNCH (basd_tridiag_ranked_gated): this workgroup was queued before its input exists, so that it has
        // its CU when the input arrives -- a workgroup that needs every VGPR of a CU is only placed on a CU that has
        // drained, and once the step's throughput launches are refilling every free slot that takes 0.3 ms.  One lane
        // polls the word the producer's stream sets behind the input (agent-scope loads: served by L2 / fabric), asleep
        // in between; the other waves wait at a barrier (no issue slots).  BOUNDED: if the word does not arrive
        // (kernels serialised by a profiler: the producer cannot run while this kernel does), the workgroup gives up
        // with status word 0 = 2 and the host queues the plain launch.
        __shared__ int go_ok;
        if (threadIdx.x == 0) {
            int ok = 0;
            for (int i = 0; i < go_budget; ++i) {
                if (__hip_atomic_load(go_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == go_value) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(64);
            }
            go_ok = ok;
        }
        __syncthreads();
        if (!go_ok) {
            if (threadIdx.x == 0 && blockIdx.x == 0) {
                if (rk.status) const_cast<int*>(rk.status)[0] = 2;
                if (rk.host_mirror) rk.host_mirror[rk.count] = 2;
            }
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the input was written after this kernel started
    }
    // diagnostics (BASD_TRIDIAG_CLOCKS=1): how long this workgroup ran, on the constant 100 MHz clock and on the shader
    // clock -- status words 6 and 7, which travel to the host with the ranks, tell a run that WAITED for a free CU from
    // one that was slow
    const long long wall0 = wall_clock64(), clk0 = clock64();
    // row operands of the pass: [wave][batch kb][v x 8 | w x 8 | u x 8], slot = i & 7 of row w + 8 i, i = 8 kb + slot
    __shared__ __attribute__((aligned(16))) float optab[WAVES][CH][24];
    // column operands per lane (c = 64 k + pi(l)): u[6], v[6], w[6], 2 pad: five b128
    __shared__ __attribute__((aligned(16))) float colf[64][20];
    __shared__ __attribute__((aligned(16))) float part[WAVES][64][8];        // per-wave column partials (6 of 8 used)
    __shared__ __attribute__((aligned(16))) float yrows[WAVES][PK_RPW];      // cross-lane row sums of row w + 8 i
    __shared__ float cap[PK_NMAX];                    // the captured pivot row, by (chunk, lane)
    __shared__ float dloc[PK_NMAX], eloc[PK_NMAX], tloc[PK_NMAX];
    const int z = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = pk_pi(lane);                       // my column inside every chunk
    float* Az = A + (long)z * a_batch_stride;
    float* Vz = Vh + (long)z * n * n;
    const float* const optab_w = &optab[wave][0][0];
    // where the lane that ends a batch's row reduction in row-of-16 rho writes its pair q = [0, 2, 1, 3][rho]
    float* const yrow_q = &yrows[wave][0] + 2 * (((lane >> 5) & 1) | ((lane >> 3) & 2));
    // where wave 0 publishes (u, v, w) of my columns as row operands: wave l >> 3, slot l & 7; per chunk + 24 floats
    float* const pub = &optab[lane >> 3][0][0] + (lane & 7);
    // where the last wave gathers u in natural column order c = lane + 64 k for the reflector row: row c is wave
    // c & 7 = lane & 7, batch k, slot (c >> 3) & 7 = lane >> 3
    const float* const ugather = &optab[lane & 7][0][0] + 16 + (lane >> 3);

    // ---- the upper triangle, zero padded to 384, diagonal halved
    tri_f2 A2[84];
#pragma unroll
    for (int P = 0; P < PK_PAIRS; ++P) {
        const int ra = wave + 16 * P, rb = ra + 8;
#pragma unroll
        for (int k = P >> 2; k < CH; ++k) {
            const int c = pl + 64 * k;
            float xa = 0.f, xb = 0.f;
            if (c < n) {
                if (ra < n && c >= ra) xa = Az[(long)ra * n + c] * (c == ra ? 0.5f : 1.f);
                if (rb < n && c >= rb) xb = Az[(long)rb * n + c] * (c == rb ? 0.5f : 1.f);
            }
            A2[pk_idx(P, k)] = tri_f2{xa, xb};
        }
    }
    // boundary masks of the pairs P & 3 = q: rows i & 7 = 2 q, 2 q + 1 sit at column wave + 8 (i & 7) of their chunk
    tri_f2 MK2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ta = wave + 16 * q, tb = ta + 8;
        MK2[q] = tri_f2{pl > ta ? 1.f : (pl == ta ? 0.5f : 0.f), pl > tb ? 1.f : (pl == tb ? 0.5f : 0.f)};
    }

    float tau = 0.f;              // wave 0: tau of the current step
    // column-form entry of column c (wave-uniform), broadcast: lane pi(c & 63), chunk c >> 6
    auto pick = [&](const float (&x)[CH], int c) {
        const int l = pk_pi(c & 63);
        return uniform_select6(lane_bcast(x[0], l), lane_bcast(x[1], l), lane_bcast(x[2], l), lane_bcast(x[3], l),
                               lane_bcast(x[4], l), lane_bcast(x[5], l), c >> 6);
    };
    // wave 0: reflector of step jl from cn[] (row jl of the current matrix, column form; dnew = its diagonal entry),
    // then the operands of the next pass: u = the new reflector, v = uc (the old one), w = wn
    auto next_reflector = [&](int jl, bool last, const float (&cn)[CH], const float (&uc)[CH], const float (&wn)[CH],
                              float dnew) {
        const int r0 = jl + 1;
        float part2 = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) part2 = (pl + 64 * k > r0) ? fmaf(cn[k], cn[k], part2) : part2;
        const float alpha = pick(cn, r0 < PK_NMAX ? r0 : PK_NMAX - 1);
        const float xn2 = wave_sum(part2);
        const bool live = !last && xn2 > 0.f;
        const float beta = live ? -copysignf(__builtin_amdgcn_sqrtf(fmaf(alpha, alpha, xn2)), alpha) : (last ? 0.f : alpha);
        tau = live ? (beta - alpha) * __builtin_amdgcn_rcpf(beta) : 0.f;
        const float scal = live ? __builtin_amdgcn_rcpf(alpha - beta) : 0.f;
        float un[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int c = pl + 64 * k;
            un[k] = last ? 0.f : (c < r0 ? 0.f : (c == r0 ? 1.f : cn[k] * scal));
            pub[24 * k] = uc[k];
            pub[24 * k + 8] = wn[k];
            pub[24 * k + 16] = un[k];
        }
        tri_f32x4* cf = (tri_f32x4*)&colf[lane][0];
        cf[0] = tri_f32x4{un[0], un[1], un[2], un[3]};
        cf[1] = tri_f32x4{un[4], un[5], uc[0], uc[1]};
        cf[2] = tri_f32x4{uc[2], uc[3], uc[4], uc[5]};
        cf[3] = tri_f32x4{wn[0], wn[1], wn[2], wn[3]};
        cf[4] = tri_f32x4{wn[4], wn[5], 0.f, 0.f};
        if (lane == 0) {
            dloc[jl] = dnew;
            eloc[jl] = beta;
            tloc[jl] = tau;
        }
    };
    // waves 1..6, behind barrier B: reflector row of step jl from the published u, natural column order, one chunk of
    // 64 columns per wave (one LDS read + one store each: nobody starts the next pass late)
    auto store_reflector = [&](int jl) {
        const int k = wave - 1;
        if (k >= 0 && k < CH && lane + 64 * k < n)
            Vz[(long)jl * n + lane + 64 * k] = ugather[24 * k];
    };
    if (wave == 0) {
        float cn[CH], zero[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            cn[k] = A2[pk_idx(0, k)].x * ((k == 0 && pl == 0) ? 2.f : 1.f);
            zero[k] = 0.f;
        }
        next_reflector(0, n <= 1, cn, zero, zero, pick(cn, 0));
    }
    lds_barrier();
    store_reflector(0);

    for (int jl = 0; jl < n - 1; ++jl) {
        const int r0 = jl + 1;
        // ---- the pass
        PK_STAMP(0);
        tri_f2 NV[3], NW[3], UC[3], acc2[CH];
        {
            const tri_f32x4* cf = (const tri_f32x4*)&colf[lane][0];
            const tri_f32x4 c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3], c4 = cf[4];   // u0-3 | u4,u5,v0,v1 | v2-5 | w0-3 | w4,w5
            UC[0] = tri_f2{c0.x, c0.y}; UC[1] = tri_f2{c0.z, c0.w}; UC[2] = tri_f2{c1.x, c1.y};
            NV[0] = tri_f2{c1.z, c1.w}; NV[1] = tri_f2{c2.x, c2.y}; NV[2] = tri_f2{c2.z, c2.w};
            NW[0] = tri_f2{c3.x, c3.y}; NW[1] = tri_f2{c3.z, c3.w}; NW[2] = tri_f2{c4.x, c4.y};
#pragma unroll
            for (int k = 0; k < CH; ++k) acc2[k] = tri_f2{0.f, 0.f};
        }
        {
            // first row index of this wave that is still live (r = wave + 8 i >= r0), and its batch of 8 row indices
            const int i0 = r0 > wave ? (r0 - wave + 7) >> 3 : 0;
            switch (i0 >> 3) {
                PK_BATCH(0)
                PK_BATCH(1)
                PK_BATCH(2)
                PK_BATCH(3)
                PK_BATCH(4)
                PK_BATCH(5)
                default: break;
            }
        }
        if (wave == (r0 & (WAVES - 1))) {
            const int i0r = r0 >> 3, half = i0r & 1;
            const float capfix = pl == (r0 & 63) ? 2.f : 1.f;     // the halved diagonal entry, whole again
            switch (i0r >> 1) {
                PK_CAP(0) PK_CAP(1) PK_CAP(2) PK_CAP(3) PK_CAP(4) PK_CAP(5) PK_CAP(6) PK_CAP(7)
                PK_CAP(8) PK_CAP(9) PK_CAP(10) PK_CAP(11) PK_CAP(12) PK_CAP(13) PK_CAP(14) PK_CAP(15)
                PK_CAP(16) PK_CAP(17) PK_CAP(18) PK_CAP(19) PK_CAP(20) PK_CAP(21) PK_CAP(22) PK_CAP(23)
                default: break;
            }
        }
        {
            tri_f32x4* pp = (tri_f32x4*)&part[wave][lane][0];
            pp[0] = tri_f32x4{acc2[0].x + acc2[0].y, acc2[1].x + acc2[1].y, acc2[2].x + acc2[2].y, acc2[3].x + acc2[3].y};
            pp[1] = tri_f32x4{acc2[4].x + acc2[4].y, acc2[5].x + acc2[5].y, 0.f, 0.f};
        }
        PK_STAMP(1);
        lds_barrier();                                                                     // A
        PK_STAMP(2);
        if (wave == 0) {
            // ---- the scalar part of the step: one wave, wave-level sums only; column pairs (k, k + 1) as packed math
            float uc[CH];
            uc[0] = UC[0].x; uc[1] = UC[0].y; uc[2] = UC[1].x; uc[3] = UC[1].y; uc[4] = UC[2].x; uc[5] = UC[2].y;
            // the diagonal entry of the captured row (column r0: chunk r0 >> 6, lane pi(r0 & 63)): a uniform read
            const float cap_r0 = cap[(r0 >> 6) * 64 + pk_pi(r0 & 63)];
            tri_f2 sc2[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const tri_f32x4* pp = (const tri_f32x4*)&part[w][lane][0];
                const tri_f32x4 p0 = pp[0];
                const tri_f2 p1 = *(const tri_f2*)&pp[1];
                sc2[0] = pk_add(sc2[0], tri_f2{p0.x, p0.y});
                sc2[1] = pk_add(sc2[1], tri_f2{p0.z, p0.w});
                sc2[2] = pk_add(sc2[2], p1);
            }
            tri_f2 pc2[3], capc2[3];
            const tri_f2 tau2 = {tau, tau};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                // row sums of row c = 64 k + pi(lane): wave c & 7 = lane >> 3, index c >> 3 = 8 k + (lane & 7)
                const tri_f2 y2 = {yrows[lane >> 3][16 * q + (lane & 7)], yrows[lane >> 3][16 * q + 8 + (lane & 7)]};
                pc2[q] = pk_mul(tau2, pk_add(sc2[q], y2));
                capc2[q] = tri_f2{cap[128 * q + lane], cap[128 * q + 64 + lane]};
            }
            PK_STAMP(3);
            tri_f2 dot2 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 3; ++q) pk_fma(dot2, pc2[q], UC[q]);
            float pc[CH];
            pc[0] = pc2[0].x; pc[1] = pc2[0].y; pc[2] = pc2[1].x; pc[3] = pc2[1].y; pc[4] = pc2[2].x; pc[5] = pc2[2].y;
            const float p_r0 = pick(pc, r0);
            const float gamma = -0.5f * tau * wave_sum(dot2.x + dot2.y);
            const float w0 = p_r0 + gamma;                     // u[r0] = 1
            // No masks for the columns c < r0 (tridiag_tail2_kernel needs them: it stores full rows): with the upper
            // triangle no live row holds such a column -- as COLUMN operands (u, v, w)_c only ever meet padding lanes
            // (masked), as ROW operands they belong to finished rows, whose u is 0 (next_reflector) and whose registers
            // nothing reads any more; everything stays finite (sums of finite products).
            float wn[CH], cn[CH];
            {
                const tri_f2 g2 = {gamma, gamma}, w02 = {w0, w0};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    tri_f2 wn2 = pc2[q];
                    pk_fma(wn2, g2, UC[q]);                    // w = p + gamma u
                    tri_f2 t2_ = wn2;
                    pk_fma(t2_, UC[q], w02);                   // u w0 + w
                    const tri_f2 cn2 = pk_add(capc2[q], tri_f2{-t2_.x, -t2_.y});
                    wn[2 * q] = wn2.x; wn[2 * q + 1] = wn2.y;
                    cn[2 * q] = cn2.x; cn[2 * q + 1] = cn2.y;
                }
            }
            PK_STAMP(4);
            // d of the next step = entry r0 of that row = cap - (1 * w0 + w0): the same arithmetic as cn[] at column r0
            next_reflector(jl + 1, jl + 1 >= n - 1, cn, uc, wn, cap_r0 - fmaf(1.f, w0, w0));
            PK_STAMP(5);
        }
        lds_barrier();                                                                     // B
        PK_STAMP(6);
        store_reflector(jl + 1);
    }
    lds_barrier();
    {
        float* dz = d + (long)z * n;
        float* ez = e + (long)z * n;
        float* tz = tau_out + (long)z * n;
        for (int c = tid; c < n; c += 64 * WAVES) {
            dz[c] = dloc[c];
            ez[c] = eloc[c];
            tz[c] = tloc[c];
        }
    }
    if (clocks && z == 0 && tid == 0 && rk.status) {
        int* st = const_cast<int*>(rk.status);
        st[6] = (int)(wall_clock64() - wall0);          // x 10 ns
        st[7] = (int)((clock64() - clk0) >> 4);         // shader cycles / 16
    }
    if (rk.rank_out && z < rk.count) {
        __syncthreads();
        mp_rank_block(dloc, eloc, n, z, rk, nullptr);
    }
}
#undef PK_CHUNK
#undef PK_PAIR
#undef PK_BATCH
#undef PK_CAP

// ---------------------------------------------------------------------------
// All eigenvalues of the symmetric tridiagonal (d, e) from Sturm counts (LAPACK sstebz), descending.
// A Sturm count is a chain of n dependent steps, so plain bisection costs ~45 chains per eigenvalue; here
// every eigenvalue is owned by one DPP row of 16 lanes that evaluates 16 interior points of its bracket per
// pass (17-section): ~8 passes instead of 45.  grid = (ceil(n/64), batch), block = 1024 = 64 eigenvalues x 16.
// ---------------------------------------------------------------------------
// Sturm count #(eigenvalues < x) of the tridiagonal (dz, ez); x may differ per lane, (dz, ez) is per wave.
// The recurrence walks the diagonal in order and every lane reads the same element: dz / ez are wave-uniform
// GLOBAL addresses, so the compiler fetches them with scalar loads (s_load_dwordx8 after unrolling) through the
// constant cache -- no LDS traffic.  That matters because these kernels share their CUs with the LDS-bound
// Procrustes Jacobi of the caller's stream: with the diagonal in LDS the rank kernel took 360 us inside a
// training step (50 alone), with scalar loads 195.  (Tried and dropped: 64 entries per VGPR broadcast with
// v_readlane -- the SGPR hazards make every step longer; 0.32 ms for the full bisection against 0.20.)
__device__ __forceinline__ int sturm_count(const float* __restrict__ dz, const float* __restrict__ ez, int n, float x,
                                           float pivmin) {
    int cnt = 0;
    float q = dz[0] - x;
    if (fabsf(q) < pivmin) q = -pivmin;
    cnt += q < 0.f;
#pragma unroll 8
    for (int r = 1; r < n; ++r) {
        // 1-ulp hardware reciprocal: the count is only ambiguous where q is round-off anyway
        const float er = ez[r - 1];
        q = (dz[r] - x) - (er * er) * __builtin_amdgcn_rcpf(q);
        if (fabsf(q) < pivmin) q = -pivmin;
        cnt += q < 0.f;
    }
    return cnt;
}

__device__ __forceinline__ float row16_allmax(float x) {
    x = fmaxf(x, dpp_get<0xB1>(x));
    x = fmaxf(x, dpp_get<0x4E>(x));
    x = fmaxf(x, dpp_get<0x141>(x));
    x = fmaxf(x, dpp_get<0x140>(x));
    return x;
}

__global__ void __launch_bounds__(1024) sturm_bisect_kernel(const float* __restrict__ d, const float* __restrict__ e,
                                                            int n, float* __restrict__ vals_desc) {
    __shared__ float red3[3][16];
    const int z = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x, nwv = nthr >> 6;
    const float* dz = d + (long)z * n;
    const float* ez = e + (long)z * n;
    float lo = 3.4e38f, hi = -3.4e38f, emax = 0.f;
    for (int i = tid; i < n; i += nthr) {
        const float di = dz[i];
        const float el = i > 0 ? fabsf(ez[i - 1]) : 0.f, er = i < n - 1 ? fabsf(ez[i]) : 0.f;
        lo = fminf(lo, di - el - er);
        hi = fmaxf(hi, di + el + er);
        emax = fmaxf(emax, er * er);
    }
    lo = -wave_max(-lo);
    hi = wave_max(hi);
    emax = wave_max(emax);
    __syncthreads();
    if ((tid & 63) == 0) { red3[0][tid >> 6] = lo; red3[1][tid >> 6] = hi; red3[2][tid >> 6] = emax; }
    __syncthreads();
    for (int i = 0; i < nwv; ++i) {
        lo = fminf(lo, red3[0][i]);
        hi = fmaxf(hi, red3[1][i]);
        emax = fmaxf(emax, red3[2][i]);
    }
    const float tnorm = fmaxf(fabsf(lo), fabsf(hi));
    const float eps = 1.1920929e-7f;
    lo -= 2.f * tnorm * eps * n + 1e-37f;
    hi += 2.f * tnorm * eps * n + 1e-37f;
    const float pivmin = fmaxf(1.1754944e-38f * fmaxf(emax, 1.f), 1e-37f);

    const int row = tid >> 4, sub = tid & 15;             // 64 eigenvalues per workgroup, 16 lanes each
    const int i = blockIdx.x * 64 + row;                  // descending index of this row's eigenvalue
    const bool live = i < n;
    const int k_asc = n - 1 - (live ? i : n - 1);         // 0-based ascending index
    float a = lo, b = hi;
    bool done = false;                                    // uniform within a row of 16 lanes
    for (int it = 0; it < 40; ++it) {
        // 16 interior points of [a, b]; every lane of the row runs one Sturm count.  The whole wave stays in the
        // loop until its four rows are done (sturm_count broadcasts across all 64 lanes); a finished row idles.
        const float x = a + (b - a) * ((float)(sub + 1) * (1.f / 17.f));
        const int cnt = sturm_count(dz, ez, n, x, pivmin);
        // new bracket: the largest point with count <= k and the smallest with count > k
        const bool below = cnt <= k_asc;
        const float na = row16_allmax(below && x > a ? x : a);
        const float nb = -row16_allmax(-((!below && x < b) ? x : b));
        if (!done) {
            const bool stalled = !(na > a) && !(nb < b);  // points collapsed onto the ends: resolution reached
            a = na;
            b = nb;
            done = stalled || b - a <= 2.f * eps * fmaxf(fabsf(a), fabsf(b)) + pivmin;
        }
        if (__all(done)) break;
    }
    if (live && sub == 0) vals_desc[(long)z * n + i] = 0.5f * (a + b);
}

// ---------------------------------------------------------------------------
// Top-k eigenvectors of the tridiagonal by inverse iteration (LAPACK sstein / slagtf / slagts), followed by
// ordered re-orthogonalisation inside clusters of close eigenvalues.  Z out: (k x n) row-major, unit rows.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float hash_unit(unsigned a, unsigned b) {
    unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return (float)(h & 0xFFFFFF) * (2.f / 16777216.f) - 1.f;     // (-1, 1)
}

// shifts actually used: the computed eigenvalues, with (numerically) coincident ones pushed apart
__device__ __forceinline__ void invit_shifts(const float* __restrict__ vz, int k, float tnorm, float* __restrict__ lam) {
    const float sep = 10.f * 1.1920929e-7f * tnorm;
    float prev = 0.f;
    for (int t = 0; t < k; ++t) {
        float l = vz[t];
        if (t > 0 && prev - l < sep) l = prev - sep;
        lam[t] = l;
        prev = l;
    }
}

__device__ __forceinline__ float tridiag_norm(const float* __restrict__ dz, const float* __restrict__ ez, int n,
                                              int tid, int nthr, float* red) {
    float tn = 0.f;
    for (int i = tid; i < n; i += nthr)
        tn = fmaxf(tn, fabsf(dz[i]) + (i > 0 ? fabsf(ez[i - 1]) : 0.f) + (i < n - 1 ? fabsf(ez[i]) : 0.f));
    tn = wave_max(tn);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = tn;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < (nthr + 63) / 64; ++i) r = fmaxf(r, red[i]);
    return r;
}

// Phase A.  grid = (ceil(k / vpw), batch), block = 64: thread t of a workgroup owns vector blockIdx.x*vpw + t.
// The LU factors (a, b, c, d2, pivot flag) and the iterate x live in LDS as [array][element][vector]: the
// recurrences are sequential in the element index, so what matters is that the operands of step i+1 do not
// depend on step i's result (they are fetched ahead), leaving a short arithmetic chain per step.
__global__ void __launch_bounds__(64) tridiag_invit_kernel(const float* __restrict__ d, const float* __restrict__ e,
                                                           const float* __restrict__ vals_desc, int n, int k, int vpw,
                                                           float* __restrict__ Z) {
    extern __shared__ float sm[];
    __shared__ float red[32];
    __shared__ float lam_s[1024];
    const int z = blockIdx.y, tid = threadIdx.x;
    const float* dz = d + (long)z * n;
    const float* ez = e + (long)z * n;
    const float tnorm = tridiag_norm(dz, ez, n, tid, 64, red);
    if (tid == 0) invit_shifts(vals_desc + (long)z * n, k, tnorm, lam_s);
    __syncthreads();
    const int t = blockIdx.x * vpw + tid;
    if (tid >= vpw || t >= k) return;
    const float eps = 1.1920929e-7f;
    const float tol = eps * tnorm + 1e-37f;
    const long S = (long)n * vpw;                  // array stride
    float* a = sm + 0 * S + tid;                   // element i at a[i * vpw]
    float* b = sm + 1 * S + tid;
    float* c = sm + 2 * S + tid;
    float* d2 = sm + 3 * S + tid;
    float* in = sm + 4 * S + tid;
    float* x = sm + 5 * S + tid;
    const float l = lam_s[t];
    for (int i = 0; i < n; ++i) {
        a[i * vpw] = dz[i] - l;
        const float ei = i < n - 1 ? ez[i] : 0.f;
        b[i * vpw] = ei;
        c[i * vpw] = ei;
        d2[i * vpw] = 0.f;
        in[i * vpw] = 0.f;
        x[i * vpw] = hash_unit((unsigned)i, (unsigned)t);
    }
    // slagtf: LU with partial pivoting of the shifted tridiagonal
    {
        float ai = a[0];
        float scale1 = fabsf(ai) + (n > 1 ? fabsf(b[0]) : 0.f);
        for (int i = 0; i < n - 1; ++i) {
            const float ci = c[i * vpw], bi = b[i * vpw];
            const float a1 = a[(i + 1) * vpw];
            const float b1 = i < n - 2 ? b[(i + 1) * vpw] : 0.f;
            const float scale2 = fabsf(ci) + fabsf(a1) + fabsf(b1);
            const float piv1 = ai == 0.f ? 0.f : fabsf(ai) / scale1;
            const float piv2 = ci == 0.f ? 0.f : fabsf(ci) / scale2;
            float a_next;
            if (ci == 0.f || piv2 <= piv1) {
                const float mult = ci == 0.f ? 0.f : ci / ai;
                c[i * vpw] = mult;
                a_next = a1 - mult * bi;
            } else {
                in[i * vpw] = 1.f;
                const float mult = ai / ci;
                a[i * vpw] = ci;
                a_next = bi - mult * a1;
                if (i < n - 2) {
                    d2[i * vpw] = b1;
                    b[(i + 1) * vpw] = -mult * b1;
                }
                b[i * vpw] = a1;
                c[i * vpw] = mult;
            }
            a[(i + 1) * vpw] = a_next;
            ai = a_next;
            scale1 = scale2;
        }
    }
    // inverse iteration: two solves with slagts-style pivot perturbation.  The shift is an eigenvalue to a few
    // eps ||T||, so the first solve already amplifies the wanted direction by ~1/eps; the iterate enters each
    // solve with unit 2-norm times n ||T|| eps (sstein's scaling), folded into the forward sweep.
    const float s0 = (float)n * tnorm * eps;
    {
        float ss = 0.f;
        for (int i = 0; i < n; ++i) ss = fmaf(x[i * vpw], x[i * vpw], ss);
        const float sc = 1.f / sqrtf(fmaxf(ss, 1e-37f));
        for (int i = 0; i < n; ++i) x[i * vpw] *= sc;
    }
    for (int it = 0; it < 2; ++it) {
        // forward substitution (P L)
        float prev = x[0] * s0;
        for (int i = 1; i < n; ++i) {
            const float ci = c[(i - 1) * vpw], flag = in[(i - 1) * vpw];
            const float xi = x[i * vpw] * s0;
            float keep, next;
            if (flag == 0.f) { keep = prev; next = xi - ci * prev; }
            else { keep = xi; next = prev - ci * xi; }
            x[(i - 1) * vpw] = keep;
            prev = next;
        }
        x[(n - 1) * vpw] = prev;
        // back substitution (U); |x| <= ~n here, so the squares cannot overflow
        float x1 = 0.f, x2 = 0.f, ss = 0.f;
        for (int i = n - 1; i >= 0; --i) {
            float tmp = x[i * vpw] - b[i * vpw] * x1 - d2[i * vpw] * x2;    // b / d2 are 0 past the end
            float ak = a[i * vpw];
            if (fabsf(ak) < tol) ak = copysignf(tol, ak == 0.f ? 1.f : ak);
            tmp *= __builtin_amdgcn_rcpf(ak);
            x[i * vpw] = tmp;
            ss = fmaf(tmp, tmp, ss);
            x2 = x1;
            x1 = tmp;
        }
        const float sc2 = 1.f / sqrtf(fmaxf(ss, 1e-37f));
        for (int i = 0; i < n; ++i) x[i * vpw] *= sc2;
    }
    float* out = Z + ((long)z * k + t) * n;
    for (int i = 0; i < n; ++i) out[i] = x[i * vpw];
}

// Phase B: ordered re-orthogonalisation (two classical Gram-Schmidt passes) inside clusters of eigenvalues
// closer than 1e-3 ||T|| (LAPACK sstein's criterion).  grid = batch, block = 256.
__global__ void __launch_bounds__(1024) cluster_orth_kernel(const float* __restrict__ d, const float* __restrict__ e,
                                                            const float* __restrict__ vals_desc, int n, int k,
                                                            float* __restrict__ Z) {
    // Panels of 16 vectors, one wave per vector, the vector in registers (n <= 1024).  (A) every wave sweeps its vector
    // over the finished vectors of the cluster (modified Gram-Schmidt, twice; no workgroup barrier: the waves are
    // independent); (B) inside the panel the vectors are finished in order -- second sweep over the panel's finished
    // ones, normalise, publish in LDS, one barrier -- and the later waves take the new vector out of theirs.
    // A cluster of c vectors costs ~c / 16 * (c / 2 + 16 * 8) dot-and-update steps of ONE wave instead of c^2 / 2 steps
    // of the whole workgroup with two barriers each: features of a real network put most of the leading subspace
    // into one cluster (gaps are measured against ||T||, which the top eigenvalue dominates) -- 158 vectors took 14 ms.
    constexpr int PW = 16, EPL = 16;
    extern __shared__ float sm[];
    float* lam = sm;                    // k
    float* panel = sm + ((k + 3) & ~3); // PW x n: the finished vectors of the current panel
    __shared__ float red[32];
    const int z = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float tnorm = tridiag_norm(d + (long)z * n, e + (long)z * n, n, tid, 1024, red);
    if (tid == 0) invit_shifts(vals_desc + (long)z * n, k, tnorm, lam);
    __syncthreads();
    float* Zz = Z + (long)z * k * n;
    auto sweep = [&](float (&x)[EPL], const float* y) {            // x -= (x . y) y
        float yv[EPL], sdot = 0.f;
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int r = lane + 64 * i;
            yv[i] = r < n ? y[r] : 0.f;
            sdot = fmaf(x[i], yv[i], sdot);
        }
        sdot = wave_sum(sdot);
#pragma unroll
        for (int i = 0; i < EPL; ++i) x[i] = fmaf(-sdot, yv[i], x[i]);
    };
    int s0 = 0;
    while (s0 < k) {
        int s1 = s0 + 1;                                            // cluster [s0, s1): consecutive gaps below 1e-3 ||T||
        while (s1 < k && lam[s1 - 1] - lam[s1] < 1e-3f * tnorm) ++s1;
        if (s1 - s0 > 1) {
            for (int p0 = s0; p0 < s1; p0 += PW) {
                const int np = min(PW, s1 - p0);
                float x[EPL];
                if (wave < np) {
                    const float* xv = Zz + (long)(p0 + wave) * n;
#pragma unroll
                    for (int i = 0; i < EPL; ++i) x[i] = lane + 64 * i < n ? xv[lane + 64 * i] : 0.f;
                    for (int pass = 0; pass < 2; ++pass)
                        for (int j = s0; j < p0; ++j) sweep(x, Zz + (long)j * n);
                }
                for (int q = 0; q < np; ++q) {
                    if (wave == q) {
                        for (int j = 0; j < q; ++j) sweep(x, panel + j * n);       // second sweep over the panel
                        float ss = 0.f;
#pragma unroll
                        for (int i = 0; i < EPL; ++i) ss = fmaf(x[i], x[i], ss);
                        ss = wave_sum(ss);
                        const float inv = 1.f / sqrtf(fmaxf(ss, 1e-37f));
                        float* out = Zz + (long)(p0 + q) * n;
#pragma unroll
                        for (int i = 0; i < EPL; ++i) {
                            const int r = lane + 64 * i;
                            x[i] *= inv;
                            if (r < n) {
                                panel[q * n + r] = x[i];
                                out[r] = x[i];
                            }
                        }
                    }
                    __syncthreads();
                    if (wave > q && wave < np) sweep(x, panel + q * n);
                }
                __syncthreads();              // global stores of this panel are read by the next panel's sweeps
            }
        }
        s0 = s1;
    }
}

// ---------------------------------------------------------------------------
// Eigenvectors of A from those of T:  x = H_0 H_1 ... H_{n-2} z.   One wave per vector, the vector in
// registers (n <= 64 * EPL).  grid = (ceil(k/4), batch), block = 256.  out: (k x n) rows.
// ---------------------------------------------------------------------------
// FWD = false: out = Q z = H_0 H_1 ... H_{n-2} z (eigenvectors of T -> eigenvectors of A);
// FWD = true : out = Q^T z = H_{n-2} ... H_0 z (a vector of the original space -> the tridiagonal's basis).
template <int EPL, bool FWD = false>
__global__ void __launch_bounds__(256) backtransform_kernel(const float* __restrict__ Vh,
                                                            const float* __restrict__ tau, int n, int k,
                                                            const float* __restrict__ Z, float* __restrict__ out,
                                                            int out_stride_k) {
    const int z = blockIdx.y, t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= k) return;
    const float* Vz = Vh + (long)z * n * n;
    const float* tz = tau + (long)z * n;
    const float* zz = Z + ((long)z * k + t) * n;
    float x[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int r = lane + 64 * i;
        x[i] = r < n ? zz[r] : 0.f;
    }
    // A chain of n - 1 dependent reflector applications, each needing one row of Vh from L2 (~1 us away): the rows are
    // fetched PF steps ahead into a ring of registers (their addresses do not depend on the chain), so a step costs its
    // dot product and wave sum, not a memory round trip (0.28 ms -> the arithmetic at n = 384).
    constexpr int PF = EPL <= 6 ? 8 : 4;
    float ring[PF][EPL];
    float tring[PF];
    auto fetch = [&](int jj, float (&dst)[EPL], float& tdst) {
        const int j = FWD ? jj : n - 2 - jj;
        const bool ok = jj <= n - 2;
        const float* vj = Vz + (long)(ok ? j : 0) * n;
        tdst = ok ? tz[j] : 0.f;
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int r = lane + 64 * i;
            dst[i] = (ok && r < n) ? vj[r] : 0.f;
        }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) fetch(p, ring[p], tring[p]);
    for (int j0 = 0; j0 <= n - 2; j0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            float vr[EPL];
            const float tj = tring[p];
#pragma unroll
            for (int i = 0; i < EPL; ++i) vr[i] = ring[p][i];
            fetch(j0 + p + PF, ring[p], tring[p]);          // the slot's next tenant is requested before this one is used
            if (tj != 0.f) {                                 // (wave-uniform; rows past the end come back with tau = 0)
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < EPL; ++i) s = fmaf(vr[i], x[i], s);
                s = wave_sum(s) * tj;
#pragma unroll
                for (int i = 0; i < EPL; ++i) x[i] = fmaf(-s, vr[i], x[i]);
            }
        }
    }
    float* o = out + ((long)z * out_stride_k + t) * n;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int r = lane + 64 * i;
        if (r < n) o[r] = x[i];
    }
}


// ---------------------------------------------------------------------------
// Marchenko-Pastur rank straight from the tridiagonal (reference layer_selector.py:16-19): the rank needs the
// lower median eigenvalue and the number of eigenvalues above median * factor -- two order statistics, not the
// spectrum.  One workgroup of 1024 threads per matrix: 1025-section for the median (every thread runs one Sturm
// count per pass: three or four passes instead of the full bisection of all n eigenvalues), then ONE Sturm count
// at the threshold.  This is what the host waits for each step; the full spectrum is only needed later, for
// the eigenvectors.  Same arithmetic as sturm_bisect_kernel / mp_rank_kernel (pivmin rule, convergence rule,
// threshold rounded to fp32); "eigenvalue > threshold" is counted as n - #(eigenvalues < threshold).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void mp_rank_block(const float* dz, const float* ez, int n, int z, const MpRankOut& o,
                                              float* thr_out) {
    __shared__ float red3[3][16];
    const int tid = threadIdx.x, nthr = blockDim.x, nwv = nthr >> 6;
    float lo = 3.4e38f, hi = -3.4e38f, emax = 0.f;
    for (int i = tid; i < n; i += nthr) {
        const float di = dz[i];
        const float el = i > 0 ? fabsf(ez[i - 1]) : 0.f, er = i < n - 1 ? fabsf(ez[i]) : 0.f;
        lo = fminf(lo, di - el - er);
        hi = fmaxf(hi, di + el + er);
        emax = fmaxf(emax, er * er);
    }
    lo = -wave_max(-lo);
    hi = wave_max(hi);
    emax = wave_max(emax);
    __syncthreads();
    if ((tid & 63) == 0) { red3[0][tid >> 6] = lo; red3[1][tid >> 6] = hi; red3[2][tid >> 6] = emax; }
    __syncthreads();
    for (int i = 0; i < nwv; ++i) {
        lo = fminf(lo, red3[0][i]);
        hi = fmaxf(hi, red3[1][i]);
        emax = fmaxf(emax, red3[2][i]);
    }
    const float tnorm = fmaxf(fabsf(lo), fabsf(hi));
    const float eps = 1.1920929e-7f;
    lo -= 2.f * tnorm * eps * n + 1e-37f;
    hi += 2.f * tnorm * eps * n + 1e-37f;
    const float pivmin = fmaxf(1.1754944e-38f * fmaxf(emax, 1.f), 1e-37f);

    const int k_asc = (n - 1) / 2;                        // lower median, as torch.median
    float a = lo, b = hi;
    for (int it = 0; it < 40; ++it) {
        const float x = a + (b - a) * ((float)(tid + 1) / (float)(nthr + 1));
        const bool below = sturm_count(dz, ez, n, x, pivmin) <= k_asc;
        // new bracket: the largest point with count <= k and the smallest with count > k
        float na = wave_max(below && x > a ? x : a);
        float nb = wave_max(-((!below && x < b) ? x : b));
        __syncthreads();                                   // previous round's readers are done with red3
        if ((tid & 63) == 0) { red3[0][tid >> 6] = na; red3[1][tid >> 6] = nb; }
        __syncthreads();
        for (int i = 0; i < nwv; ++i) {
            na = fmaxf(na, red3[0][i]);
            nb = fmaxf(nb, red3[1][i]);
        }
        nb = -nb;
        const bool stalled = !(na > a) && !(nb < b);      // points collapsed onto the ends: resolution reached
        a = na;
        b = nb;
        if (stalled || b - a <= 2.f * eps * fmaxf(fabsf(a), fabsf(b)) + pivmin) break;   // uniform
    }
    const float sigma2 = 0.5f * (a + b);
    const float lam = (float)((double)sigma2 * o.factor);
    const int above = n - sturm_count(dz, ez, n, lam, pivmin);     // whole waves: the count broadcasts across lanes
    if (tid == 0) {
        o.rank_out[z] = above < o.cap ? above : o.cap;
        if (thr_out) thr_out[z] = lam;
        // the host's copy, written straight into pinned memory (no copy engine round on the critical path):
        // [ranks x count, 8 status words of the factorisation]
        if (o.host_mirror) o.host_mirror[z] = above < o.cap ? above : o.cap;
    }
    if (o.host_mirror && o.status && z == 0 && tid < 8) o.host_mirror[o.count + tid] = o.status[tid];
}

// ---------------------------------------------------------------------------
// Marchenko-Pastur rank of T + rho w w^T -- the UNCENTRED Gram in the basis that tridiagonalises the CENTRED one -- from
// the centred factorisation alone: z^T z = G_c + M zbar zbar^T (layer_selector.py:13 vs :35), G_c = Q T Q^T, so
// z^T z = Q (T + M w w^T) Q^T with w = Q^T zbar: a teacher layer then needs ONE factorisation instead of two (cfg-4: 24
// matrices of order 768 instead of 48; the shared stage is L2-bandwidth-bound there: half the time).
// Counting eigenvalues of a rank-one modification needs no new factorisation: for A = T - x I (LDL^T pivots q_i: the
// Sturm sequence) the bordered matrix [[A, w], [w^T, -1/rho]] has inertia(A) + inertia(s), s = -1/rho - w^T A^-1 w, and
// also inertia(-1/rho) + inertia(A + rho w w^T), hence
//     #{eigenvalues of T + rho w w^T below x} = #{q_i < 0} + [s < 0] - 1,     w^T A^-1 w = sum_i y_i^2 / q_i,  L y = w.
// Everything in fp64 (inputs are the fp32 d, e, w): next to a tiny pivot y^2 / q passes 1e70, and the sign of s is
// what is wanted.  Same order statistics and tie policy as mp_rank_block (lower median, float64 factor, threshold
// rounded to fp32 on the 1 / M scale the reference works on, strict >).  grid = count, block = 1024.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int sturm_count_rank1(const float* __restrict__ dz, const float* __restrict__ ez,
                                                 const float* __restrict__ wz, int n, double x, double pivmin,
                                                 double rho) {
    int cnt = 0;
    double q = (double)dz[0] - x;
    if (fabs(q) < pivmin) q = -pivmin;
    cnt += q < 0.;
    // (v_rcp_f64: one instruction, ~1 ulp -- an IEEE fp64 division is a ~40-instruction sequence, twice per step)
    double rq = __builtin_amdgcn_rcp(q);
    double y = (double)wz[0], acc = y * y * rq;
    for (int r = 1; r < n; ++r) {
        const double er = (double)ez[r - 1], l = er * rq;
        q = ((double)dz[r] - x) - er * l;
        if (fabs(q) < pivmin) q = -pivmin;
        cnt += q < 0.;
        rq = __builtin_amdgcn_rcp(q);
        y = (double)wz[r] - l * y;
        acc = fma(y * y, rq, acc);
    }
    const double sc = -1. / rho - acc;
    return cnt + (sc < 0. ? 1 : 0) - 1;
}

__global__ void __launch_bounds__(1024) tridiag_mp_rank1_kernel(const float* __restrict__ d, const float* __restrict__ e,
                                                                const float* __restrict__ w, int n, double rho,
                                                                double factor, int cap, int* __restrict__ rank_out,
                                                                const int* __restrict__ status,
                                                                int* __restrict__ host_mirror) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ double red2[2][16];
    __shared__ float red3[3][16];
    const int z = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x, nwv = nthr >> 6, count = gridDim.x;
    const float* dz = d + (long)z * n;
    const float* ez = e + (long)z * n;
    const float* wz = w + (long)z * n;
    float lo = 3.4e38f, hi = -3.4e38f, emax = 0.f, w2 = 0.f;
    for (int i = tid; i < n; i += nthr) {
        const float di = dz[i];
        const float el = i > 0 ? fabsf(ez[i - 1]) : 0.f, er = i < n - 1 ? fabsf(ez[i]) : 0.f;
        lo = fminf(lo, di - el - er);
        hi = fmaxf(hi, di + el + er);
        emax = fmaxf(emax, er * er);
        w2 = fmaf(wz[i], wz[i], w2);
    }
    lo = -wave_max(-lo);
    hi = wave_max(hi);
    emax = wave_max(emax);
    w2 = wave_sum(w2);
    if ((tid & 63) == 0) { red3[0][tid >> 6] = lo; red3[1][tid >> 6] = hi; red3[2][tid >> 6] = emax; red2[0][tid >> 6] = (double)w2; }
    __syncthreads();
    double wn2 = 0.;
    for (int i = 0; i < nwv; ++i) {
        lo = fminf(lo, red3[0][i]);
        hi = fmaxf(hi, red3[1][i]);
        emax = fmaxf(emax, red3[2][i]);
        wn2 += red2[0][i];
    }
    // T + rho w w^T is T plus a positive semi-definite matrix of norm rho |w|^2
    const double tnorm = fmax(fabs((double)lo), fabs((double)hi)) + rho * wn2;
    const double eps = 1.1920929e-7;
    double a = (double)lo - 2. * tnorm * eps * n - 1e-37, b = (double)hi + rho * wn2 + 2. * tnorm * eps * n + 1e-37;
    const double pivmin = fmax(1.1754944e-38 * fmax((double)emax, 1.), 1e-37);
    const int k_asc = (n - 1) / 2;                        // lower median, as torch.median
    for (int it = 0; it < 48; ++it) {
        const double x = a + (b - a) * ((double)(tid + 1) / (double)(nthr + 1));
        const bool below = sturm_count_rank1(dz, ez, wz, n, x, pivmin, rho) <= k_asc;
        double na = below && x > a ? x : a, nb = (!below && x < b) ? x : b;
        for (int m = 32; m > 0; m >>= 1) {
            na = fmax(na, __shfl_xor(na, m, 64));
            nb = fmin(nb, __shfl_xor(nb, m, 64));
        }
        __syncthreads();
        if ((tid & 63) == 0) { red2[0][tid >> 6] = na; red2[1][tid >> 6] = nb; }
        __syncthreads();
        for (int i = 0; i < nwv; ++i) {
            na = fmax(na, red2[0][i]);
            nb = fmin(nb, red2[1][i]);
        }
        const bool stalled = !(na > a) && !(nb < b);
        a = na;
        b = nb;
        // the fp32 resolution of the eigenvalue, as the fp32 bisection of mp_rank_block stops at
        if (stalled || b - a <= 2. * eps * fmax(fabs(a), fabs(b)) + pivmin) break;   // uniform
    }
    // the reference works on z^T z / M: median and threshold on that scale, in fp32 like its tensors
    const float sigma2 = (float)(0.5 * (a + b) / rho);
    const float lam = (float)((double)sigma2 * factor);
    const int above = n - sturm_count_rank1(dz, ez, wz, n, (double)lam * rho, pivmin, rho);
    if (tid == 0) {
        rank_out[z] = above < cap ? above : cap;
        if (host_mirror) host_mirror[z] = above < cap ? above : cap;
    }
    if (host_mirror && status && z == 0 && tid < 8) host_mirror[count + tid] = status[tid];
}

__global__ void __launch_bounds__(1024) tridiag_mp_rank_kernel(const float* __restrict__ d, const float* __restrict__ e,
                                                               int n, double factor, int cap,
                                                               int* __restrict__ rank_out, float* __restrict__ thr_out,
                                                               const int* __restrict__ status,
                                                               int* __restrict__ host_mirror) {
    __builtin_amdgcn_s_setprio(3);     // the host is waiting for this kernel: see tridiag_kernel
    const int z = blockIdx.x;
    const MpRankOut o{rank_out, host_mirror, status, factor, cap, (int)gridDim.x};
    mp_rank_block(d + (long)z * n, e + (long)z * n, n, z, o, thr_out);
}


// ---------------------------------------------------------------------------
// x = (T - shift I)^{-1} rhs for k right-hand sides per tridiagonal, each with its own shift that IS (to round-off)
// an eigenvalue of T: slagtf LU with partial pivoting + ONE slagts-style solve with the pivot perturbation that
// inverse iteration uses, so the answer is finite: rhs's component along the shift's own eigenvector is amplified
// by at most 1 / (eps ||T||) -- the caller projects that direction out (it asks for the solve on the orthogonal
// complement).  Same thread-per-vector LDS layout as tridiag_invit_kernel.  grid = (ceil(k/vpw), batch), block 64.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) tridiag_shifted_solve_kernel(const float* __restrict__ d,
                                                                   const float* __restrict__ e,
                                                                   const float* __restrict__ shifts, int shift_stride,
                                                                   int n, int k, int vpw,
                                                                   const float* __restrict__ rhs, float* __restrict__ X) {
    extern __shared__ float sm[];
    __shared__ float red[32];
    const int z = blockIdx.y, tid = threadIdx.x;
    const float* dz = d + (long)z * n;
    const float* ez = e + (long)z * n;
    const float tnorm = tridiag_norm(dz, ez, n, tid, 64, red);
    const int t = blockIdx.x * vpw + tid;
    if (tid >= vpw || t >= k) return;
    const float eps = 1.1920929e-7f;
    const float tol = eps * tnorm + 1e-37f;
    const long S = (long)n * vpw;
    float* a = sm + 0 * S + tid;                   // element i at a[i * vpw]
    float* b = sm + 1 * S + tid;
    float* c = sm + 2 * S + tid;
    float* d2 = sm + 3 * S + tid;
    float* in = sm + 4 * S + tid;
    float* x = sm + 5 * S + tid;
    const float l = shifts[(long)z * shift_stride + t];
    const float* r = rhs + ((long)z * k + t) * n;
    for (int i = 0; i < n; ++i) {
        a[i * vpw] = dz[i] - l;
        const float ei = i < n - 1 ? ez[i] : 0.f;
        b[i * vpw] = ei;
        c[i * vpw] = ei;
        d2[i * vpw] = 0.f;
        in[i * vpw] = 0.f;
        x[i * vpw] = r[i];
    }
    {   // slagtf
        float ai = a[0];
        float scale1 = fabsf(ai) + (n > 1 ? fabsf(b[0]) : 0.f);
        for (int i = 0; i < n - 1; ++i) {
            const float ci = c[i * vpw], bi = b[i * vpw];
            const float a1 = a[(i + 1) * vpw];
            const float b1 = i < n - 2 ? b[(i + 1) * vpw] : 0.f;
            const float scale2 = fabsf(ci) + fabsf(a1) + fabsf(b1);
            const float piv1 = ai == 0.f ? 0.f : fabsf(ai) / scale1;
            const float piv2 = ci == 0.f ? 0.f : fabsf(ci) / scale2;
            float a_next;
            if (ci == 0.f || piv2 <= piv1) {
                const float mult = ci == 0.f ? 0.f : ci / ai;
                c[i * vpw] = mult;
                a_next = a1 - mult * bi;
            } else {
                in[i * vpw] = 1.f;
                const float mult = ai / ci;
                a[i * vpw] = ci;
                a_next = bi - mult * a1;
                if (i < n - 2) {
                    d2[i * vpw] = b1;
                    b[(i + 1) * vpw] = -mult * b1;
                }
                b[i * vpw] = a1;
                c[i * vpw] = mult;
            }
            a[(i + 1) * vpw] = a_next;
            ai = a_next;
            scale1 = scale2;
        }
    }
    // forward substitution (P L)
    float prev = x[0];
    for (int i = 1; i < n; ++i) {
        const float ci = c[(i - 1) * vpw], flag = in[(i - 1) * vpw];
        const float xi = x[i * vpw];
        float keep, next;
        if (flag == 0.f) { keep = prev; next = xi - ci * prev; }
        else { keep = xi; next = prev - ci * xi; }
        x[(i - 1) * vpw] = keep;
        prev = next;
    }
    x[(n - 1) * vpw] = prev;
    // back substitution (U) with perturbed tiny pivots
    float x1 = 0.f, x2 = 0.f;
    float* out = X + ((long)z * k + t) * n;
    for (int i = n - 1; i >= 0; --i) {
        float tmp = x[i * vpw] - b[i * vpw] * x1 - d2[i * vpw] * x2;
        float ak = a[i * vpw];
        if (fabsf(ak) < tol) ak = copysignf(tol, ak == 0.f ? 1.f : ak);
        tmp /= ak;
        out[i] = tmp;
        x2 = x1;
        x1 = tmp;
    }
}

}  // namespace basd

using namespace basd;

extern "C" {

// Householder tridiagonalisation of `batch` symmetric matrices (destroyed).  d, e, tau: (batch, n);
// vh: (batch, n, n) reflector rows.
//
// Two stages.  The trailing block of order <= 256 is factored by tridiag_tail_kernel in the registers of ONE CU per
// matrix (no exchange, nothing spins: n <= 256 never leaves that kernel).  For n > 256 the first n - 256 steps --
// where one CU's issue rate and L2 bandwidth bound a step -- run on tridiag_kernel with the matrix shared by up to 16
// workgroups ("members", one 32-row block each at n = 384; measured at n = 384, 6 matrices, whole factorisation:
// 3.12 ms with 1 member, 1.95 with 4, 1.63 with 12).  Members poll each other's granules, so all of them must be
// resident: at most 128 workgroups per launch (two concurrent launches still fit the 256 CUs).
static int nblk_of(int n) { return (n + TRI_BLK - 1) / TRI_BLK; }

constexpr int TRI_TAIL_MAX = 256;      // order of the register-resident trailing block (64 * CPL of the tail kernel)

// Tuning / test hooks, read ONCE when the library is loaded (never inside an entry point):
//   BASD_TRIDIAG_MEMBERS  members per matrix in the shared stage (default: one per 32-row block, <= 16)
//   BASD_TRIDIAG_PAD      workgroup-id padding between matrices (scatters the members over XCDs; tests)
//   BASD_TRIDIAG_LAG      member that sleeps every step (tests the hand-off under uneven progress)
//   BASD_TRIDIAG_THREADS  threads per member
//   BASD_TRIDIAG_TAIL     1 (default): orders 257..384 whole in tridiag_packed_kernel (one CU's registers, upper triangle),
//                         other orders shared stage + tridiag_tail2_kernel; 3: shared stage + tail2 for every order (the
//                         round-2 path); 2: the same with the four-barrier tail kernel; 0: whole factorisation in the
//                         shared stage (the round-1 path).  Tests compare all of them.
struct TridiagTuning {
    int members = 0, pad = -1, lag = -1, threads = 0, tail = 1;
    TridiagTuning() {
        if (const char* s = getenv("BASD_TRIDIAG_MEMBERS")) members = atoi(s);
        if (const char* s = getenv("BASD_TRIDIAG_PAD")) pad = atoi(s);
        if (const char* s = getenv("BASD_TRIDIAG_LAG")) lag = atoi(s);
        if (const char* s = getenv("BASD_TRIDIAG_THREADS")) threads = atoi(s);
        if (const char* s = getenv("BASD_TRIDIAG_TAIL")) tail = atoi(s);
    }
};
static TridiagTuning g_tuning;
static const int g_pk_clocks = getenv("BASD_TRIDIAG_CLOCKS") ? atoi(getenv("BASD_TRIDIAG_CLOCKS")) : 0;   // diagnostics

// Workgroups of one shared-stage launch that may spin on each other: all of them must be resident together, beside
// whatever the other streams keep on the chip.  Half of what the device can hold of this kernel (occupancy query for
// the actual kernel, 1024 threads, its LDS; once per process and n), never more than 128 -- on a 256-CU MI355X that
// is the old rule (two concurrent launches fit even at one workgroup per CU); a smaller or partitioned device
// gets a smaller budget, down to one member per matrix (no exchange at all).
static int tridiag_resident_budget(int n) {
    static std::atomic<int> cached_n{0}, cached_budget{0};
    if (cached_n.load(std::memory_order_acquire) == n) return cached_budget.load(std::memory_order_relaxed);
    int dev = 0, cus = 0, per_cu = 0, budget = 128;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)tridiag_kernel<true, true>, 1024,
                                                     sizeof(float) * 9 * (size_t)n) == hipSuccess &&
        cus > 0 && per_cu > 0) {
        budget = per_cu * cus / 2;
        if (budget > 128) budget = 128;
    }
    if (const char* sb = getenv("BASD_TRIDIAG_BUDGET")) budget = atoi(sb);      // experiment hook, read once per n
    cached_budget.store(budget, std::memory_order_relaxed);
    cached_n.store(n, std::memory_order_release);
    return budget;
}

static int tridiag_members(int n, int batch) {
    const int nblk = nblk_of(n);
    int p = g_tuning.members > 0 ? g_tuning.members : nblk;
    if (p > 16) p = 16;
    const int budget = tridiag_resident_budget(n);
    while (p > 1 && p * batch > budget) --p;
    if (p > nblk) p = nblk;
    return p < 1 ? 1 : p;
}

// Test / tuning hook: overrides the values read from the environment at load time (negative = keep the current value;
// `reset` != 0 first restores the load-time environment values).  Process-wide; meant for tests and experiments, not for
// use while factorisations are being queued from other threads.
int basd_tridiag_tuning(int members, int pad, int lag, int threads, int tail, int reset) {
    if (reset) g_tuning = TridiagTuning();
    if (members >= 0) g_tuning.members = members;
    if (pad >= 0) g_tuning.pad = pad;
    if (lag >= 0) g_tuning.lag = lag;
    if (threads >= 0) g_tuning.threads = threads;
    if (tail >= 0) g_tuning.tail = tail;
    return BASD_OK;
}

long basd_tridiag_workspace_bytes(int n, int batch) {
    // per matrix: 2 parities x n granules of 16 bytes, then the pending (v, w) handed to the tail stage; at the very
    // end the status word and 7 words of give-up trace
    return (long)batch * 2 * n * 16 + (long)batch * 2 * n * 4 + 32;
}

static int tridiag_impl(float* a, long a_batch_stride, int n, int batch, float* d, float* e, float* tau, float* vh,
                        void* work, MpRankOut rk, hipStream_t stream, hipEvent_t mid_event = nullptr,
                        const unsigned* go_flag = nullptr, unsigned go_value = 0, int go_budget = 0) {
    BASD_CHECK_ARG(a && d && e && tau && vh && work && n > 1 && batch > 0);
    BASD_CHECK_ARG((((uintptr_t)work) & 15) == 0);
    if (n > 4096) return BASD_EUNSUPPORTED;
    const long gran_bytes = (long)batch * 2 * n * 16;
    uint4* xg = (uint4*)work;
    float* pend = (float*)((char*)work + gran_bytes);
    int* err = (int*)((char*)work + gran_bytes + (long)batch * 2 * n * 4);
    const bool tail = g_tuning.tail != 0;
    // orders 257..384: the whole factorisation in one CU's registers (upper triangle): no shared stage at all
    const bool packed = g_tuning.tail == 1 && n > TRI_TAIL_MAX && n <= PK_NMAX;
    if (go_flag && !packed) return BASD_EUNSUPPORTED;       // only the one-kernel factorisation can be queued early
    const int j_stop = !tail ? n - 1 : (packed ? 0 : (n > TRI_TAIL_MAX ? n - TRI_TAIL_MAX : 0));
    if (j_stop > 0) {
        const bool vec = (n & 3) == 0 && (a_batch_stride & 3) == 0 && (((uintptr_t)a) & 15) == 0;
        const int P = tridiag_members(n, batch);
        int batch_pad = P > 1 ? (batch + 7) & ~7 : batch;
        if (g_tuning.pad >= 0) batch_pad = batch + g_tuning.pad;
        // 20-bit launch nonce in the granule tags (12 bits of step below it: n <= 4096), never 0: a zero-filled
        // granule of fresh memory must not look like step 0 of any launch.  The counter only has to differ between
        // launches that can see each other's granules (same recycled buffer), not to be unique for ever.
        static std::atomic<unsigned> launches{0};
        const unsigned tag_base = (1u + launches.fetch_add(1, std::memory_order_relaxed) % 0xFFFFFu) << 12;
        const size_t lds = sizeof(float) * 9 * (size_t)n;
        // Always 1024 threads per member.  A member with one 32-row block keeps only four waves busy in the pass and
        // 512 threads made its barriers cheaper (n = 384, 12 members: 1.55 ms against 1.63) -- but round 3 found every
        // workgroup size BELOW 1024 (256, 512, 768; 4, 6 or 12 members) delivering wrong factorisations (status words
        // clean, granule check words consistent, all members of a matrix on one XCD) when the split-operand Gram launch
        // runs beside it, 5-10 times in 10, and never with the fp32 Gram launch or alone; 1024 threads: never, under
        // the same load (test_tridiag_members_under_uneven_load; 800 factorisations of orders 384 and 768 beside either
        // split-operand kernel: none wrong).  What was ruled out: torn granules (a check word travels with them), members on
        // different XCDs (HW_REG_XCC_ID recorded: one XCD per matrix), stale LDS, barriers that leave VMEM in flight (all
        // replaced by __syncthreads: same), wave priority, the LDS footprint of the neighbour (fp32 Gram launch padded to
        // the same 48 KB: clean); the fp32 MFMA kernels, the LDS / VALU-bound Jacobi and HBM streams as neighbours: clean.
        // Cause not found; the smaller sizes stay reachable through basd_tridiag_tuning(threads) for whoever wants to hunt
        // it, production never takes them.
        int threads = 1024;
        if (g_tuning.threads > 0) threads = g_tuning.threads;
        const int lag = g_tuning.lag;
        if (vec && (n & 7) == 0) tridiag_kernel<true, true><<<P * batch_pad, threads, lds, stream>>>(a, a_batch_stride, n, batch, batch_pad, P, d, e, tau, vh, xg, err, tag_base, lag, j_stop, pend);
        else if (vec) tridiag_kernel<true, false><<<P * batch_pad, threads, lds, stream>>>(a, a_batch_stride, n, batch, batch_pad, P, d, e, tau, vh, xg, err, tag_base, lag, j_stop, pend);
        else tridiag_kernel<false, false><<<P * batch_pad, threads, lds, stream>>>(a, a_batch_stride, n, batch, batch_pad, P, d, e, tau, vh, xg, err, tag_base, lag, j_stop, pend);
    } else {
        hipError_t me = hipMemsetAsync(err, 0, 32, stream);        // no shared stage: the status words stay clean
        if (me != hipSuccess) return (int)me;
    }
    // recorded behind the shared (multi-workgroup) stage: from here on the factorisation only occupies one CU per
    // matrix, so work that was held back to keep the chip quiet for the spinning members may start
    if (mid_event) {
        hipError_t ee = hipEventRecord(mid_event, stream);
        if (ee != hipSuccess) return (int)ee;
    }
    rk.status = err;
    const bool fused_rank = tail && rk.rank_out && 2 * n <= 8 * TRI_TAIL_MAX;
    if (tail) {
        MpRankOut in_tail = rk;
        if (!fused_rank) in_tail.rank_out = nullptr;
        if (packed)
            tridiag_packed_kernel<<<batch, 64 * PK_WAVES, 0, stream>>>(a, a_batch_stride, n, d, e, tau, vh, in_tail, g_pk_clocks, go_flag, go_value, go_budget);
        else if (g_tuning.tail == 2)      // the four-barrier form (16 waves): kept for the tests that compare the two
            tridiag_tail_kernel<16, 16, 4><<<batch, 1024, 0, stream>>>(a, a_batch_stride, n, j_stop, j_stop > 0 ? pend : nullptr, d, e, tau, vh, in_tail);
        else
            tridiag_tail2_kernel<<<batch, 512, 0, stream>>>(a, a_batch_stride, n, j_stop, j_stop > 0 ? pend : nullptr, d, e, tau, vh, in_tail);
    }
    if (rk.rank_out && !fused_rank)
        tridiag_mp_rank_kernel<<<rk.count, 1024, 0, stream>>>(d, e, n, rk.factor, rk.cap, rk.rank_out, nullptr, rk.host_mirror ? err : nullptr, rk.host_mirror);
    BASD_RETURN_LAST();
}

int basd_tridiag(float* a, long a_batch_stride, int n, int batch, float* d, float* e, float* tau, float* vh,
                 void* work, hipStream_t stream) {
    return tridiag_impl(a, a_batch_stride, n, batch, d, e, tau, vh, work, MpRankOut{nullptr, nullptr, nullptr, 0., 0, 0}, stream);
}

// basd_tridiag + the Marchenko-Pastur ranks (basd_tridiag_mp_rank) of the FIRST rank_count matrices of the batch, computed
// by the workgroup that finishes the factorisation (no separate launch on the path the host waits for).  host_mirror
// (nullable): pinned host memory of rank_count + 8 ints -- the ranks, then the factorisation's 8 status words.
int basd_tridiag_ranked(float* a, long a_batch_stride, int n, int batch, float* d, float* e, float* tau, float* vh,
                        void* work, int rank_count, double factor, int cap, int* rank_out, int* host_mirror,
                        void* mid_event, hipStream_t stream) {
    BASD_CHECK_ARG(rank_out && rank_count > 0 && rank_count <= batch);
    if (n > 8192) return BASD_EUNSUPPORTED;
    return tridiag_impl(a, a_batch_stride, n, batch, d, e, tau, vh, work,
                        MpRankOut{rank_out, host_mirror, nullptr, factor, cap, rank_count}, stream,
                        (hipEvent_t)mid_event);
}

// basd_tridiag_ranked QUEUED BEFORE ITS INPUT EXISTS (orders 257..384 only: BASD_EUNSUPPORTED otherwise): the kernel waits,
// bounded, until *go_flag == go_value (agent scope) and only then reads the matrices; see tridiag_packed_kernel.  The
// caller queues this on a stream of its own and sets the word with basd_flag_set on the stream that produces the
// matrices, behind them.  go_budget: polls of ~1.7 us before the workgroups give up (status word 0 = 2, mirrored).
int basd_tridiag_ranked_gated(float* a, long a_batch_stride, int n, int batch, float* d, float* e, float* tau, float* vh,
                              void* work, int rank_count, double factor, int cap, int* rank_out, int* host_mirror,
                              const unsigned* go_flag, unsigned go_value, int go_budget, hipStream_t stream) {
    BASD_CHECK_ARG(rank_out && rank_count > 0 && rank_count <= batch && go_flag && go_value != 0 && go_budget > 0);
    return tridiag_impl(a, a_batch_stride, n, batch, d, e, tau, vh, work,
                        MpRankOut{rank_out, host_mirror, nullptr, factor, cap, rank_count}, stream, nullptr, go_flag,
                        go_value, go_budget);
}

namespace basd {
__global__ void flag_set_kernel(unsigned* flag, unsigned value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace basd
int basd_flag_set(unsigned* flag, unsigned value, hipStream_t stream) {
    BASD_CHECK_ARG(flag);
    basd::flag_set_kernel<<<1, 1, 0, stream>>>(flag, value);
    BASD_RETURN_LAST();
}

// All eigenvalues (descending) of the tridiagonals by Sturm bisection.
int basd_tridiag_eigenvalues(const float* d, const float* e, int n, int batch, float* vals_desc, hipStream_t stream) {
    BASD_CHECK_ARG(d && e && vals_desc && n > 0 && batch > 0);
    if (n > 8192) return BASD_EUNSUPPORTED;
    sturm_bisect_kernel<<<dim3((n + 63) / 64, batch), 1024, 0, stream>>>(d, e, n, vals_desc);
    BASD_RETURN_LAST();
}

// out (batch, k_stride, n) rows = Q x or Q^T x for the rows of x (batch, k, n); Q = H_0 ... H_{n-2} of basd_tridiag.
int basd_tridiag_apply_q(const float* tau, const float* vh, int n, int k, int batch, const float* x, float* out,
                         int k_stride, int transpose, hipStream_t stream) {
    BASD_CHECK_ARG(tau && vh && x && out && n > 1 && k > 0 && batch > 0 && k_stride >= k);
    const dim3 grid((k + 3) / 4, batch);
#define BASD_APPLY_Q(E)                                                                                          \
    do {                                                                                                          \
        if (transpose) backtransform_kernel<E, true><<<grid, 256, 0, stream>>>(vh, tau, n, k, x, out, k_stride);  \
        else backtransform_kernel<E, false><<<grid, 256, 0, stream>>>(vh, tau, n, k, x, out, k_stride);           \
    } while (0)
    if (n <= 192) BASD_APPLY_Q(3);
    else if (n <= 384) BASD_APPLY_Q(6);
    else if (n <= 768) BASD_APPLY_Q(12);
    else if (n <= 1024) BASD_APPLY_Q(16);
    else return BASD_EUNSUPPORTED;
#undef BASD_APPLY_Q
    BASD_RETURN_LAST();
}

// x[z][t] = (T_z - shifts[z][t] I)^{-1} rhs[z][t] for t < k (see tridiag_shifted_solve_kernel); rhs, x: (batch, k, n).
int basd_tridiag_shifted_solve(const float* d, const float* e, const float* shifts, int shift_stride, int n, int k,
                               int batch, const float* rhs, float* x, hipStream_t stream) {
    BASD_CHECK_ARG(d && e && shifts && rhs && x && n > 1 && k > 0 && batch > 0 && shift_stride >= k);
    int vpw = (int)((144 * 1024) / (6 * sizeof(float) * (size_t)n));
    if (vpw > 64) vpw = 64;
    if (vpw < 1) return BASD_EUNSUPPORTED;
    const size_t lds_a = sizeof(float) * 6 * (size_t)n * vpw;
    if (lds_a > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)tridiag_shifted_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a);
    tridiag_shifted_solve_kernel<<<dim3((k + vpw - 1) / vpw, batch), 64, lds_a, stream>>>(d, e, shifts, shift_stride, n, k, vpw, rhs, x);
    BASD_RETURN_LAST();
}

// MP ranks of `batch` tridiagonals (d, e: batch x n) without computing their spectra; `factor` = (1 + sqrt(D/M))^2
// in float64 from the host as in basd_mp_rank.  thr_out nullable.  host_mirror (nullable): device-visible pinned
// host memory of batch + 8 ints that receives the ranks and the 8 words at `status` (nullable).
int basd_tridiag_mp_rank(const float* d, const float* e, int n, int batch, double factor, int cap, int* rank_out,
                         float* thr_out, const int* status, int* host_mirror, hipStream_t stream) {
    BASD_CHECK_ARG(d && e && rank_out && n > 0 && batch > 0);
    if (n > 8192) return BASD_EUNSUPPORTED;
    tridiag_mp_rank_kernel<<<batch, 1024, 0, stream>>>(d, e, n, factor, cap, rank_out, thr_out, status, host_mirror);
    BASD_RETURN_LAST();
}

// MP ranks of the matrices Q_z (T_z + rho w_z w_z^T) Q_z^T from the tridiagonals (d, e) and the vectors w (batch x n, fp32):
// see tridiag_mp_rank1_kernel.  rho = M (rows of the projected tokens), factor as basd_mp_rank.
int basd_tridiag_mp_rank_rank1(const float* d, const float* e, const float* w, int n, int batch, double rho,
                               double factor, int cap, int* rank_out, const int* status, int* host_mirror,
                               hipStream_t stream) {
    BASD_CHECK_ARG(d && e && w && rank_out && n > 0 && batch > 0 && rho > 0.);
    if (n > 8192) return BASD_EUNSUPPORTED;
    tridiag_mp_rank1_kernel<<<batch, 1024, 0, stream>>>(d, e, w, n, rho, factor, cap, rank_out, status, host_mirror);
    BASD_RETURN_LAST();
}

#ifdef BASD_TAIL_DBG
// stamped builds only (tools/tail_stamps_in_step.py): the s_memtime stamps of the last tail-stage launches
int basd_debug_tail_stamps(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(basd::g_tail_dbg), sizeof(long long) * 2 * 8 * 2 * 1024);
}
#endif

// Top-k eigenvectors of the ORIGINAL matrices (rows of vecs: (batch, k_stride, n), first k rows written).
// z: batch * k * n floats of scratch (eigenvectors of the tridiagonals).
int basd_tridiag_eigenvectors(const float* d, const float* e, const float* tau, const float* vh,
                              const float* vals_desc, int n, int k, int batch, float* z, float* vecs, int k_stride,
                              hipStream_t stream) {
    BASD_CHECK_ARG(d && e && tau && vh && vals_desc && z && vecs && n > 1 && k > 0 && k <= n && batch > 0);
    BASD_CHECK_ARG(k_stride >= k && k <= 1024);
    int vpw = (int)((144 * 1024) / (6 * sizeof(float) * (size_t)n));
    if (vpw > 64) vpw = 64;
    if (vpw < 1) return BASD_EUNSUPPORTED;
    const size_t lds_a = sizeof(float) * 6 * (size_t)n * vpw;
    if (lds_a > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)tridiag_invit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a);
    tridiag_invit_kernel<<<dim3((k + vpw - 1) / vpw, batch), 64, lds_a, stream>>>(d, e, vals_desc, n, k, vpw, z);
    {
        if (n > 1024) return BASD_EUNSUPPORTED;
        const size_t lds_c = sizeof(float) * (((size_t)k + 3) / 4 * 4 + 16 * (size_t)n);
        if (lds_c > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)cluster_orth_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
        cluster_orth_kernel<<<batch, 1024, lds_c, stream>>>(d, e, vals_desc, n, k, z);
    }
    const dim3 grid((k + 3) / 4, batch);
    if (n <= 192) backtransform_kernel<3><<<grid, 256, 0, stream>>>(vh, tau, n, k, z, vecs, k_stride);
    else if (n <= 384) backtransform_kernel<6><<<grid, 256, 0, stream>>>(vh, tau, n, k, z, vecs, k_stride);
    else if (n <= 768) backtransform_kernel<12><<<grid, 256, 0, stream>>>(vh, tau, n, k, z, vecs, k_stride);
    else if (n <= 1024) backtransform_kernel<16><<<grid, 256, 0, stream>>>(vh, tau, n, k, z, vecs, k_stride);
    else return BASD_EUNSUPPORTED;
    BASD_RETURN_LAST();
}

}  // extern "C"
