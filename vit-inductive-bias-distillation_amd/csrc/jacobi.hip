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

// (4 eps)^2: squared relative size below which a column is numerically null
constexpr float kNull2 = 2.2737368e-13f;

struct Rot {
    float c, s, t; // fp32 cosine / sine / tangent of the plane rotation
    float h;       // (c^2 + s^2)(1 + h)^2 = 1 to ~1e-14: the rotation as applied shrinks both columns by (1 - h);
                   // h is accumulated per column and folded back in once, at write-back
    bool apply;
    bool strong;   // |cos| > sqrt(tol): see the stopping rule of jacobi_lds_kernel
};

// Rotation that makes columns p,q orthogonal given alpha=|p|^2, beta=|q|^2, gamma=p.q.
// The angle only has to be good enough for quadratic convergence, so it is built from the 1-ulp hardware
// reciprocal / square-root instructions; what must be exact is c^2 + s^2 = 1, restored by the low parts.
__device__ __forceinline__ Rot make_rotation(float alpha, float beta, float gamma, float tol) {
    Rot r{1.f, 0.f, 0.f, 0.f, false, false};
    const float cosv = gamma * __builtin_amdgcn_rsqf(alpha) * __builtin_amdgcn_rsqf(beta);
    if (!(fabsf(cosv) > tol)) return r;                      // also catches NaN and zero columns
    r.strong = cosv * cosv > tol;
    const float zeta = (beta - alpha) * __builtin_amdgcn_rcpf(2.f * gamma);
    const float az = fabsf(zeta);
    float t = az > 1e8f ? 0.5f * __builtin_amdgcn_rcpf(az)
                        : __builtin_amdgcn_rcpf(az + __builtin_amdgcn_sqrtf(fmaf(zeta, zeta, 1.f)));
    t = copysignf(t, zeta);
    r.t = t;
    r.c = __builtin_amdgcn_rsqf(fmaf(t, t, 1.f));
    r.s = r.c * t;
    // defect = 1 - c^2 - s^2 with error-free squares (c^2 in [0.5, 1]: the subtractions are exact);
    // over the ~n * sweeps rotations a column sees, an O(eps) defect is a random walk of its norm
    // (measured 6e-6 relative at n = 96 before this correction).
    const float pc = r.c * r.c, ec = fmaf(r.c, r.c, -pc);
    const float ps = r.s * r.s, es = fmaf(r.s, r.s, -ps);
    r.h = 0.5f * ((((1.f - pc) - ps) - ec) - es);
    r.apply = true;
    return r;
}

template <int LPP>
__device__ __forceinline__ float pair_allsum(float x) {
    static_assert(LPP == 4 || LPP == 8 || LPP == 16 || LPP == 32 || LPP == 64,
                  "a pair is owned by a quad, half a DPP row, a row, two rows or a wave");
    if constexpr (LPP == 4 || LPP == 8) {
        x += dpp_get<0xB1>(x);    // quad_perm [1,0,3,2]
        x += dpp_get<0x4E>(x);    // quad_perm [2,3,0,1]
        if constexpr (LPP == 8) x += dpp_get<0x141>(x);   // row_half_mirror
        return x;
    } else if constexpr (LPP == 32) {
        x = row16_allsum(x);
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else {
        return LPP == 16 ? row16_allsum(x) : wave64_allsum(x);
    }
}

// Orthogonalise LDS columns p and q with one DPP row of 16 lanes (gl = lane index in the row).
// LDS columns are zero-padded to 16 * EPL rows, dot rows first (padded to 16 * DOT), so lane gl owns rows
// gl + 16 i, i < EPL, of which the first DOT enter the dot product: no per-element predicate at all, and
// the elements stay in registers between the dot product and the update (one LDS read + one write each).
//
// Squared column norms are cached in LDS (n2p / n2q point at the two entries): they are recomputed from the
// data once per sweep (LDS solver) or launch (block solver) and follow the rotations analytically in
// between (alpha' = alpha - t gamma, beta' = beta + t gamma), so a pair-step needs ONE dot product.
// devp / devq accumulate the normalisation defect h of every rotation applied to the column.
//
// null2: columns whose squared norm is below it are numerically null (sigma < 4 eps sigma_max, measured in
// the previous sweep); they are pure round-off, never become "orthogonal relative to their own size", and
// would keep the sweep loop alive for ever, so pairs involving them count as converged.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int EPL, int DOT, int LPP, bool LOG = false>
__device__ __forceinline__ int rotate_pair(float* __restrict__ cp, float* __restrict__ cq, int gl, float tol,
                                            float null2, float& norm2_max, float* __restrict__ n2p,
                                            float* __restrict__ n2q, float* __restrict__ devp,
                                            float* __restrict__ devq, f32x2* __restrict__ logp = nullptr) {
    // lane gl owns the row pairs (2 gl, 2 gl + 1) + 2 LPP i: 8-byte LDS accesses and packed fp32 math
    static_assert(EPL % 2 == 0 && DOT % 2 == 0, "row chunks come in pairs");
    constexpr int H = EPL / 2, HD = DOT / 2;
    // the dot rows stay in registers between the dot product and the update; the riding rows (stacked shapes) are
    // read, rotated and written afterwards, a pair at a time: half the registers of holding both halves (the
    // Procrustes kernel then fits 64 VGPRs, see jacobi_lds_lowreg_kernel)
    f32x2 x[HD], y[HD];
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < HD; ++i) {
        x[i] = *(const f32x2*)(cp + 2 * gl + 2 * LPP * i);
        y[i] = *(const f32x2*)(cq + 2 * gl + 2 * LPP * i);
        acc = __builtin_elementwise_fma(x[i], y[i], acc);
    }
    const float g = pair_allsum<LPP>(acc.x + acc.y);
    const float a = *n2p, b = *n2q;
    norm2_max = fmaxf(norm2_max, fmaxf(a, b));
    // LOG: (c, s) of every pair-step goes to a log that jacobi_apply_log_kernel replays on the riding rows
    if (fminf(a, b) <= null2) {
        if (LOG && gl == 0) *logp = f32x2{1.f, 0.f};
        return 0;
    }
    const Rot rot = make_rotation(a, b, g, tol);   // identical in every lane of the group
    if (!rot.apply) {
        if (LOG && gl == 0) *logp = f32x2{1.f, 0.f};
        return 0;
    }
    const f32x2 c2 = {rot.c, rot.c}, s2 = {rot.s, rot.s};
#pragma unroll
    for (int i = 0; i < HD; ++i) {
        *(f32x2*)(cp + 2 * gl + 2 * LPP * i) = __builtin_elementwise_fma(c2, x[i], -(s2 * y[i]));
        *(f32x2*)(cq + 2 * gl + 2 * LPP * i) = __builtin_elementwise_fma(s2, x[i], c2 * y[i]);
    }
#pragma unroll
    for (int i = HD; i < H; ++i) {
        const f32x2 xr = *(const f32x2*)(cp + 2 * gl + 2 * LPP * i);
        const f32x2 yr = *(const f32x2*)(cq + 2 * gl + 2 * LPP * i);
        *(f32x2*)(cp + 2 * gl + 2 * LPP * i) = __builtin_elementwise_fma(c2, xr, -(s2 * yr));
        *(f32x2*)(cq + 2 * gl + 2 * LPP * i) = __builtin_elementwise_fma(s2, xr, c2 * yr);
    }
    if (gl == 0) {
        *n2p = fmaxf(a - rot.t * g, 0.f);
        *n2q = b + rot.t * g;
        *devp += rot.h;
        *devq += rot.h;
        if (LOG) *logp = f32x2{rot.c, rot.s};
    }
    return rot.strong ? 3 : 1;      // bit 0: rotated, bit 1: by more than sqrt(tol)
}

// exact squared norm (over the dot rows) of one padded LDS column, by one lane group
template <int DOT, int LPP>
__device__ __forceinline__ float column_norm2(const float* __restrict__ col, int gl) {
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < DOT / 2; ++i) {
        const f32x2 v = *(const f32x2*)(col + 2 * gl + 2 * LPP * i);
        acc = __builtin_elementwise_fma(v, v, acc);
    }
    return pair_allsum<LPP>(acc.x + acc.y);
}

// LDS row of global row r: dot rows first, the riding rows start at the next multiple of 16
template <int EPL, int DOT, int LPP>
__device__ __forceinline__ int lds_row(int r, int rows_dot) {
    return (DOT == EPL || r < rows_dot) ? r : LPP * DOT + (r - rows_dot);
}

// ---------------------------------------------------------------------------
// LDS-resident solver.  grid = batch, block = multiple of 64; a pair = LPP adjacent lanes (one DPP row of 16, or one
// quad).  DOT == EPL: rows_dot == rows_tot <= LPP EPL.   DOT < EPL: rows_dot <= LPP DOT, rows_tot - rows_dot <=
// LPP (EPL - DOT).
// LPP = 16 keeps the latency of ONE solve low (few matrices, e.g. the k x k principal-angle problems).
// LPP = 4 is the throughput shape for large batches of small matrices (the Procrustes cores, 1024 x (98 x 49) at
// cfg-2): the rotation (~30 scalar-like instructions) is computed once per 4 lanes instead of once per 16 and the
// cross-lane sum is 2 DPP steps instead of 4, so a matrix round costs about a third of the instruction issues --
// the kernel is bound by VALU / LDS issue with every CU holding several matrices, not by the latency of a round.
// ---------------------------------------------------------------------------
template <int EPL, int DOT, int LPP = 16>
__global__ void __launch_bounds__((LPP == 4 && EPL >= 48) ? 512 : 1024) jacobi_lds_kernel(float* __restrict__ W, long batch_stride, int rows_dot,
                                                           int rows_tot, int n_fixed, const int* __restrict__ n_arr,
                                                           int max_sweeps, float tol, float* __restrict__ colnorm,
                                                           int colnorm_stride, int* __restrict__ sweeps_out) {
    // Column stride.  16 lanes per pair: + 2 (even: 8-byte aligned columns; PMC: no bank conflicts).  4 lanes per pair: a
    // half-wave's ds_read_b64 touches 8 different columns, 8 banks each; with the stride 8 x odd the 8 consecutive
    // columns of a round-robin round tile the 64 banks exactly (PMC with + 2: 40 % of the LDS cycles were conflicts).
    // (8 lanes per pair: 4 columns of 16 banks per half-wave, stride 16 x odd.)
    // (4 lanes per pair, 4 EPL already 8 x odd -- the 196-row columns of cfg-4's cores, EPL = 50: no padding at all, which
    // is what lets a 196 x 196 matrix fit one CU's LDS)
    constexpr int LD = LPP * EPL + (LPP == 4 ? (((LPP * EPL / 8) % 2 == 0) ? 8 : 0) : LPP == 8 ? 16 : 2);
    static_assert(LPP != 4 || ((LD / 8) % 2 == 1 && LD % 8 == 0), "stride must come out as 8 x odd");
    static_assert(LPP != 8 || ((LPP * EPL / 16) % 2 == 0), "stride must come out as 16 x odd");
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
    const int n_even = (n + 1) & ~1;
    for (int idx = tid; idx < n_even * LD; idx += nthr) lds[idx] = 0.f;
    __syncthreads();
    // global (leading dim rows_tot) -> padded LDS columns
    for (int idx = tid; idx < n * rt; idx += nthr) {
        const int c = idx / rt, r = idx - c * rt;
        lds[c * LD + lds_row<EPL, DOT, LPP>(r, rd)] = Wm[(long)c * rows_tot + r];
    }
    __syncthreads();

    const int groups = nthr / LPP, grp = tid / LPP, gl = tid % LPP;
    float* n2 = lds + n_even * LD;      // cached squared column norms
    float* dev = n2 + n_even;           // accumulated normalisation defects
    __shared__ int s_norm2_bits;
    if (tid == 0) s_norm2_bits = 0;
    for (int c = tid; c < n_even; c += nthr) dev[c] = 0.f;
    float null2 = 0.f;
    int sweep = 0;
    for (; sweep < max_sweeps && n > 1; ++sweep) {
        int rotated = 0;
        float norm2_max = 0.f;
        for (int c = grp; c < n_even; c += groups) {          // refresh the cached norms from the data
            const float v = column_norm2<DOT, LPP>(lds + c * LD, gl);
            if (gl == 0) n2[c] = v;
        }
        __syncthreads();
        for (int r = 0; r < n_even - 1; ++r) {
            for (int t = grp; t < n_even / 2; t += groups) {
                int p, q;
                rr_pair(n_even, r, t, p, q);
                if (p >= n || q >= n) continue;  // padding column of an odd-order matrix
                if (p > q) { const int tmp = p; p = q; q = tmp; }
                rotated |= rotate_pair<EPL, DOT, LPP>(lds + p * LD, lds + q * LD, gl, tol, null2, norm2_max, n2 + p, n2 + q,
                                                 dev + p, dev + q);
            }
            __syncthreads();
        }
        if (gl == 0 && norm2_max > 0.f) atomicMax(&s_norm2_bits, __float_as_int(norm2_max));
        // Stopping rule: the sweep applied every rotation above tol, and Jacobi converges quadratically --
        // if none of them exceeded sqrt(tol), what is left afterwards is O(tol): done, without the extra
        // sweep that would only verify it (one sweep in ~9 at n = 49).
        if (!__syncthreads_or(rotated & 2)) {
            ++sweep;
            break;
        }
        null2 = kNull2 * __int_as_float(s_norm2_bits);   // every column norm of this sweep has been folded in
    }
    if (sweeps_out && tid == 0) sweeps_out[m] = sweep;

    // column norms over the dot rows + write back, both with the accumulated defect folded in
    for (int c = grp; c < n; c += groups) {
        const float a = column_norm2<DOT, LPP>(lds + c * LD, gl);
        if (gl == 0) {
            const float nv = sqrtf(a);
            colnorm[(long)m * colnorm_stride + c] = fmaf(nv, dev[c], nv);
        }
    }
    for (int idx = tid; idx < n * rt; idx += nthr) {
        const int c = idx / rt, r = idx - c * rt;
        const float v = lds[c * LD + lds_row<EPL, DOT, LPP>(r, rd)];
        Wm[(long)c * rows_tot + r] = fmaf(v, dev[c], v);
    }
}

// ---------------------------------------------------------------------------
// Register-resident solver with the odd-even ("brick wall") ordering, for plain matrices (no riding rows) in batches:
// the transposed Procrustes cores.
//
// The round-robin solver above keeps the columns in LDS and every pair-step reads two columns and writes two: at 196
// rows the matrix fills a CU's LDS (one matrix per CU, all its waves in the same phase: LDS read, rotate, LDS write),
// and the LDS pipe is what a round costs.  Here the columns sit on a line of seats 0 .. n - 1; even rounds rotate the
// seats (2g, 2g + 1), odd rounds (2g + 1, 2g + 2), and after every rotation the two columns trade seats.  n rounds are one
// sweep: every pair meets exactly once (the exchanges of odd-even transposition on a reversed sequence), and the line
// ends up reversed.  Group g (4 lanes) holds BOTH its columns in registers; from one round to the next it keeps one of
// them and passes the other to a neighbour through LDS: ONE column written and ONE read per pair-step instead of two
// and two, and only the even seats ever travel -- the mail slots are half a matrix (78 KB at 196 x 196), so TWO
// matrices share a CU and one's LDS phase runs under the other's arithmetic.  One barrier per round: a slot is written
// by the group that read it last.
//   round e (even): rotate (X = seat 2g, Y = seat 2g + 1); they trade seats; Y (now seat 2g) -> slot g; barrier;
//                   slot g + 1 -> Y (seat 2g + 2)
//   round o (odd) : rotate (X = seat 2g + 1, Y = seat 2g + 2); trade; X (now seat 2g + 2) -> slot g + 1; barrier;
//                   slot g -> X (seat 2g).                          [slot G is never written and stays zero; the last
//                   group's odd "rotation" with that zero column is the fixed (c, s) = (0, 1): its column moves over]
// Squared norms and normalisation defects travel with their columns.  After an odd number of sweeps the columns are
// written back in reversed order (the zero column that pads an odd order then sits where it started and is dropped).
// ---------------------------------------------------------------------------
template <int H>
__device__ __forceinline__ int rotate_regs(f32x2 (&x)[H], f32x2 (&y)[H], float& a, float& b, float& dx, float& dy,
                                           float tol, float null2, float& norm2_max, bool forced) {
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < H; ++i) acc = __builtin_elementwise_fma(x[i], y[i], acc);
    const float g = pair_allsum<4>(acc.x + acc.y);
    norm2_max = fmaxf(norm2_max, fmaxf(a, b));
    Rot rot{1.f, 0.f, 0.f, 0.f, false, false};
    if (fminf(a, b) > null2) rot = make_rotation(a, b, g, tol);
    float c = rot.c, s = rot.s;
    if (forced) { c = 0.f; s = 1.f; }
    if (rot.apply || forced) {
        const f32x2 c2 = {c, c}, s2 = {s, s};
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const f32x2 xi = x[i], yi = y[i];
            x[i] = __builtin_elementwise_fma(c2, xi, -(s2 * yi));
            y[i] = __builtin_elementwise_fma(s2, xi, c2 * yi);
        }
    }
    if (forced) {
        b = a; a = 0.f; dy = dx; dx = 0.f;
    } else if (rot.apply) {
        a = fmaxf(a - rot.t * g, 0.f);
        b = b + rot.t * g;
        dx += rot.h;
        dy += rot.h;
    }
    return rot.strong ? 3 : (rot.apply ? 1 : 0);
}

template <int EPL>
__global__ void __launch_bounds__(512, 4) jacobi_oe_kernel(float* __restrict__ W, long batch_stride, int rows, int n,
                                                           int max_sweeps, float tol, float* __restrict__ colnorm,
                                                           int colnorm_stride, int* __restrict__ sweeps_out) {
    static_assert(EPL % 2 == 0, "row chunks come in pairs");
    constexpr int H = EPL / 2;
    constexpr int LD = 4 * EPL + (((4 * EPL / 8) % 2 == 0) ? 8 : 0);   // 8 x odd: see jacobi_lds_kernel
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int m = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int n_even = (n + 1) & ~1, G = n_even / 2;
    float* mail_n2 = lds + (G + 1) * LD;
    float* mail_dev = mail_n2 + (G + 1);
    __shared__ int s_norm2_bits;
    float* Wm = W + (long)m * batch_stride;
    const int g = tid >> 2, gl = tid & 3;
    const bool active = g < G, last = g == G - 1;
    for (int idx = tid; idx < (G + 1) * (LD + 2); idx += nthr) lds[idx] = 0.f;
    if (tid == 0) s_norm2_bits = 0;

    f32x2 x[H], y[H];
    float a = 0.f, b = 0.f, dx = 0.f, dy = 0.f;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const int r = 2 * gl + 8 * i;
        const int cx = 2 * g, cy = 2 * g + 1;
        x[i].x = (active && cx < n && r < rows) ? Wm[(long)cx * rows + r] : 0.f;
        x[i].y = (active && cx < n && r + 1 < rows) ? Wm[(long)cx * rows + r + 1] : 0.f;
        y[i].x = (active && cy < n && r < rows) ? Wm[(long)cy * rows + r] : 0.f;
        y[i].y = (active && cy < n && r + 1 < rows) ? Wm[(long)cy * rows + r + 1] : 0.f;
    }
    float* slot_lo = lds + (active ? g : 0) * LD + 2 * gl;           // slot g
    float* slot_hi = lds + (active ? g + 1 : 0) * LD + 2 * gl;       // slot g + 1
    __syncthreads();

    float null2 = 0.f;
    int sweep = 0;
    for (; sweep < max_sweeps && n > 1; ++sweep) {
        int rotated = 0;
        float norm2_max = 0.f;
        if (active) {       // the cached norms, refreshed from the data once per sweep
            f32x2 sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < H; ++i) {
                sa = __builtin_elementwise_fma(x[i], x[i], sa);
                sb = __builtin_elementwise_fma(y[i], y[i], sb);
            }
            a = pair_allsum<4>(sa.x + sa.y);
            b = pair_allsum<4>(sb.x + sb.y);
        }
        for (int dr = 0; dr < G; ++dr) {
            if (active) {
                rotated |= rotate_regs<H>(x, y, a, b, dx, dy, tol, null2, norm2_max, false);
#pragma unroll
                for (int i = 0; i < H; ++i) *(f32x2*)(slot_lo + 8 * i) = y[i];
                if (gl == 0) { mail_n2[g] = b; mail_dev[g] = dy; }
            }
            __syncthreads();
            if (active) {
#pragma unroll
                for (int i = 0; i < H; ++i) y[i] = *(const f32x2*)(slot_hi + 8 * i);
                b = mail_n2[g + 1];
                dy = mail_dev[g + 1];
                rotated |= rotate_regs<H>(x, y, a, b, dx, dy, tol, null2, norm2_max, last);
                if (!last) {
#pragma unroll
                    for (int i = 0; i < H; ++i) *(f32x2*)(slot_hi + 8 * i) = x[i];
                    if (gl == 0) { mail_n2[g + 1] = a; mail_dev[g + 1] = dx; }
                }
            }
            __syncthreads();
            if (active) {
#pragma unroll
                for (int i = 0; i < H; ++i) x[i] = *(const f32x2*)(slot_lo + 8 * i);
                a = mail_n2[g];
                dx = mail_dev[g];
            }
        }
        if (active && gl == 0 && norm2_max > 0.f) atomicMax(&s_norm2_bits, __float_as_int(norm2_max));
        if (!__syncthreads_or(rotated & 2)) {       // the stopping rule of jacobi_lds_kernel
            ++sweep;
            break;
        }
        null2 = kNull2 * __int_as_float(s_norm2_bits);
    }
    if (sweeps_out && tid == 0) sweeps_out[m] = sweep;
    if (!active) return;

    // exact norms + write back with the accumulated defects folded in; the line is reversed after an odd number of sweeps
    f32x2 sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < H; ++i) {
        sa = __builtin_elementwise_fma(x[i], x[i], sa);
        sb = __builtin_elementwise_fma(y[i], y[i], sb);
    }
    const float na = sqrtf(pair_allsum<4>(sa.x + sa.y)), nb = sqrtf(pair_allsum<4>(sb.x + sb.y));
    const bool rev = sweep & 1;
    const int cx = rev ? n_even - 1 - 2 * g : 2 * g, cy = rev ? n_even - 2 - 2 * g : 2 * g + 1;
    if (gl == 0) {
        if (cx < n) colnorm[(long)m * colnorm_stride + cx] = fmaf(na, dx, na);
        if (cy < n) colnorm[(long)m * colnorm_stride + cy] = fmaf(nb, dy, nb);
    }
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const int r = 2 * gl + 8 * i;
        if (cx < n && r < rows) Wm[(long)cx * rows + r] = fmaf(x[i].x, dx, x[i].x);
        if (cx < n && r + 1 < rows) Wm[(long)cx * rows + r + 1] = fmaf(x[i].y, dx, x[i].y);
        if (cy < n && r < rows) Wm[(long)cy * rows + r] = fmaf(y[i].x, dy, y[i].x);
        if (cy < n && r + 1 < rows) Wm[(long)cy * rows + r + 1] = fmaf(y[i].y, dy, y[i].y);
    }
}

// ---------------------------------------------------------------------------
// Two-pass solver for stacked cores [M; L_b] (2n x n) whose 2n rows do not fit LDS while n rows do (n = 144 at cfg-5,
// 196 at cfg-4: the block solver above moves the matrix through L2 / HBM n_blocks - 1 times per sweep).
//   pass 1, jacobi_top_logged_kernel: the LDS-resident solver on the top n x n alone, every pair-step's (c, s) written
//           to a log (8 bytes per pair-step; identity where no rotation was applied), plus the accumulated
//           normalisation defects and the number of sweeps.
//   pass 2, jacobi_apply_log_kernel: the riding rows in LDS, the log replayed on them -- no dot products, no
//           reductions, the same rotations in the same order, so the result is what the one-kernel form computes.
// 4 lanes per column pair; column stride LPP EPL (+ 8) = 8 x odd (see jacobi_lds_kernel).
// Log entry of (sweep s, round r, pair seat t): ((s (n_even - 1) + r) n_even / 2 + t).
// ---------------------------------------------------------------------------
template <int EPL>
struct TwoPassShape {
    static constexpr int LPP = 4;
    static constexpr int LD = LPP * EPL + (((LPP * EPL / 8) % 2 == 0) ? 8 : 0);
    static_assert((LD / 8) % 2 == 1 && LD % 8 == 0, "stride must come out as 8 x odd");
};

template <int EPL>
__global__ void __launch_bounds__(512) jacobi_top_logged_kernel(float* __restrict__ W, long batch_stride, int ld, int n,
                                                                int max_sweeps, float tol, float* __restrict__ colnorm,
                                                                int colnorm_stride, f32x2* __restrict__ rotlog,
                                                                long log_stride, float* __restrict__ dev_out,
                                                                int* __restrict__ sweeps_out) {
    constexpr int LPP = TwoPassShape<EPL>::LPP, LD = TwoPassShape<EPL>::LD;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int m = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    float* Wm = W + (long)m * batch_stride;
    f32x2* logm = rotlog + (long)m * log_stride;
    const int n_even = (n + 1) & ~1;
    for (int idx = tid; idx < n_even * LD; idx += nthr) lds[idx] = 0.f;
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nthr) {
        const int c = idx / n, r = idx - c * n;
        lds[c * LD + r] = Wm[(long)c * ld + r];
    }
    __syncthreads();
    const int groups = nthr / LPP, grp = tid / LPP, gl = tid % LPP;
    float* n2 = lds + n_even * LD;
    float* dev = n2 + n_even;
    __shared__ int s_norm2_bits;
    if (tid == 0) s_norm2_bits = 0;
    for (int c = tid; c < n_even; c += nthr) dev[c] = 0.f;
    float null2 = 0.f;
    int sweep = 0;
    const int pairs = n_even / 2;
    for (; sweep < max_sweeps && n > 1; ++sweep) {
        int rotated = 0;
        float norm2_max = 0.f;
        for (int c = grp; c < n_even; c += groups) {
            const float v = column_norm2<EPL, LPP>(lds + c * LD, gl);
            if (gl == 0) n2[c] = v;
        }
        __syncthreads();
        for (int r = 0; r < n_even - 1; ++r) {
            f32x2* logr = logm + ((long)sweep * (n_even - 1) + r) * pairs;
            for (int t = grp; t < pairs; t += groups) {
                int p, q;
                rr_pair(n_even, r, t, p, q);
                if (p >= n || q >= n) {
                    if (gl == 0) logr[t] = f32x2{1.f, 0.f};
                    continue;
                }
                if (p > q) { const int tmp = p; p = q; q = tmp; }
                rotated |= rotate_pair<EPL, EPL, LPP, true>(lds + p * LD, lds + q * LD, gl, tol, null2, norm2_max, n2 + p,
                                                            n2 + q, dev + p, dev + q, logr + t);
            }
            __syncthreads();
        }
        if (gl == 0 && norm2_max > 0.f) atomicMax(&s_norm2_bits, __float_as_int(norm2_max));
        if (!__syncthreads_or(rotated & 2)) {
            ++sweep;
            break;
        }
        null2 = kNull2 * __int_as_float(s_norm2_bits);
    }
    if (tid == 0) sweeps_out[m] = sweep;
    for (int c = grp; c < n; c += groups) {
        const float a = column_norm2<EPL, LPP>(lds + c * LD, gl);
        if (gl == 0) {
            const float nv = sqrtf(a);
            colnorm[(long)m * colnorm_stride + c] = fmaf(nv, dev[c], nv);
            dev_out[(long)m * n + c] = dev[c];
        }
    }
    for (int idx = tid; idx < n * n; idx += nthr) {
        const int c = idx / n, r = idx - c * n;
        const float v = lds[c * LD + r];
        Wm[(long)c * ld + r] = fmaf(v, dev[c], v);
    }
}

// blockIdx.y = chunk of 4 EPL riding rows: rows transform independently under column rotations, so a core's riding
// block is cut into row chunks -- several workgroups per CU whose load / rotate / store phases overlap (one workgroup
// per core with all n rows ran at half the LDS rate: every wave in the same phase between two barriers).
template <int EPL>
__global__ void __launch_bounds__(512) jacobi_apply_log_kernel(float* __restrict__ W, long batch_stride, int ld, int row0,
                                                               int n, const f32x2* __restrict__ rotlog, long log_stride,
                                                               const float* __restrict__ dev_in,
                                                               const int* __restrict__ sweeps) {
    constexpr int LPP = TwoPassShape<EPL>::LPP, LD = TwoPassShape<EPL>::LD, H = EPL / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int m = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int r_lo = blockIdx.y * LPP * EPL, rows = n - r_lo < LPP * EPL ? n - r_lo : LPP * EPL;
    float* Wm = W + (long)m * batch_stride + row0 + r_lo;
    const f32x2* logm = rotlog + (long)m * log_stride;
    const int n_even = (n + 1) & ~1, pairs = n_even / 2;
    for (int idx = tid; idx < n_even * LD; idx += nthr) lds[idx] = 0.f;
    __syncthreads();
    for (int idx = tid; idx < n * rows; idx += nthr) {
        const int c = idx / rows, r = idx - c * rows;
        lds[c * LD + r] = Wm[(long)c * ld + r];
    }
    __syncthreads();
    const int grp = tid / LPP, gl = tid % LPP;          // one pair seat per lane group: nthr >= LPP * pairs
    const long steps = (long)sweeps[m] * (n_even - 1);
    const bool seat = grp < pairs;
    const f32x2 ident = {1.f, 0.f};
    // the (c, s) of this seat, two rounds ahead of their use (the log comes from L2 / HBM)
    f32x2 cs0 = seat && steps > 0 ? logm[grp] : ident;
    f32x2 cs1 = seat && steps > 1 ? logm[pairs + grp] : ident;
    for (long st = 0; st < steps; ++st) {
        const f32x2 cs2 = seat && st + 2 < steps ? logm[(st + 2) * pairs + grp] : ident;
        const int r = (int)(st % (n_even - 1));
        if (seat && (cs0.y != 0.f || cs0.x != 1.f)) {
            int p, q;
            rr_pair(n_even, r, grp, p, q);
            if (p > q) { const int tmp = p; p = q; q = tmp; }
            float* cp = lds + p * LD;
            float* cq = lds + q * LD;
            const f32x2 c2 = {cs0.x, cs0.x}, s2 = {cs0.y, cs0.y};
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const f32x2 xr = *(const f32x2*)(cp + 2 * gl + 2 * LPP * i);
                const f32x2 yr = *(const f32x2*)(cq + 2 * gl + 2 * LPP * i);
                *(f32x2*)(cp + 2 * gl + 2 * LPP * i) = __builtin_elementwise_fma(c2, xr, -(s2 * yr));
                *(f32x2*)(cq + 2 * gl + 2 * LPP * i) = __builtin_elementwise_fma(s2, xr, c2 * yr);
            }
        }
        __syncthreads();
        cs0 = cs1;
        cs1 = cs2;
    }
    const float* dv = dev_in + (long)m * n;
    for (int idx = tid; idx < n * rows; idx += nthr) {
        const int c = idx / rows, r = idx - c * rows;
        const float v = lds[c * LD + r];
        Wm[(long)c * ld + r] = fmaf(v, dv[c], v);
    }
}

// ---------------------------------------------------------------------------
// Block solver: one launch = one round-robin round over column blocks of BW columns.
// grid = (nblk/2, batch), block = LPP * BW threads: LPP lanes per column pair, BW pairs at a time.  The panel of 2 BW
// columns is staged in LDS; what a sweep costs is the trips of the matrix through L2 / HBM -- nblk - 1 rounds, each
// reading and writing every column once -- so the host picks the widest panel that fits (few lanes per pair when the
// columns are short: 288 rows x 112 columns at cfg-5 is 3 rounds per sweep where one wave per pair and 32 columns
// took 9).  Columns are padded to LPP * EPL rows.  flags[m * max_sweeps + s] != 0 <=> sweep s applied a rotation above
// sqrt(tol), i.e. another sweep follows.
// ---------------------------------------------------------------------------
template <int EPL, int DOT, int LPP>
__global__ void __launch_bounds__(1024) jacobi_block_round_kernel(float* __restrict__ W, long batch_stride,
                                                                  int rows_dot, int rows_tot, int n, int nblk, int BW,
                                                                  int round, int sweep, int max_sweeps,
                                                                  float tol, int* __restrict__ flags,
                                                                  int* __restrict__ norm2_bits) {
    constexpr int LD = LPP * EPL + 2;   // even: 8-byte aligned columns for the paired accesses
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int m = blockIdx.y;
    if (sweep > 0 && flags[m * max_sweeps + sweep - 1] == 0) return;  // converged in an earlier sweep
    const float null2 = sweep > 0 ? kNull2 * __int_as_float(norm2_bits[m * max_sweeps + sweep - 1]) : 0.f;
    float norm2_max = 0.f;
    int bi, bj;
    rr_pair(nblk, round, blockIdx.x, bi, bj);
    if (bi > bj) { const int t = bi; bi = bj; bj = t; }
    float* Wm = W + (long)m * batch_stride;
    const int tid = threadIdx.x;
    const int grp = tid / LPP, gl = tid % LPP;

    // stage the 2*BW columns (zero-padded rows, zero columns >= n).  The panel is tiny next to the launch's
    // latency budget, so what matters is loads in flight: every thread issues several independent 16-byte
    // loads before the first LDS write.
    const int NT = LPP * BW;
    const bool padded = rows_tot != LPP * EPL || (bi + 1) * BW > n || (bj + 1) * BW > n;
    if (padded) {
        for (int idx = tid; idx < 2 * BW * LD; idx += NT) lds[idx] = 0.f;
        __syncthreads();
    }
    const bool vec4 = (rows_tot & 3) == 0 && (batch_stride & 3) == 0 && (((uintptr_t)W) & 15) == 0;
    if (vec4) {
        const int q4 = rows_tot >> 2, total4 = 2 * BW * q4;
#pragma unroll 4
        for (int idx = tid; idx < total4; idx += NT) {
            const int lc = idx / q4, r = (idx - lc * q4) << 2;
            const int gc = (lc < BW ? bi * BW + lc : bj * BW + (lc - BW));
            if (gc < n) {
                const float4 v = *(const float4*)(Wm + (long)gc * rows_tot + r);
                float* col = lds + lc * LD;
                col[lds_row<EPL, DOT, LPP>(r, rows_dot)] = v.x;
                col[lds_row<EPL, DOT, LPP>(r + 1, rows_dot)] = v.y;
                col[lds_row<EPL, DOT, LPP>(r + 2, rows_dot)] = v.z;
                col[lds_row<EPL, DOT, LPP>(r + 3, rows_dot)] = v.w;
            }
        }
    } else {
#pragma unroll 4
        for (int idx = tid; idx < 2 * BW * rows_tot; idx += NT) {
            const int lc = idx / rows_tot, r = idx - lc * rows_tot;
            const int gc = (lc < BW ? bi * BW + lc : bj * BW + (lc - BW));
            if (gc < n) lds[lc * LD + lds_row<EPL, DOT, LPP>(r, rows_dot)] = Wm[(long)gc * rows_tot + r];
        }
    }
    float* n2 = lds + 2 * BW * LD;      // cached squared column norms
    float* dev = n2 + 2 * BW;           // accumulated normalisation defects
    __syncthreads();
    for (int c = grp; c < 2 * BW; c += BW) {
        const float v = column_norm2<DOT, LPP>(lds + c * LD, gl);
        if (gl == 0) { n2[c] = v; dev[c] = 0.f; }
    }
    __syncthreads();

    int rotated = 0;
    if (round == 0) {
        // pairs inside block I (rows 0..BW/2-1) and inside block J (rows BW/2..BW-1)
        const int half = grp / (BW / 2), t = grp % (BW / 2), base = half * BW;
        for (int r = 0; r < BW - 1; ++r) {
            int p, q;
            rr_pair(BW, r, t, p, q);
            if (p > q) { const int tmp = p; p = q; q = tmp; }
            const int gp = (half ? bj : bi) * BW + p, gq = (half ? bj : bi) * BW + q;
            if (gp < n && gq < n)
                rotated |= rotate_pair<EPL, DOT, LPP>(lds + (base + p) * LD, lds + (base + q) * LD, gl, tol, null2, norm2_max,
                                                 n2 + base + p, n2 + base + q, dev + base + p, dev + base + q);
            __syncthreads();
        }
    }
    // cross pairs: column `grp` of I with column (grp + r) % BW of J
    for (int r = 0; r < BW; ++r) {
        const int p = grp, q = (grp + r) % BW;
        const int gp = bi * BW + p, gq = bj * BW + q;
        if (gp < n && gq < n)
            rotated |= rotate_pair<EPL, DOT, LPP>(lds + p * LD, lds + (BW + q) * LD, gl, tol, null2, norm2_max, n2 + p,
                                             n2 + BW + q, dev + p, dev + BW + q);
        __syncthreads();
    }
    if (gl == 0 && norm2_max > 0.f) atomicMax(&norm2_bits[m * max_sweeps + sweep], __float_as_int(norm2_max));
    // the stopping rule of jacobi_lds_kernel: a sweep whose rotations all stayed below sqrt(tol) leaves O(tol) behind
    // (quadratic convergence) and is the last one -- no extra sweep that would only verify it
    if (__syncthreads_or(rotated & 2) && tid == 0) atomicOr(&flags[m * max_sweeps + sweep], 1);

    if (vec4) {
        const int q4 = rows_tot >> 2, total4 = 2 * BW * q4;
#pragma unroll 4
        for (int idx = tid; idx < total4; idx += NT) {
            const int lc = idx / q4, r = (idx - lc * q4) << 2;
            const int gc = (lc < BW ? bi * BW + lc : bj * BW + (lc - BW));
            if (gc < n) {
                const float* col = lds + lc * LD;
                const float dv = dev[lc];
                float4 v;
                v.x = col[lds_row<EPL, DOT, LPP>(r, rows_dot)];
                v.y = col[lds_row<EPL, DOT, LPP>(r + 1, rows_dot)];
                v.z = col[lds_row<EPL, DOT, LPP>(r + 2, rows_dot)];
                v.w = col[lds_row<EPL, DOT, LPP>(r + 3, rows_dot)];
                v.x = fmaf(v.x, dv, v.x); v.y = fmaf(v.y, dv, v.y); v.z = fmaf(v.z, dv, v.z); v.w = fmaf(v.w, dv, v.w);
                *(float4*)(Wm + (long)gc * rows_tot + r) = v;
            }
        }
    } else {
#pragma unroll 4
        for (int idx = tid; idx < 2 * BW * rows_tot; idx += NT) {
            const int lc = idx / rows_tot, r = idx - lc * rows_tot;
            const int gc = (lc < BW ? bi * BW + lc : bj * BW + (lc - BW));
            if (gc < n) {
                const float v = lds[lc * LD + lds_row<EPL, DOT, LPP>(r, rows_dot)];
                Wm[(long)gc * rows_tot + r] = fmaf(v, dev[lc], v);
            }
        }
    }
}

// sweeps the block solver ran on matrix m: those that asked for another one, plus the last
__global__ void block_sweeps_kernel(const int* __restrict__ flags, int batch, int max_sweeps, int* __restrict__ sweeps_out) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= batch) return;
    int s = 0;
    while (s < max_sweeps && flags[m * max_sweeps + s] != 0) ++s;
    sweeps_out[m] = s < max_sweeps ? s + 1 : max_sweeps;
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


extern "C" {

// Largest LDS the resident solver may use (bytes); leaves room for the runtime.
#define BASD_JACOBI_LDS_LIMIT (156 * 1024)

int basd_jacobi_workspace_ints(int batch, int max_sweeps) { return 2 * batch * max_sweeps; }

// 1 when basd_jacobi_onesided solves a batch (>= 128) of plain n x n matrices in LDS with 4 lanes per column pair
// (orders 40..144): the transposed Procrustes cores then need no riding rows (basd_procrustes_forward_fused).
// column stride (floats) of the plain 4-lane solver for e elements per lane: 4 e, + 8 unless that is 8 x odd already
static inline size_t plain4_ld(int e) { return 4 * (size_t)e + (((4 * e / 8) % 2 == 0) ? 8 : 0); }
int basd_jacobi_plain4_fits(int n) {
    const int n_even = (n + 1) & ~1;
    if (n_even < 40 || n > 200) return 0;
    static const int p_epl[] = {8, 16, 24, 36, 50};
    for (int e : p_epl)
        if (4 * e >= n) return (size_t)n_even * (plain4_ld(e) + 2) * sizeof(float) <= BASD_JACOBI_LDS_LIMIT;
    return 0;
}

// 1 when basd_jacobi_onesided keeps square matrices of order n in LDS with 16 lanes per column pair: the only form that
// takes per-matrix orders (n_arr), i.e. the principal-angle matrices of basd_selector_tail / basd_selector_chain_tail.
int basd_jacobi_lds_square_fits(int n) {
    if (n < 1) return 0;
    const int n_even = (n + 1) & ~1, epl = (n + 15) / 16;
    static const int lds_epl[] = {2, 4, 6, 8, 12, 16, 20};
    for (int e : lds_epl)
        if (e >= epl) return (size_t)n_even * (16 * e + 4) * sizeof(float) <= BASD_JACOBI_LDS_LIMIT;
    return 0;
}

// Test / tuning hook: lanes per column pair of the LDS-resident solver -- 0 = automatic, 4 / 8 / 16 = forced where the
// shape allows (4 and 8: stacked matrices only).  Process-wide.
static int g_jacobi_lanes = 0;
int basd_jacobi_tuning(int lanes_per_pair) {
    if (lanes_per_pair != 0 && lanes_per_pair != 4 && lanes_per_pair != 8 && lanes_per_pair != 16) return BASD_EINVAL;
    g_jacobi_lanes = lanes_per_pair;
    return BASD_OK;
}

// Test / tuning hook: ordering of the plain batched solver -- 1 (default) = odd-even, columns in registers
// (jacobi_oe_kernel); 0 = round-robin through LDS (jacobi_lds_kernel<E, E, 4>).  Process-wide.
static int g_jacobi_ordering = 1;
int basd_jacobi_ordering(int odd_even) {
    if (odd_even != 0 && odd_even != 1) return BASD_EINVAL;
    g_jacobi_ordering = odd_even;
    return BASD_OK;
}

// One-sided Jacobi on `batch` column-major matrices (rows_tot x n, leading dim rows_tot).
//   n_arr (device, nullable): per-matrix order for square problems (rows = n_arr[m]); the
//   storage still uses rows_tot / batch_stride of the largest problem.
//   colnorm: (batch, colnorm_stride) column norms over the first rows_dot rows.
//   flags: device scratch of basd_jacobi_workspace_ints() ints (block path only, may be null
//   when the LDS path is taken).
//   tol_cos: stop once every pair has |cos| <= tol_cos in a full sweep (<= 0: eps * sqrt(rows_dot)).  Column
//   norms (singular / eigen values) are second-order accurate in it, vectors first-order.
int basd_jacobi_onesided(float* W, long batch_stride, int rows_dot, int rows_tot, int n, int batch,
                         const int* n_arr, float* colnorm, int colnorm_stride, int max_sweeps, float tol_cos,
                         int* flags, int* sweeps_out, hipStream_t stream) {
    BASD_CHECK_ARG(W && colnorm && rows_dot > 0 && rows_tot >= rows_dot && n > 0 && batch > 0 && max_sweeps > 0);
    // a pair counts as orthogonal when |cos| <= tol; <= 0 selects the round-off level eps * sqrt(rows)
    const float tol = tol_cos > 0.f ? tol_cos : 1.2e-7f * sqrtf((float)rows_dot);
    const int n_even = (n + 1) & ~1;
    // per-lane element counts of the zero-padded LDS columns (16 lanes per column pair)
    const bool stacked = rows_tot > rows_dot;
    const int dot16 = (rows_dot + 15) / 16, ride16 = (rows_tot - rows_dot + 15) / 16;
    const int half = dot16 > ride16 ? dot16 : ride16;          // stacked layouts: EPL = 2 * DOT
    const int epl = stacked ? 2 * half : dot16;
    BASD_CHECK_ARG(!(stacked && n_arr));

    // ---- LDS-resident, 8 lanes per column pair (stacked matrices) ----
    // automatic choice for large batches of stacked matrices (measured on MI355X, 1024 x (98 x 49): 0.87 ms with 16 lanes
    // per pair, 0.80 with 8, 0.70 with 4; 512 x (72 x 36): 0.45 / 0.36 / 0.41)
    int lanes = g_jacobi_lanes;
    if (lanes == 0 && stacked && batch >= 256) lanes = n_even >= 40 ? 4 : (n_even >= 16 ? 8 : 16);
    if (stacked && n_even >= 8 && lanes == 8) {
        const int dot16 = (rows_dot + 15) / 16, ride16 = (rows_tot - rows_dot + 15) / 16;
        const int half16 = dot16 > ride16 ? dot16 : ride16;       // 16-row chunks per half; DOT = 2 * half16
        const int d8 = 2 * half16;
        const size_t lds8 = (size_t)n_even * (8 * 2 * d8 + 16 + 2) * sizeof(float);
        if (d8 <= 8 && lds8 <= BASD_JACOBI_LDS_LIMIT) {
            int threads = (((n_even / 2) * 8 + 63) / 64) * 64;
            if (threads > 1024) threads = 1024;
#define LAUNCH_LDS8(D)                                                                                               \
    do {                                                                                                             \
        if (lds8 > 48 * 1024)                                                                                        \
            (void)hipFuncSetAttribute((const void*)jacobi_lds_kernel<2 * (D), D, 8>,                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BASD_JACOBI_LDS_LIMIT);            \
        jacobi_lds_kernel<2 * (D), D, 8><<<batch, threads, lds8, stream>>>(W, batch_stride, rows_dot, rows_tot, n,   \
                                                                           n_arr, max_sweeps, tol, colnorm,          \
                                                                           colnorm_stride, sweeps_out);              \
    } while (0)
            switch (d8) {
                case 2: LAUNCH_LDS8(2); break;
                case 4: LAUNCH_LDS8(4); break;
                case 6: LAUNCH_LDS8(6); break;
                default: LAUNCH_LDS8(8); break;
            }
#undef LAUNCH_LDS8
            BASD_RETURN_LAST();
        }
    }

    // ---- LDS-resident, throughput shape: large batches of small stacked matrices, 4 lanes per column pair ----
    if (stacked && n_even >= 8 && lanes == 4) {
        const int dot8 = (rows_dot + 7) / 8, ride8 = (rows_tot - rows_dot + 7) / 8;
        const int half8 = dot8 > ride8 ? dot8 : ride8;            // 8-row chunks per half; DOT = 2 * half8 elements / lane
        static const int q_dot[] = {8, 12, 14, 16};
        int d4 = 0;
        for (int dq : q_dot)
            if (dq >= 2 * half8) { d4 = dq; break; }
        const size_t lds4 = (size_t)n_even * (4 * 2 * d4 + 8 + 2) * sizeof(float);
        if (d4 && lds4 <= BASD_JACOBI_LDS_LIMIT) {
            int threads = (((n_even / 2) * 4 + 63) / 64) * 64;
            if (threads > 1024) threads = 1024;
#define LAUNCH_LDS4(D)                                                                                               \
    do {                                                                                                             \
        if (lds4 > 48 * 1024)                                                                                        \
            (void)hipFuncSetAttribute((const void*)jacobi_lds_kernel<2 * (D), D, 4>,                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BASD_JACOBI_LDS_LIMIT);            \
        jacobi_lds_kernel<2 * (D), D, 4><<<batch, threads, lds4, stream>>>(W, batch_stride, rows_dot, rows_tot, n,   \
                                                                           n_arr, max_sweeps, tol, colnorm,          \
                                                                           colnorm_stride, sweeps_out);              \
    } while (0)
            switch (d4) {
                case 8: LAUNCH_LDS4(8); break;
                case 12: LAUNCH_LDS4(12); break;
                case 14: LAUNCH_LDS4(14); break;
                default: LAUNCH_LDS4(16); break;
            }
#undef LAUNCH_LDS4
            BASD_RETURN_LAST();
        }
    }

    // ---- (round-robin ordering only) the same with 8 lanes per pair for long columns at about one matrix per CU (cfg-5:
    // 256 cores of 144 x 144): what counts there is the latency of one round, and half the elements per lane shorten it.
    // The odd-even solver below beats it there as well (cfg-5: 6.54 -> 5.9 ms per step) ----
    if (g_jacobi_ordering == 0 && !stacked && !n_arr && rows_tot > 96 && rows_tot <= 160 && n_even <= 160 &&
        (lanes == 8 || (lanes == 0 && batch >= 128 && batch <= 320))) {
        const size_t lds8p = (size_t)n_even * (8 * 20 + 16 + 2) * sizeof(float);
        if (lds8p <= BASD_JACOBI_LDS_LIMIT) {
            int threads = (((n_even / 2) * 8 + 63) / 64) * 64;
            if (threads > 1024) threads = 1024;
            (void)hipFuncSetAttribute((const void*)jacobi_lds_kernel<20, 20, 8>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BASD_JACOBI_LDS_LIMIT);
            jacobi_lds_kernel<20, 20, 8><<<batch, threads, lds8p, stream>>>(W, batch_stride, rows_dot, rows_tot, n, n_arr,
                                                                            max_sweeps, tol, colnorm, colnorm_stride,
                                                                            sweeps_out);
            BASD_RETURN_LAST();
        }
    }

    // ---- register-resident, odd-even ordering: plain matrices in batches (the transposed Procrustes cores) ----
    if (g_jacobi_ordering == 1 && !stacked && !n_arr && n_even >= 8 && rows_tot == rows_dot &&
        (lanes == 0 || lanes == 4) && (g_jacobi_lanes == 4 || (batch >= 128 && n_even >= 40) || n_even >= 96)) {
        static const int o_epl[] = {8, 16, 24, 36, 50};
        int e4 = 0;
        for (int e : o_epl)
            if (4 * e >= rows_tot) { e4 = e; break; }
        const int G = n_even / 2;
        const size_t lds_oe = (size_t)(G + 1) * (plain4_ld(e4) + 2) * sizeof(float);
        const int threads = ((G * 4 + 63) / 64) * 64;
        if (e4 && threads <= 512 && lds_oe <= BASD_JACOBI_LDS_LIMIT) {
#define LAUNCH_OE(E)                                                                                                  \
    do {                                                                                                              \
        if (lds_oe > 48 * 1024)                                                                                       \
            (void)hipFuncSetAttribute((const void*)jacobi_oe_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                      BASD_JACOBI_LDS_LIMIT);                                                         \
        jacobi_oe_kernel<E><<<batch, threads, lds_oe, stream>>>(W, batch_stride, rows_tot, n, max_sweeps, tol, colnorm, \
                                                                colnorm_stride, sweeps_out);                          \
    } while (0)
            switch (e4) {
                case 8: LAUNCH_OE(8); break;
                case 16: LAUNCH_OE(16); break;
                case 24: LAUNCH_OE(24); break;
                case 36: LAUNCH_OE(36); break;
                default: LAUNCH_OE(50); break;
            }
#undef LAUNCH_OE
            BASD_RETURN_LAST();
        }
    }

    // ---- LDS-resident, plain square-ish matrices in large batches (the transposed Procrustes cores: no riding rows),
    // 4 lanes per column pair.  Elements per lane: multiples of 4 (column stride 4 EPL + 8 = 8 x odd). ----
    if (!stacked && !n_arr && n_even >= 8 && (lanes == 4 || (lanes == 0 && batch >= 128 && n_even >= 40))) {
        static const int p_epl[] = {8, 16, 24, 36, 50};
        int e4 = 0;
        for (int e : p_epl)
            if (4 * e >= rows_tot) { e4 = e; break; }
        const size_t lds4 = (size_t)n_even * (plain4_ld(e4) + 2) * sizeof(float);
        if (e4 && lds4 <= BASD_JACOBI_LDS_LIMIT) {
            int threads = (((n_even / 2) * 4 + 63) / 64) * 64;
            if (threads > 1024) threads = 1024;
#define LAUNCH_P4(E)                                                                                                  \
    do {                                                                                                              \
        if (lds4 > 48 * 1024)                                                                                         \
            (void)hipFuncSetAttribute((const void*)jacobi_lds_kernel<E, E, 4>,                                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BASD_JACOBI_LDS_LIMIT);             \
        jacobi_lds_kernel<E, E, 4><<<batch, threads, lds4, stream>>>(W, batch_stride, rows_dot, rows_tot, n, n_arr,   \
                                                                     max_sweeps, tol, colnorm, colnorm_stride,        \
                                                                     sweeps_out);                                     \
    } while (0)
            switch (e4) {
                case 8: LAUNCH_P4(8); break;
                case 16: LAUNCH_P4(16); break;
                case 24: LAUNCH_P4(24); break;
                case 36: LAUNCH_P4(36); break;
                default: LAUNCH_P4(50); break;
            }
#undef LAUNCH_P4
            BASD_RETURN_LAST();
        }
    }

    // ---- LDS-resident: the whole (padded) matrix in one CU ----
    static const int lds_epl[] = {2, 4, 6, 8, 12, 16, 20};
    int e_lds = 0;
    for (int e : lds_epl)
        if (e >= epl && (!stacked || e % 4 == 0)) { e_lds = e; break; }
    if (e_lds && (size_t)n_even * (16 * e_lds + 4) * sizeof(float) <= BASD_JACOBI_LDS_LIMIT) {
        const size_t lds_bytes = (size_t)n_even * (16 * e_lds + 4) * sizeof(float);
        // every pair of a round-robin round in flight at once when it fits the block
        int threads = (((n_even / 2) * 16 + 63) / 64) * 64;
        if (threads > 1024) threads = 1024;
#define LAUNCH_LDS(E, D)                                                                                          \
    do {                                                                                                          \
        if (lds_bytes > 48 * 1024)                                                                                \
            (void)hipFuncSetAttribute((const void*)jacobi_lds_kernel<E, D>,                                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BASD_JACOBI_LDS_LIMIT);         \
        jacobi_lds_kernel<E, D><<<batch, threads, lds_bytes, stream>>>(W, batch_stride, rows_dot, rows_tot, n,    \
                                                                        n_arr, max_sweeps, tol, colnorm,           \
                                                                        colnorm_stride, sweeps_out);               \
    } while (0)
#define LAUNCH_LDS_E(E)                                      \
    do {                                                     \
        if (stacked) {                                       \
            if constexpr ((E) % 4 == 0) LAUNCH_LDS(E, (E) / 2); \
        } else {                                             \
            LAUNCH_LDS(E, E);                                \
        }                                                    \
    } while (0)
        switch (e_lds) {
            case 2: LAUNCH_LDS_E(2); break;
            case 4: LAUNCH_LDS_E(4); break;
            case 6: LAUNCH_LDS_E(6); break;
            case 8: LAUNCH_LDS_E(8); break;
            case 12: LAUNCH_LDS_E(12); break;
            case 16: LAUNCH_LDS_E(16); break;
            default: LAUNCH_LDS_E(20); break;
        }
#undef LAUNCH_LDS_E
#undef LAUNCH_LDS
        BASD_RETURN_LAST();
    }

    // ---- block path: one launch per round-robin round over blocks of BW columns ----
    BASD_CHECK_ARG(n_arr == nullptr && flags != nullptr);
    // Panel shape: lanes per column pair (64 / 32 / 16), elements per lane, columns per block.  A sweep is nblk - 1
    // trips of the whole matrix through L2 / HBM, so: the widest panel that fits LDS; ties go to more lanes per pair.
    struct Shape { int lpp, epl, dot, bw, nblk; };
    Shape best{0, 0, 0, 0, 1 << 30};
    // instantiated element counts per lane: `dot` of the stacked shapes (EPL = 2 dot), EPL of the plain ones
    static const int st64[] = {2, 4, 6, 8, 10, 12, 16, 0}, pl64[] = {2, 4, 6, 8, 12, 16, 0};
    static const int st32[] = {4, 8, 12, 16, 20, 0}, pl32[] = {4, 8, 12, 16, 20, 0};
    static const int st16[] = {8, 10, 14, 16, 20, 0}, pl16[] = {8, 10, 14, 16, 20, 28, 32, 0};
    static const int lpps[] = {64, 32, 16};
    for (int lpp : lpps) {
        const int cd = (rows_dot + 2 * lpp - 1) / (2 * lpp), cr = (rows_tot - rows_dot + 2 * lpp - 1) / (2 * lpp);
        const int need = 2 * (stacked && cr > cd ? cr : cd);
        const int* q = lpp == 64 ? (stacked ? st64 : pl64) : lpp == 32 ? (stacked ? st32 : pl32) : (stacked ? st16 : pl16);
        int sel = 0;
        for (; *q; ++q)
            if (*q >= need) { sel = *q; break; }
        if (!sel) continue;
        const int epl = stacked ? 2 * sel : sel;
        const size_t col_bytes = sizeof(float) * (size_t)(lpp * epl + 2);
        int bw = (int)((BASD_JACOBI_LDS_LIMIT - 64) / (2 * (col_bytes + 8)));      // + norm / defect per column
        if (bw > 1024 / lpp) bw = 1024 / lpp;
        if (bw > ((n + 1) & ~1)) bw = (n + 1) & ~1;
        bw &= ~1;
        if (bw < 2) continue;
        int nb = (n + bw - 1) / bw;
        nb = (nb + 1) & ~1;
        if (nb < best.nblk) best = Shape{lpp, epl, sel, bw, nb};
    }
    if (!best.lpp) return BASD_EUNSUPPORTED;                     // columns too long for a two-column panel in LDS
    const int BW = best.bw, nblk = best.nblk;
    const size_t panel_bytes = (size_t)2 * BW * (best.lpp * best.epl + 2 + 2) * sizeof(float);
    hipError_t err = hipMemsetAsync(flags, 0, sizeof(int) * (size_t)2 * batch * max_sweeps, stream);
    if (err != hipSuccess) return (int)err;
#define LAUNCH_BLOCK(E, D, LP)                                                                                   \
    do {                                                                                                         \
        if (panel_bytes > 48 * 1024)                                                                             \
            (void)hipFuncSetAttribute((const void*)jacobi_block_round_kernel<E, D, LP>,                          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BASD_JACOBI_LDS_LIMIT);        \
        for (int s = 0; s < max_sweeps; ++s)                                                                     \
            for (int r = 0; r < nblk - 1; ++r)                                                                   \
                jacobi_block_round_kernel<E, D, LP><<<dim3(nblk / 2, batch), LP * BW, panel_bytes, stream>>>(    \
                    W, batch_stride, rows_dot, rows_tot, n, nblk, BW, r, s, max_sweeps, tol, flags,              \
                    flags + (size_t)batch * max_sweeps);                                                         \
    } while (0)
#define LAUNCH_BLOCK_SEL(Q, LP)                      \
    do {                                             \
        if (stacked) LAUNCH_BLOCK(2 * (Q), Q, LP);   \
        else LAUNCH_BLOCK(Q, Q, LP);                 \
    } while (0)
    bool launched = true;
    if (best.lpp == 64) {
        switch (best.dot) {
            case 2: LAUNCH_BLOCK_SEL(2, 64); break;
            case 4: LAUNCH_BLOCK_SEL(4, 64); break;
            case 6: LAUNCH_BLOCK_SEL(6, 64); break;
            case 8: LAUNCH_BLOCK_SEL(8, 64); break;
            case 10: LAUNCH_BLOCK(20, 10, 64); break;      // stacked cores of 577..640 rows (576 tokens)
            case 12: LAUNCH_BLOCK_SEL(12, 64); break;
            case 16: LAUNCH_BLOCK_SEL(16, 64); break;
            default: launched = false; break;
        }
    } else if (best.lpp == 32) {
        switch (best.dot) {
            case 4: LAUNCH_BLOCK_SEL(4, 32); break;
            case 8: LAUNCH_BLOCK_SEL(8, 32); break;
            case 12: LAUNCH_BLOCK_SEL(12, 32); break;
            case 16: LAUNCH_BLOCK_SEL(16, 32); break;
            case 20: LAUNCH_BLOCK_SEL(20, 32); break;
            default: launched = false; break;
        }
    } else {
        switch (best.dot) {
            case 8: LAUNCH_BLOCK_SEL(8, 16); break;
            case 10: LAUNCH_BLOCK_SEL(10, 16); break;
            case 14: LAUNCH_BLOCK_SEL(14, 16); break;
            case 16: LAUNCH_BLOCK_SEL(16, 16); break;
            case 20: LAUNCH_BLOCK_SEL(20, 16); break;
            case 28: LAUNCH_BLOCK(28, 28, 16); break;
            case 32: LAUNCH_BLOCK(32, 32, 16); break;
            default: launched = false; break;
        }
    }
#undef LAUNCH_BLOCK_SEL
#undef LAUNCH_BLOCK
    if (!launched) return BASD_EUNSUPPORTED;
    if (sweeps_out) block_sweeps_kernel<<<(batch + 255) / 256, 256, 0, stream>>>(flags, batch, max_sweeps, sweeps_out);
    colnorm_kernel<<<dim3((n + 3) / 4, batch), 256, 0, stream>>>(W, batch_stride, rows_dot, rows_tot, n, colnorm,
                                                                 colnorm_stride);
    BASD_RETURN_LAST();
}

// Two-pass solver (see jacobi_top_logged_kernel) for stacked square cores [M; L_b] (2n x n, leading dimension 2n).
// Lane-element counts instantiated: n <= 4 EPL.
static int twopass_epl(int n) {
    static const int epls[] = {28, 36, 44, 50};
    const size_t n_even = (n + 1) & ~1;
    for (int e : epls)
        if (n <= 4 * e) {
            const size_t ld = 4 * e + (((4 * e / 8) % 2 == 0) ? 8 : 0);          // TwoPassShape<e>::LD
            return sizeof(float) * (n_even * ld + 2 * n_even) <= BASD_JACOBI_LDS_LIMIT ? e : 0;   // n <= 196
        }
    return 0;
}
static bool stacked_lds_fits(int n) {              // the one-kernel LDS solver of basd_jacobi_onesided for a 2n x n core
    const int n_even = (n + 1) & ~1, epl = 2 * ((n + 15) / 16);
    static const int lds_epl[] = {4, 8, 12, 16, 20};
    for (int e : lds_epl)
        if (e >= epl) return (size_t)n_even * (16 * e + 4) * sizeof(float) <= BASD_JACOBI_LDS_LIMIT;
    return false;
}
// Bytes of device workspace basd_jacobi_stacked_twopass needs (rotation log + defects + sweep counts); 0 when the
// shape is not one it covers (cores that fit the one-kernel LDS solver: n <= 128, or n > 196).
long basd_jacobi_twopass_workspace_bytes(int n, int batch, int max_sweeps) {
    if (n < 2 || batch <= 0 || max_sweeps <= 0 || stacked_lds_fits(n) || !twopass_epl(n)) return 0;
    const long n_even = (n + 1) & ~1;
    const long entries = (long)max_sweeps * (n_even - 1) * (n_even / 2);
    return (long)batch * (entries * 8 + (long)n * 4 + 4) + 64;
}
int basd_jacobi_stacked_twopass(float* W, long batch_stride, int n, int batch, float* colnorm, int colnorm_stride,
                                int max_sweeps, float tol_cos, void* workspace, int* sweeps_out, hipStream_t stream) {
    BASD_CHECK_ARG(W && colnorm && workspace && n > 1 && batch > 0 && max_sweeps > 0);
    BASD_CHECK_ARG(((uintptr_t)workspace & 7) == 0);
    const int epl = twopass_epl(n);
    if (!epl || stacked_lds_fits(n)) return BASD_EUNSUPPORTED;
    const float tol = tol_cos > 0.f ? tol_cos : 1.2e-7f * sqrtf((float)n);
    const long n_even = (n + 1) & ~1;
    const long entries = (long)max_sweeps * (n_even - 1) * (n_even / 2);
    f32x2* rotlog = (f32x2*)workspace;
    float* dev = (float*)(rotlog + (long)batch * entries);
    int* sweeps = (int*)(dev + (long)batch * n);
    int threads = (int)(((n_even / 2) * 4 + 63) / 64) * 64;
    BASD_CHECK_ARG(threads <= 512);
#define LAUNCH_2P(E)                                                                                                  \
    do {                                                                                                              \
        const size_t lds = sizeof(float) * ((size_t)n_even * TwoPassShape<E>::LD + 2 * (size_t)n_even);               \
        if (lds > BASD_JACOBI_LDS_LIMIT) return BASD_EUNSUPPORTED;                                                    \
        (void)hipFuncSetAttribute((const void*)jacobi_top_logged_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  BASD_JACOBI_LDS_LIMIT);                                                             \
        jacobi_top_logged_kernel<E><<<batch, threads, lds, stream>>>(W, batch_stride, 2 * n, n, max_sweeps, tol, colnorm, \
                                                                     colnorm_stride, rotlog, entries, dev, sweeps);   \
        if (n <= 160) {                                                                                               \
            const size_t lds2 = sizeof(float) * (size_t)n_even * TwoPassShape<10>::LD;                                \
            jacobi_apply_log_kernel<10><<<dim3(batch, (n + 39) / 40), threads, lds2, stream>>>(                       \
                W, batch_stride, 2 * n, n, n, rotlog, entries, dev, sweeps);                                          \
        } else {                                                                                                      \
            const size_t lds2 = sizeof(float) * (size_t)n_even * TwoPassShape<14>::LD;                                \
            jacobi_apply_log_kernel<14><<<dim3(batch, (n + 55) / 56), threads, lds2, stream>>>(                       \
                W, batch_stride, 2 * n, n, n, rotlog, entries, dev, sweeps);                                          \
        }                                                                                                             \
    } while (0)
    switch (epl) {
        case 28: LAUNCH_2P(28); break;
        case 36: LAUNCH_2P(36); break;
        case 44: LAUNCH_2P(44); break;
        default: LAUNCH_2P(50); break;
    }
#undef LAUNCH_2P
    if (sweeps_out) {
        hipError_t e = hipMemcpyAsync(sweeps_out, sweeps, sizeof(int) * (size_t)batch, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return (int)e;
    }
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
