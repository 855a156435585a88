// fp32-in / fp32-accumulate MFMA contractions for the BASD loss path.
//
//   reference call sites replaced:
//     tokens.reshape(-1, D_t) @ proj_t.T          src/losses/layer_selector.py:72, :135   (gemm_nt)
//     features.T @ features / M  (and x @ x.T)    src/losses/layer_selector.py:13, :15    (gemm_tn / gemm_nt)
//     centred z^T z feeding the thin SVD           src/losses/layer_selector.py:35-36, :90-92 (gemm_tn with mean)
//     U_s.T @ U_t                                  src/losses/layer_selector.py:99          (gemm_nt)
//
// Both kernels use v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 157 TFLOP/s peak
// on MI355X): a 128x128 block tile, 4 waves as 2x2, each wave 2x2 MFMA tiles.
//
//   gemm_nt : C[M,N] = A[M,K] * B[N,K]^T     K contiguous in both operands.
//             LDS image [row][k] padded to 36 floats, fragments by ds_read_b128:
//             one 16-byte read feeds four k-steps (lane half h takes k = 8g+4h+s,
//             the same permutation for A and B, so the sum over k is unchanged).
//   gemm_tn : C[M,N] = A[K,M]^T * B[K,N]     contraction index is the slow one
//             (Gram matrices over tokens).  LDS image [k/4][m][4] (4x4 register
//             transposes while staging), fragments by ds_read_b128.  Optional
//             per-column mean subtraction while staging, split over K with one
//             slab per split (deterministic reduction by reduce_slabs_kernel).
#include "basd_common.h"

namespace basd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128;

// ---- split operands: fp32-grade products on the bf16 matrix cores (see syrk_tn_split_kernel) ----
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// (a, b) -> packed bf16 pieces (hi, mid, lo) with a = hi_a + mid_a + lo_a exactly, likewise b
__device__ __forceinline__ void split3(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    mid = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);
    lo = cvt_pk_bf16(sa, sb);
}


struct GemmOperand {
    const void* ptr;
    long sb, sn, sd;     // (batch, row-in-batch, column) strides in elements
    int rows_per_batch;  // rows are indexed m = b * rows_per_batch + n
    long batch_stride;   // extra stride for blockIdx.z-batched problems (elements)
};

__device__ __forceinline__ long row_off(const GemmOperand& o, long m) {
    const long b = m / o.rows_per_batch, n = m - b * o.rows_per_batch;
    return b * o.sb + n * o.sn;
}

__device__ __forceinline__ void store_tile(float* __restrict__ C, long ldc, int M, int N, int m0, int n0,
                                           const f32x16 (&acc)[2][2], int wm, int wn, int lane, float scale,
                                           const float* __restrict__ bias = nullptr, float beta = 0.f) {
    const int col_l = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wn * 64 + ni * 32 + col_l;
            if (col >= N) continue;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row < M) {
                    float v = acc[mi][ni][r] * scale - bv;
                    if (beta != 0.f) v = fmaf(beta, C[(long)row * ldc + col], v);
                    C[(long)row * ldc + col] = v;
                }
            }
        }
}

// ---------------------------------------------------------------------------
// NT: grid = (ceil(N/128), ceil(M/128), batch), block = 256
// ---------------------------------------------------------------------------
constexpr int NT_BK = 32, NT_LD = 36;

// Row offsets are fixed for the whole K loop: computed once (rows past the end clamp to the last row; their
// products land in output rows that store_tile never writes).  The interior K loop is branch-free so that
// all of a thread's loads are in flight together; only a ragged last chunk takes the predicated form.
template <bool VEC>
__device__ __forceinline__ void nt_rows(const GemmOperand& o, int rows_total, int row0, int tid, long (&off)[4]) {
    if (VEC) {
        // lane -> (row, 4 consecutive k): 8 lanes cover one 128-byte row segment
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int row = row0 + (tid >> 3) + 32 * j;
            if (row >= rows_total) row = rows_total - 1;
            off[j] = row_off(o, row) + (long)((tid & 7) * 4) * o.sd;
        }
    } else {
        // lanes along rows (row-contiguous / generic layouts): one row per thread, k = k0 + (tid >> 7) + 2 q
        int row = row0 + (tid & 127);
        if (row >= rows_total) row = rows_total - 1;
        off[0] = row_off(o, row) + (long)(tid >> 7) * o.sd;
        off[1] = off[2] = off[3] = 0;
    }
}

template <typename TA, bool VEC>
__device__ __forceinline__ void nt_load(const GemmOperand& o, const long (&off)[4], int K, int k0, int tid,
                                        float (&reg)[4][4]) {
    const TA* base = (const TA*)o.ptr;
    const bool full = k0 + NT_BK <= K;     // uniform
    if (VEC) {
        if (full) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const TA* p = base + off[j] + k0;
                if (sizeof(TA) == 4) {
                    const float4 v = *(const float4*)p;
                    reg[j][0] = v.x; reg[j][1] = v.y; reg[j][2] = v.z; reg[j][3] = v.w;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) reg[j][c] = to_f32(p[c]);
                }
            }
        } else {
            const int k = k0 + (tid & 7) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) reg[j][c] = (k + c < K) ? to_f32(base[off[j] + k0 + c]) : 0.f;
        }
    } else {
        const TA* p = base + off[0] + (long)k0 * o.sd;
        if (full) {
#pragma unroll
            for (int q = 0; q < 16; ++q) reg[q >> 2][q & 3] = to_f32(p[(long)(2 * q) * o.sd]);
        } else {
            const int k = k0 + (tid >> 7);
#pragma unroll
            for (int q = 0; q < 16; ++q) reg[q >> 2][q & 3] = (k + 2 * q < K) ? to_f32(p[(long)(2 * q) * o.sd]) : 0.f;
        }
    }
}

template <bool VEC>
__device__ __forceinline__ void nt_store_lds(float* __restrict__ tile, int tid, const float (&reg)[4][4]) {
    if (VEC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j;
            *(float4*)(tile + (f >> 3) * NT_LD + (f & 7) * 4) = make_float4(reg[j][0], reg[j][1], reg[j][2], reg[j][3]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int f = tid + 256 * (j * 4 + c);
                tile[(f & 127) * NT_LD + (f >> 7)] = reg[j][c];
            }
    }
}

template <typename TA, bool VEC_A>
__global__ void __launch_bounds__(256) gemm_nt_kernel(GemmOperand A, GemmOperand B, int M, int N, int K,
                                                      float* __restrict__ C, long ldc, long c_batch_stride,
                                                      float scale, const float* __restrict__ bias, float beta,
                                                      float* __restrict__ colsum_part, int nx, int ny) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 128 * NT_LD];
    float* tA = lds;
    float* tB = lds + 128 * NT_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // Tile of this workgroup.  1-D grids (gridDim.y == 1 < ny) are decoded XCD-aware: workgroup ids are dealt round-
    // robin over the 8 XCDs, so the nx column tiles of one row tile get ids of ONE residue class mod 8 and adjacent
    // slots -- the row tile's A panel (the large, often strided operand) then crosses the fabric into one L2 only
    // (teacher projection at cfg-2, nx = 3: 464 MB of HBM-side traffic per launch before, for 125 MB of operands).
    int bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.y == 1 && ny > 1) {
        const int r = bx & 7, s = bx >> 3;
        bx = s % nx;
        by = (s / nx) * 8 + r;
        if (by >= ny) return;
    }
    const int m0 = by * BM, n0 = bx * BN;
    A.ptr = (const TA*)A.ptr + (long)blockIdx.z * A.batch_stride;
    B.ptr = (const float*)B.ptr + (long)blockIdx.z * B.batch_stride;
    C += (long)blockIdx.z * c_batch_stride;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[4][4], rb[4][4];
    long offa[4], offb[4];
    nt_rows<VEC_A>(A, M, m0, tid, offa);
    nt_rows<true>(B, N, n0, tid, offb);
    nt_load<TA, VEC_A>(A, offa, K, 0, tid, ra);
    nt_load<float, true>(B, offb, K, 0, tid, rb);
    const int i = lane & 31, h = lane >> 5;
    for (int k0 = 0; k0 < K; k0 += NT_BK) {
        __syncthreads();
        nt_store_lds<VEC_A>(tA, tid, ra);
        nt_store_lds<true>(tB, tid, rb);
        __syncthreads();
        if (k0 + NT_BK < K) {
            nt_load<TA, VEC_A>(A, offa, K, k0 + NT_BK, tid, ra);
            nt_load<float, true>(B, offb, K, k0 + NT_BK, tid, rb);
        }
#pragma unroll
        for (int g = 0; g < NT_BK / 8; ++g) {
            float4 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = *(const float4*)(tA + (wm * 64 + mi * 32 + i) * NT_LD + g * 8 + 4 * h);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = *(const float4*)(tB + (wn * 64 + ni * 32 + i) * NT_LD + g * 8 + 4 * h);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].x, b[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].y, b[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].z, b[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].w, b[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
    }
    store_tile(C, ldc, M, N, m0, n0, acc, wm, wn, lane, scale, bias, beta);
    if (colsum_part) {
        // column sums of this row tile of C (the values just stored; beta == 0): the caller's column means
        // then cost one small fold instead of another pass over C.  Fixed order: deterministic.
        const int col_l = lane & 31, hi = lane >> 5;
        float s[2] = {0.f, 0.f};
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wn * 64 + ni * 32 + col_l;
            const float bv = (bias && col < N) ? bias[col] : 0.f;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row < M) s[ni] += acc[mi][ni][r] * scale - bv;
                }
            s[ni] += __shfl_xor(s[ni], 32, 64);      // the two lane halves hold different rows of one column
        }
        __syncthreads();                             // the K loop is done with the LDS tiles
        if (hi == 0) {
            lds[wm * 128 + wn * 64 + col_l] = s[0];
            lds[wm * 128 + wn * 64 + 32 + col_l] = s[1];
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < N)
            colsum_part[((long)blockIdx.z * ny + by) * N + n0 + tid] = lds[tid] + lds[128 + tid];
    }
}

// ---------------------------------------------------------------------------
// NT with split operands (see syrk_tn_split_kernel): both tiles are staged as three bf16 planes [row][k], row stride 40
// bf16 = 80 bytes (20 dwords: the 16 lanes of a ds_read_b128 group land on disjoint bank quads), a lane's MFMA operand
// is ONE 16-byte read of 8 consecutive k.  Rows map to MFMA lanes directly: store_tile and the column-sum epilogue are
// those of gemm_nt_kernel.  The strided-operand loader (channel-major teacher tokens: lanes along rows) fetches k in
// adjacent PAIRS (k0 + 2 (tid >> 7) + 4 q + {0, 1}) so that a pair packs into one 4-byte LDS store per plane.
// ---------------------------------------------------------------------------
constexpr int NTS_LD = 40;                       // bf16 elements per LDS row (32 used)
constexpr int NTS_PLANE = 128 * NTS_LD;          // bf16 elements of one plane of one operand tile

template <bool VEC>
__device__ __forceinline__ void nts_rows(const GemmOperand& o, int rows_total, int row0, int tid, long (&off)[4]) {
    if (VEC) {
        nt_rows<true>(o, rows_total, row0, tid, off);
    } else {
        int row = row0 + (tid & 127);
        if (row >= rows_total) row = rows_total - 1;
        off[0] = row_off(o, row) + (long)(2 * (tid >> 7)) * o.sd;
        off[1] = off[2] = off[3] = 0;
    }
}

template <typename TA, bool VEC>
__device__ __forceinline__ void nts_load(const GemmOperand& o, const long (&off)[4], int K, int k0, int tid,
                                         float (&reg)[4][4]) {
    if (VEC) {
        nt_load<TA, true>(o, off, K, k0, tid, reg);
    } else {
        const TA* p = (const TA*)o.ptr + off[0] + (long)k0 * o.sd;
        const int k = k0 + 2 * (tid >> 7);
        if (k0 + NT_BK <= K) {     // uniform
#pragma unroll
            for (int q = 0; q < 16; ++q) reg[q >> 2][q & 3] = to_f32(p[(long)(4 * (q >> 1) + (q & 1)) * o.sd]);
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int dk = 4 * (q >> 1) + (q & 1);
                reg[q >> 2][q & 3] = (k + dk < K) ? to_f32(p[(long)dk * o.sd]) : 0.f;
            }
        }
    }
}

template <bool VEC>
__device__ __forceinline__ void nts_store_lds(unsigned short* __restrict__ tile, int tid, const float (&reg)[4][4]) {
    if (VEC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j;
            unsigned h0, m0, l0, h1, m1, l1;
            split3(reg[j][0], reg[j][1], h0, m0, l0);
            split3(reg[j][2], reg[j][3], h1, m1, l1);
            unsigned short* p = tile + (f >> 3) * NTS_LD + (f & 7) * 4;
            *(uint2*)(p) = make_uint2(h0, h1);
            *(uint2*)(p + NTS_PLANE) = make_uint2(m0, m1);
            *(uint2*)(p + 2 * NTS_PLANE) = make_uint2(l0, l1);
        }
    } else {
        unsigned short* row = tile + (tid & 127) * NTS_LD + 2 * (tid >> 7);
#pragma unroll
        for (int pq = 0; pq < 8; ++pq) {
            unsigned h, m, l;
            split3(reg[pq >> 1][2 * (pq & 1)], reg[pq >> 1][2 * (pq & 1) + 1], h, m, l);
            *(unsigned*)(row + 4 * pq) = h;
            *(unsigned*)(row + 4 * pq + NTS_PLANE) = m;
            *(unsigned*)(row + 4 * pq + 2 * NTS_PLANE) = l;
        }
    }
}

template <typename TA, bool VEC_A>
__global__ void __launch_bounds__(256) gemm_nt_split_kernel(GemmOperand A, GemmOperand B, int M, int N, int K,
                                                            float* __restrict__ C, long ldc, long c_batch_stride,
                                                            float scale, const float* __restrict__ bias, float beta,
                                                            float* __restrict__ colsum_part, int nx, int ny) {
    __shared__ __attribute__((aligned(16))) unsigned short lds16[2 * 3 * NTS_PLANE];
    unsigned short* tA = lds16;
    unsigned short* tB = lds16 + 3 * NTS_PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y;       // XCD-aware decode of 1-D grids: see gemm_nt_kernel
    if (gridDim.y == 1 && ny > 1) {
        const int r = bx & 7, s = bx >> 3;
        bx = s % nx;
        by = (s / nx) * 8 + r;
        if (by >= ny) return;
    }
    const int m0 = by * BM, n0 = bx * BN;
    A.ptr = (const TA*)A.ptr + (long)blockIdx.z * A.batch_stride;
    B.ptr = (const float*)B.ptr + (long)blockIdx.z * B.batch_stride;
    C += (long)blockIdx.z * c_batch_stride;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[4][4], rb[4][4];
    long offa[4], offb[4];
    nts_rows<VEC_A>(A, M, m0, tid, offa);
    nts_rows<true>(B, N, n0, tid, offb);
    nts_load<TA, VEC_A>(A, offa, K, 0, tid, ra);
    nts_load<float, true>(B, offb, K, 0, tid, rb);
    const int i = lane & 31, h = lane >> 5;
    for (int k0 = 0; k0 < K; k0 += NT_BK) {
        __syncthreads();
        nts_store_lds<VEC_A>(tA, tid, ra);
        nts_store_lds<true>(tB, tid, rb);
        __syncthreads();
        if (k0 + NT_BK < K) {
            nts_load<TA, VEC_A>(A, offa, K, k0 + NT_BK, tid, ra);
            nts_load<float, true>(B, offb, K, k0 + NT_BK, tid, rb);
        }
#pragma unroll
        for (int s16 = 0; s16 < NT_BK / 16; ++s16) {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    a[mi][p] = *(const bf16x8*)(tA + p * NTS_PLANE + (wm * 64 + mi * 32 + i) * NTS_LD + 16 * s16 + 8 * h);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    b[ni][p] = *(const bf16x8*)(tB + p * NTS_PLANE + (wn * 64 + ni * 32 + i) * NTS_LD + 16 * s16 + 8 * h);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    f32x16 c = acc[mi][ni];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][2], b[ni][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][0], c, 0, 0, 0);
                    acc[mi][ni] = c;
                }
        }
    }
    store_tile(C, ldc, M, N, m0, n0, acc, wm, wn, lane, scale, bias, beta);
    if (colsum_part) {       // column sums of this row tile of C: see gemm_nt_kernel
        float* lds = (float*)lds16;
        const int col_l = lane & 31, hi = lane >> 5;
        float s[2] = {0.f, 0.f};
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wn * 64 + ni * 32 + col_l;
            const float bv = (bias && col < N) ? bias[col] : 0.f;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row < M) s[ni] += acc[mi][ni][r] * scale - bv;
                }
            s[ni] += __shfl_xor(s[ni], 32, 64);
        }
        __syncthreads();
        if (hi == 0) {
            lds[wm * 128 + wn * 64 + col_l] = s[0];
            lds[wm * 128 + wn * 64 + 32 + col_l] = s[1];
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < N)
            colsum_part[((long)blockIdx.z * ny + by) * N + n0 + tid] = lds[tid] + lds[128 + tid];
    }
}

// ---------------------------------------------------------------------------
// TN: grid = (ceil(N/128), ceil(M/128), batch * splits), block = 256
//   A: (Krows x M), B: (Krows x N); contraction over rows [k_begin, k_end) of the split.
//   out: slab (split) or batch element: C + z * c_z_stride.
// ---------------------------------------------------------------------------
constexpr int TN_BK = 32;

// Stage a (TN_BK rows x 128 columns) slab as [k/4][column][4]: one ds_read_b128 then yields the four
// k-steps of a lane (lane half h takes k-group 2g + h; same permutation for A and B).  Threads 0..127 stage
// A, 128..255 stage B; each thread transposes two 4x4 blocks in registers.
// A thread stages 8 rows (2 groups of 4 consecutive k) x 4 consecutive columns.  TnCursor carries the
// element offset and the in-batch row index of each of its rows and advances by TN_BK rows per chunk with
// no division; interior chunks of interior column tiles load branch-free (all 8 loads in flight together),
// ragged chunks / tiles take the predicated form.  The mean is subtracted while staging into LDS (not at
// load time, which would put a wait on the loads ahead of the MFMA block); rows past the end are filled
// with the mean so that they stage as exact zeros.
struct TnCursor {
    long off[2];   // element offset of the first row of each 4-row group
    int n[2];      // its row index inside the batch item
};

__device__ __forceinline__ void tn_cursor_init(TnCursor& cu, const GemmOperand& o, int k_begin, int t) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k_begin + ((t >> 5) + 4 * j) * 4;
        const int b = k / o.rows_per_batch, n = k - b * o.rows_per_batch;
        cu.n[j] = n;
        cu.off[j] = (long)b * o.sb + (long)n * o.sn;
    }
}

__device__ __forceinline__ void tn_cursor_advance(TnCursor& cu, const GemmOperand& o) {
    const long wrap = o.sb - (long)o.rows_per_batch * o.sn;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        cu.n[j] += TN_BK;
        cu.off[j] += (long)TN_BK * o.sn;
        while (cu.n[j] >= o.rows_per_batch) { cu.n[j] -= o.rows_per_batch; cu.off[j] += wrap; }
    }
}

// offset of row r (0..3) of group j: the group may straddle a batch-item boundary
__device__ __forceinline__ long tn_row(const TnCursor& cu, const GemmOperand& o, int j, int r) {
    int n = cu.n[j] + r;
    long off = cu.off[j] + (long)r * o.sn;
    const long wrap = o.sb - (long)o.rows_per_batch * o.sn;
    while (n >= o.rows_per_batch) { n -= o.rows_per_batch; off += wrap; }
    return off;
}

// mean4: the thread's four column means (0 when not centred / column out of range)
__device__ __forceinline__ void tn_mean4(const float* __restrict__ mean, int cols, int col0, int t, float (&m4)[4]) {
    const int c = col0 + (t & 31) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) m4[e] = (mean && c + e < cols) ? mean[c + e] : 0.f;
}

template <typename T, bool VEC>
__device__ __forceinline__ void tn_load(const GemmOperand& o, const TnCursor& cu, int krows_end, int cols, int k0,
                                        int col0, int t, const float (&m4)[4], float (&reg)[2][4][4]) {
    const T* base = (const T*)o.ptr;
    const int c = col0 + (t & 31) * 4;
    if (VEC && k0 + TN_BK <= krows_end && col0 + 128 <= cols) {      // uniform
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const T* p = base + tn_row(cu, o, j, r) + c;
                if (sizeof(T) == 4) {
                    const float4 v = *(const float4*)p;
                    reg[j][r][0] = v.x; reg[j][r][1] = v.y; reg[j][r][2] = v.z; reg[j][r][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) reg[j][r][e] = to_f32(p[e]);
                }
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kg = (t >> 5) + 4 * j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool row_ok = k0 + kg * 4 + r < krows_end;
            const long ro = row_ok ? tn_row(cu, o, j, r) : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cc = c + e;
                reg[j][r][e] = !row_ok ? m4[e] : (cc < cols ? to_f32(base[ro + (long)cc * o.sd]) : 0.f);
            }
        }
    }
}

__device__ __forceinline__ void tn_store_lds(float* __restrict__ tile, int t, const float (&reg)[2][4][4],
                                             const float (&m4)[4]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kg = (t >> 5) + 4 * j, m = (t & 31) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            *(float4*)(tile + ((kg * 128 + m + e) << 2)) =
                make_float4(reg[j][0][e] - m4[e], reg[j][1][e] - m4[e], reg[j][2][e] - m4[e], reg[j][3][e] - m4[e]);
    }
}

template <typename T, bool VEC>
__global__ void __launch_bounds__(256) gemm_tn_kernel(GemmOperand A, GemmOperand B, int M, int N, int Krows,
                                                      int splits, const float* __restrict__ mean_a,
                                                      const float* __restrict__ mean_b, float* __restrict__ C,
                                                      long ldc, long c_z_stride, float scale) {
    __shared__ __attribute__((aligned(16))) float lds[2 * TN_BK * 128];
    float* tA = lds;
    float* tB = lds + TN_BK * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int bz = blockIdx.z / splits, sp = blockIdx.z - bz * splits;
    A.ptr = (const T*)A.ptr + (long)bz * A.batch_stride;
    B.ptr = (const T*)B.ptr + (long)bz * B.batch_stride;
    C += (long)blockIdx.z * c_z_stride;
    // rows of this split, in multiples of TN_BK
    const int chunks = (Krows + TN_BK - 1) / TN_BK;
    const int per = (chunks + splits - 1) / splits;
    const int k_begin = sp * per * TN_BK;
    int k_end = k_begin + per * TN_BK;
    if (k_end > Krows) k_end = Krows;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // threads 0..127 own operand A, 128..255 operand B (wave-uniform split)
    const bool is_b = tid >= 128;
    const int t = tid & 127;
    const GemmOperand& op = is_b ? B : A;
    const int cols = is_b ? N : M, col0 = is_b ? n0 : m0;
    const float* mean = is_b ? mean_b : mean_a;
    float* tile = is_b ? tB : tA;
    float reg[2][4][4], m4[4];
    TnCursor cu;
    tn_cursor_init(cu, op, k_begin, t);
    tn_mean4(mean, cols, col0, t, m4);
    if (k_begin < k_end) tn_load<T, VEC>(op, cu, k_end, cols, k_begin, col0, t, m4, reg);
    const int i = lane & 31, h = lane >> 5;
    for (int k0 = k_begin; k0 < k_end; k0 += TN_BK) {
        __syncthreads();
        tn_store_lds(tile, t, reg, m4);
        __syncthreads();
        if (k0 + TN_BK < k_end) {
            tn_cursor_advance(cu, op);
            tn_load<T, VEC>(op, cu, k_end, cols, k0 + TN_BK, col0, t, m4, reg);
        }
#pragma unroll
        for (int g = 0; g < TN_BK / 8; ++g) {
            float4 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = *(const float4*)(tA + (((2 * g + h) * 128 + wm * 64 + mi * 32 + i) << 2));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = *(const float4*)(tB + (((2 * g + h) * 128 + wn * 64 + ni * 32 + i) << 2));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].x, b[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].y, b[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].z, b[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].w, b[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
    }
    store_tile(C, ldc, M, N, m0, n0, acc, wm, wn, lane, scale);
}


// ---------------------------------------------------------------------------
// Multi-matrix symmetric TN (centred Gram / SYRK):  G[z] = scale[z] * (X_z - 1 mean_z^T)^T (X_z - 1 mean_z^T)
//   grid = T(T+1)/2 lower-triangular 128x128 tile pairs x splits x n_mats (1-D, decoded XCD-aware), block = 256.
//   X_z share shape and strides, base pointers come from a device table.  Only tiles (mi >= ni) are
//   computed; a diagonal tile stages its column slab once and feeds both MFMA operands from it.  Half the
//   column-slab traffic of the general TN kernel (9 instead of 18 slab reads at 384 columns).
//   syrk_reduce_kernel folds the split slabs in fixed order and mirrors the strict lower tiles.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void tri_tile(int p, int& mi, int& ni) {
    mi = 0;
    while ((mi + 1) * (mi + 2) / 2 <= p) ++mi;
    ni = p - mi * (mi + 1) / 2;
}

template <typename T, bool VEC>
__global__ void __launch_bounds__(256) syrk_tn_kernel(const void* const* __restrict__ ptrs, GemmOperand X, int cols,
                                                      int Krows, int splits, int n_mats, int pairs,
                                                      const float* __restrict__ means, float* __restrict__ slabs,
                                                      const float* __restrict__ fold, int fold_parts, int fold_from) {
    __shared__ __attribute__((aligned(16))) float lds[2 * TN_BK * 128];
    // XCD-aware decode of the 1-D grid: workgroup ids that agree mod 8 run on one XCD (one L2).  The tile
    // pairs of one (matrix, split) unit read the same rows, so they are given ids of one residue class and
    // adjacent dispatch slots: the unit's rows come over the fabric once, the other pairs hit in that L2
    // (measured: 975 MB -> see profiles/ per launch at cfg-2 against 308 MB of operands).
    int pair, unit;
    {
        const int units = splits * n_mats, id = blockIdx.x;
        if ((units & 7) == 0) {
            const int xcd = id & 7, slot = id >> 3;
            pair = slot % pairs;
            unit = xcd + 8 * (slot / pairs);
        } else {
            pair = id % pairs;
            unit = id / pairs;
        }
    }
    int tmi, tni;
    tri_tile(pair, tmi, tni);
    const bool diag = tmi == tni;
    float* tA = lds;
    float* tB = diag ? lds : lds + TN_BK * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = tmi * BM, n0 = tni * BN;
    const int z = unit / splits, sp = unit - z * splits;
    X.ptr = ptrs[z];
    const float* mean = means ? means + (long)z * cols : nullptr;
    float* C = slabs + ((long)z * splits + sp) * (long)cols * cols;
    const int chunks = (Krows + TN_BK - 1) / TN_BK;
    const int per = (chunks + splits - 1) / splits;
    const int k_begin = sp * per * TN_BK;
    int k_end = k_begin + per * TN_BK;
    if (k_end > Krows) k_end = Krows;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool is_b = tid >= 128;
    const bool loader = !(is_b && diag);       // wave-uniform
    const int t = tid & 127;
    const int col0 = is_b ? n0 : m0;
    float* tile = is_b ? tB : tA;
    float reg[2][4][4], m4[4];
    TnCursor cu;
    tn_cursor_init(cu, X, k_begin, t);
    if (fold && z >= fold_from) {
        // column means folded here from the row-tile sums that the producing GEMM's epilogue left behind
        // (fixed order: deterministic) -- one small kernel less on the teacher chain, which is the step's
        // critical path and gets its small launches starved by the student chain's long Gram launch
        const float* fp = fold + (long)(z - fold_from) * fold_parts * cols;
        const int c = col0 + (t & 31) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = 0.f;
            if (c + e < cols)
                for (int p = 0; p < fold_parts; ++p) s += fp[(long)p * cols + c + e];
            m4[e] = s * (1.f / Krows);
        }
    } else {
        tn_mean4(mean, cols, col0, t, m4);
    }
    if (loader && k_begin < k_end) tn_load<T, VEC>(X, cu, k_end, cols, k_begin, col0, t, m4, reg);
    const int i = lane & 31, h = lane >> 5;
    for (int k0 = k_begin; k0 < k_end; k0 += TN_BK) {
        __syncthreads();
        if (loader) tn_store_lds(tile, t, reg, m4);
        __syncthreads();
        if (loader && k0 + TN_BK < k_end) {
            tn_cursor_advance(cu, X);
            tn_load<T, VEC>(X, cu, k_end, cols, k0 + TN_BK, col0, t, m4, reg);
        }
#pragma unroll
        for (int g = 0; g < TN_BK / 8; ++g) {
            float4 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = *(const float4*)(tA + (((2 * g + h) * 128 + wm * 64 + mi * 32 + i) << 2));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = *(const float4*)(tB + (((2 * g + h) * 128 + wn * 64 + ni * 32 + i) << 2));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].x, b[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].y, b[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].z, b[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].w, b[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
    }
    store_tile(C, cols, cols, cols, m0, n0, acc, wm, wn, lane, 1.f);
}

// ---------------------------------------------------------------------------
// The same Gram launch on the bf16 matrix cores, with fp32 results: "split" operands.
//   fp32 MFMA runs at 157 TF dense on this chip, bf16 MFMA at 2.5 PF.  Every staged value (x - mean, fp32) is cut into
//   three bf16 pieces hi + mid + lo = x EXACTLY (round-to-nearest pieces of 8 significant bits each: 24 in all), once,
//   while the slab goes to LDS; a product x y is then the six piece products of order <= 2^-16 (hi hi, hi mid, mid hi,
//   mid mid, hi lo, lo hi: each exact in fp32), accumulated in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  What is
//   dropped (mid lo, lo mid, lo lo) is below 2^-24 |x y| with either sign -- less than the rounding of one fp32 product.
//   6 bf16 MFMAs replace 8 fp32 ones per 16 contraction steps at 16x the rate each: 2.7x the fp32 MFMA ceiling.
//   LDS image per operand: three planes [octet of k][column position][8 bf16] (24 KB per 32-row slab): one ds_read_b128
//   IS the 8-element operand of a lane.  Column c = 4 l + e of the 128-column slab sits at position 32 e + l, so that the
//   staging threads (4 consecutive columns each, float4 global loads) store conflict-free; the MFMA lane i of 32-block
//   b therefore holds column 4 i + b, and the epilogue stores C with that map.
// ---------------------------------------------------------------------------
constexpr int SPLIT_PLANE = (TN_BK / 8) * 128 * 8;       // bf16 elements of one plane of one operand slab

__device__ __forceinline__ void tn_store_lds_split(unsigned short* __restrict__ tile, int t, const float (&reg)[2][4][4],
                                                   const float (&m4)[4]) {
    const int l = t & 31;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kg = (t >> 5) + 4 * j, o = kg >> 1, half = kg & 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned h0, m0, l0, h1, m1, l1;
            split3(reg[j][0][e] - m4[e], reg[j][1][e] - m4[e], h0, m0, l0);
            split3(reg[j][2][e] - m4[e], reg[j][3][e] - m4[e], h1, m1, l1);
            unsigned short* p = tile + ((o * 128 + 32 * e + l) << 3) + 4 * half;
            *(uint2*)(p) = make_uint2(h0, h1);
            *(uint2*)(p + SPLIT_PLANE) = make_uint2(m0, m1);
            *(uint2*)(p + 2 * SPLIT_PLANE) = make_uint2(l0, l1);
        }
    }
}

template <typename T, bool VEC>
__global__ void __launch_bounds__(256) syrk_tn_split_kernel(const void* const* __restrict__ ptrs, GemmOperand X, int cols,
                                                            int Krows, int splits, int n_mats, int pairs,
                                                            const float* __restrict__ means, float* __restrict__ slabs,
                                                            const float* __restrict__ fold, int fold_parts, int fold_from) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * 3 * SPLIT_PLANE];
    int pair, unit;       // XCD-aware decode: see syrk_tn_kernel
    {
        const int units = splits * n_mats, id = blockIdx.x;
        if ((units & 7) == 0) {
            const int xcd = id & 7, slot = id >> 3;
            pair = slot % pairs;
            unit = xcd + 8 * (slot / pairs);
        } else {
            pair = id % pairs;
            unit = id / pairs;
        }
    }
    int tmi, tni;
    tri_tile(pair, tmi, tni);
    const bool diag = tmi == tni;
    unsigned short* tA = lds;
    unsigned short* tB = diag ? lds : lds + 3 * SPLIT_PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = tmi * BM, n0 = tni * BN;
    const int z = unit / splits, sp = unit - z * splits;
    X.ptr = ptrs[z];
    const float* mean = means ? means + (long)z * cols : nullptr;
    float* C = slabs + ((long)z * splits + sp) * (long)cols * cols;
    const int chunks = (Krows + TN_BK - 1) / TN_BK;
    const int per = (chunks + splits - 1) / splits;
    const int k_begin = sp * per * TN_BK;
    int k_end = k_begin + per * TN_BK;
    if (k_end > Krows) k_end = Krows;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool is_b = tid >= 128;
    const bool loader = !(is_b && diag);       // wave-uniform
    const int t = tid & 127;
    const int col0 = is_b ? n0 : m0;
    unsigned short* tile = is_b ? tB : tA;
    float reg[2][4][4], m4[4];
    TnCursor cu;
    tn_cursor_init(cu, X, k_begin, t);
    if (fold && z >= fold_from) {
        const float* fp = fold + (long)(z - fold_from) * fold_parts * cols;
        const int c = col0 + (t & 31) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = 0.f;
            if (c + e < cols)
                for (int p = 0; p < fold_parts; ++p) s += fp[(long)p * cols + c + e];
            m4[e] = s * (1.f / Krows);
        }
    } else {
        tn_mean4(mean, cols, col0, t, m4);
    }
    if (loader && k_begin < k_end) tn_load<T, VEC>(X, cu, k_end, cols, k_begin, col0, t, m4, reg);
    const int i = lane & 31, h = lane >> 5;
    for (int k0 = k_begin; k0 < k_end; k0 += TN_BK) {
        __syncthreads();
        if (loader) tn_store_lds_split(tile, t, reg, m4);
        __syncthreads();
        if (loader && k0 + TN_BK < k_end) {
            tn_cursor_advance(cu, X);
            tn_load<T, VEC>(X, cu, k_end, cols, k0 + TN_BK, col0, t, m4, reg);
        }
#pragma unroll
        for (int s16 = 0; s16 < TN_BK / 16; ++s16) {
            bf16x8 a[2][3], b[2][3];
            const int o = 2 * s16 + h;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    a[mi][p] = *(const bf16x8*)(tA + p * SPLIT_PLANE + ((o * 128 + (wm * 2 + mi) * 32 + i) << 3));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    b[ni][p] = *(const bf16x8*)(tB + p * SPLIT_PLANE + ((o * 128 + (wn * 2 + ni) * 32 + i) << 3));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    f32x16 c = acc[mi][ni];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][2], b[ni][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][0], c, 0, 0, 0);
                    acc[mi][ni] = c;
                }
        }
    }
    // epilogue: MFMA lane i of 32-block b holds slab column 4 i + b (rows of C from the A side, columns from the B side)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + 4 * i + (wn * 2 + ni);
            if (col >= cols) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * h) + (wm * 2 + mi);
                if (row < cols) C[(long)row * cols + col] = acc[mi][ni][r];
            }
        }
}

// General TN with split operands: gemm_tn_kernel's grid, loaders and splits, syrk_tn_split_kernel's LDS image and MFMA
// block (column 4 l + e of a slab at position 32 e + l; the epilogue undoes the map on both sides).
template <typename T, bool VEC>
__global__ void __launch_bounds__(256) gemm_tn_split_kernel(GemmOperand A, GemmOperand B, int M, int N, int Krows,
                                                            int splits, const float* __restrict__ mean_a,
                                                            const float* __restrict__ mean_b, float* __restrict__ C,
                                                            long ldc, long c_z_stride, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * 3 * SPLIT_PLANE];
    unsigned short* tA = lds;
    unsigned short* tB = lds + 3 * SPLIT_PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int bz = blockIdx.z / splits, sp = blockIdx.z - bz * splits;
    A.ptr = (const T*)A.ptr + (long)bz * A.batch_stride;
    B.ptr = (const T*)B.ptr + (long)bz * B.batch_stride;
    C += (long)blockIdx.z * c_z_stride;
    const int chunks = (Krows + TN_BK - 1) / TN_BK;
    const int per = (chunks + splits - 1) / splits;
    const int k_begin = sp * per * TN_BK;
    int k_end = k_begin + per * TN_BK;
    if (k_end > Krows) k_end = Krows;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool is_b = tid >= 128;
    const int t = tid & 127;
    const GemmOperand& op = is_b ? B : A;
    const int cols = is_b ? N : M, col0 = is_b ? n0 : m0;
    const float* mean = is_b ? mean_b : mean_a;
    unsigned short* tile = is_b ? tB : tA;
    float reg[2][4][4], m4[4];
    TnCursor cu;
    tn_cursor_init(cu, op, k_begin, t);
    tn_mean4(mean, cols, col0, t, m4);
    if (k_begin < k_end) tn_load<T, VEC>(op, cu, k_end, cols, k_begin, col0, t, m4, reg);
    const int i = lane & 31, h = lane >> 5;
    for (int k0 = k_begin; k0 < k_end; k0 += TN_BK) {
        __syncthreads();
        tn_store_lds_split(tile, t, reg, m4);
        __syncthreads();
        if (k0 + TN_BK < k_end) {
            tn_cursor_advance(cu, op);
            tn_load<T, VEC>(op, cu, k_end, cols, k0 + TN_BK, col0, t, m4, reg);
        }
#pragma unroll
        for (int s16 = 0; s16 < TN_BK / 16; ++s16) {
            bf16x8 a[2][3], b[2][3];
            const int o = 2 * s16 + h;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    a[mi][p] = *(const bf16x8*)(tA + p * SPLIT_PLANE + ((o * 128 + (wm * 2 + mi) * 32 + i) << 3));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    b[ni][p] = *(const bf16x8*)(tB + p * SPLIT_PLANE + ((o * 128 + (wn * 2 + ni) * 32 + i) << 3));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    f32x16 c = acc[mi][ni];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][2], b[ni][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][0], c, 0, 0, 0);
                    acc[mi][ni] = c;
                }
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + 4 * i + (wn * 2 + ni);
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * h) + (wm * 2 + mi);
                if (row < M) C[(long)row * ldc + col] = acc[mi][ni][r] * scale;
            }
        }
}

// grid = (tile pairs * 64, n_mats), block 256: one element of a lower tile per thread.
__global__ void __launch_bounds__(256) syrk_reduce_kernel(const float* __restrict__ slabs, int cols, int splits,
                                                          const float* __restrict__ scales, float* __restrict__ out,
                                                          long out_stride) {
    int tmi, tni;
    tri_tile(blockIdx.x >> 6, tmi, tni);
    const int e = (blockIdx.x & 63) * 256 + threadIdx.x;
    const int r = tmi * BM + (e >> 7), c = tni * BN + (e & 127);
    if (r >= cols || c >= cols) return;
    const int z = blockIdx.y;
    const long mat = (long)cols * cols;
    // a diagonal tile holds both triangles; the upper one is taken from the lower (the split-operand kernel sums the
    // piece products of (r, c) and (c, r) in different orders: equal to rounding, not bit for bit)
    const bool up = tmi == tni && r < c;
    const float* in = slabs + (long)z * splits * mat + (up ? (long)c * cols + r : (long)r * cols + c);
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += in[k * mat];
    s *= scales ? scales[z] : 1.f;
    float* o = out + (long)z * out_stride;
    o[(long)r * cols + c] = s;
    if (tmi != tni) o[(long)c * cols + r] = s;
}

// out[i] = scale * sum_s slabs[s][i]   (fixed order: deterministic)
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, long slab_stride, int splits, long count,
                                    float scale, float* __restrict__ out, long batch_in_stride,
                                    long batch_out_stride) {
    const float* in = slabs + (long)blockIdx.y * batch_in_stride;
    float* o = out + (long)blockIdx.y * batch_out_stride;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += in[(long)k * slab_stride + i];
        o[i] = s * scale;
    }
}

// Column means of a (rows x cols) strided token matrix: two-stage, deterministic.
//   stage 1: grid = (ceil(cols/256), parts, batch) partial sums; stage 2 folds the parts.
template <typename T>
__global__ void __launch_bounds__(256) colsum_partial_kernel(GemmOperand X, int rows, int cols, int parts,
                                                             float* __restrict__ partial,
                                                             const void* const* __restrict__ ptrs = nullptr) {
    // block = 64 columns x 4 row groups; grid = (ceil(cols/64), parts, batch)
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const T* base = ptrs ? (const T*)ptrs[blockIdx.z] : (const T*)X.ptr + (long)blockIdx.z * X.batch_stride;
    const int per = (rows + parts - 1) / parts;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    float s = 0.f;
    if (c < cols)
        for (int r = r0 + rg; r < r1; r += 4) s += to_f32(base[row_off(X, r) + (long)c * X.sd]);
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < cols)
        partial[((long)blockIdx.z * parts + blockIdx.y) * cols + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// Vector form (unit column stride, 16-byte aligned rows): a wave reads 1 KB of one row per load and keeps four
// rows in flight; block = 4 row groups x 64 column quads; grid = (ceil(cols/256), parts, batch).
__device__ __forceinline__ float4 load4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ float4 load4(const __hip_bfloat16* p) {
    const uint2 v = *(const uint2*)p;      // four bf16: widen by shifting into the high half
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
}

template <typename T>
__global__ void __launch_bounds__(256) colsum_partial_vec_kernel(GemmOperand X, int rows, int cols, int parts,
                                                                 float* __restrict__ partial,
                                                                 const void* const* __restrict__ ptrs) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6, c = (blockIdx.x * 64 + lane) * 4;
    const T* base = ptrs ? (const T*)ptrs[blockIdx.z] : (const T*)X.ptr + (long)blockIdx.z * X.batch_stride;
    const int per = (rows + parts - 1) / parts;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    const bool active = c < cols;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    // (batch item, row in item) of this wave's current row: advanced without division
    int r = r0 + rg;
    int b = r / X.rows_per_batch, n = r - b * X.rows_per_batch;
    long off = (long)b * X.sb + (long)n * X.sn;
    const long wrap = X.sb - (long)X.rows_per_batch * X.sn;
    auto next = [&]() {
        r += 4; n += 4; off += 4 * X.sn;
        while (n >= X.rows_per_batch) { n -= X.rows_per_batch; off += wrap; }     // wave-uniform
    };
    while (r + 12 < r1) {
        long o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { o[u] = off; next(); }
        if (active) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = load4(base + o[u] + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
    }
    while (r < r1) {
        if (active) { const float4 v = load4(base + off + c); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        next();
    }
    red[rg][lane] = s;
    __syncthreads();
    if (rg == 0 && active) {
        const float4 a = red[0][lane], b4 = red[1][lane], c4 = red[2][lane], d4 = red[3][lane];
        *(float4*)(partial + ((long)blockIdx.z * parts + blockIdx.y) * cols + c) =
            make_float4((a.x + b4.x) + (c4.x + d4.x), (a.y + b4.y) + (c4.y + d4.y), (a.z + b4.z) + (c4.z + d4.z),
                        (a.w + b4.w) + (c4.w + d4.w));
    }
}

__global__ void __launch_bounds__(256) colsum_final_kernel(const float* __restrict__ partial, int cols, int parts,
                                                           float inv_rows, float* __restrict__ mean) {
    // block = 64 columns x 4 part groups (fixed summation order: deterministic)
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const float* p = partial + (long)blockIdx.y * parts * cols;
    float s = 0.f;
    if (c < cols) {
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four independent partial folds: loads overlap
        int k = pg;
        for (; k + 12 < parts; k += 16) {
            s += p[(long)k * cols + c];
            s1 += p[(long)(k + 4) * cols + c];
            s2 += p[(long)(k + 8) * cols + c];
            s3 += p[(long)(k + 12) * cols + c];
        }
        for (; k < parts; k += 4) s += p[(long)k * cols + c];
        s = (s + s1) + (s2 + s3);
    }
    red[pg][cl] = s;
    __syncthreads();
    if (pg == 0 && c < cols)
        mean[(long)blockIdx.y * cols + c] = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) * inv_rows;
}

}  // namespace basd

using namespace basd;

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <typename T>
static void launch_colsum(const GemmOperand& X, const void* const* ptrs, const void* x, int rows, int cols, int parts,
                          int batch, bool vec, float* partial, hipStream_t stream) {
    if (vec)
        colsum_partial_vec_kernel<T><<<dim3((cols + 255) / 256, parts, batch), 256, 0, stream>>>(X, rows, cols, parts, partial, ptrs);
    else
        colsum_partial_kernel<T><<<dim3((cols + 63) / 64, parts, batch), 256, 0, stream>>>(X, rows, cols, parts, partial, ptrs);
}

extern "C" {

// C[z] (M x N, ldc) = beta * C[z] + scale * A[z] (M x K) * B[z]^T (N x K) - 1 bias^T.
// Test / tuning hook: 1 (default) = Gram launches and NT products on the bf16 matrix cores with three-way split operands
// (fp32 results: syrk_tn_split_kernel, gemm_nt_split_kernel); 0 = fp32 MFMA.  Process-wide.
static int g_gemm_split = 1;
int basd_gemm_tuning_get(void) { return g_gemm_split; }
int basd_gemm_tuning(int split_bf16) {
    if (split_bf16 != 0 && split_bf16 != 1) return BASD_EINVAL;
    g_gemm_split = split_bf16;
    return BASD_OK;
}

//   A: element (m, k) at a + z*a_batch_stride + (m / a_rows_per_batch)*a_sb + (m % a_rows_per_batch)*a_sn + k*a_sd,
//      dtype fp32 or bf16.   B: fp32, row-major with leading dimension ldb.  bias (nullable): N floats.
int basd_gemm_nt(const void* a, int a_dtype, long a_sb, long a_sn, long a_sd, int a_rows_per_batch,
                 long a_batch_stride, const float* b, long ldb, long b_batch_stride, int M, int N, int K, int batch,
                 float* c, long ldc, long c_batch_stride, float scale, const float* bias, float beta,
                 float* colsum_part, float* col_mean, hipStream_t stream) {
    BASD_CHECK_ARG(a && b && c && M > 0 && N > 0 && K > 0 && batch > 0 && a_rows_per_batch > 0);
    BASD_CHECK_ARG(aligned16(b) && ldb % 4 == 0 && b_batch_stride % 4 == 0);
    BASD_CHECK_ARG(!colsum_part || beta == 0.f);
    BASD_CHECK_ARG(!col_mean || colsum_part);
    GemmOperand A{a, a_sb, a_sn, a_sd, a_rows_per_batch, a_batch_stride};
    GemmOperand B{b, 0, ldb, 1, 1 << 30, b_batch_stride};
    const int nx = (N + BN - 1) / BN, ny = (M + BM - 1) / BM;
    // many row tiles x several column tiles: 1-D grid with the XCD-aware decode (see the kernel); else the plain 3-D grid
    const bool xcd = nx > 1 && ny >= 16;
    const dim3 grid = xcd ? dim3(8 * nx * ((ny + 7) / 8), 1, batch) : dim3(nx, ny, batch);
    if (g_gemm_split && a_dtype == BASD_DTYPE_F32) {
        const bool vec = a_sd == 1 && aligned16(a) && a_sb % 4 == 0 && a_sn % 4 == 0 && a_batch_stride % 4 == 0;
        if (vec) gemm_nt_split_kernel<float, true><<<grid, 256, 0, stream>>>(A, B, M, N, K, c, ldc, c_batch_stride, scale, bias, beta, colsum_part, nx, ny);
        else gemm_nt_split_kernel<float, false><<<grid, 256, 0, stream>>>(A, B, M, N, K, c, ldc, c_batch_stride, scale, bias, beta, colsum_part, nx, ny);
    } else if (g_gemm_split && a_dtype == BASD_DTYPE_BF16) {
        if (a_sd == 1) gemm_nt_split_kernel<__hip_bfloat16, true><<<grid, 256, 0, stream>>>(A, B, M, N, K, c, ldc, c_batch_stride, scale, bias, beta, colsum_part, nx, ny);
        else gemm_nt_split_kernel<__hip_bfloat16, false><<<grid, 256, 0, stream>>>(A, B, M, N, K, c, ldc, c_batch_stride, scale, bias, beta, colsum_part, nx, ny);
    } else if (a_dtype == BASD_DTYPE_F32) {
        const bool vec = a_sd == 1 && aligned16(a) && a_sb % 4 == 0 && a_sn % 4 == 0 && a_batch_stride % 4 == 0;
        if (vec) gemm_nt_kernel<float, true><<<grid, 256, 0, stream>>>(A, B, M, N, K, c, ldc, c_batch_stride, scale, bias, beta, colsum_part, nx, ny);
        else gemm_nt_kernel<float, false><<<grid, 256, 0, stream>>>(A, B, M, N, K, c, ldc, c_batch_stride, scale, bias, beta, colsum_part, nx, ny);
    } else if (a_dtype == BASD_DTYPE_BF16) {
        if (a_sd == 1) gemm_nt_kernel<__hip_bfloat16, true><<<grid, 256, 0, stream>>>(A, B, M, N, K, c, ldc, c_batch_stride, scale, bias, beta, colsum_part, nx, ny);
        else gemm_nt_kernel<__hip_bfloat16, false><<<grid, 256, 0, stream>>>(A, B, M, N, K, c, ldc, c_batch_stride, scale, bias, beta, colsum_part, nx, ny);
    } else {
        return BASD_EINVAL;
    }
    if (col_mean)      // fold the row-tile sums: col_mean[z][n] = (1/M) sum over the row tiles
        colsum_final_kernel<<<dim3((N + 63) / 64, batch), 256, 0, stream>>>(colsum_part, N, (M + BM - 1) / BM, 1.f / M, col_mean);
    BASD_RETURN_LAST();
}

int basd_gemm_tn_splits(int krows) {
    int s = krows / 256;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

// C[z] (M x N) = scale * (A[z] - 1 mean_a^T)^T (B[z] - 1 mean_b^T), contraction over `krows` rows.
//   A element (k, m) at a + z*a_batch_stride + (k / rows_per_batch)*sb + (k % rows_per_batch)*sn + m*sd; B likewise
//   with its own base/strides; both share dtype.  `slabs` (nullable when splits == 1) holds
//   batch*splits*M*N floats of scratch.
int basd_gemm_tn(const void* a, const void* b, int dtype, long a_sb, long a_sn, long a_sd, long a_batch_stride,
                 long b_sb, long b_sn, long b_sd, long b_batch_stride, int rows_per_batch, int krows, int M, int N,
                 int batch, const float* mean_a, const float* mean_b, int splits, float* slabs, float* c, long ldc,
                 long c_batch_stride, float scale, hipStream_t stream) {
    BASD_CHECK_ARG(a && b && c && krows > 0 && M > 0 && N > 0 && batch > 0 && splits >= 1 && rows_per_batch > 0);
    BASD_CHECK_ARG(splits == 1 || (slabs != nullptr && ldc == N));
    GemmOperand A{a, a_sb, a_sn, a_sd, rows_per_batch, a_batch_stride};
    GemmOperand B{b, b_sb, b_sn, b_sd, rows_per_batch, b_batch_stride};
    const dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch * splits);
    float* out = splits == 1 ? c : slabs;
    const long out_ld = splits == 1 ? ldc : N;
    const long z_stride = splits == 1 ? c_batch_stride : (long)M * N;
    const float k_scale = splits == 1 ? scale : 1.f;
    const bool vec = dtype == BASD_DTYPE_F32 && a_sd == 1 && b_sd == 1 && aligned16(a) && aligned16(b) &&
                     a_sb % 4 == 0 && a_sn % 4 == 0 && b_sb % 4 == 0 && b_sn % 4 == 0 && a_batch_stride % 4 == 0 &&
                     b_batch_stride % 4 == 0;
    if (g_gemm_split && dtype == BASD_DTYPE_F32) {
        if (vec) gemm_tn_split_kernel<float, true><<<grid, 256, 0, stream>>>(A, B, M, N, krows, splits, mean_a, mean_b, out, out_ld, z_stride, k_scale);
        else gemm_tn_split_kernel<float, false><<<grid, 256, 0, stream>>>(A, B, M, N, krows, splits, mean_a, mean_b, out, out_ld, z_stride, k_scale);
    } else if (g_gemm_split && dtype == BASD_DTYPE_BF16) {
        if (a_sd == 1 && b_sd == 1) gemm_tn_split_kernel<__hip_bfloat16, true><<<grid, 256, 0, stream>>>(A, B, M, N, krows, splits, mean_a, mean_b, out, out_ld, z_stride, k_scale);
        else gemm_tn_split_kernel<__hip_bfloat16, false><<<grid, 256, 0, stream>>>(A, B, M, N, krows, splits, mean_a, mean_b, out, out_ld, z_stride, k_scale);
    } else if (dtype == BASD_DTYPE_F32) {
        if (vec) gemm_tn_kernel<float, true><<<grid, 256, 0, stream>>>(A, B, M, N, krows, splits, mean_a, mean_b, out, out_ld, z_stride, k_scale);
        else gemm_tn_kernel<float, false><<<grid, 256, 0, stream>>>(A, B, M, N, krows, splits, mean_a, mean_b, out, out_ld, z_stride, k_scale);
    } else if (dtype == BASD_DTYPE_BF16) {
        if (a_sd == 1 && b_sd == 1) gemm_tn_kernel<__hip_bfloat16, true><<<grid, 256, 0, stream>>>(A, B, M, N, krows, splits, mean_a, mean_b, out, out_ld, z_stride, k_scale);
        else gemm_tn_kernel<__hip_bfloat16, false><<<grid, 256, 0, stream>>>(A, B, M, N, krows, splits, mean_a, mean_b, out, out_ld, z_stride, k_scale);
    } else {
        return BASD_EINVAL;
    }
    if (splits > 1) {
        const long count = (long)M * N;
        int blocks = (int)((count + 255) / 256);
        if (blocks > 1024) blocks = 1024;
        reduce_slabs_kernel<<<dim3(blocks, batch), 256, 0, stream>>>(slabs, count, splits, count, scale, c,
                                                                    (long)splits * count, c_batch_stride);
    }
    BASD_RETURN_LAST();
}

int basd_colmean_parts(int rows) {
    // enough row slices to fill the chip (x column blocks x matrices), few enough that the second stage's
    // serial fold stays short
    int p = rows / 256;
    if (p < 1) p = 1;
    if (p > 64) p = 64;
    return p;
}

// mean[z][c] = (1/rows) sum_r X[z](r, c).  `partial`: batch*parts*cols floats of scratch.
int basd_colmean(const void* x, int dtype, long sb, long sn, long sd, int rows_per_batch, long batch_stride,
                 int rows, int cols, int batch, int parts, float* partial, float* mean, hipStream_t stream) {
    BASD_CHECK_ARG(x && partial && mean && rows > 0 && cols > 0 && batch > 0 && parts >= 1);
    GemmOperand X{x, sb, sn, sd, rows_per_batch, batch_stride};
    const int esz = dtype == BASD_DTYPE_F32 ? 4 : 2;
    const bool vec = sd == 1 && cols % 4 == 0 && ((uintptr_t)x & 15) == 0 && (sb * esz) % 16 == 0 &&
                     (sn * esz) % 16 == 0 && (batch_stride * esz) % 16 == 0;
    if (dtype == BASD_DTYPE_F32) launch_colsum<float>(X, nullptr, x, rows, cols, parts, batch, vec, partial, stream);
    else if (dtype == BASD_DTYPE_BF16) launch_colsum<__hip_bfloat16>(X, nullptr, x, rows, cols, parts, batch, vec, partial, stream);
    else return BASD_EINVAL;
    colsum_final_kernel<<<dim3((cols + 63) / 64, batch), 256, 0, stream>>>(partial, cols, parts, 1.f / rows, mean);
    BASD_RETURN_LAST();
}

// Split count for the symmetric Gram launch: the grid (tile pairs x splits x matrices) should fill whole
// rounds of the chip (256 CUs x 3 resident workgroups at this kernel's register budget) -- a 1.3-round
// grid costs two rounds -- with every split keeping >= 8 k-chunks.  Among (near-)equal fills the smallest
// split count wins (less slab traffic).
int basd_syrk_splits(int krows, int cols, int n_mats) {
    const int tiles = (cols + BM - 1) / BM, pairs = tiles * (tiles + 1) / 2;
    const long w = (long)pairs * n_mats;
    const int chunks = (krows + TN_BK - 1) / TN_BK;
    int cap = chunks / 8;
    if (cap > 64) cap = 64;
    if (cap < 1) cap = 1;
    const double slots = 768.0;
    auto fill = [&](int s) {
        const double x = (double)(w * s) / slots;
        const double rounds = x <= 1.0 ? 1.0 : (double)(long)(x + 0.999999);
        return x / rounds;
    };
    double best_eff = 0.0;
    for (int s = 1; s <= cap; ++s) best_eff = fill(s) > best_eff ? fill(s) : best_eff;
    int best = cap;
    for (int s = 1; s <= cap; ++s)
        if (fill(s) >= 0.97 * best_eff) { best = s; break; }
    return best;
}

// out[z] (cols x cols, symmetric) = scales[z] * (X_z - 1 means[z]^T)^T (X_z - 1 means[z]^T),  z < n_mats.
//   x_ptrs: DEVICE array of n_mats base pointers; all X_z share dtype, shape (krows x cols) and strides
//   (element (k, c) at x_ptrs[z] + (k / rows_per_batch)*sb + (k % rows_per_batch)*sn + c*sd).
//   means (nullable): n_mats*cols floats;  scales (nullable): n_mats floats, both on the device.
//   slabs: n_mats*splits*cols*cols floats of scratch, splits from basd_syrk_splits.
int basd_syrk_multi(const void* const* x_ptrs, int dtype, long sb, long sn, long sd, int rows_per_batch, int krows,
                    int cols, int n_mats, const float* means, const float* scales, int splits, float* slabs,
                    float* out, long out_stride, int vec_ok, const float* fold, int fold_parts, int fold_from,
                    hipStream_t stream) {
    BASD_CHECK_ARG(x_ptrs && slabs && out && krows > 0 && cols > 0 && n_mats > 0 && splits >= 1 && rows_per_batch > 0);
    BASD_CHECK_ARG(out_stride >= (long)cols * cols);
    BASD_CHECK_ARG(!fold || (fold_parts > 0 && fold_from >= 0 && fold_from <= n_mats));
    GemmOperand X{nullptr, sb, sn, sd, rows_per_batch, 0};
    const int tiles = (cols + BM - 1) / BM, pairs = tiles * (tiles + 1) / 2;
    const dim3 grid(pairs * splits * n_mats);
    if (g_gemm_split && dtype == BASD_DTYPE_F32) {
        const bool vec = vec_ok && sd == 1 && sb % 4 == 0 && sn % 4 == 0;
        if (vec) syrk_tn_split_kernel<float, true><<<grid, 256, 0, stream>>>(x_ptrs, X, cols, krows, splits, n_mats, pairs, means, slabs, fold, fold_parts, fold_from);
        else syrk_tn_split_kernel<float, false><<<grid, 256, 0, stream>>>(x_ptrs, X, cols, krows, splits, n_mats, pairs, means, slabs, fold, fold_parts, fold_from);
    } else if (g_gemm_split && dtype == BASD_DTYPE_BF16) {
        if (sd == 1) syrk_tn_split_kernel<__hip_bfloat16, true><<<grid, 256, 0, stream>>>(x_ptrs, X, cols, krows, splits, n_mats, pairs, means, slabs, fold, fold_parts, fold_from);
        else syrk_tn_split_kernel<__hip_bfloat16, false><<<grid, 256, 0, stream>>>(x_ptrs, X, cols, krows, splits, n_mats, pairs, means, slabs, fold, fold_parts, fold_from);
    } else if (dtype == BASD_DTYPE_F32) {
        const bool vec = vec_ok && sd == 1 && sb % 4 == 0 && sn % 4 == 0;
        if (vec) syrk_tn_kernel<float, true><<<grid, 256, 0, stream>>>(x_ptrs, X, cols, krows, splits, n_mats, pairs, means, slabs, fold, fold_parts, fold_from);
        else syrk_tn_kernel<float, false><<<grid, 256, 0, stream>>>(x_ptrs, X, cols, krows, splits, n_mats, pairs, means, slabs, fold, fold_parts, fold_from);
    } else if (dtype == BASD_DTYPE_BF16) {
        if (sd == 1) syrk_tn_kernel<__hip_bfloat16, true><<<grid, 256, 0, stream>>>(x_ptrs, X, cols, krows, splits, n_mats, pairs, means, slabs, fold, fold_parts, fold_from);
        else syrk_tn_kernel<__hip_bfloat16, false><<<grid, 256, 0, stream>>>(x_ptrs, X, cols, krows, splits, n_mats, pairs, means, slabs, fold, fold_parts, fold_from);
    } else {
        return BASD_EINVAL;
    }
    syrk_reduce_kernel<<<dim3(pairs * 64, n_mats), 256, 0, stream>>>(slabs, cols, splits, scales, out, out_stride);
    BASD_RETURN_LAST();
}

// means[z][c] = (1/rows) sum_r X_z(r, c) for a device table of same-layout matrices.
//   partial: n_mats*parts*cols floats of scratch (parts from basd_colmean_parts).
int basd_colmean_multi(const void* const* x_ptrs, int dtype, long sb, long sn, long sd, int rows_per_batch, int rows,
                       int cols, int n_mats, int parts, float* partial, float* means, int vec_ok,
                       hipStream_t stream) {
    BASD_CHECK_ARG(x_ptrs && partial && means && rows > 0 && cols > 0 && n_mats > 0 && parts >= 1);
    GemmOperand X{nullptr, sb, sn, sd, rows_per_batch, 0};
    const int esz = dtype == BASD_DTYPE_F32 ? 4 : 2;
    const bool vec = vec_ok && sd == 1 && cols % 4 == 0 && (sb * esz) % 16 == 0 && (sn * esz) % 16 == 0;
    if (dtype == BASD_DTYPE_F32) launch_colsum<float>(X, x_ptrs, nullptr, rows, cols, parts, n_mats, vec, partial, stream);
    else if (dtype == BASD_DTYPE_BF16) launch_colsum<__hip_bfloat16>(X, x_ptrs, nullptr, rows, cols, parts, n_mats, vec, partial, stream);
    else return BASD_EINVAL;
    colsum_final_kernel<<<dim3((cols + 63) / 64, n_mats), 256, 0, stream>>>(partial, cols, parts, 1.f / rows, means);
    BASD_RETURN_LAST();
}

}  // extern "C"
