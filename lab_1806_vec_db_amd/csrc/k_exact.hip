// k_exact.hip -- reference-order (strict left fold, no FMA) f32 distance kernels for gfx950.
//
// Everything in this file must reproduce the arithmetic of /root/reference/src/distance/mod.rs
// bit for bit: products and sums are separately rounded (mod.rs:72-77), accumulation is strictly
// ascending in the dimension.  The file is compiled with -ffp-contract=off and the pragma below
// repeats that, so no v_fma is formed from a*b+c.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace vdb {

enum Fold : int { FOLD_L2 = 0, FOLD_DOT = 1 };

template <int FOLD>
__device__ __forceinline__ float fold1(float acc, float x, float q) {
    if (FOLD == FOLD_L2) {
        float df = x - q;  // (x-q)^2 == (q-x)^2 exactly, so the argument order of flat_index.rs:52 is immaterial
        float sq = df * df;
        return acc + sq;
    } else {
        float p = x * q;
        return acc + p;
    }
}

// epilogue shared by scan / rerank
__device__ __forceinline__ float epilogue(int metric, float acc, float xsq, float qsq) {
    if (metric == MET_L2_DIRECT) return acc;
    if (metric == MET_COSINE) {
        // distance/mod.rs:60-69: 1 - dot / max(|a|*|b|, 1e-10)
        float den = fmaxf(sqrtf(qsq) * sqrtf(xsq), 1e-10f);
        float r = acc / den;
        return 1.0f - r;
    }
    // distance/mod.rs:54-57 with a = stored row, b = query (hnsw_index.rs:351-355)
    float s = xsq + qsq;
    float t = 2.0f * acc;
    return s - t;
}

// ---------------------------------------------------------------------------------------------
// row squared norms (dist_cache, distance/mod.rs:31-36): one thread per row, sequential fold
// ---------------------------------------------------------------------------------------------
template <int BS>
__global__ __launch_bounds__(BS) void k_row_sqnorm(const float *__restrict__ X, uint64_t n, uint32_t dim,
                                                   float *__restrict__ sq) {
    uint64_t r = uint64_t(blockIdx.x) * BS + threadIdx.x;
    if (r >= n) return;
    const float *x = X + r * dim;
    float acc = 0.0f;
    if ((dim & 3) == 0) {
        // the fold is a strict chain, the loads are not: 8 x 16 B per lane in flight (a lane walks its own row, so
        // every load is a separate line; latency, not bandwidth, is what this kernel waits for)
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        const uint32_t nv = dim / 4;
        uint32_t j = 0;
        for (; j + 8 <= nv; j += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = x4[j + u];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                acc = fold1<FOLD_DOT>(acc, v[u].x, v[u].x);
                acc = fold1<FOLD_DOT>(acc, v[u].y, v[u].y);
                acc = fold1<FOLD_DOT>(acc, v[u].z, v[u].z);
                acc = fold1<FOLD_DOT>(acc, v[u].w, v[u].w);
            }
        }
        for (; j < nv; j++) {
            float4 v = x4[j];
            acc = fold1<FOLD_DOT>(acc, v.x, v.x);
            acc = fold1<FOLD_DOT>(acc, v.y, v.y);
            acc = fold1<FOLD_DOT>(acc, v.z, v.z);
            acc = fold1<FOLD_DOT>(acc, v.w, v.w);
        }
    } else {
        for (uint32_t j = 0; j < dim; j++) acc = fold1<FOLD_DOT>(acc, x[j], x[j]);
    }
    sq[r] = acc;
}

void launch_row_sqnorm(const float *X, uint64_t n, uint32_t dim, float *sq, hipStream_t s) {
    if (n == 0) return;
    if (n <= 16384)  // query batches: one wave per workgroup spreads the rows over more CUs
        hipLaunchKernelGGL(k_row_sqnorm<64>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, X, n, dim, sq);
    else
        hipLaunchKernelGGL(k_row_sqnorm<256>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, X, n, dim, sq);
}

// ---------------------------------------------------------------------------------------------
// exact scan, generic variant: one thread per row reading its row straight from global memory.
// Correct for every dim; used for dim % 4 != 0 and as the small-input path.
// ---------------------------------------------------------------------------------------------
template <int BQ, int FOLD>
__global__ __launch_bounds__(256) void k_scan_exact_simple(const float *__restrict__ X, uint64_t n, uint32_t dim,
                                                           const float *__restrict__ Q, int metric,
                                                           const float *__restrict__ xsq,
                                                           const float *__restrict__ qsq, float *__restrict__ out,
                                                           uint64_t ld) {
    uint64_t r = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (r >= n) return;
    const float *x = X + r * dim;
    float acc[BQ];
#pragma unroll
    for (int b = 0; b < BQ; b++) acc[b] = 0.0f;
    if ((dim & 3) == 0) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        for (uint32_t j = 0; j < dim / 4; j++) {
            float4 v = x4[j];
#pragma unroll
            for (int b = 0; b < BQ; b++) {
                const float *q = Q + size_t(b) * dim + 4 * j;  // wave-uniform -> scalar loads
                acc[b] = fold1<FOLD>(acc[b], v.x, q[0]);
                acc[b] = fold1<FOLD>(acc[b], v.y, q[1]);
                acc[b] = fold1<FOLD>(acc[b], v.z, q[2]);
                acc[b] = fold1<FOLD>(acc[b], v.w, q[3]);
            }
        }
    } else {
        for (uint32_t j = 0; j < dim; j++) {
            float v = x[j];
#pragma unroll
            for (int b = 0; b < BQ; b++) acc[b] = fold1<FOLD>(acc[b], v, Q[size_t(b) * dim + j]);
        }
    }
    float xs = (metric == MET_L2_DIRECT) ? 0.0f : xsq[r];
#pragma unroll
    for (int b = 0; b < BQ; b++) {
        float qs = (metric == MET_L2_DIRECT) ? 0.0f : qsq[b];
        out[uint64_t(b) * ld + r] = epilogue(metric, acc[b], xs, qs);
    }
}

// ---------------------------------------------------------------------------------------------
// exact scan, LDS-staged variant (dim % 4 == 0): a 256-row tile is streamed in 32-column chunks
// with fully used 128-B lines (8 lanes x 16 B per row segment), transposed through LDS so that
// thread t folds row t in order.  Row stride 36 floats: for ds_read_b128 the 16 lanes of a
// group hit banks 36*t mod 64 = distinct multiples of 4 -> conflict-free.
// ---------------------------------------------------------------------------------------------
constexpr int TR = 256;
constexpr int CW = 32;
constexpr int LDT = CW + 4;

template <int BQ, int FOLD>
__global__ __launch_bounds__(256) void k_scan_exact_lds(const float *__restrict__ X, uint64_t n, uint32_t dim,
                                                        const float *__restrict__ Q, int metric,
                                                        const float *__restrict__ xsq,
                                                        const float *__restrict__ qsq, float *__restrict__ out,
                                                        uint64_t ld) {
    __shared__ __attribute__((aligned(16))) float tile[2][TR * LDT];
    const uint32_t tid = threadIdx.x;
    const uint64_t row0 = uint64_t(blockIdx.x) * TR;
    const uint32_t nchunk = (dim + CW - 1) / CW;

    float4 stage[8];
    auto load_chunk = [&](uint32_t c) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t f = i * 256 + tid;
            uint32_t r = f >> 3, c4 = f & 7;
            uint64_t row = row0 + r;
            if (row >= n) row = n - 1;
            uint32_t col = c * CW + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < dim) v = *reinterpret_cast<const float4 *>(X + row * dim + col);
            stage[i] = v;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t f = i * 256 + tid;
            uint32_t r = f >> 3, c4 = f & 7;
            *reinterpret_cast<float4 *>(&tile[buf][r * LDT + c4 * 4]) = stage[i];
        }
    };

    float acc[BQ];
#pragma unroll
    for (int b = 0; b < BQ; b++) acc[b] = 0.0f;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (uint32_t c = 0; c < nchunk; c++) {
        int buf = c & 1;
        if (c + 1 < nchunk) load_chunk(c + 1);
        uint32_t cols = dim - c * CW;
        if (cols > CW) cols = CW;
        const float *trow = &tile[buf][tid * LDT];
        for (uint32_t j = 0; j < cols; j += 4) {
            float4 v = *reinterpret_cast<const float4 *>(trow + j);
#pragma unroll
            for (int b = 0; b < BQ; b++) {
                const float *q = Q + size_t(b) * dim + c * CW + j;  // wave-uniform
                acc[b] = fold1<FOLD>(acc[b], v.x, q[0]);
                acc[b] = fold1<FOLD>(acc[b], v.y, q[1]);
                acc[b] = fold1<FOLD>(acc[b], v.z, q[2]);
                acc[b] = fold1<FOLD>(acc[b], v.w, q[3]);
            }
        }
        if (c + 1 < nchunk) store_chunk(buf ^ 1);
        __syncthreads();
    }
    uint64_t r = row0 + tid;
    if (r < n) {
        float xs = (metric == MET_L2_DIRECT) ? 0.0f : xsq[r];
#pragma unroll
        for (int b = 0; b < BQ; b++) {
            float qs = (metric == MET_L2_DIRECT) ? 0.0f : qsq[b];
            out[uint64_t(b) * ld + r] = epilogue(metric, acc[b], xs, qs);
        }
    }
}

template <int BQ>
static void scan_exact_bq(const float *X, uint64_t n, uint32_t dim, const float *Q, int metric, const float *xsq,
                          const float *qsq, float *out, uint64_t ld, bool use_lds, hipStream_t s) {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    bool lds = use_lds && (dim & 3) == 0;
    if (metric == MET_L2_DIRECT) {
        if (lds)
            hipLaunchKernelGGL((k_scan_exact_lds<BQ, FOLD_L2>), grid, block, 0, s, X, n, dim, Q, metric, xsq, qsq, out, ld);
        else
            hipLaunchKernelGGL((k_scan_exact_simple<BQ, FOLD_L2>), grid, block, 0, s, X, n, dim, Q, metric, xsq, qsq, out, ld);
    } else {
        if (lds)
            hipLaunchKernelGGL((k_scan_exact_lds<BQ, FOLD_DOT>), grid, block, 0, s, X, n, dim, Q, metric, xsq, qsq, out, ld);
        else
            hipLaunchKernelGGL((k_scan_exact_simple<BQ, FOLD_DOT>), grid, block, 0, s, X, n, dim, Q, metric, xsq, qsq, out, ld);
    }
}

void launch_scan_exact(const float *X, uint64_t n, uint32_t dim, const float *Q, uint32_t nq, int metric,
                       const float *xsq, const float *qsq, float *out, uint64_t ld, bool use_lds, hipStream_t s) {
    if (n == 0 || nq == 0) return;
    switch (nq) {
        case 1: scan_exact_bq<1>(X, n, dim, Q, metric, xsq, qsq, out, ld, use_lds, s); break;
        case 2: scan_exact_bq<2>(X, n, dim, Q, metric, xsq, qsq, out, ld, use_lds, s); break;
        case 3: scan_exact_bq<3>(X, n, dim, Q, metric, xsq, qsq, out, ld, use_lds, s); break;
        case 4: scan_exact_bq<4>(X, n, dim, Q, metric, xsq, qsq, out, ld, use_lds, s); break;
        case 5: scan_exact_bq<5>(X, n, dim, Q, metric, xsq, qsq, out, ld, use_lds, s); break;
        case 6: scan_exact_bq<6>(X, n, dim, Q, metric, xsq, qsq, out, ld, use_lds, s); break;
        case 7: scan_exact_bq<7>(X, n, dim, Q, metric, xsq, qsq, out, ld, use_lds, s); break;
        case 8: scan_exact_bq<8>(X, n, dim, Q, metric, xsq, qsq, out, ld, use_lds, s); break;
        default: throw Error(1, "launch_scan_exact: nq must be 1..8");
    }
}

// ---------------------------------------------------------------------------------------------
// re-rank (ResultSet::pq_resort's dist_fn, candidate_pair.rs:102-108; also the exact stage after
// the MFMA shortlist): one thread per (query, candidate) folds its row in reference order.
// ---------------------------------------------------------------------------------------------
template <int FOLD>
__global__ __launch_bounds__(64) void k_rerank(const float *__restrict__ X, uint32_t dim,
                                               const float *__restrict__ Q, int metric,
                                               const float *__restrict__ xsq, const float *__restrict__ qsq,
                                               const uint64_t *__restrict__ cand, uint64_t *__restrict__ out,
                                               uint32_t ncand, uint32_t ldc) {
    uint32_t q = blockIdx.y;
    uint32_t j = blockIdx.x * 64 + threadIdx.x;
    if (j >= ncand) {
        if (j < ldc) out[uint64_t(q) * ldc + j] = PAIR_NONE;  // padding of the row: no separate memset of `out`
        return;
    }
    uint64_t c = cand[uint64_t(q) * ldc + j];
    if (c == PAIR_NONE) {
        out[uint64_t(q) * ldc + j] = PAIR_NONE;
        return;
    }
    uint32_t idx = uint32_t(c);
    const float *x = X + uint64_t(idx) * dim;
    const float *qv = Q + uint64_t(q) * dim;  // block-uniform
    float acc = 0.0f;
    if ((dim & 3) == 0) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
#pragma unroll 8
        for (uint32_t t = 0; t < dim / 4; t++) {
            float4 v = x4[t];
            acc = fold1<FOLD>(acc, v.x, qv[4 * t + 0]);
            acc = fold1<FOLD>(acc, v.y, qv[4 * t + 1]);
            acc = fold1<FOLD>(acc, v.z, qv[4 * t + 2]);
            acc = fold1<FOLD>(acc, v.w, qv[4 * t + 3]);
        }
    } else {
        for (uint32_t t = 0; t < dim; t++) acc = fold1<FOLD>(acc, x[t], qv[t]);
    }
    float xs = (metric == MET_L2_DIRECT) ? 0.0f : xsq[idx];
    float qs = (metric == MET_L2_DIRECT) ? 0.0f : qsq[q];
    out[uint64_t(q) * ldc + j] = pair_key(epilogue(metric, acc, xs, qs), idx);
}

// The same fold with the work of a row spread over the 8 lanes that fetch it (k_flat_tail64, round 3).  A wave takes 8 rows:
// lane 8g + p fetches piece p (16 B) of every 32-column chunk of row g -- a whole 128-B line per row and instruction, DEPTH
// chunks ahead -- and turns it into PRODUCTS itself: (x - q)^2 or x * q, each separately rounded exactly as distance/mod.rs:
// 72-77 rounds them; which lane computes a product is immaterial.  The products of a chunk meet in a [8][9]-float4 LDS tile and
// every lane of group g adds row g's 32 values in reference order (the 8 lanes compute the same sum; only one chain matters).
// Against a wave that folds 64 rows with every lane subtracting, multiplying AND adding its own row (96 dependent-issue VALU
// operations per chunk) a chunk costs 4 packed product operations + 32 adds, and the four waves of a workgroup take 8 rows each
// instead of leaving 3 waves idle.  Columns past dim contribute +0.0 (exact: the running sum is never -0.0).
template <int FOLD, int DEPTH>
__device__ __forceinline__ float group_rerank_fold(const float *__restrict__ X, uint32_t dim, uint32_t idx, bool live, const float4 *qs4,
                                                   float4 *tile /* this wave's [8][9] */, uint32_t lane) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const uint32_t d4 = dim / 4, g = lane >> 3, p = lane & 7;
    // EVERY load below is executed by every lane: pieces past the row (dim % 32 != 0) and chunks past the end re-read the
    // row's last piece, rows that are not live read row `idx` = 0 all the same, and what must not count is zeroed as a PRODUCT.
    // A load under a lane condition becomes a branch, and the wait-count pass then gives up counting: it puts
    // `s_waitcnt vmcnt(0)` in front of the first use in every chunk, i.e. every chunk waited for the load issued one chunk
    // before it -- a full memory round trip per chunk whatever DEPTH was (measured with cache-resident rows, -DVDB_TAIL_ABLATE=1:
    // ~960 shader-clock ticks per chunk where the 32 adds take 350).
    (void)live;
    const f4 *rp = reinterpret_cast<const f4 *>(X + uint64_t(idx) * dim);
    const f4 *q4 = reinterpret_cast<const f4 *>(qs4);
    f4 *t4 = reinterpret_cast<f4 *>(tile);
    const uint32_t nch = (d4 + 7) / 8, last = d4 - 1;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    auto piece = [&](uint32_t ch) -> uint32_t {
        const uint32_t i = ch * 8 + p;
        return i < last ? i : last;
    };
    f4 stg[DEPTH];
#pragma unroll
    for (int st = 0; st < DEPTH; st++) stg[st] = rp[piece(uint32_t(st))];
    float acc = 0.0f;
    auto chunk = [&](uint32_t ch, f4 &sg) {
        const bool inside = ch * 8 + p < d4;
        const f4 x = sg, w = q4[piece(ch)];
        f4 pr;
        if (FOLD == FOLD_L2) {
            const f4 df = x - w;
            pr = df * df;
        } else {
            pr = x * w;
        }
        t4[g * 9 + p] = inside ? pr : zero;
        sg = rp[piece(ch + DEPTH)];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const f4 v = t4[g * 9 + j];
            acc = acc + v.x;
            acc = acc + v.y;
            acc = acc + v.z;
            acc = acc + v.w;
        }
    };
    // whole groups of DEPTH chunks without a branch between them (the register ring keeps its places and the waits their counts),
    // then the last nch % DEPTH chunks
    uint32_t ch0 = 0;
    for (; ch0 + DEPTH <= nch; ch0 += DEPTH) {
#pragma unroll
        for (int st = 0; st < DEPTH; st++) chunk(ch0 + st, stg[st]);
    }
#pragma unroll
    for (int st = 0; st < DEPTH - 1; st++)
        if (ch0 + st < nch) chunk(ch0 + st, stg[st]);  // uniform
    return acc;
}

// group_rerank_fold for a wave that takes RW = 8 NP rows (k_flat_tail_lb with fewer than 8 waves per query).  The fetch side is the
// same in NP passes -- in pass i lane 8g + p fetches piece p of every chunk of the wave's row 8i + g and turns it into products --
// and the 64 / RW lanes of chain r add row r's 32 values of a chunk in reference order.  What changes is the instruction count: a
// chunk costs a wave 32 adds + 8 LDS reads whatever RW is, so 63 rows cost 8 waves x 8 rows 8 x (40 + 7) issue slots per chunk and
// 2 waves x 32 rows 2 x (40 + 28).  The exact stage of a 1000-query call keeps every SIMD of the chip busy with several waves,
// where issue slots, not the length of one chain, are what the stage waits for.
template <int FOLD, int DEPTH, int RW>
__device__ __forceinline__ float group_rerank_fold_n(const float *__restrict__ X, uint32_t dim, const uint32_t (&idx)[RW / 8],
                                                     const bool (&live)[RW / 8], const float4 *qs4, float4 *tile /* this wave's [RW][9] */,
                                                     uint32_t lane) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int NP = RW / 8;
    const uint32_t d4 = dim / 4, g = lane >> 3, p = lane & 7, r = lane / (64 / RW);
    (void)live;  // (unconditional loads: see group_rerank_fold)
    const f4 *rp[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) rp[i] = reinterpret_cast<const f4 *>(X + uint64_t(idx[i]) * dim);
    const f4 *q4 = reinterpret_cast<const f4 *>(qs4);
    f4 *t4 = reinterpret_cast<f4 *>(tile);
    const uint32_t nch = (d4 + 7) / 8, last = d4 - 1;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    auto piece = [&](uint32_t ch) -> uint32_t {
        const uint32_t i = ch * 8 + p;
        return i < last ? i : last;
    };
    f4 stg[DEPTH][NP];
#pragma unroll
    for (int st = 0; st < DEPTH; st++)
#pragma unroll
        for (int i = 0; i < NP; i++) stg[st][i] = rp[i][piece(uint32_t(st))];
    float acc = 0.0f;
    auto chunk = [&](uint32_t ch, f4(&sg)[NP]) {
        const bool inside = ch * 8 + p < d4;
        const f4 w = q4[piece(ch)];
        const uint32_t nxp = piece(ch + DEPTH);
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const f4 x = sg[i];
            f4 pr;
            if (FOLD == FOLD_L2) {
                const f4 df = x - w;
                pr = df * df;
            } else {
                pr = x * w;
            }
            t4[(8 * i + g) * 9 + p] = inside ? pr : zero;
            sg[i] = rp[i][nxp];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const f4 v = t4[r * 9 + j];
            acc = acc + v.x;
            acc = acc + v.y;
            acc = acc + v.z;
            acc = acc + v.w;
        }
    };
    uint32_t ch0 = 0;
    for (; ch0 + DEPTH <= nch; ch0 += DEPTH) {
#pragma unroll
        for (int st = 0; st < DEPTH; st++) chunk(ch0 + st, stg[st]);
    }
#pragma unroll
    for (int st = 0; st < DEPTH - 1; st++)
        if (ch0 + st < nch) chunk(ch0 + st, stg[st]);  // uniform
    return acc;
}

// The fold of a 64-row round with ONE chain per lane (k_flat_tail_lb, 4 waves).  In the forms above a row's chain is added by
// all the lanes that fetched the row (8, 4 or 2 of them): a round of 63 rows spends 4 waves x 32 add instructions per chunk on 16
// distinct chains each, and with four workgroups per CU the stage is bound by issue slots (measured: 48k shader-clock ticks per
// round where 960 dependent adds take 10.5k).  Here the four waves only PRODUCE: wave w fetches rows 16w .. 16w + 15 as before (8
// lanes per row and chunk, a whole 128-B line per instruction, DEPTH chunks ahead) and writes the products of a 32-column chunk into
// one of two [64 rows][9] float4 tiles; after the chunk's barrier ONE wave (`consumer`, rotated over the workgroups so that the
// consumers of a CU's workgroups spread over its SIMDs) adds row `lane`'s 32 values in reference order -- 8 LDS reads + 32 adds
// per chunk for all 64 rows.  The reads are split in halves that alternate with the adds of the previous half, so the chain
// never waits for LDS; __syncthreads() is `s_waitcnt lgkmcnt(0); s_barrier` here (row loads stay in flight across it), which
// is also what makes two tiles enough: a tile is rewritten two barriers after the consumer's reads of it were waited for.
// Returns row `lane`'s sum in the consumer wave (garbage elsewhere).  All 256 threads must call it.
template <int FOLD, int DEPTH>
__device__ __forceinline__ float pc_rerank_fold(const float *__restrict__ X, uint32_t dim, const uint32_t (&idx)[2], const bool (&live)[2],
                                                const float4 *qs4, float4 *tiles /* [2][64][9] */, uint32_t wave, uint32_t lane,
                                                bool consumer) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const uint32_t d4 = dim / 4, g = lane >> 3, p = lane & 7;
    (void)live;  // (unconditional loads: see group_rerank_fold)
    const f4 *rp[2];
#pragma unroll
    for (int i = 0; i < 2; i++) rp[i] = reinterpret_cast<const f4 *>(X + uint64_t(idx[i]) * dim);
    const f4 *q4 = reinterpret_cast<const f4 *>(qs4);
    f4 *t4 = reinterpret_cast<f4 *>(tiles);
    const uint32_t nch = (d4 + 7) / 8, last = d4 - 1;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    auto piece = [&](uint32_t ch) -> uint32_t {
        const uint32_t i = ch * 8 + p;
        return i < last ? i : last;
    };
    f4 stg[DEPTH][2];
#pragma unroll
    for (int st = 0; st < DEPTH; st++)
#pragma unroll
        for (int i = 0; i < 2; i++) stg[st][i] = rp[i][piece(uint32_t(st))];
    float acc = 0.0f;
    f4 hb[4] = {zero, zero, zero, zero};  // second half of the previous chunk (+0.0 before the first: exact, the sum is never -0.0)
    uint32_t buf = 0;
    auto chunk = [&](uint32_t ch, f4(&sg)[2]) {
        const bool inside = ch * 8 + p < d4;
        const f4 w = q4[piece(ch)];
        const uint32_t nxp = piece(ch + DEPTH);
        f4 *tb = t4 + buf * (64 * 9);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const f4 x = sg[i];
            f4 pr;
            if (FOLD == FOLD_L2) {
                const f4 df = x - w;
                pr = df * df;
            } else {
                pr = x * w;
            }
            tb[(16 * wave + 8 * i + g) * 9 + p] = inside ? pr : zero;
            sg[i] = rp[i][nxp];
        }
        __syncthreads();
        if (consumer) {
            f4 ha[4];
#pragma unroll
            for (int j = 0; j < 4; j++) ha[j] = tb[lane * 9 + j];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc = acc + hb[j].x;
                acc = acc + hb[j].y;
                acc = acc + hb[j].z;
                acc = acc + hb[j].w;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) hb[j] = tb[lane * 9 + 4 + j];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc = acc + ha[j].x;
                acc = acc + ha[j].y;
                acc = acc + ha[j].z;
                acc = acc + ha[j].w;
            }
        }
        buf ^= 1u;
    };
    // whole groups of DEPTH chunks without a branch between them (the register ring keeps its places and the waits their counts),
    // then the last nch % DEPTH chunks
    uint32_t ch0 = 0;
    for (; ch0 + DEPTH <= nch; ch0 += DEPTH) {
#pragma unroll
        for (int st = 0; st < DEPTH; st++) chunk(ch0 + st, stg[st]);
    }
#pragma unroll
    for (int st = 0; st < DEPTH - 1; st++)
        if (ch0 + st < nch) chunk(ch0 + st, stg[st]);  // uniform
    if (consumer) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc = acc + hb[j].x;
            acc = acc + hb[j].y;
            acc = acc + hb[j].z;
            acc = acc + hb[j].w;
        }
    }
    return acc;
}

// The re-rank of candidate rows with coalesced row fetches, dim % 4 == 0 (PQ-Flat's ef rows per query, the separate-kernels exact
// stage of the Flat pipeline; k_rerank above lets every lane walk its own row 16 B at a time: one load instruction touches 64
// different 128-B lines).  Producer / consumer fold: a workgroup of four waves takes 64 candidates of one query, the four waves
// fetch 16 rows each (8 lanes per row: whole lines) and write products, one wave adds row `lane`.  It replaced k_rerank_t (rounds
// 1 - 3: one wave per 64 candidates with its 8 loads, 8 LDS writes and 96 fold operations per chunk in ONE instruction stream
// and loads under lane conditions, i.e. one memory round trip per chunk): 1000 queries x 100 rows at dim 960 361 -> 70 us
// (profiles/r03_probe_tail_lb_waves.txt), PQ-Flat step 6.49 -> 6.10 ms on the same box.
template <int FOLD>
__global__ __launch_bounds__(256, 4) void k_rerank_pc(const float *__restrict__ X, uint32_t dim, const float *__restrict__ Q,
                                                      int metric, const float *__restrict__ xsq, const float *__restrict__ qsq,
                                                      const uint64_t *__restrict__ cand, uint64_t *__restrict__ out,
                                                      uint32_t ncand_max, uint32_t ldc, const uint32_t *__restrict__ cnt) {
    extern __shared__ float4 rp_smem[];  // [dim/4] query, then two [64 rows][9] float4 product tiles
    __shared__ uint64_t scand[64];
    const uint32_t q = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j0 = blockIdx.x * 64;
    // cnt != null: counted lists (the first cnt[q] slots of a row hold candidates; a count above the capacity = an overflowed list: nothing
    // of it is evaluated, the caller flags the query)
    const uint32_t ncand = cnt ? (cnt[q] <= ncand_max ? cnt[q] : 0u) : ncand_max;
    if (j0 >= ncand) {  // block-uniform: this block only pads the row
        if (threadIdx.x < 64 && j0 + threadIdx.x < ldc) out[uint64_t(q) * ldc + j0 + threadIdx.x] = PAIR_NONE;
        return;
    }
    if (threadIdx.x < 64) scand[threadIdx.x] = j0 + threadIdx.x < ncand ? cand[uint64_t(q) * ldc + j0 + threadIdx.x] : PAIR_NONE;
    const uint32_t d4 = dim / 4;
    float4 *qs4 = rp_smem;
    for (uint32_t i = threadIdx.x; i < d4; i += 256) qs4[i] = reinterpret_cast<const float4 *>(Q + uint64_t(q) * dim)[i];
    __syncthreads();
    uint32_t idx[2];
    bool live[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint64_t c = scand[16 * wave + 8 * i + (lane >> 3)];
        live[i] = c != PAIR_NONE;
        idx[i] = live[i] ? uint32_t(c) : 0u;
    }
    const bool consumer = wave == ((blockIdx.x + blockIdx.y) & 3u);
    const float acc = pc_rerank_fold<FOLD, 3>(X, dim, idx, live, qs4, rp_smem + d4, wave, lane, consumer);
    if (consumer && j0 + lane < ldc) {
        const uint64_t c = scand[lane];
        uint64_t r = PAIR_NONE;
        if (c != PAIR_NONE) {
            const uint32_t ix = uint32_t(c);
            const float xs = (metric == MET_L2_DIRECT) ? 0.0f : xsq[ix];
            const float qs = (metric == MET_L2_DIRECT) ? 0.0f : qsq[q];
            r = pair_key(epilogue(metric, acc, xs, qs), ix);
        }
        out[uint64_t(q) * ldc + j0 + lane] = r;
    }
}

void launch_rerank(const float *X, uint32_t dim, const float *Q, uint32_t nq, int metric, const float *xsq,
                   const float *qsq, const uint64_t *cand, uint64_t *out, uint32_t ncand, uint32_t ldc,
                   hipStream_t s, const uint32_t *cnt) {
    if (nq == 0 || ncand == 0) return;
    if ((dim & 3) == 0 && dim >= 64 && dim <= 8192) {
        dim3 grid((std::max(ncand, ldc) + 63) / 64, nq), block(256);
        const size_t lds = (size_t(dim / 4) + 2 * 64 * 9) * sizeof(float4);
        if (metric == MET_L2_DIRECT)
            hipLaunchKernelGGL((k_rerank_pc<FOLD_L2>), grid, block, lds, s, X, dim, Q, metric, xsq, qsq, cand, out, ncand, ldc, cnt);
        else
            hipLaunchKernelGGL((k_rerank_pc<FOLD_DOT>), grid, block, lds, s, X, dim, Q, metric, xsq, qsq, cand, out, ncand, ldc, cnt);
        return;
    }
    VDB_REQUIRE(cnt == nullptr, "rerank: counted lists need dim % 4 == 0");
    dim3 grid((std::max(ncand, ldc) + 63) / 64, nq), block(64);  // the tail blocks only pad
    if (metric == MET_L2_DIRECT)
        hipLaunchKernelGGL((k_rerank<FOLD_L2>), grid, block, 0, s, X, dim, Q, metric, xsq, qsq, cand, out, ncand, ldc);
    else
        hipLaunchKernelGGL((k_rerank<FOLD_DOT>), grid, block, 0, s, X, dim, Q, metric, xsq, qsq, cand, out, ncand, ldc);
}

// ids 0..n-1 as a candidate row per query (PAIR_NONE pads): "every row is a candidate"
__global__ void k_iota_keys(uint64_t *__restrict__ rows, uint32_t n, uint32_t ld) {
    uint64_t *row = rows + uint64_t(blockIdx.x) * ld;
    for (uint32_t j = threadIdx.x; j < ld; j += blockDim.x) row[j] = j < n ? uint64_t(j) : PAIR_NONE;
}
void launch_iota_keys(uint64_t *rows, uint32_t nq, uint32_t n, uint32_t ld, hipStream_t s) {
    if (nq == 0) return;
    hipLaunchKernelGGL(k_iota_keys, dim3(nq), dim3(256), 0, s, rows, n, ld);
}

// ---------------------------------------------------------------------------------------------
// pair keys -> reference-shaped outputs
// ---------------------------------------------------------------------------------------------
__global__ void k_finalize(const uint64_t *__restrict__ keys, uint32_t ldk, uint32_t nq, uint32_t ksel,
                           uint32_t kstride, uint64_t id_offset, uint64_t *__restrict__ out_idx,
                           float *__restrict__ out_dist, uint64_t *__restrict__ out_count) {
    uint32_t q = blockIdx.x;
    uint32_t cnt = 0;
    for (uint32_t j = threadIdx.x; j < ksel; j += blockDim.x) {
        uint64_t c = keys[uint64_t(q) * ldk + j];
        bool ok = c != PAIR_NONE;
        out_idx[uint64_t(q) * kstride + j] = ok ? uint64_t(uint32_t(c)) + id_offset : 0;
        out_dist[uint64_t(q) * kstride + j] = ok ? f32_from_orderable(uint32_t(c >> 32)) : 0.0f;
        cnt += ok;
    }
    // keys are sorted with PAIR_NONE last, so the count is the number of valid entries
    __shared__ uint32_t total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    atomicAdd(&total, cnt);
    __syncthreads();
    if (threadIdx.x == 0 && out_count) out_count[q] = total;
}

void launch_finalize(const uint64_t *keys, uint32_t ldk, uint32_t nq, uint32_t ksel, uint32_t kstride,
                     uint64_t id_offset, uint64_t *out_idx, float *out_dist, uint64_t *out_count, hipStream_t s) {
    if (nq == 0) return;
    hipLaunchKernelGGL(k_finalize, dim3(nq), dim3(64), 0, s, keys, ldk, nq, ksel, kstride, id_offset, out_idx,
                       out_dist, out_count);
}

// ---------------------------------------------------------------------------------------------
// certification of an MFMA shortlist.
//   approx key(r)  = xsq[r] - 2*S(r,q) (S = MFMA fma chain), approx distance a(r) = key(r) + qsq[q]
//   exact distance e(r) = strict-order f32 fold (the reference's value)
// Both differ from the real-number distance by at most
//   E = 2.5 * (dim+8) * 2^-24 * (sqrt(xsq_max) + sqrt(qsq))^2  +  5e-5 * sqrt(xsq_max * qsq)
// first term: standard gamma_n bounds on an n-term f32 sum of products, doubled for the two computations;
// second term: the split-bf16 contraction of k_flat_mfma -- each product x_j*q_j is off by at most
// 3*2^-17 |x_j q_j| (hi/lo representation of both factors + the dropped lo*lo term), the key carries
// -2*S, and sum |x_j q_j| <= |x||q|: 2 * 3 * 2^-17 = 4.6e-5, rounded up.
// Every row outside the shortlist has key >= kappa (the largest shortlisted key), hence
// e(r) >= kappa + qsq - E.  If the k-th smallest exact distance D_k among the shortlisted rows
// satisfies D_k < kappa + qsq - E, no outside row can enter the exact top-k: certified.
// When the shortlist holds every row (n <= k') it is trivially certified.
// flat_certify_flag (the rule in use) replaces sqrt(xsq_max) by min(sqrt(xsq_max), |q| + sqrt(D_k)): a row whose norm
// exceeds |q| + sqrt(D_k) is farther than D_k by the triangle inequality, whatever its key says.
// ---------------------------------------------------------------------------------------------
//
// Cosine (keys -S/|x|, approximate distance a = 1 + key/|q|): S is off by at most
// ((dim+8)*2^-24 + 3*2^-17) |x||q| in the MFMA form and dim*2^-24 |x||q| in the strict fold, the norms by
// dim*2^-25 relative each, so both distances are within E_cos = 2.5*(dim+8)*2^-24 + 6e-5 of the real one
// -- provided the reference's max(|x||q|, 1e-10) clamp (distance/mod.rs:68) is inactive for every non-zero
// row, which is checked through the smallest positive row norm; otherwise the query is not certified.
// ---------------------------------------------------------------------------------------------
__global__ void k_certify(const uint64_t *__restrict__ exact_sorted, uint32_t lde,
                          const uint64_t *__restrict__ approx_sorted, uint32_t lda, uint32_t nq, uint32_t k,
                          uint32_t kprime, uint64_t n_rows, const float *__restrict__ qsq, float xsq_max,
                          float xsq_min_pos, int cosine, uint32_t dim, uint8_t *__restrict__ flags) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    if (n_rows <= kprime) {
        flags[q] = 0;
        return;
    }
    uint32_t kk = k < kprime ? k : kprime;
    uint64_t ek = exact_sorted[uint64_t(q) * lde + (kk - 1)];
    uint64_t ak = approx_sorted[uint64_t(q) * lda + (kprime - 1)];
    if (ek == PAIR_NONE || ak == PAIR_NONE) {  // cannot happen for n_rows > kprime; be safe
        flags[q] = 1;
        return;
    }
    float dk = f32_from_orderable(uint32_t(ek >> 32));
    float kappa = f32_from_orderable(uint32_t(ak >> 32));
    float qs = qsq[q];
    bool ok;
    if (cosine) {
        float qn = sqrtf(qs);
        float E = 2.5f * float(dim + 8) * 5.9604645e-8f + 6e-5f;
        bool clamp_free = sqrtf(xsq_min_pos) * qn > 1e-9f;  // 10x margin over the 1e-10 clamp
        ok = clamp_free && dk < (1.0f + kappa / qn) - E;
    } else {
        float nrm = sqrtf(xsq_max) + sqrtf(qs);
        float E = 2.5f * float(dim + 8) * 5.9604645e-8f * nrm * nrm + 5e-5f * sqrtf(xsq_max) * sqrtf(qs);
        ok = dk < (kappa + qs) - E;  // NaN anywhere -> not ok
    }
    flags[q] = ok ? 0 : 1;
}

// tau[q] = key of the kprime-th smallest sampled pair (+inf when the sample holds fewer)
__global__ void k_extract_tau(const uint64_t *__restrict__ sorted, uint32_t ld, uint32_t nq, uint32_t kprime,
                              float *__restrict__ tau) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    uint64_t e = sorted[uint64_t(q) * ld + (kprime - 1)];
    tau[q] = e == PAIR_NONE ? INFINITY : f32_from_orderable(uint32_t(e >> 32));
}
void launch_extract_tau(const uint64_t *sorted, uint32_t ld, uint32_t nq, uint32_t kprime, float *tau, hipStream_t s) {
    if (nq == 0) return;
    hipLaunchKernelGGL(k_extract_tau, dim3((nq + 63) / 64), dim3(64), 0, s, sorted, ld, nq, kprime, tau);
}

// flags[q] |= 1 when the filter pass dropped candidates of query q (more hits than slots) or returned fewer than the
// k' the shortlist needs (possible only with a thinned threshold sample, see mfma_sample_plan)
__global__ void k_flag_overflow(const uint32_t *__restrict__ cnt, uint32_t cap, uint32_t min_hits, uint32_t nq,
                                uint8_t *__restrict__ flags) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    if (cnt[q] > cap || cnt[q] < min_hits) flags[q] = 1;
}
void launch_flag_overflow(const uint32_t *cnt, uint32_t cap, uint32_t min_hits, uint32_t nq, uint8_t *flags, hipStream_t s) {
    if (nq == 0) return;
    hipLaunchKernelGGL(k_flag_overflow, dim3((nq + 63) / 64), dim3(64), 0, s, cnt, cap, min_hits, nq, flags);
}

// the certification rule (see the comment above k_certify): ek = the kk-th smallest exact pair of the shortlist, ak = the
// k'-th smallest approximate pair; returns 1 when the exact top-k might not be contained in the shortlist
__device__ __forceinline__ uint8_t flat_certify_flag(uint64_t ek, uint64_t ak, uint32_t q, uint32_t kprime, uint64_t n_rows,
                                                     const float *__restrict__ qsq, float xsq_max, float xsq_min_pos, int cosine,
                                                     uint32_t dim, const SplitErr &se, uint32_t cnt_q, uint32_t cap) {
    uint8_t flag = 0;
    if (n_rows > kprime) {
            if (ek == PAIR_NONE || ak == PAIR_NONE) {
                flag = 1;  // fewer than k' hits (thinned sample) or an emptied overflow list
            } else {
                float dk = f32_from_orderable(uint32_t(ek >> 32));
                float kappa = f32_from_orderable(uint32_t(ak >> 32));
                float qs = qsq[q];
                bool ok;
                if (cosine) {
                    float qn = sqrtf(qs);
                    float split = 6e-5f;  // split-bf16: 3 * 2^-18 relative to |x||q|, with slack
                    if (se.qerr) {        // fp16 operands: measured |dx|/|x| and |dq| (k_half.hip); NaN / inf -> not certified
                        const float qr = se.qerr[q] / qn;
                        split = (se.dx_rel + qr + se.dx_rel * qr) * 1.001f;
                    }
                    float E = 2.5f * float(dim + 8) * 5.9604645e-8f + split;
                    bool clamp_free = sqrtf(xsq_min_pos) * qn > 1e-9f;
                    ok = clamp_free && dk < (1.0f + kappa / qn) - E;
                } else {
                    // Only rows with |x| <= |q| + sqrt(D_k) can have a distance below D_k (triangle inequality; 0.1 % slack
                    // for the roundings of the norms and of the fold), so the error terms are those of rows up to that norm,
                    // not of the largest row of the index: outlier norms do not loosen the bound of ordinary queries.
                    const float qn = sqrtf(qs);
                    const float rx = fminf(sqrtf(xsq_max), (qn + sqrtf(fmaxf(dk, 0.0f))) * 1.001f);
                    float nrm = rx + qn;
                    float split = 5e-5f * rx * qn;  // split-bf16: 2 * 3 * 2^-18 |x||q|, with slack
                    if (se.qerr) {  // the key holds -2S: twice |dx||q| + |x||dq| + |dx||dq|; |dx_r| <= dx_rel |x_r|
                        const float qe = se.qerr[q];
                        const float dxa = fminf(se.dx_abs, se.dx_rel * rx);
                        split = 2.0f * (dxa * qn + rx * qe + dxa * qe) * 1.001f;
                    }
                    float E = 2.5f * float(dim + 8) * 5.9604645e-8f * nrm * nrm + split;
                    ok = dk < (kappa + qs) - E;
                }
                flag = ok ? 0 : 1;
            }
        }
    if (cnt_q > cap || cnt_q < (n_rows > kprime ? kprime : 0)) flag = 1;
    return flag;
}

// certification + overflow check + output formatting of the Flat MFMA pipeline in one launch (one block per query):
// k_certify's test, k_flag_overflow's test and k_finalize's writes (three 5-us launches per step otherwise)
__global__ __launch_bounds__(64) void k_flat_finish(const uint64_t *__restrict__ exact_sorted, uint32_t lde,
                                                    const uint64_t *__restrict__ approx_sorted, uint32_t lda,
                                                    uint32_t ksel, uint32_t kstride, uint32_t kprime, uint64_t n_rows,
                                                    const float *__restrict__ qsq, float xsq_max, float xsq_min_pos,
                                                    int cosine, uint32_t dim, SplitErr se, const uint32_t *__restrict__ cnt,
                                                    uint32_t cap, uint64_t id_offset, uint8_t *__restrict__ flags,
                                                    uint64_t *__restrict__ out_idx, float *__restrict__ out_dist,
                                                    uint64_t *__restrict__ out_count) {
    const uint32_t q = blockIdx.x;
    if (threadIdx.x == 0) {
        uint32_t kk = ksel < kprime ? ksel : kprime;
        uint64_t ek = exact_sorted[uint64_t(q) * lde + (kk - 1)];
        uint64_t ak = approx_sorted[uint64_t(q) * lda + (kprime - 1)];
        flags[q] = flat_certify_flag(ek, ak, q, kprime, n_rows, qsq, xsq_max, xsq_min_pos, cosine, dim, se, cnt[q], cap);
    }
    uint32_t c = 0;
    for (uint32_t j = threadIdx.x; j < ksel; j += blockDim.x) {
        uint64_t e = exact_sorted[uint64_t(q) * lde + j];
        bool ok = e != PAIR_NONE;
        out_idx[uint64_t(q) * kstride + j] = ok ? uint64_t(uint32_t(e)) + id_offset : 0;
        out_dist[uint64_t(q) * kstride + j] = ok ? f32_from_orderable(uint32_t(e >> 32)) : 0.0f;
        c += ok;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
    if (threadIdx.x == 0 && out_count) out_count[q] = c;
}
void launch_flat_finish(const uint64_t *exact_sorted, uint32_t lde, const uint64_t *approx_sorted, uint32_t lda,
                        uint32_t nq, uint32_t ksel, uint32_t kstride, uint32_t kprime, uint64_t n_rows, const float *qsq,
                        float xsq_max, float xsq_min_pos, int cosine, uint32_t dim, SplitErr se, const uint32_t *cnt, uint32_t cap,
                        uint64_t id_offset, uint8_t *flags, uint64_t *out_idx, float *out_dist, uint64_t *out_count,
                        hipStream_t s) {
    if (nq == 0) return;
    hipLaunchKernelGGL(k_flat_finish, dim3(nq), dim3(64), 0, s, exact_sorted, lde, approx_sorted, lda, ksel, kstride,
                       kprime, n_rows, qsq, xsq_max, xsq_min_pos, cosine, dim, se, cnt, cap, id_offset, flags, out_idx,
                       out_dist, out_count);
}

// ---------------------------------------------------------------------------------------------
// The whole exact stage of the Flat pipeline in ONE launch (shortlists of at most 64 rows, dim % 4 == 0): per query a
// workgroup (i) selects the k' smallest approximate pairs of the query's hit list (block_top64, 4 waves), then wave 0
// (ii) re-ranks them in reference order (group_rerank_fold), (iii) sorts the exact pairs across its
// lanes and writes the first ksel, (iv) certifies (flat_certify_flag).  Replaces k_top64_counted + the re-rank kernel +
// k_topk_merge + k_flat_finish: 4 launches and 3 round trips through HBM scratch per step (~110 -> ~60 us per 1000
// queries; at a 125k-row shard the step is 0.5 ms, so this is what is left to trim there).
// ---------------------------------------------------------------------------------------------
template <int FOLD>
__global__ __launch_bounds__(256) void k_flat_tail64(FlatTailArgs a) {
    extern __shared__ float4 ft_smem[];  // [dim/4] query, then 4 waves x [8 rows][9] float4 product tiles
    __shared__ uint64_t sbest[4][64];
    __shared__ uint64_t skeys[64];       // exact pair keys of the shortlist, by shortlist position
    __shared__ uint32_t s_flag;
    const uint32_t q = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t d4 = a.dim / 4;
    float4 *qs4 = ft_smem, *tile = ft_smem + d4 + wave * 72;
    for (uint32_t i = threadIdx.x; i < d4; i += 256) qs4[i] = reinterpret_cast<const float4 *>(a.Q + uint64_t(q) * a.dim)[i];
    const uint32_t cnt_q = a.cnt[q];
    const uint32_t total = cnt_q <= a.cap ? cnt_q : 0;  // cnt > cap: slots are not all written, the query is redone
    const uint64_t best0 = block_top64(a.cand + uint64_t(q) * a.cap, total, sbest);  // (its barrier also covers qs4)
    // the shortlist (ascending approximate pairs, valid in wave 0) for all four waves
    if (wave == 0) sbest[0][lane] = lane < a.kprime ? best0 : PAIR_NONE;
    if (threadIdx.x < 64) skeys[threadIdx.x] = PAIR_NONE;
    __syncthreads();
    const uint32_t kk = a.ksel < a.kprime ? a.ksel : a.kprime;
    // exact pair keys of shortlist positions [j0, j0 + 32): wave w folds positions j0 + 8w .. + 7, one per group of 8 lanes
    auto rerank32 = [&](uint32_t j0) {
        const uint32_t j = j0 + wave * 8 + (lane >> 3);
        const uint64_t c = sbest[0][j];
        const bool live = c != PAIR_NONE;
        const uint32_t idx = live ? uint32_t(c) : 0u;
        const float acc = group_rerank_fold<FOLD, 8>(a.X, a.dim, idx, live, qs4, tile, lane);
        if (live && (lane & 7) == 0) {
            const float xs = (a.metric == MET_L2_DIRECT) ? 0.0f : a.xsq[idx];
            const float qs = (a.metric == MET_L2_DIRECT) ? 0.0f : a.qsq[q];
            skeys[j] = pair_key(epilogue(a.metric, acc, xs, qs), idx);
        }
    };
    // Two stages when the shortlist is longer than 32 rows: the 32 best approximate candidates are re-ranked first and
    // certified against the 33rd-smallest approximate key (every row outside those 32 -- the other shortlisted rows
    // included -- has a key at least that large); only a query that fails this test pays for the other rows' gathers.
    // The exact stage is bound by those gathers (64 rows x dim x 4 B per query), so most queries cost half.
    const bool two_stage = a.kprime > 32 && kk <= 32 && cnt_q <= a.cap;  // block-uniform
    uint64_t sorted = PAIR_NONE;
    uint8_t flag = 1;
    rerank32(0);
    if (!two_stage && a.kprime > 32) rerank32(32);
    __syncthreads();
    if (wave == 0) {
        sorted = sort64(skeys[lane], lane);  // (distance, index) order, PAIR_NONE last
        const uint32_t kp = two_stage ? 32u : a.kprime;
        const uint64_t ek = __shfl(sorted, kk - 1), ak = sbest[0][kp - 1 + (two_stage ? 1 : 0)];  // two-stage: the smallest key OUTSIDE the first 32
        uint8_t f = 1;
        if (lane == 0) {
            f = flat_certify_flag(ek, ak, q, kp, a.n_rows, a.qsq, a.xsq_max, a.xsq_min_pos, a.cosine, a.dim, a.se, cnt_q, a.cap);
            s_flag = f;
        }
        flag = (uint8_t)__shfl((int)f, 0);
    }
    if (two_stage) {
        __syncthreads();
        if (s_flag) {  // block-uniform: the rest of the shortlist
            rerank32(32);
            __syncthreads();
            if (wave == 0) {
                sorted = sort64(skeys[lane], lane);
                const uint64_t ek = __shfl(sorted, kk - 1), ak = sbest[0][a.kprime - 1];
                uint8_t f = 1;
                if (lane == 0)
                    f = flat_certify_flag(ek, ak, q, a.kprime, a.n_rows, a.qsq, a.xsq_max, a.xsq_min_pos, a.cosine, a.dim, a.se, cnt_q, a.cap);
                flag = (uint8_t)__shfl((int)f, 0);
            }
        }
    }
    if (wave != 0) return;
    const bool ok = lane < a.ksel && sorted != PAIR_NONE;
    if (lane < a.ksel) {
        a.out_idx[uint64_t(q) * a.kstride + lane] = ok ? uint64_t(uint32_t(sorted)) + a.id_offset : 0;
        a.out_dist[uint64_t(q) * a.kstride + lane] = ok ? f32_from_orderable(uint32_t(sorted >> 32)) : 0.0f;
    }
    const uint32_t count = __builtin_popcountll(__ballot(ok));
    if (lane == 0) {
        if (a.out_count) a.out_count[q] = count;
        a.flags[q] = flag;
    }
}
bool flat_tail64_supported(uint32_t dim, uint32_t kprime, uint32_t ksel) {
    return (dim & 3) == 0 && dim >= 64 && dim <= 8192 && kprime >= 1 && kprime <= 64 && ksel >= 1 && ksel <= 64;
}
void launch_flat_tail64(const FlatTailArgs &a, uint32_t nq, hipStream_t s) {
    if (nq == 0) return;
    VDB_REQUIRE(flat_tail64_supported(a.dim, a.kprime, a.ksel), "flat_tail64: unsupported shape");
    const size_t lds = (size_t(a.dim / 4) + 4 * 72) * sizeof(float4);
    if (a.metric == MET_L2_DIRECT)
        hipLaunchKernelGGL((k_flat_tail64<FOLD_L2>), dim3(nq), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL((k_flat_tail64<FOLD_DOT>), dim3(nq), dim3(256), lds, s, a);
    VDB_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// The exact stage behind the 8-bit pass (k_gemm8.hip).  Its keys are not estimates but LOWER BOUNDS: D(r, q) >= key(r, q) +
// O_q for every row (k_i8.hip), so nothing has to be known about the rows that were never looked at except the smallest key
// among them.  The workgroup walks the query's hit list in key order, 64 keys per round (block_top64_above: the 64 smallest
// keys above the previous round's last one), re-ranks 63 of them at once (8 waves x 8 rows) in the reference's order, keeps the k smallest
// exact pairs, and stops as soon as
//     D_k  <  (kappa + O_q) (1 - (d + 8) u) - 4 u (|x|max + |q| + 2 |mu|)^2
// with kappa = the smallest key not yet evaluated (tau[q] once the list is exhausted: every row outside it has key > tau).
// The first factor covers the strict fold of a row outside (e >= D (1 - gamma_{d+2})), the last term the two roundings of
// the key's own evaluation.  Most queries stop after 32 .. 96 rows; one that is still open after max_rounds is flagged and
// redone by the next tier.  Cosine (round 4): the same walk with the dot fold and the reference's epilogue on the cached norms;
// the keys then bound the L2Sqr distance of the unit vectors, i.e. twice the cosine distance (flat_certify_lb).
// ---------------------------------------------------------------------------------------------
// Variant of k_flat_tail_lb ("flat_tail_lb_nw": 8 / 4 / 2 / 1 = waves per query with the chains on the fetching lanes, 40 / 41 = four
// waves with one chain per lane, loads 3 / 5 chunks deep).  Measured (profiles/r03_probe_tail_lb_waves.txt; 1000 queries on a 125k-row
// shard | at 1M rows | 32 queries | 1 query, us): 8 waves 97 | 125 | 50 | 35, 4 waves 72 | 110 | 50 | 35, 40: 67 | 106 | 44 | 31,
// 41: 71 | 108 | 45 | 32 (its extra depth costs the spills of a 128-register budget).
#define TAIL_LB_AUTO_NW(nq) 40
template <int NW>
__device__ __forceinline__ uint64_t block_top64_above(const uint64_t *__restrict__ src, uint32_t total, uint64_t (*sbest)[64], uint64_t above) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t batches = (total + 255) / 256;
    uint64_t best = PAIR_NONE;
    for (uint32_t bt = wave; bt < batches; bt += NW) {
        uint64_t r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = (bt * 4 + u) * 64 + lane;
            const uint64_t v = i < total ? src[i] : PAIR_NONE;
            r[u] = v > above ? v : PAIR_NONE;
        }
        const uint64_t tau = __shfl(best, 63);
        uint64_t lo = r[0] < r[1] ? r[0] : r[1], lo2 = r[2] < r[3] ? r[2] : r[3];
        lo = lo < lo2 ? lo : lo2;
        if (__ballot(lo < tau) == 0) continue;  // wave-uniform
        bitonic_sort_from<2, 4>(r, lane);
        const uint64_t a = merge64(r[0], r[1], lane), b = merge64(r[2], r[3], lane);
        best = merge64(best, merge64(a, b, lane), lane);
    }
    sbest[wave][lane] = best;
    __syncthreads();
    if (wave != 0) return PAIR_NONE;
    uint64_t m = sbest[0][lane];
#pragma unroll
    for (int w = 1; w < NW; w++) m = merge64(m, sbest[w][lane], lane);
    return m;
}
__device__ __forceinline__ uint8_t flat_certify_lb(uint64_t ek, float kappa, uint32_t q, const FlatTailArgs &a) {
    if (ek == PAIR_NONE) return 1;
    const float dk = f32_from_orderable(uint32_t(ek >> 32));
    if (a.cosine) {
        // The keys bound the L2Sqr distance of the UNIT vectors: 1 - cos >= (kappa + O_q) / 2 for every row outside (k_i8.hip).  The
        // reference's f32 value of a row outside (strict dot fold, two strict norm folds, sqrt, product, quotient, 1 - r:
        // distance/mod.rs:60-69) is within (2 d + 8) u of the real-number cosine distance -- |dot_f - <x, q>| <= d u |x||q|, each norm
        // (d / 2 + 1) u relative, |cos| <= 1 -- provided the max(|x||q|, 1e-10) clamp is inactive and no norm leaves [1e-30, 1e30]
        // (checked here through the smallest positive row norm of the index; rows the cached |x|^2 does not describe carry keys of
        // -FLT_MAX and are always evaluated).  The key's own two roundings: 2 u (|x~||q~| + |key|) <= 4 u (2 + 2 |mu|)^2 as for
        // L2Sqr with unit norms, halved with the key; 4 u |lower| for kappa + O_q.
        const float qs = a.qsq[q], qn = sqrtf(qs);
        const bool plain = a.xsq_min_pos >= 1e-30f && qs >= 1e-30f && qs <= 1e30f && sqrtf(a.xsq_min_pos) * qn > 1e-9f;
        const float nr = 2.0f + 2.0f * a.se.mu_norm;
        float lower = 0.5f * (kappa + a.se.qoff[q]);
        lower = lower - 4.0f * 5.9604645e-8f * fabsf(lower) - float(2 * a.dim + 16) * 5.9604645e-8f * 1.01f - 2.0f * 5.9604645e-8f * nr * nr;
        return (plain && dk < lower) ? 0 : 1;  // NaN anywhere -> not certified
    }
    const float qn = sqrtf(a.qsq[q]);
    const float rx = fminf(sqrtf(a.xsq_max), (qn + sqrtf(fmaxf(dk, 0.0f))) * 1.001f);  // (flat_certify_flag: why)
    const float nr = rx + qn + 2.0f * a.se.mu_norm;
    float lower = kappa + a.se.qoff[q];
    lower = lower - float(a.dim + 8) * 5.9604645e-8f * 1.01f * fabsf(lower) - 4.0f * 5.9604645e-8f * nr * nr;
    return dk < lower ? 0 : 1;  // NaN anywhere -> not certified
}
// NW waves of 64 (8, 4, 2 or 1): a round = one select + ONE re-rank stage of 63 rows, 64 / NW per wave (the last position of
// the round's 64 keys opens the next round)
template <int FOLD, int NW, int PCD>  // PCD: 0, or the load depth of the producer / consumer fold
__global__ __launch_bounds__(NW * 64, PCD ? 4 : 1) void k_flat_tail_lb(FlatTailArgs a) {
    constexpr bool PC = PCD != 0;
    static_assert(NW == 8 || NW == 4 || NW == 2 || NW == 1, "a stage covers the 64 keys of a round");
    static_assert(!PC || NW == 4, "producer / consumer fold: four waves");
    constexpr int RW = 64 / NW, NP = RW / 8;
    constexpr int DEPTH = PC ? PCD : (NW == 8 ? 8 : (NW == 4 ? 4 : (NW == 2 ? 3 : 2)));  // chunks of row loads in flight per wave (NP KB each)
    extern __shared__ float4 ftl_smem[];  // [dim/4] query, then NW waves x [RW rows][9] float4 product tiles (PC: two [64][9] tiles)
    __shared__ uint64_t sbest[NW][64];
    __shared__ uint64_t skeys[64];
    __shared__ uint32_t s_flag;
    const uint32_t q = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t d4 = a.dim / 4;
    float4 *qs4 = ftl_smem, *tile = ftl_smem + d4 + wave * (RW * 9);
    for (uint32_t i = threadIdx.x; i < d4; i += NW * 64) qs4[i] = reinterpret_cast<const float4 *>(a.Q + uint64_t(q) * a.dim)[i];
    const uint32_t cnt_q = a.cnt[q];
    const uint32_t total = cnt_q <= a.cap ? cnt_q : 0;  // cnt > cap: slots are not all written, the query is redone
    const uint32_t kk = a.ksel;                         // (<= 64)
    const float tau_q = a.tau[q];
    uint64_t run = PAIR_NONE;  // wave 0: the 64 smallest exact pairs so far, ascending across the lanes
    uint64_t above = 0;        // every pair key is > 0 (the smallest orderable float is 0x007fffff)
    uint8_t flag = 1;
    const uint32_t rounds = cnt_q <= a.cap ? a.kprime / 64 : 0;
// measurement builds (make EXTRA=-DVDB_TAIL_STAMPS, run with VDB_TAIL_STAMPS=<call number>): thread 0 of every workgroup stamps the
// shader clock at the phase boundaries of every round (start | select | fold | merge + certify ...), launch_flat_tail_lb prints
// the averages.  -DVDB_TAIL_ABLATE=1 on top: every row index is taken modulo 64 (row fetches hit the caches; results wrong by design).
#ifdef VDB_TAIL_STAMPS
    uint32_t ns = 0;
#define TAIL_STAMP()                                                                                      \
    do {                                                                                                  \
        if (a.stamps && threadIdx.x == 0 && ns < 30) a.stamps[q * 32 + ns++] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define TAIL_STAMP() do { } while (0)
#endif
    TAIL_STAMP();
    uint32_t rounds_done = 0;
    for (uint32_t rd = 0; rd < rounds; rd++) {
        rounds_done = rd + 1;
        const uint64_t best = block_top64_above<NW>(a.cand + uint64_t(q) * a.cap, total, sbest, above);  // (its barrier also covers qs4)
        __syncthreads();  // wave 0 is done with sbest[1..]
        if (wave == 0) sbest[0][lane] = best;
        if (threadIdx.x < 64) skeys[threadIdx.x] = PAIR_NONE;
        __syncthreads();
        TAIL_STAMP();
        // position 63 is not evaluated: it opens the next round
        uint32_t idx[NP];
        bool live[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const uint32_t j = wave * RW + 8 * i + (lane >> 3);
            const uint64_t c = sbest[0][j];
            live[i] = c != PAIR_NONE && j < 63;
            idx[i] = live[i] ? uint32_t(c) : 0u;
#if defined(VDB_TAIL_ABLATE) && VDB_TAIL_ABLATE == 1
            idx[i] &= 63u;
#endif
        }
        float acc;
        if constexpr (PC) {
            const bool consumer = wave == (blockIdx.x & 3u);
            acc = pc_rerank_fold<FOLD, DEPTH>(a.X, a.dim, idx, live, qs4, ftl_smem + d4, wave, lane, consumer);
            const uint64_t c = sbest[0][lane];  // the consumer's lane added row `lane`
            if (consumer && c != PAIR_NONE && lane < 63)
                skeys[lane] = pair_key(FOLD == FOLD_L2 ? acc : epilogue(MET_COSINE, acc, a.xsq[uint32_t(c)], a.qsq[q]), uint32_t(c));
        } else {
            if constexpr (NW == 8)
                acc = group_rerank_fold<FOLD, 8>(a.X, a.dim, idx[0], live[0], qs4, tile, lane);
            else
                acc = group_rerank_fold_n<FOLD, DEPTH, RW>(a.X, a.dim, idx, live, qs4, tile, lane);
            const uint32_t j = wave * RW + lane / NW;  // the row whose chain this lane added (NW lanes each)
            const uint64_t c = sbest[0][j];
            if (c != PAIR_NONE && j < 63 && lane % NW == 0)
                skeys[j] = pair_key(FOLD == FOLD_L2 ? acc : epilogue(MET_COSINE, acc, a.xsq[uint32_t(c)], a.qsq[q]), uint32_t(c));
        }
        __syncthreads();
        TAIL_STAMP();
        if (wave == 0) {
            run = merge64(run, sort64(skeys[lane], lane), lane);
            const uint64_t ek = __shfl(run, kk - 1), nxt = sbest[0][63];
            if (lane == 0) {
                const float kappa = nxt == PAIR_NONE ? tau_q : f32_from_orderable(uint32_t(nxt >> 32));
                s_flag = flat_certify_lb(ek, kappa, q, a) | (nxt == PAIR_NONE ? 2u : 0u);
            }
        }
        __syncthreads();
        TAIL_STAMP();
        const uint32_t f = s_flag;  // block-uniform
        if ((f & 1u) == 0) {
            flag = 0;
            break;
        }
        if (f & 2u) break;  // the hit list is exhausted and the k-th distance is still above tau's bound
        above = sbest[0][62];
        __syncthreads();  // sbest is rewritten by the next round
    }
#ifdef VDB_TAIL_STAMPS
    if (a.stamps && threadIdx.x == 0) {
        a.stamps[q * 32 + 30] = ns;
        a.stamps[q * 32 + 31] = total;
    }
#endif
    if (wave != 0) return;
    const bool ok = lane < a.ksel && run != PAIR_NONE;
    if (lane < a.ksel) {
        a.out_idx[uint64_t(q) * a.kstride + lane] = ok ? uint64_t(uint32_t(run)) + a.id_offset : 0;
        a.out_dist[uint64_t(q) * a.kstride + lane] = ok ? f32_from_orderable(uint32_t(run >> 32)) : 0.0f;
    }
    const uint32_t count = __builtin_popcountll(__ballot(ok));
    if (lane == 0) {
        if (a.out_count) a.out_count[q] = count;
        a.flags[q] = uint8_t(flag | ((rounds_done < 127u ? rounds_done : 127u) << 1));  // bit 0: not certified; bits 1..7: rounds walked (the host's refinement rule)
        if (a.qstat) a.qstat[q] = rounds_done | ((cnt_q < 0xFFFFFFu ? cnt_q : 0xFFFFFFu) << 8);
    }
}
bool flat_tail_lb_supported(uint32_t dim, uint32_t kprime, uint32_t ksel) {
    return (dim & 3) == 0 && dim >= 64 && dim <= 8192 && kprime >= 64 && kprime % 64 == 0 && kprime <= 8192 && ksel >= 1 && ksel <= 64;
}
static std::atomic<int> g_tail_lb_nw{0};  // 0 auto; 8 / 4 / 2 / 1 waves per query
void flat_tail_lb_set_nw(int v) { g_tail_lb_nw = v; }
#ifdef VDB_TAIL_STAMPS
static void tail_stamps_report(unsigned long long *d, uint32_t nq, hipStream_t s) {
    VDB_SYNC(s);
    std::vector<unsigned long long> h(size_t(nq) * 32);
    VDB_HIP(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    VDB_HIP(hipFree(d));
    uint32_t hist[8] = {0}, nr = 0;
    double sel = 0, fold = 0, cert = 0, life = 0;
    for (uint32_t q = 0; q < nq; q++) {
        const unsigned long long *t = &h[size_t(q) * 32];
        const uint32_t ns = (uint32_t)t[30], r = ns ? (ns - 1) / 3 : 0;
        hist[r < 7 ? r : 7]++;
        if (ns) life += double(t[ns - 1] - t[0]);
        for (uint32_t i = 0; i < r; i++, nr++) {
            sel += double(t[1 + 3 * i] - t[3 * i]);
            fold += double(t[2 + 3 * i] - t[1 + 3 * i]);
            cert += double(t[3 + 3 * i] - t[2 + 3 * i]);
        }
    }
    fprintf(stderr, "TAIL_STAMPS nq %u  queries by rounds 0:%u 1:%u 2:%u 3:%u 4:%u 5+:%u  ticks per round: select %.0f fold %.0f merge+certify %.0f  per workgroup %.0f\n",
            nq, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7], sel / (nr ? nr : 1), fold / (nr ? nr : 1),
            cert / (nr ? nr : 1), life / nq);
}
#endif
void launch_flat_tail_lb(const FlatTailArgs &a0, uint32_t nq, hipStream_t s) {
    if (nq == 0) return;
    FlatTailArgs a = a0;
#ifdef VDB_TAIL_STAMPS
    static std::atomic<int> calls{0};
    const char *want = getenv("VDB_TAIL_STAMPS");
    unsigned long long *stamps = nullptr;
    if (want && ++calls == atoi(want)) {
        VDB_HIP(hipMalloc(&stamps, size_t(nq) * 32 * 8));
        VDB_HIP(hipMemset(stamps, 0, size_t(nq) * 32 * 8));
        a.stamps = stamps;
    }
#endif
    VDB_REQUIRE(flat_tail_lb_supported(a.dim, a.kprime, a.ksel) && (a.metric == MET_L2_DIRECT || a.metric == MET_COSINE) && a.se.qoff && a.tau,
                "flat_tail_lb: unsupported shape");
    VDB_REQUIRE((a.metric == MET_COSINE) == (a.cosine != 0) && (a.metric == MET_L2_DIRECT || (a.xsq && a.qsq)), "flat_tail_lb: metric / norms");
    const size_t lds = (size_t(a.dim / 4) + 2 * 64 * 9) * sizeof(float4);
    int nw = g_tail_lb_nw;
    if (nw != 8 && nw != 4 && nw != 2 && nw != 1 && nw != 40 && nw != 41) nw = TAIL_LB_AUTO_NW(nq);
    if (a.metric == MET_COSINE) {  // (Cosine: the producer / consumer form and the plain 8-wave form)
        if (nw == 8)
            hipLaunchKernelGGL((k_flat_tail_lb<FOLD_DOT, 8, 0>), dim3(nq), dim3(512), lds, s, a);
        else if (nw == 41)
            hipLaunchKernelGGL((k_flat_tail_lb<FOLD_DOT, 4, 5>), dim3(nq), dim3(256), lds, s, a);
        else
            hipLaunchKernelGGL((k_flat_tail_lb<FOLD_DOT, 4, 3>), dim3(nq), dim3(256), lds, s, a);
    } else if (nw == 40)  // four waves, producer / consumer fold
        hipLaunchKernelGGL((k_flat_tail_lb<FOLD_L2, 4, 3>), dim3(nq), dim3(256), lds, s, a);
    else if (nw == 41)
        hipLaunchKernelGGL((k_flat_tail_lb<FOLD_L2, 4, 5>), dim3(nq), dim3(256), lds, s, a);
    else if (nw == 8)
        hipLaunchKernelGGL((k_flat_tail_lb<FOLD_L2, 8, 0>), dim3(nq), dim3(512), lds, s, a);
    else if (nw == 4)
        hipLaunchKernelGGL((k_flat_tail_lb<FOLD_L2, 4, 0>), dim3(nq), dim3(256), lds, s, a);
    else if (nw == 2)
        hipLaunchKernelGGL((k_flat_tail_lb<FOLD_L2, 2, 0>), dim3(nq), dim3(128), lds, s, a);
    else
        hipLaunchKernelGGL((k_flat_tail_lb<FOLD_L2, 1, 0>), dim3(nq), dim3(64), lds, s, a);
    VDB_HIP(hipGetLastError());
#ifdef VDB_TAIL_STAMPS
    if (stamps) tail_stamps_report(stamps, nq, s);
#endif
}

// ---------------------------------------------------------------------------------------------------
// The exact stage of the SECOND 8-bit attempt for a handful of queries (k_redo.hip).  Their thresholds were derived from an upper bound
// of the k-th distance, so every row that can matter is in the candidate list; walking it in rounds of 63 rows is a chain of 30 - 130
// dependent rounds for the long lists such queries have (measured: the ~2 % stragglers of loosely clustered data cost 0.5 ms of a 1.5-ms
// step).  Few queries leave the chip idle, so ALL their candidates are evaluated at once instead (k_rerank_pc over the counted lists),
// the k smallest selected, and the query certified against its threshold: every row outside the list has key > tau.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_flat_finish_lb(FlatTailArgs a, const uint64_t *__restrict__ exact_sorted, uint32_t lde) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const uint32_t cnt_q = a.cnt[q];
    const uint64_t e = lane < a.ksel ? exact_sorted[uint64_t(q) * lde + lane] : PAIR_NONE;
    const uint64_t ek = __shfl(e, a.ksel - 1);
    uint8_t flag = 1;
    if (lane == 0 && cnt_q <= a.cap) flag = flat_certify_lb(ek, a.tau[q], q, a);
    const bool ok = lane < a.ksel && e != PAIR_NONE;
    if (lane < a.ksel) {
        a.out_idx[uint64_t(q) * a.kstride + lane] = ok ? uint64_t(uint32_t(e)) + a.id_offset : 0;
        a.out_dist[uint64_t(q) * a.kstride + lane] = ok ? f32_from_orderable(uint32_t(e >> 32)) : 0.0f;
    }
    const uint32_t count = __builtin_popcountll(__ballot(ok));
    if (lane == 0) {
        if (a.out_count) a.out_count[q] = count;
        a.flags[q] = flag;
        if (a.qstat) a.qstat[q] = 0xFFu | ((cnt_q < 0xFFFFFFu ? cnt_q : 0xFFFFFFu) << 8);
    }
}
void launch_flat_full_lb(const FlatTailArgs &a, uint32_t nq, uint64_t *exact_keys /* nq x a.cap */, uint64_t *topk /* nq x topk_capacity(ksel) */,
                         hipStream_t s) {
    if (nq == 0) return;
    VDB_REQUIRE(a.ksel >= 1 && a.ksel <= 64 && (a.dim & 3) == 0 && a.se.qoff && a.tau, "flat_full_lb: unsupported shape");
    launch_rerank(a.X, a.dim, a.Q, nq, a.metric, a.xsq, a.qsq, a.cand, exact_keys, a.cap, a.cap, s, a.cnt);
    launch_topk_merge_counted(exact_keys, a.cap, a.cnt, nq, a.ksel, topk, s);  // (four waves, bitonic: the serial one-wave merge took 49 us over 8192 slots)
    hipLaunchKernelGGL(k_flat_finish_lb, dim3(nq), dim3(64), 0, s, a, topk, topk_capacity(a.ksel));
    VDB_HIP(hipGetLastError());
}

void launch_certify(const uint64_t *exact_sorted, uint32_t lde, const uint64_t *approx_sorted, uint32_t lda,
                    uint32_t nq, uint32_t k, uint32_t kprime, uint64_t n_rows, const float *qsq, float xsq_max,
                    float xsq_min_pos, int cosine, uint32_t dim, uint8_t *flags, hipStream_t s) {
    if (nq == 0) return;
    hipLaunchKernelGGL(k_certify, dim3((nq + 63) / 64), dim3(64), 0, s, exact_sorted, lde, approx_sorted, lda, nq, k,
                       kprime, n_rows, qsq, xsq_max, xsq_min_pos, cosine, dim, flags);
}

}  // namespace vdb
