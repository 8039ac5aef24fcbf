// hnsw.hip -- HNSW on gfx950: host-side graph builder (the reference builds on the CPU too) and the
// GPU search kernel.  Reference: src/index_algorithm/hnsw_index.rs.
//
// Search (knn_with_ef :619-634, knn_pq :672-697) must replay the reference's data-dependent walk
// exactly, because the visit order decides which nodes are ever scored:
//   * distances are the reference's values bit for bit: cached form (cache[idx] + qcache) - 2*dot(row, q)
//     with dot a strict left fold (distance/mod.rs:54-57, hnsw_index.rs:351-355), or the ADC sum;
//   * greedy descent (:306-330): one sweep scores every link of the current node in stored order and
//     moves to each strictly closer one in turn;
//   * level 0 (:258-291): pop the smallest unexpanded pair; stop when it is not `< worst` under the full
//     (distance, index) order (check_candidate, candidate_pair.rs:55-57); score unvisited links in stored
//     order; ResultSet::add admits on strictly smaller DISTANCE (candidate_pair.rs:61-74).
// One 64-lane wave owns one query: the result set is a sorted register list (as in k_topk.hip), the
// candidate queue is an unsorted LDS pool with a wave-parallel min scan (only pairs that pass
// check_candidate when they are found are kept: a pair that fails it then can never pass it later, because
// the worst result only improves, so the walk is unchanged), the visited set is a per-query bitmap in HBM
// updated with atomicOr, and one lane scores one neighbour.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <thread>

#include <atomic>
#include <functional>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "half_rows.hpp"
#include "heap.hpp"
#include "pq_hnsw.hpp"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#pragma clang fp contract(off)

namespace vdb {

// ===================================================================================================
// device side
// ===================================================================================================

struct HnswDev {
    const float *rows;
    const float *xsq;  // dot(x,x) per row
    const uint32_t *level0;
    const uint32_t *len0;
    const uint32_t *upper;
    const uint32_t *upper_len;
    const uint64_t *upper_off;
    uint64_t n;
    uint32_t dim, m, max_m0;
    uint32_t enter_point, enter_level;
    int cosine;
    // PQ (ADC walk)
    const uint8_t *codes;
    const float *cent_cache;
    uint32_t enc_dim, pq_m, pq_kc, n_bits;
    int dma;  // exact level-0 distances through LDS-DMA staging (max_m0 <= 32, dim % 32 == 0)
    const uint32_t *entry0;  // optional, per query of the launch: level-0 entry point (the builder's searches of members above level 0); 0xFFFFFFFF / null = greedy descent from the enter point
    const uint16_t *rows_h;  // row-major fp16 image of the rows (null: no pre-pass), values fp16(x / inv_sx)
    float inv_sx, dx_abs, dx_rel;  // its scale and measured rounding error (Index::half_dx_*)
    uint32_t pool_cap;  // live candidates the LDS pool may hold (<= HNSW_POOL; a test hook lowers it to reach the heap walk)
};

constexpr uint32_t HNSW_POOL = 2048;  // most candidate pool entries per query (LDS)
__host__ __device__ inline uint32_t hnsw_lds_query_off(uint32_t pool_cap) { return (pool_cap * 8u + 15u) & ~15u; }

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint64_t o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}

// exact cached-form distance of row idx to the query held in LDS (strict left fold over the dimension)
__device__ __forceinline__ float hnsw_exact_dist(const HnswDev &g, const float *qlds, float qsq, uint32_t idx) {
    const float *x = g.rows + uint64_t(idx) * g.dim;
    float acc = 0.0f;
    if ((g.dim & 3) == 0) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        const float4 *q4 = reinterpret_cast<const float4 *>(qlds);
#pragma unroll 8
        for (uint32_t t = 0; t < g.dim / 4; t++) {  // (products first, then the chain of adds: see hnsw_exact_dists_regs)
            const float4 v = x4[t];
            const float4 q = q4[t];
            const float p0 = v.x * q.x, p1 = v.y * q.y, p2 = v.z * q.z, p3 = v.w * q.w;
            acc = acc + p0;
            acc = acc + p1;
            acc = acc + p2;
            acc = acc + p3;
        }
    } else {
        for (uint32_t t = 0; t < g.dim; t++) {
            float p = x[t] * qlds[t];
            acc = acc + p;
        }
    }
    float xs = g.xsq[idx];
    if (g.cosine) {  // cosine_distance_cached with norms sqrt(dot) (distance/mod.rs:66-69, :31-36)
        float den = fmaxf(sqrtf(xs) * sqrtf(qsq), 1e-10f);
        float r = acc / den;
        return 1.0f - r;
    }
    float s = xs + qsq;
    float t2 = 2.0f * acc;
    return s - t2;
}

// ADC distance of code row idx (pq_table.rs:239-301); lut in LDS or global.  COS is the table's metric as a compile-time fact: with
// `if (g.cosine)` inside the unrolled lookups every one of a row's 320 lookups carried two branches (1 187 in the kernel).
template <bool COS>
__device__ __forceinline__ float hnsw_adc_dist_t(const HnswDev &g, const float *lut, float qsq, uint32_t idx);
__device__ __forceinline__ float hnsw_adc_dist(const HnswDev &g, const float *lut, float qsq, uint32_t idx) {
    return g.cosine ? hnsw_adc_dist_t<true>(g, lut, qsq, idx) : hnsw_adc_dist_t<false>(g, lut, qsq, idx);
}
template <bool COS>
__device__ __forceinline__ float hnsw_adc_dist_t(const HnswDev &g, const float *lut, float qsq, uint32_t idx) {
    const uint8_t *cr = g.codes + uint64_t(idx) * g.enc_dim;
    float sum = 0.0f, cdp = 0.0f;
    const uint32_t kc = g.pq_kc, m = g.pq_m;
    if (g.n_bits == 4 && (g.enc_dim & 15) == 0 && m == 2 * g.enc_dim) {
        // every nibble is a group (the Gist1M table: 160-B code rows): 16-B code loads, the 8 lookups of a code word
        // issued together, then the strict-order adds (pq_table.rs:254-292).  The byte-at-a-time loop below costs 160
        // dependent byte loads and 320 branches per neighbour (measured 43 us per expansion).
        // all code words of the row are requested before the first lookup (10 words for the Gist1M table): one round trip
        // instead of one per word (measured 11.4 us of distance evaluation per expansion with a one-word look-ahead)
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        constexpr int CW = 10;
        const v4u *cw = reinterpret_cast<const v4u *>(cr);
        const uint32_t nw = g.enc_dim / 16;
        for (uint32_t w0 = 0; w0 < nw; w0 += CW) {
            v4u cv[CW];
            static_for<CW>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                cv[i] = cw[w0 + i < nw ? w0 + i : nw - 1];
            });
            static_for<CW>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const uint32_t w = w0 + i;
                if (w < nw) {  // uniform
                    const uint32_t words[4] = {cv[i].x, cv[i].y, cv[i].z, cv[i].w};
#pragma unroll
                    for (int wi = 0; wi < 4; wi++) {
                        float t[8], c[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) {  // nibble j of the word = group 32w + 8wi + j (low nibble of a byte first)
                            const uint32_t at = (w * 32 + 8 * wi + j) * 16 + ((words[wi] >> (4 * j)) & 0xf);
                            t[j] = lut[at];
                            if (COS) c[j] = g.cent_cache[at];
                        }
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            sum = sum + t[j];
                            if (COS) cdp = cdp + c[j];
                        }
                    }
                }
            });
        }
        if (!COS) return sum;
        float den0 = fmaxf(sqrtf(cdp) * sqrtf(qsq), 1e-10f);
        float r0 = sum / den0;
        return 1.0f - r0;
    }
    for (uint32_t b = 0; b < g.enc_dim; b++) {
        uint32_t u = cr[b];
        if (g.n_bits == 4) {
            uint32_t i = 2 * b;
            if (i < m) {
                sum = sum + lut[i * kc + (u & 0xf)];
                if (COS) cdp = cdp + g.cent_cache[i * kc + (u & 0xf)];
            }
            i++;
            if (i < m) {
                sum = sum + lut[i * kc + (u >> 4)];
                if (COS) cdp = cdp + g.cent_cache[i * kc + (u >> 4)];
            }
        } else {
            sum = sum + lut[b * kc + u];
            if (COS) cdp = cdp + g.cent_cache[b * kc + u];
        }
    }
    if (!COS) return sum;
    float den = fmaxf(sqrtf(cdp) * sqrtf(qsq), 1e-10f);
    float r = sum / den;
    return 1.0f - r;
}

// ---- exact distances of up to 32 neighbours at once, rows staged through LDS by DMA ----------------------------
// The walk's time is the 30 dependent 128-B lines of every neighbour row: one lane folds one row (the fold is a strict
// chain, distance/mod.rs:72-74) and with register destinations it keeps a single line in flight (more lines per lane
// thrash the L1: measured).  `global_load_lds_dwordx4` has no register destination: lane l's 16-B piece lands at
// base + 16*l, so one instruction moves chunk c of all (<= 32) fresh rows into LDS as [chunk][lane] -- which is also
// the conflict-free layout for the lane-private fold.  HNSW_DMA_BUF line buffers of 8 chunks rotate, HNSW_DMA_BUF-1
// lines per row are in flight while the oldest is folded.  The DMAs are issued through inline asm (hipcc would wait
// vmcnt(0) before every LDS read after a tracked LDS-DMA); their completion is counted here: 8 instructions per line.
constexpr int HNSW_DMA_BUF = 4;  // 3 lines per row in flight (measured 2/3/4 buffers: 3.33 / 3.29 / 3.22 ms per 1000 queries; 5 costs a wave of occupancy)
constexpr uint32_t HNSW_DMA_BYTES = HNSW_DMA_BUF * 8 * 512 + 32 * 4;    // line buffers (32-lane stride) + |x|^2 of the rows

#ifndef HNSW_NT
#define HNSW_NT 0  // 1: row lines fetched with the non-temporal policy -- measured 2.3x SLOWER (200k rows, ef = 128: 346k -> 150k QPS): the walk revisits rows through L2; kept as a measurement switch
#endif
#if HNSW_NT
#define HNSW_NT_SUFFIX " nt"
#else
#define HNSW_NT_SUFFIX ""
#endif
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" HNSW_NT_SUFFIX "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void glds4(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// lanes with fresh == true (all < 32) get the cached-form distance of row nb; dim % 32 == 0; at least one lane fresh
__device__ __forceinline__ float hnsw_exact_dists_dma(const HnswDev &g, const float *qlds, float qsq, uint32_t nb, bool fresh,
                                                      unsigned char *stage, uint32_t lane) {
    typedef __attribute__((address_space(3))) unsigned char *lds_p;
    const uint32_t sbase = (uint32_t)(uintptr_t)(lds_p)stage;  // LDS byte address of the staging area
    const uint32_t xs_off = HNSW_DMA_BUF * 8 * 512;
    const char *row = reinterpret_cast<const char *>(g.rows + uint64_t(fresh ? nb : 0) * g.dim);
    const uint32_t nlines = g.dim / 32;
    auto issue = [&](uint32_t L) {  // 8 DMA instructions: the 8 chunks of line L of every fresh row
        const uint32_t b = L % HNSW_DMA_BUF;
        if (fresh) {
#pragma unroll
            for (int c = 0; c < 8; c++) glds16(row + uint64_t(L) * 128 + c * 16, sbase + (b * 8 + c) * 512);
        }
    };
    if (fresh) glds4(g.xsq + nb, sbase + xs_off);  // the oldest DMA: done when the first line is
#pragma unroll
    for (int L = 0; L < HNSW_DMA_BUF - 1; L++)
        if ((uint32_t)L < nlines) issue(L);
    float acc = 0.0f;
    const float4 *q4 = reinterpret_cast<const float4 *>(qlds);
    for (uint32_t L = 0; L < nlines; L++) {
        const uint32_t ahead = L + HNSW_DMA_BUF - 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads of the buffer about to be refilled are done
        if (ahead < nlines) {
            issue(ahead);
            wait_vm<8 * (HNSW_DMA_BUF - 1)>();  // all but the newest HNSW_DMA_BUF-1 lines have landed: line L is in LDS
        } else {
            wait_vm<0>();  // tail of the row: drain
        }
        const uint32_t b = L % HNSW_DMA_BUF;
        const unsigned char *src = stage + (b * 8) * 512 + lane * 16;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float4 v = *reinterpret_cast<const float4 *>(src + c * 512);
            const float4 qq = q4[L * 8 + c];
            float p;
            p = v.x * qq.x; acc = acc + p;
            p = v.y * qq.y; acc = acc + p;
            p = v.z * qq.z; acc = acc + p;
            p = v.w * qq.w; acc = acc + p;
        }
    }
    const float xs = *reinterpret_cast<const float *>(stage + xs_off + lane * 4);
    if (g.cosine) {  // cosine_distance_cached with norms sqrt(dot) (distance/mod.rs:66-69, :31-36)
        float den = fmaxf(sqrtf(xs) * sqrtf(qsq), 1e-10f);
        float r = acc / den;
        return 1.0f - r;
    }
    float s2 = xs + qsq;
    float t2 = 2.0f * acc;
    return s2 - t2;
}

// ---- the same, rows staged through REGISTERS -------------------------------------------------------------------
// What bounds the walk is memory latency per wave, not a rate: 256 queries (one wave per CU) take 2.98 ms, 1024 queries
// (four per CU, all the LDS staging above allows) 3.45 ms, and from 2048 queries on the rate is flat (tools/
// probe_hnsw_batch.py); phase stamps put 13.7 of the 20 us of an expansion in these 30 dependent line waits.  The
// bytes in flight per CU are what to raise, and LDS is the small memory for that (a line of 32 rows is 4 KB): the
// vector registers hold 512 KB per CU.  Here the lines are fetched to registers, HNSW_REG_DEPTH lines (16 registers
// each) in flight per wave, and only the line being folded passes through a 4-KB LDS block to be transposed to the
// lane-per-row layout of the strict fold.  With the candidate pool sized by ef a query needs ~10 KB of LDS, so the
// register file bounds the residency: 2 waves per SIMD x 8 lines = 64 lines in flight per CU against 12 above.
// Loads are line-major (lanes 8g .. 8g+7 fetch the 8 chunks of one row's line: one L1 access per line; a chunk-major
// form like the DMA's touches every line 8 times -- with LDS-DMA that made no difference, 3.58 vs 3.57 ms).  The fresh
// rows are compacted to lanes 0 .. nfresh-1 first, so ceil(nfresh/8) load instructions per line are issued.  LDS
// layout of a line: row (k, g) = 8k + g owns slots 8g .. 8g+7 of block k, chunk c in slot 8g + ((c - g - (k & 1)) mod 8):
// the writes are linear (lane l writes slot l of block k: lane 8g+j fetched chunk (j + g + (k & 1)) mod 8), and for any
// c the 16 rows of a ds_read_b128 lane group sit in 16 different 4-bank groups.
#ifndef HNSW_REG_DEPTH
#define HNSW_REG_DEPTH 8
#endif
#ifndef HNSW_REG_DEPTH4
#define HNSW_REG_DEPTH4 (HNSW_REG_DEPTH / 2)
#endif
#ifndef HNSW_LB
#define HNSW_LB 2
#endif
#ifndef HNSW_FOLD_PRODUCTS
#define HNSW_FOLD_PRODUCTS 1  // 0: the round-3 / early round-4 fold (raw rows through the transpose, products in the folding layout)
#endif
#ifndef HNSW_NG3
#define HNSW_NG3 1  // 0: expansions of 17 .. 24 fresh rows take the four-group form
#endif
#ifndef HNSW_FOLD_FREE
#define HNSW_FOLD_FREE 1  // no scheduling barrier between the chunks of a line (0: pinned chunk by chunk -- 8 % slower per one-query call)
#endif
#ifndef HNSW_ABL
#define HNSW_ABL 0  // measurement builds (tools/hnsw_ablate.sh; WRONG distances): 1 = no row-line loads inside the fold, 2 = no query re-reads, 4 = no LDS transpose
#endif
#ifdef HNSW_STAMP2  // measurement build: wall-clock ticks (100 MHz) inside hnsw_exact_dists_regs: set-up + first loads issued | first line arrived and
                    // transposed | the fold's lines | epilogue; [4] = calls
__device__ unsigned long long g_hnsw_st2[8];
#define HNSW_T2(i)                                                     \
    {                                                                  \
        const unsigned long long _n = wall_clock64();                  \
        if (lane == 0) atomicAdd(&g_hnsw_st2[i], _n - _t2);            \
        _t2 = _n;                                                      \
    }
#else
#define HNSW_T2(i)
#endif
constexpr uint32_t HNSW_REG_STAGE = 2 * 4096;  // two line blocks: the transpose runs one line ahead of the fold
// NG = groups of 8 compacted rows that are fetched (ceil(nfresh / 8) <= NG): after the half-precision pre-pass most
// expansions are left with a handful of rows, and a whole-instruction `if` would cost the counted waits (see below)
template <int NG>
__device__ __forceinline__ float hnsw_exact_dists_regs(const HnswDev &g, const float *qlds, float qsq, uint32_t nb, bool fresh,
                                                       unsigned char *stage, uint32_t lane) {
    constexpr int D = NG >= 3 ? HNSW_REG_DEPTH4 : HNSW_REG_DEPTH;  // 16 NG D registers of lines in flight + 64 of the line being folded
#ifdef HNSW_STAMP2
    unsigned long long _t2 = wall_clock64();
    if (lane == 0) atomicAdd(&g_hnsw_st2[4], 1ull);
#endif
    const uint32_t nlines = g.dim / 32;
    const uint64_t fm = __ballot(fresh);
    const uint32_t nfresh = (uint32_t)__builtin_popcountll(fm);  // <= 32: the fresh lanes are all below max_m0 <= 32
    const uint32_t rank = (uint32_t)__builtin_popcountll(fm & ((1ull << lane) - 1));
    // a permutation of the lanes that brings the r-th fresh neighbour to lane r
    const uint32_t cnb = (uint32_t)__builtin_amdgcn_ds_permute(int((fresh ? rank : nfresh + (lane - rank)) * 4), int(nb));
    const uint32_t gg = lane >> 3, jj = lane & 7;
    // Straight-line code from here on: no lane or line is predicated.  (With `if (fresh)` / `if (L < nlines)` around the
    // loads hipcc's wait-count insertion gave up and put vmcnt(0) before every use: one line in flight, 2x slower than
    // the DMA form.)  Lanes beyond the last fresh row fetch the first fresh row's line again (an L1 hit), line indices
    // are clamped to the last line (the tail refills are L1 hits too) and the fold of a padding line is computed and
    // dropped by a select.
    // (native vectors: with HIP's float4 struct the copies become memcpys the optimiser leaves in scratch)
    const v4f *rp[NG];
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const uint32_t src = 8 * k + gg;
        const uint32_t nbk = __shfl(cnb, src < nfresh ? src : 0u);
        rp[k] = reinterpret_cast<const v4f *>(g.rows + uint64_t(nbk) * g.dim) + ((jj + gg + (k & 1)) & 7);
    }
    const float xs = g.xsq[cnb];  // (lanes >= nfresh hold other valid row numbers: visited neighbours or 0)
    const uint32_t last = nlines - 1;
    v4f buf[D][NG];
    static_for<D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const uint32_t Li = (uint32_t)i < last ? (uint32_t)i : last;
#pragma unroll
        for (int k = 0; k < NG; k++) buf[i][k] = rp[k][Li * 8];
        __builtin_amdgcn_sched_barrier(0);  // loads return in issue order: line 0 has to be the first one issued
    });
    float acc = 0.0f;
    HNSW_T2(0)
    const v4f *q4 = reinterpret_cast<const v4f *>(qlds);
    // reader: lane r < 32 folds compacted row r = 8 kr + gr (lanes >= 32 mirror them, their result is not used)
    const uint32_t kr = (lane >> 3) & 3, gr = lane & 7, rot = (gr + (kr & 1)) & 7;
    uint32_t off[8];
#pragma unroll
    for (int c = 0; c < 8; c++) off[c] = kr * 1024 + (8 * gr + ((c + 8 - rot) & 7)) * 16;
    // The fold of a line is a chain of 32 dependent adds (~256 cycles); an LDS write -> read round trip in front of every
    // line costs about as much again (measured: ~10 us of an expansion's 18 for 30 lines).  So the transpose runs one line
    // ahead through two LDS blocks: while line L is folded from registers (cur / qv), line L+1 is written to the other block
    // and every chunk register is re-read for line L+1 right after its last use.
    static_assert(D % 2 == 0, "the LDS block of a line is chosen by the parity of its ring slot");
#if HNSW_FOLD_PRODUCTS
    // What the transpose carries is the PRODUCTS.  tools/hnsw_ablate.sh: a one-query call's fold is the sum of its instructions' issue
    // times, not a chain with shadows to hide work in -- 30 lines x (32 adds + 16 packed multiplies + 16 ds_read_b128 + NG writes + NG
    // loads) took 7.2 us where the 960 adds alone take 2.05 (5 cycles per dependent add).  In the loaded layout all 64 lanes hold
    // different data (lane 8g+j: chunk (j + g + (k & 1)) mod 8 of row 8k + g), in the folding layout lanes 32 .. 63 mirror lanes 0 .. 31:
    // multiplying BEFORE the transpose takes 2 NG packed multiplies per line instead of 16, and the query is read as the two chunks a
    // lane needs per line (even / odd k) instead of all eight.  Same IEEE products, same order of the adds.
    v4f cur[8];
    const uint32_t cqe = (jj + gg) & 7, cqo = (jj + gg + 1) & 7;  // this lane's query chunk of a line for even / odd k
    v4f qe = q4[cqe], qo = q4[cqo];                               // ... of line 0
    {
#pragma unroll
        for (int k = 0; k < NG; k++) *reinterpret_cast<v4f *>(stage + k * 1024 + 16 * lane) = buf[0][k] * ((k & 1) ? qo : qe);
        const uint32_t Ld = (uint32_t)D < last ? (uint32_t)D : last;
#pragma unroll
        for (int k = 0; k < NG; k++) buf[0][k] = rp[k][Ld * 8];
        const uint32_t Lq = 1u < last ? 1u : last;
        qe = q4[Lq * 8 + cqe];
        qo = q4[Lq * 8 + cqo];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 8; c++) cur[c] = *reinterpret_cast<const v4f *>(stage + off[c]);
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef HNSW_STAMP2
    acc += cur[0].x * 0.0f;  // (the first line must have arrived before the clock is read)
#endif
    HNSW_T2(1)
    for (uint32_t L0 = 0; L0 < nlines; L0 += D) {
        static_for<D>([&](auto ic) {  // (compile-time indices: the line buffers must stay in registers)
            constexpr int i = decltype(ic)::value, in = (i + 1) % D;
            const uint32_t L = L0 + i;
            const uint32_t L2 = L + 2 < last ? L + 2 : last, Lr = L + 1 + D < last ? L + 1 + D : last;
            unsigned char *sb = stage + ((i + 1) & 1) * 4096;  // block of line L+1
#pragma unroll
            for (int k = 0; k < NG; k++) *reinterpret_cast<v4f *>(sb + k * 1024 + 16 * lane) = buf[in][k] * ((k & 1) ? qo : qe);  // (qe / qo: line L+1's)
#pragma unroll
            for (int k = 0; k < NG; k++) buf[in][k] = rp[k][Lr * 8];
            qe = q4[L2 * 8 + cqe];
            qo = q4[L2 * 8 + cqo];
            __builtin_amdgcn_sched_barrier(0);
            float a = acc;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const v4f pc = cur[c];
                a = a + pc.x;
                a = a + pc.y;
                a = a + pc.z;
                a = a + pc.w;
                cur[c] = *reinterpret_cast<const v4f *>(sb + off[c]);
#if !HNSW_FOLD_FREE
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            acc = L < nlines ? a : acc;
#if HNSW_FOLD_FREE < 2
            __builtin_amdgcn_sched_barrier(0);  // (the next line's products stay below this line's adds: they need the query chunks read above)
#endif
        });
    }
#else
    v4f cur[8], qv[8];
    {
#pragma unroll
        for (int k = 0; k < NG; k++) *reinterpret_cast<v4f *>(stage + k * 1024 + 16 * lane) = buf[0][k];
        const uint32_t Ld = (uint32_t)D < last ? (uint32_t)D : last;
#pragma unroll
        for (int k = 0; k < NG; k++) buf[0][k] = rp[k][Ld * 8];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            cur[c] = *reinterpret_cast<const v4f *>(stage + off[c]);
            qv[c] = q4[c];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef HNSW_STAMP2
    acc += cur[0].x * 0.0f;  // (the first line must have arrived before the clock is read)
#endif
    HNSW_T2(1)
    for (uint32_t L0 = 0; L0 < nlines; L0 += D) {
        static_for<D>([&](auto ic) {  // (compile-time indices: the line buffers must stay in registers)
            constexpr int i = decltype(ic)::value, in = (i + 1) % D;
            const uint32_t L = L0 + i;
            const uint32_t L1 = L + 1 < last ? L + 1 : last, Lr = L + 1 + D < last ? L + 1 + D : last;
            unsigned char *sb = stage + ((i + 1) & 1) * 4096;  // block of line L+1
            if (!(HNSW_ABL & 4)) {
#pragma unroll
                for (int k = 0; k < NG; k++) *reinterpret_cast<v4f *>(sb + k * 1024 + 16 * lane) = buf[in][k];
            }
            if (!(HNSW_ABL & 1)) {
#pragma unroll
                for (int k = 0; k < NG; k++) buf[in][k] = rp[k][Lr * 8];
            }
            __builtin_amdgcn_sched_barrier(0);
            // The chain is the 32 dependent adds of a line (10.9 cycles each: tools/fold_chain_probe.cpp); a multiply in front of every
            // add -- what `p = v.x * q.x; a = a + p` compiles to, each add waiting for the multiply issued just before it -- made an
            // element 15.6 cycles.  The products of a chunk are formed first, packed (2 x v_pk_mul_f32: the same IEEE products), one chunk
            // ahead of their adds, so that they issue in the shadow of the previous chunk's adds: 11.9 cycles per element.
            // The chain is the 32 dependent adds of a line (10.9 cycles each: tools/fold_chain_probe.cpp); a multiply in front of every
            // add -- what `p = v.x * q.x; a = a + p` compiles to, each add waiting for the multiply issued just before it -- made an
            // element 15.6 cycles.  The products of a chunk are formed first, packed (2 x v_pk_mul_f32: the same IEEE products), one chunk
            // ahead of their adds: 11.9 cycles per element in the probe.  (Pinning the issue order further -- every LDS read and multiply
            // behind one particular add, statement by statement -- measured 4.5 % SLOWER per call on the same box and was dropped.)
            float a = acc;
            v4f pr = cur[0] * qv[0];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const v4f pc = pr;
                if (c + 1 < 8) pr = cur[c + 1] * qv[c + 1];
                a = a + pc.x;
                a = a + pc.y;
                a = a + pc.z;
                a = a + pc.w;
                if (!(HNSW_ABL & 4)) cur[c] = *reinterpret_cast<const v4f *>(sb + off[c]);
                if (!(HNSW_ABL & 2)) qv[c] = q4[L1 * 8 + c];
                __builtin_amdgcn_sched_barrier(0);
            }
            acc = L < nlines ? a : acc;
        });
    }
#endif
#ifdef HNSW_STAMP2
    acc = __shfl(acc, lane);
#endif
    HNSW_T2(2)
    float d;
    if (g.cosine) {
        float den = fmaxf(sqrtf(xs) * sqrtf(qsq), 1e-10f);
        float r = acc / den;
        d = 1.0f - r;
    } else {
        float s2 = xs + qsq;
        float t2 = 2.0f * acc;
        d = s2 - t2;
    }
    d = __shfl(d, rank);  // back to the lane the neighbour came from
    HNSW_T2(3)
    return d;
}


// ---- certified half-precision pre-pass (half_rows.hpp) ------------------------------------------------------------
// Large calls are bound by the bytes of the row gathers, and 85 % of the rows an expansion scores fail check_candidate
// (measured: 498k of 3.3M pass at ef = 128).  Once the result list is full an expansion first scores all fresh neighbours from
// the row-major fp16 image (half_dots32) and drops every row whose reference distance is certainly above the worst result
// (half_rules_out: a - E > worst  =>  e > worst, so check_candidate and add both fail in the reference too, distance / index
// ties included since the inequality is strict; tau only moves down while the expansion is replayed).  Rows that survive are
// scored exactly (hnsw_exact_dists_regs); the counters count every fresh row as before.

// LUT_LDS (ADC walk, round 4): the query's table is in LDS as a COMPILE-TIME fact.  With a run-time choice between the LDS copy and the
// global table the lookups were generic-address loads (flat_load_dword: 64-bit address arithmetic per lookup, and flat operations complete
// out of order, so every use waited for vmcnt(0) AND lgkmcnt(0)): 969 of them in the kernel.  Known to be LDS they are ds_read_b32 with
// the group's offset as an immediate and counted waits.
template <int R, bool ADC, bool LUT_LDS = false>
__global__ __launch_bounds__(64, HNSW_LB) void k_hnsw_search(HnswDev g, const float *__restrict__ Q,
                                                    const float *__restrict__ qsq_all,
                                                    const float *__restrict__ lut_all, uint32_t lut_in_lds,
                                                    uint32_t ef, uint32_t *__restrict__ visited,
                                                    uint64_t visited_words, uint64_t *__restrict__ out /*[nq][64R]*/,
                                                    unsigned long long *__restrict__ stats /*[2]*/,
                                                    uint32_t *__restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // [ candidate pool: pool_cap pairs | query (or the ADC table) | staging of the exact walk's row lines, 512-B aligned ]
    uint64_t *pool = reinterpret_cast<uint64_t *>(smem_raw);
    const uint32_t fl_off = hnsw_lds_query_off(g.pool_cap);
    float *fl = reinterpret_cast<float *>(smem_raw + fl_off);
    unsigned char *stage = smem_raw + ((fl_off + g.dim * sizeof(float) + 511) & ~size_t(511));
    const bool dma = !ADC && g.dma != 0;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t q = blockIdx.x;
    const float qsq = qsq_all[q];
    uint32_t *vis = visited + uint64_t(q) * visited_words;

    const float *lg = ADC ? lut_all + uint64_t(q) * g.pq_m * g.pq_kc : nullptr;
    (void)lut_in_lds;  // (the launcher picks LUT_LDS from it)
    if (ADC) {
        if constexpr (LUT_LDS)
            for (uint32_t i = lane; i < g.pq_m * g.pq_kc; i += 64) fl[i] = lg[i];
    } else {
        for (uint32_t i = lane; i < g.dim; i += 64) fl[i] = Q[uint64_t(q) * g.dim + i];
    }
    __syncthreads();

    auto dist_of = [&](uint32_t idx) -> float {
        if constexpr (ADC && LUT_LDS)
            return hnsw_adc_dist(g, fl, qsq, idx);  // (fl: provably shared memory -> LDS instructions)
        else
            return ADC ? hnsw_adc_dist(g, lg, qsq, idx) : hnsw_exact_dist(g, fl, qsq, idx);
    };

    unsigned long long n_dist = 0, n_exp = 0, n_drop = 0, n_half = 0;  // (n_half: rows scored by the half-precision pre-pass, n_drop: ruled out by it)

    // ---- greedy descent, levels enter_level .. 1 (hnsw_index.rs:306-350) ---------------------------
    uint32_t cur = g.enter_point;
    uint32_t top_level = g.enter_level;
    if (g.entry0 != nullptr && g.entry0[q] != 0xFFFFFFFFu) {  // block-uniform: start level 0 at the given node
        cur = g.entry0[q];
        top_level = 0;
    }
    float cur_d = 0.0f;
    {
        float d = 0.0f;
        if (lane == 0) d = dist_of(cur);
        cur_d = __shfl(d, 0);
        n_dist++;
    }
    for (uint32_t level = top_level; level >= 1; level--) {
        n_dist++;  // the reference re-scores the current node at the start of every level (:313)
        for (;;) {
            uint64_t slot = g.upper_off[cur] + level - 1;
            uint32_t len = g.upper_len[slot];
            const uint32_t *lk = g.upper + slot * g.m;
            bool moved = false;
            for (uint32_t base = 0; base < len; base += 64) {  // m <= 64 in practice; loop keeps it general
                uint32_t j = base + lane;
                uint32_t nb = j < len ? lk[j] : 0;
                float d = 0.0f;
                if (j < len) d = dist_of(nb);
                uint32_t cnt = len - base < 64 ? len - base : 64;
                n_dist += cnt;
                for (uint32_t t = 0; t < cnt; t++) {  // stored order, strict improvement
                    float dt = __shfl(d, t);
                    uint32_t nt = __shfl(nb, t);
                    if (dt < cur_d) {
                        cur_d = dt;
                        cur = nt;
                        moved = true;
                    }
                }
            }
            if (!moved) break;
        }
    }

    // ---- level 0 best-first search (hnsw_index.rs:258-291) -----------------------------------------
    uint64_t rv[R];
#pragma unroll
    for (int r = 0; r < R; r++) rv[r] = PAIR_NONE;
    uint64_t tau = PAIR_NONE;  // pair at position ef-1 (NONE until the set is full)
    auto rs_insert = [&](uint64_t e) {
        bool placed = false;
#pragma unroll
        for (int r = 0; r < R; r++) {
            uint64_t c = rv[r];
            uint64_t mask = placed ? ~0ull : __ballot(c > e);
            if (mask != 0) {
                uint32_t pos = placed ? 0u : (uint32_t)__builtin_ctzll(mask);
                uint64_t carry = __shfl(c, 63);
                uint64_t up = __shfl_up(c, 1);
                rv[r] = lane < pos ? c : (lane == pos ? e : up);
                e = carry;
                placed = true;
            }
        }
        uint32_t p = ef - 1;
        uint64_t t = PAIR_NONE;
#pragma unroll
        for (int r = 0; r < R; r++)
            if ((p >> 6) == (uint32_t)r) t = __shfl(rv[r], p & 63);
        tau = t;
    };
    uint32_t pool_n = 0;
    bool overflow = false;
    auto pool_push = [&](uint64_t e) {
        if (pool_n == g.pool_cap) {
            // drop pairs that can no longer be expanded (>= worst); they would only ever end the walk
            uint32_t kept = 0;
            for (uint32_t base = 0; base < pool_n; base += 64) {
                uint32_t i = base + lane;
                uint64_t v = i < pool_n ? pool[i] : PAIR_NONE;
                bool live = v < tau;
                uint64_t mask = __ballot(live);
                uint32_t before = __builtin_popcountll(mask & ((1ull << lane) - 1));
                __builtin_amdgcn_wave_barrier();
                if (live) pool[kept + before] = v;  // kept + before <= i: never overwrites unread entries
                __builtin_amdgcn_wave_barrier();
                kept += __builtin_popcountll(mask);
            }
            pool_n = kept;
            if (pool_n == g.pool_cap) {
                overflow = true;
                return;
            }
        }
        if (lane == 0) pool[pool_n] = e;
        pool_n++;
    };

    {
        if (lane == 0) atomicOr(&vis[cur >> 5], 1u << (cur & 31));
        uint64_t e = pair_key(cur_d, cur);
        rs_insert(e);
        pool_push(e);
    }
#ifdef HNSW_STAMP  // measurement build: wall-clock ticks (100 MHz) per phase of the level-0 loop, summed into stats[4..7]
    unsigned long long tk_pop = 0, tk_links = 0, tk_dist = 0, tk_ins = 0;
#define HNSW_TICK(acc, t_prev)                         \
    {                                                  \
        const unsigned long long _n = wall_clock64();  \
        acc += _n - t_prev;                            \
        t_prev = _n;                                   \
    }
#else
#define HNSW_TICK(acc, t_prev)
#endif
    // (Round 4 tried a look-ahead here: the next pop is known before the inserts of an expansion -- the smaller of the pool's runner-up and
    // the smallest new candidate -- so its link row and visited test-and-set can be requested under the inserts and the next pop scan.
    // Correct, counters equal, and worth nothing: -1.5 % ... +2 % per call on the same box, profiles/r04_ab_hnsw_lookahead.txt.  Removed.)
    while (pool_n > 0 && !overflow) {
#ifdef HNSW_STAMP
        unsigned long long t_prev = wall_clock64();
#endif
        __builtin_amdgcn_wave_barrier();
        // pop_first: smallest pair in the pool.  The same pass drops the pairs that can no longer be expanded (>= worst:
        // popping one of them would only end the walk, and only when nothing better is left), so the pool stays at the
        // few hundred live candidates instead of growing towards its capacity and being re-scanned every expansion.
        uint64_t best = PAIR_NONE;  // per lane: its smallest live pair and where the compaction put it
        uint32_t best_pos = 0;
        {
            uint32_t kept = 0;
            for (uint32_t base = 0; base < pool_n; base += 64) {
                uint32_t i = base + lane;
                uint64_t v = i < pool_n ? pool[i] : PAIR_NONE;
                bool live = v < tau;
                uint64_t mask = __ballot(live);
                uint32_t before = __builtin_popcountll(mask & ((1ull << lane) - 1));
                __builtin_amdgcn_wave_barrier();
                if (live) pool[kept + before] = v;  // kept + before <= i: never overwrites unread entries
                __builtin_amdgcn_wave_barrier();
                if (live && v < best) {
                    best = v;
                    best_pos = kept + before;
                }
                kept += __builtin_popcountll(mask);
            }
            pool_n = kept;
        }
        const uint64_t mine = best;
        best = wave_min_u64(best);
        if (best == PAIR_NONE) break;  // check_candidate fails for everything left (size == ef and every pair >= worst)
        // remove it (pairs are unique): its slot is overwritten with PAIR_NONE, which the next scan drops like any pair >= worst
        // (no second pass over the pool to find and move it, as rounds 1 - 3 had)
        if (mine == best) pool[best_pos] = PAIR_NONE;
        n_exp++;
        HNSW_TICK(tk_pop, t_prev)
        const uint32_t p = uint32_t(best);
        const uint32_t *lk = g.level0 + uint64_t(p) * g.max_m0;
        // the link row is fetched whole, in the same round trip as its length (every row has max_m0 slots): a load guarded
        // by `j < len` would wait for the length first -- one more dependent trip in a loop that is a chain of them
        const uint32_t nb0 = lane < g.max_m0 ? lk[lane] : 0u;
        const uint32_t len = g.len0[p];
        for (uint32_t base = 0; base < len && !overflow; base += 64) {
            uint32_t j = base + lane;
            uint32_t nb = 0;
            bool fresh = false;
            if (j < len) {
                nb = base == 0 ? nb0 : lk[j];
                uint32_t bit = 1u << (nb & 31);
                uint32_t old = atomicOr(&vis[nb >> 5], bit);
                fresh = (old & bit) == 0;
            }
            uint64_t fm = __ballot(fresh);
            HNSW_TICK(tk_links, t_prev)
            float d = 0.0f;
            if (dma) {
                if (fm) {  // wave-uniform branches
                    if (g.dma == 2) {
                        d = hnsw_exact_dists_dma(g, fl, qsq, nb, fresh, stage, lane);
                    } else {
                        bool need = fresh;
                        if (g.rows_h != nullptr && tau != PAIR_NONE) {  // (until the result list is full every row is admitted)
                            const float S = half_dots32(g.rows_h, g.dim, g.inv_sx, fl, nb, fresh, lane);
                            const float xs = g.xsq[nb];
                            need = fresh && !half_rules_out(g.cosine ? MET_COSINE : MET_L2_CACHED, g.dim, S, xs, qsq, g.dx_abs, g.dx_rel, f32_from_orderable(uint32_t(tau >> 32)));
                            n_drop += (uint32_t)__builtin_popcountll(__ballot(fresh && !need));
                            n_half += (uint32_t)__builtin_popcountll(fm);
                            d = INFINITY;  // a dropped row fails check_candidate below, as its exact distance would
                        }
                        const uint32_t nneed = (uint32_t)__builtin_popcountll(__ballot(need));
                        float de = 0.0f;
                        if (nneed > 24)
                            de = hnsw_exact_dists_regs<4>(g, fl, qsq, nb, need, stage, lane);
#if HNSW_NG3
                        else if (nneed > 16)  // (17 .. 24 rows -- the typical expansion of a lone walk: three groups of 8, a load, a write and two multiplies less per line)
                            de = hnsw_exact_dists_regs<3>(g, fl, qsq, nb, need, stage, lane);
#else
                        else if (nneed > 16)
                            de = hnsw_exact_dists_regs<4>(g, fl, qsq, nb, need, stage, lane);
#endif
                        else if (nneed > 8)
                            de = hnsw_exact_dists_regs<2>(g, fl, qsq, nb, need, stage, lane);
                        else if (nneed > 0)
                            de = hnsw_exact_dists_regs<1>(g, fl, qsq, nb, need, stage, lane);
                        if (need) d = de;
                    }
                }
            } else if (fresh) {
                d = dist_of(nb);
            }
#ifdef HNSW_STAMP
            d = __shfl(d, lane);  // (the value must have arrived before the clock is read)
#endif
            HNSW_TICK(tk_dist, t_prev)
            n_dist += __builtin_popcountll(fm);
            // the worst pair only ever moves down, so a neighbour that fails check_candidate now fails it at its turn too
            fm = __ballot(fresh && pair_key(d, nb) < tau);
            while (fm) {  // stored order
                uint32_t t = (uint32_t)__builtin_ctzll(fm);
                fm &= fm - 1;
                uint64_t e = pair_key(__shfl(d, t), __shfl(nb, t));
                bool cand = e < tau;                                  // check_candidate before the add
                bool admit = uint32_t(e >> 32) < uint32_t(tau >> 32);  // ResultSet::add: strictly closer or not full
                if (admit) rs_insert(e);
                if (cand) pool_push(e);
            }
            HNSW_TICK(tk_ins, t_prev)
        }
    }
#ifdef HNSW_STAMP
    if (lane == 0) {
        atomicAdd(&stats[4], tk_pop);
        atomicAdd(&stats[5], tk_links);
        atomicAdd(&stats[6], tk_dist);
        atomicAdd(&stats[7], tk_ins);
    }
#endif
    if (overflow && lane == 0) err[q] = 1u;  // this query is answered again by k_hnsw_search_big
#pragma unroll
    for (int r = 0; r < R; r++) out[uint64_t(q) * (64 * R) + r * 64 + lane] = rv[r];
    if (lane == 0 && !overflow) {  // (an overflowed walk is repeated by k_hnsw_search_big, which counts it)
        atomicAdd(&stats[0], n_dist);
        atomicAdd(&stats[1], n_exp);
        if (n_half) {
            atomicAdd(&stats[2], n_drop);
            atomicAdd(&stats[3], n_half);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same walk without size limits: ef beyond the 1024 pairs of the register list, candidate pools beyond the LDS
// pool.  The reference keeps both sets in BTreeSets (hnsw_index.rs:266-268); here the result set is a max-heap and the
// candidate queue a min-heap in global memory, both driven by lane 0 (O(log n) per operation), while the distances of
// a node's links are still computed one per lane.  A node enters the queue at most once (visited set), so n entries
// per query bound it.  Same replay rules as k_hnsw_search: stored link order, check_candidate by the full order at pop
// time, ResultSet::add by distance only; the work counters are the same ones.  The result heap leaves the kernel
// unsorted (PAIR_NONE padded); a row sort follows.
// ---------------------------------------------------------------------------------------------------
template <bool ADC>
__global__ __launch_bounds__(64) void k_hnsw_search_big(HnswDev g, const float *__restrict__ Q, const float *__restrict__ qsq_all,
                                                        const float *__restrict__ lut_all, uint32_t lut_in_lds, uint32_t ef,
                                                        uint32_t *__restrict__ visited, uint64_t visited_words,
                                                        const uint32_t *__restrict__ qlist, uint64_t *__restrict__ cand_heap,
                                                        uint64_t cand_cap, uint64_t *__restrict__ res_heap, uint32_t res_ld,
                                                        unsigned long long *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_big[];
    float *fl = reinterpret_cast<float *>(smem_big);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t slot = blockIdx.x;
    const uint32_t q = qlist[slot];
    const float qsq = qsq_all[q];
    uint32_t *vis = visited + uint64_t(slot) * visited_words;
    uint64_t *ch = cand_heap + uint64_t(slot) * cand_cap;
    uint64_t *rh = res_heap + uint64_t(slot) * res_ld;
    for (uint32_t j = lane; j < res_ld; j += 64) rh[j] = PAIR_NONE;
    const float *lut = nullptr;
    if (ADC) {
        const float *lg = lut_all + uint64_t(q) * g.pq_m * g.pq_kc;
        if (lut_in_lds) {
            for (uint32_t i = lane; i < g.pq_m * g.pq_kc; i += 64) fl[i] = lg[i];
            lut = fl;
        } else {
            lut = lg;
        }
    } else {
        for (uint32_t i = lane; i < g.dim; i += 64) fl[i] = Q[uint64_t(q) * g.dim + i];
    }
    __syncthreads();
    auto dist_of = [&](uint32_t idx) -> float { return ADC ? hnsw_adc_dist(g, lut, qsq, idx) : hnsw_exact_dist(g, fl, qsq, idx); };
    unsigned long long n_dist = 0, n_exp = 0;

    // greedy descent, levels enter_level .. 1 (hnsw_index.rs:306-350): as in k_hnsw_search
    uint32_t cur = g.enter_point;
    float cur_d = 0.0f;
    {
        float d = 0.0f;
        if (lane == 0) d = dist_of(cur);
        cur_d = __shfl(d, 0);
        n_dist++;
    }
    for (uint32_t level = g.enter_level; level >= 1; level--) {
        n_dist++;
        for (;;) {
            const uint64_t sl = g.upper_off[cur] + level - 1;
            const uint32_t len = g.upper_len[sl];
            const uint32_t *lk = g.upper + sl * g.m;
            bool moved = false;
            for (uint32_t base = 0; base < len; base += 64) {
                const uint32_t j = base + lane;
                const uint32_t nb = j < len ? lk[j] : 0;
                float d = 0.0f;
                if (j < len) d = dist_of(nb);
                const uint32_t cnt = len - base < 64 ? len - base : 64;
                n_dist += cnt;
                for (uint32_t t = 0; t < cnt; t++) {
                    const float dt = __shfl(d, t);
                    const uint32_t nt = __shfl(nb, t);
                    if (dt < cur_d) {
                        cur_d = dt;
                        cur = nt;
                        moved = true;
                    }
                }
            }
            if (!moved) break;
        }
    }

    // level 0 (hnsw_index.rs:258-291)
    uint32_t cn = 0, rn = 0;       // heap sizes (lane 0's copies are the live ones)
    uint64_t tau = PAIR_NONE;      // results.last() once the set is full (uniform across the wave)
    {
        const uint64_t e = pair_key(cur_d, cur);
        if (lane == 0) {
            atomicOr(&vis[cur >> 5], 1u << (cur & 31));
            (void)result_heap_add(rh, rn, ef, e);
            heap_min_push(ch, cn, e);
            cn++;
            tau = rn == ef ? rh[0] : PAIR_NONE;
        }
        tau = __shfl(tau, 0);
    }
    for (;;) {
        uint64_t best = PAIR_NONE;
        if (lane == 0 && cn > 0 && ch[0] < tau) best = heap_min_pop(ch, cn);  // check_candidate (candidate_pair.rs:55-57)
        best = __shfl(best, 0);
        if (best == PAIR_NONE) break;  // queue empty, or its smallest pair is not < the worst result
        n_exp++;
        const uint32_t p = uint32_t(best);
        const uint32_t len = g.len0[p];
        const uint32_t *lk = g.level0 + uint64_t(p) * g.max_m0;
        for (uint32_t base = 0; base < len; base += 64) {
            const uint32_t j = base + lane;
            uint32_t nb = 0;
            bool fresh = false;
            if (j < len) {
                nb = lk[j];
                const uint32_t bit = 1u << (nb & 31);
                const uint32_t old = atomicOr(&vis[nb >> 5], bit);
                fresh = (old & bit) == 0;
            }
            uint64_t fm = __ballot(fresh);
            float d = 0.0f;
            if (fresh) d = dist_of(nb);
            n_dist += __builtin_popcountll(fm);
            while (fm) {  // stored order
                const uint32_t t = (uint32_t)__builtin_ctzll(fm);
                fm &= fm - 1;
                const uint64_t e = pair_key(__shfl(d, t), __shfl(nb, t));
                if (lane == 0) {
                    const bool cand = e < tau;  // check_candidate against the set BEFORE this add (never true later if false now)
                    (void)result_heap_add(rh, rn, ef, e);
                    if (cand && cn < cand_cap) {
                        heap_min_push(ch, cn, e);
                        cn++;
                    }
                    tau = rn == ef ? rh[0] : PAIR_NONE;
                }
                tau = __shfl(tau, 0);
            }
        }
    }
    if (lane == 0) {
        atomicAdd(&stats[0], n_dist);
        atomicAdd(&stats[1], n_exp);
    }
}

// ===================================================================================================
// host side: graph builder (hnsw_index.rs:143-256, :391-475, :538-611)
// ===================================================================================================

namespace {

struct Pair {
    float d;
    uint64_t i;
};
inline int f32_total_cmp(float a, float b) {
    bool an = std::isnan(a), bn = std::isnan(b);
    if (an || bn) return int(an) - int(bn);
    return a < b ? -1 : (a > b ? 1 : 0);
}
inline int pair_cmp(const Pair &a, const Pair &b) {
    int c = f32_total_cmp(a.d, b.d);
    if (c) return c;
    return a.i < b.i ? -1 : (a.i > b.i ? 1 : 0);
}
// ResultSet (candidate_pair.rs:43-82) as an ascending vector
struct RSet {
    size_t k;
    std::vector<Pair> v;
    explicit RSet(size_t kk) : k(kk) {}
    void insert_sorted(const Pair &p) {
        auto it = std::lower_bound(v.begin(), v.end(), p, [](const Pair &a, const Pair &b) { return pair_cmp(a, b) < 0; });
        if (it != v.end() && pair_cmp(*it, p) == 0) return;
        v.insert(it, p);
    }
    bool add(const Pair &p) {
        if (v.size() < k) {
            insert_sorted(p);
            return true;
        }
        if (!v.empty() && f32_total_cmp(p.d, v.back().d) < 0) {
            v.pop_back();
            insert_sorted(p);
            return true;
        }
        return false;
    }
    bool check_candidate(const Pair &p) const { return v.size() < k || pair_cmp(p, v.back()) < 0; }
};

struct Scratch {
    std::vector<uint32_t> stamp;
    uint32_t epoch = 0;
    std::vector<Pair> heap;
    void prepare(size_t n) {
        if (stamp.size() < n) {
            stamp.assign(n, 0);
            epoch = 0;
        }
        if (++epoch == 0) {
            std::fill(stamp.begin(), stamp.end(), 0);
            epoch = 1;
        }
        heap.clear();
    }
};

// ---- 8 strict-order dot products at once (host) ---------------------------------------------------------
// The builder spends its time in dist_cached(): a left fold whose adds form one dependent chain per distance
// (distance/mod.rs:72-74), so a scalar core retires one product-add per ~4 cycles.  Eight DIFFERENT distances
// against one fixed vector are eight independent chains: one AVX2 lane each, products and sums rounded separately
// exactly as in the scalar fold (mul then add, no FMA), so every result is bit-identical to dot().  Rows are
// transposed 8 x 8 in registers (each row is read in 32-B pieces).
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void dot8_avx2(const float *fixed, const float *const rows[8], uint64_t dim, float out[8]) {
    __m256 acc = _mm256_setzero_ps();
    uint64_t i = 0;
    for (; i + 8 <= dim; i += 8) {
        __m256 r0 = _mm256_loadu_ps(rows[0] + i), r1 = _mm256_loadu_ps(rows[1] + i);
        __m256 r2 = _mm256_loadu_ps(rows[2] + i), r3 = _mm256_loadu_ps(rows[3] + i);
        __m256 r4 = _mm256_loadu_ps(rows[4] + i), r5 = _mm256_loadu_ps(rows[5] + i);
        __m256 r6 = _mm256_loadu_ps(rows[6] + i), r7 = _mm256_loadu_ps(rows[7] + i);
        __m256 t0 = _mm256_unpacklo_ps(r0, r1), t1 = _mm256_unpackhi_ps(r0, r1);
        __m256 t2 = _mm256_unpacklo_ps(r2, r3), t3 = _mm256_unpackhi_ps(r2, r3);
        __m256 t4 = _mm256_unpacklo_ps(r4, r5), t5 = _mm256_unpackhi_ps(r4, r5);
        __m256 t6 = _mm256_unpacklo_ps(r6, r7), t7 = _mm256_unpackhi_ps(r6, r7);
        __m256 u0 = _mm256_shuffle_ps(t0, t2, 0x44), u1 = _mm256_shuffle_ps(t0, t2, 0xEE);
        __m256 u2 = _mm256_shuffle_ps(t1, t3, 0x44), u3 = _mm256_shuffle_ps(t1, t3, 0xEE);
        __m256 u4 = _mm256_shuffle_ps(t4, t6, 0x44), u5 = _mm256_shuffle_ps(t4, t6, 0xEE);
        __m256 u6 = _mm256_shuffle_ps(t5, t7, 0x44), u7 = _mm256_shuffle_ps(t5, t7, 0xEE);
        __m256 c[8];  // c[k] lane j = rows[j][i + k]
        c[0] = _mm256_permute2f128_ps(u0, u4, 0x20);
        c[1] = _mm256_permute2f128_ps(u1, u5, 0x20);
        c[2] = _mm256_permute2f128_ps(u2, u6, 0x20);
        c[3] = _mm256_permute2f128_ps(u3, u7, 0x20);
        c[4] = _mm256_permute2f128_ps(u0, u4, 0x31);
        c[5] = _mm256_permute2f128_ps(u1, u5, 0x31);
        c[6] = _mm256_permute2f128_ps(u2, u6, 0x31);
        c[7] = _mm256_permute2f128_ps(u3, u7, 0x31);
        for (int k = 0; k < 8; k++) {
            __m256 p = _mm256_mul_ps(_mm256_broadcast_ss(fixed + i + k), c[k]);
            acc = _mm256_add_ps(acc, p);
        }
    }
    alignas(32) float a[8];
    _mm256_store_ps(a, acc);
    for (; i < dim; i++)
        for (int j = 0; j < 8; j++) {
            float p = fixed[i] * rows[j][i];
            a[j] = a[j] + p;
        }
    for (int j = 0; j < 8; j++) out[j] = a[j];
}
static const bool g_have_avx2 = __builtin_cpu_supports("avx2");
#else
static const bool g_have_avx2 = false;
#endif

struct Builder {
    HNSWState &h;
    const float *rows;
    const float *sq;  // dot(x,x) per row; dist_cache = sq (L2Sqr) or sqrt(sq) (Cosine), distance/mod.rs:31-36
    uint64_t dim;
    int dist;
    struct CacheView {
        const float *sq;
        int dist;
        float operator[](uint64_t i) const { return dist == 0 ? sq[i] : std::sqrt(sq[i]); }
    };
    CacheView cache{nullptr, 0};
    double t_search = 0, t_connect = 0;  // wall seconds of the parallel candidate phase / the serial linking phase
    double t_gpu = 0, t_sync = 0;        // of which: inside GpuAssist::search / ::sync
    // GPU assistance of add_batch (hnsw_build): the level-0 candidate search of a whole batch against the pre-batch
    // graph IS HNSWIndex::knn_with_ef(k = ef = ef_construction) -- one k_hnsw_search launch over the device mirror of
    // the graph -- and the distances between batch members are one all-pairs launch.  Same values, same sets as the
    // host search (the kernel replays the reference's search bit for bit, asserted by the GPU parity tests), so the graph does not change.
    struct GpuAssist {
        uint64_t min_batch = 256;  // smaller batches cannot fill the GPU: one query is one wave
        // keys [nb][cape]: the sorted result set of every batch member (PAIR_NONE padded); ok[i] = 0: search it on the host
        // (pool overflow); cross [nb][ldx]: pair keys (distance of member i to member r, first + r)
        // entry0 [nb]: level-0 entry of a member whose upper levels were searched on the host, 0xFFFFFFFF = greedy descent
        std::function<void(uint64_t first, uint64_t nb, uint64_t enter_point, uint64_t enter_level, const std::vector<uint32_t> &entry0,
                           std::vector<uint64_t> &keys, uint32_t &cape, std::vector<uint64_t> &cross, uint32_t &ldx,
                           std::vector<uint8_t> &ok)> search;
        std::function<void(const std::vector<uint64_t> &dirty /* (owner << 8) | level */)> sync;
    };
    GpuAssist *gpu = nullptr;
    Builder(HNSWState &hh, const float *r, const float *s, uint64_t d, int ds)
        : h(hh), rows(r), sq(s), dim(d), dist(ds), cache{s, ds} {}

    float dot(const float *a, const float *b) const {
        float acc = 0.0f;
        for (uint64_t i = 0; i < dim; i++) {
            float p = a[i] * b[i];
            acc = acc + p;
        }
        return acc;
    }
    // DistanceAdapter<(&[T],f32),(&[T],f32)> (distance/mod.rs:120-129)
    float dist_cached(const float *a, const float *b, float ca, float cb) const {
        if (dist == 0) {
            float s = ca + cb;
            float t = 2.0f * dot(a, b);
            return s - t;
        }
        float den = std::fmax(ca * cb, 1e-10f);
        return 1.0f - dot(a, b) / den;
    }
    float to_query(uint64_t idx, const float *q, float qc) const { return dist_cached(rows + idx * dim, q, cache[idx], qc); }
    // dist_cached(rows[ids[j]], fixed) for j < cnt: the same values as to_query / inner one by one (every factor pair
    // and every sum is rounded identically; the operands of each commutative step are merely swapped)
    void dist_many(const float *fixed, float cfixed, const uint32_t *ids, size_t cnt, float *out) const {
        size_t j = 0;
#if defined(__x86_64__)
        if (g_have_avx2) {
            while (cnt - j >= 3) {  // groups of up to 8; a short group is padded with its first row
                const size_t m = cnt - j < 8 ? cnt - j : 8;
                const float *r[8];
                for (size_t t = 0; t < 8; t++) r[t] = rows + uint64_t(ids[j + (t < m ? t : 0)]) * dim;
                float d8[8];
                dot8_avx2(fixed, r, dim, d8);
                for (size_t t = 0; t < m; t++) {
                    const float cb = cache[ids[j + t]];
                    if (dist == 0) {
                        float s2 = cb + cfixed;
                        float t2 = 2.0f * d8[t];
                        out[j + t] = s2 - t2;
                    } else {
                        float den = std::fmax(cb * cfixed, 1e-10f);
                        out[j + t] = 1.0f - d8[t] / den;
                    }
                }
                j += m;
            }
        }
#endif
        for (; j < cnt; j++) out[j] = dist_cached(rows + uint64_t(ids[j]) * dim, fixed, cache[ids[j]], cfixed);
    }
    float inner(uint64_t a, uint64_t b) const { return dist_cached(rows + a * dim, rows + b * dim, cache[a], cache[b]); }

    const uint32_t *links(uint64_t v, uint64_t level, size_t &len) const {
        if (level == 0) {
            len = h.len0[v];
            return h.level0.data() + v * h.max_m0;
        }
        uint64_t slot = h.upper_off[v] + level - 1;
        len = h.upper_len[slot];
        return h.upper.data() + slot * h.m;
    }
    void put_links(uint64_t v, uint64_t level, const std::vector<uint32_t> &l) {
        if (level == 0) {
            h.len0[v] = l.size();
            std::copy(l.begin(), l.end(), h.level0.begin() + v * h.max_m0);
        } else {
            uint64_t slot = h.upper_off[v] + level - 1;
            h.upper_len[slot] = l.size();
            std::copy(l.begin(), l.end(), h.upper.begin() + slot * h.m);
        }
    }
    // search_on_level (hnsw_index.rs:258-304)
    RSet search_on_level(uint64_t ep, uint64_t level, size_t ef, const float *q, float qc, Scratch &s) const {
        s.prepare(h.len0.size());
        RSet res(ef);
        auto cmp = [](const Pair &a, const Pair &b) { return pair_cmp(a, b) > 0; };  // min-heap
        s.stamp[ep] = s.epoch;
        Pair e{to_query(ep, q, qc), ep};
        res.add(e);
        s.heap.push_back(e);
        while (!s.heap.empty()) {
            std::pop_heap(s.heap.begin(), s.heap.end(), cmp);
            Pair p = s.heap.back();
            s.heap.pop_back();
            if (!res.check_candidate(p)) break;
            size_t len;
            const uint32_t *lk = links(p.i, level, len);
            // unvisited neighbours in link order, their distances 8 at a time, then the same add / push sequence
            uint32_t fresh[256];
            float fd[256];
            size_t nf = 0;
            for (size_t j = 0; j < len; j++) {
                uint64_t nb = lk[j];
                if (s.stamp[nb] == s.epoch) continue;
                s.stamp[nb] = s.epoch;
                fresh[nf++] = (uint32_t)nb;
                if (nf == 256) {  // max_m0 <= 20000 in the reference; flush in blocks
                    dist_many(q, qc, fresh, nf, fd);
                    for (size_t t = 0; t < nf; t++) {
                        Pair np{fd[t], fresh[t]};
                        res.add(np);
                        s.heap.push_back(np);
                        std::push_heap(s.heap.begin(), s.heap.end(), cmp);
                    }
                    nf = 0;
                }
            }
            dist_many(q, qc, fresh, nf, fd);
            for (size_t t = 0; t < nf; t++) {
                Pair np{fd[t], fresh[t]};
                res.add(np);
                s.heap.push_back(np);
                std::push_heap(s.heap.begin(), s.heap.end(), cmp);
            }
        }
        return res;
    }
    // greedy_search_until_level (hnsw_index.rs:306-350)
    uint64_t greedy_until(uint64_t target, const float *q, float qc) const {
        uint64_t level = h.enter_level, cur = h.enter_point;
        while (level > target) {
            float cur_d = to_query(cur, q, qc);
            for (;;) {
                bool flag = false;
                size_t len;
                const uint32_t *lk = links(cur, level, len);
                std::vector<float> nd(len);
                dist_many(q, qc, lk, len, nd.data());
                for (size_t j = 0; j < len; j++) {
                    if (nd[j] < cur_d) {
                        cur_d = nd[j];
                        cur = lk[j];
                        flag = true;
                    }
                }
                if (!flag) break;
            }
            level--;
        }
        return cur;
    }
    // ResultSet::heuristic (candidate_pair.rs:85-99)
    std::vector<uint32_t> heuristic(const RSet &set, size_t m) const {
        std::vector<uint32_t> out;
        std::vector<float> dd(m);
        for (const Pair &p : set.v) {
            if (out.size() >= m) break;
            bool ok = true;
            // all inner(p, t) of the selected t at once; the first failing one decides, as in the early-exit loop
            dist_many(rows + p.i * dim, cache[p.i], out.data(), out.size(), dd.data());
            for (size_t t = 0; t < out.size(); t++)
                if (!(dd[t] >= p.d)) {
                    ok = false;
                    break;
                }
            if (ok) out.push_back((uint32_t)p.i);
        }
        return out;
    }
    // arrange_links (hnsw_index.rs:204-224)
    void arrange_links(uint64_t v, uint64_t level, uint64_t newv) {
        size_t limit = level == 0 ? h.max_m0 : h.m;
        size_t len;
        const uint32_t *lk = links(v, level, len);
        std::vector<uint32_t> l(lk, lk + len);
        l.push_back((uint32_t)newv);
        if (l.size() <= limit) {
            put_links(v, level, l);
            return;
        }
        RSet set(limit + 1);
        std::vector<float> dv(l.size());
        dist_many(rows + v * dim, cache[v], l.data(), l.size(), dv.data());
        for (size_t t = 0; t < l.size(); t++) set.add(Pair{dv[t], l[t]});
        put_links(v, level, heuristic(set, limit));
    }
    // connect_new_links (hnsw_index.rs:226-239)
    void connect_new_links(uint64_t v, uint64_t level, const RSet &cand) {
        auto nb = heuristic(cand, h.m);  // M, not max_m0, even on level 0
        put_links(v, level, nb);
        for (uint32_t t : nb) arrange_links(t, level, v);
    }
    // push_init (hnsw_index.rs:244-256) for a row that is already in the VecSet
    void push_init(uint64_t idx, uint64_t level) {
        h.level0.resize((idx + 1) * h.max_m0, 0);
        h.len0.resize(idx + 1, 0);
        h.vec_level.resize(idx + 1, level);
        h.vec_level[idx] = level;
        h.upper_off.resize(idx + 2);
        uint64_t tot = h.upper_off[idx];
        h.upper.resize((tot + level) * h.m, 0);
        h.upper_len.resize(tot + level, 0);
        h.upper_off[idx + 1] = tot + level;
    }
    // HNSWIndex::add (hnsw_index.rs:538-572), row idx already pushed
    void add(uint64_t idx, uint64_t level, Scratch &s) {
        push_init(idx, level);
        if (!h.has_enter) {
            h.has_enter = true;
            h.enter_level = level;
            h.enter_point = idx;
            return;
        }
        const float *q = rows + idx * dim;
        float qc = cache[idx];
        uint64_t enter_level = h.enter_level;
        uint64_t cur = level < enter_level ? greedy_until(level, q, qc) : h.enter_point;
        uint64_t top = std::min(level, enter_level);
        for (uint64_t l = top + 1; l-- > 0;) {
            RSet cand = search_on_level(cur, l, h.ef_construction, q, qc, s);
            cur = cand.v.front().i;
            connect_new_links(idx, l, cand);
        }
        if (level > enter_level) {
            h.enter_level = level;
            h.enter_point = idx;
        }
    }
    // add_parallel (hnsw_index.rs:399-457): candidates against the pre-batch graph in parallel
    void add_batch(uint64_t first, uint64_t nb, const uint64_t *levels, int nthreads, Scratch &s0) {
        if (first < 1000 /* start_batch_since :506 */ || nb == 1) {
            for (uint64_t i = 0; i < nb; i++) add(first + i, levels[i], s0);
            return;
        }
        for (uint64_t i = 0; i < nb; i++) push_init(first + i, levels[i]);
        const uint64_t enter_point = h.enter_point, enter_level = h.enter_level;
        const auto t_begin = std::chrono::steady_clock::now();
        std::vector<std::vector<RSet>> cands(nb);
        std::vector<uint64_t> gkeys, gcross;
        std::vector<uint8_t> gok;
        std::vector<uint32_t> entry0(nb, 0xFFFFFFFFu);
        uint32_t gcape = 0, gld = 0;
        const bool use_gpu = gpu && nb >= gpu->min_batch;
        int nt = std::max(1, std::min<int>(nthreads, (int)nb));
        auto for_members = [&](const std::function<void(uint64_t, Scratch &)> &fn) {
            if (nt == 1) {
                for (uint64_t i = 0; i < nb; i++) fn(i, s0);
                return;
            }
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++)
                th.emplace_back([&, t]() {
                    Scratch s;
                    for (uint64_t i = t; i < nb; i += nt) fn(i, s);
                });
            for (auto &x : th) x.join();
        };
        // one level of one member: search_on_level from cur, then the earlier batch members of that level offered in order
        // (rhs_idx < idx && vec_level[rhs] >= level, :431-437); returns the next level's entry (taken before the offers)
        auto level_step = [&](uint64_t i, uint64_t l, uint64_t cur, Scratch &s) -> uint64_t {
            const uint64_t idx = first + i;
            const float *q = rows + idx * dim;
            const float qc = cache[idx];
            RSet c = search_on_level(cur, l, h.ef_construction, q, qc, s);
            const uint64_t nxt = c.v.front().i;
            std::vector<uint32_t> rhs;
            for (uint64_t r = 0; r < i; r++)
                if (h.vec_level[first + r] >= l) rhs.push_back((uint32_t)(first + r));
            std::vector<float> rd(rhs.size());
            dist_many(q, qc, rhs.data(), rhs.size(), rd.data());
            for (size_t r = 0; r < rhs.size(); r++) c.add(Pair{rd[r], rhs[r]});
            cands[i].push_back(std::move(c));
            return nxt;
        };
        auto upper_levels = [&](uint64_t i, Scratch &s) -> uint64_t {  // levels top .. 1 on the host; returns the level-0 entry
            const uint64_t idx = first + i, level = h.vec_level[idx];
            uint64_t cur = level < enter_level ? greedy_until(level, rows + idx * dim, cache[idx]) : enter_point;
            for (uint64_t l = std::min(level, enter_level); l >= 1; l--) cur = level_step(i, l, cur, s);
            return cur;
        };
        if (use_gpu) {
            // (A) members above level 0: their upper-level searches (small graphs) on the host, which yields their level-0
            // entry; (B) ONE launch searches level 0 for every member -- greedy descent for the level-0 members, the given
            // entry for the others -- and one launch gives the distances between members; (C) the sets are assembled.
            for_members([&](uint64_t i, Scratch &s) {
                if (h.vec_level[first + i] >= 1) entry0[i] = (uint32_t)upper_levels(i, s);
            });
            const auto tg = std::chrono::steady_clock::now();
            gpu->search(first, nb, enter_point, enter_level, entry0, gkeys, gcape, gcross, gld, gok);
            t_gpu += std::chrono::duration<double>(std::chrono::steady_clock::now() - tg).count();
            for_members([&](uint64_t i, Scratch &s) {
                const uint64_t idx = first + i;
                if (!gok[i]) {  // the LDS pool of the GPU walk overflowed: this member's level 0 on the host
                    const uint64_t cur = entry0[i] != 0xFFFFFFFFu ? entry0[i]
                                                                  : (0 < enter_level ? greedy_until(0, rows + idx * dim, cache[idx]) : enter_point);
                    (void)level_step(i, 0, cur, s);
                    return;
                }
                RSet c(h.ef_construction);
                const uint64_t *kr = gkeys.data() + i * gcape;
                // (the kernel's register list is 64 R entries wide: positions past ef hold pairs that were pushed out of the set)
                for (uint32_t j = 0; j < gcape && j < h.ef_construction && kr[j] != PAIR_NONE; j++)
                    c.v.push_back(Pair{f32_from_orderable(uint32_t(kr[j] >> 32)), uint64_t(uint32_t(kr[j]))});
                const uint64_t *xr = gcross.data() + i * gld;  // every earlier member has level >= 0
                for (uint64_t r = 0; r < i; r++) c.add(Pair{f32_from_orderable(uint32_t(xr[r] >> 32)), first + r});
                cands[i].push_back(std::move(c));
            });
        } else {
            for_members([&](uint64_t i, Scratch &s) { (void)level_step(i, 0, upper_levels(i, s), s); });
        }
        const auto t_mid = std::chrono::steady_clock::now();
        // Linking (hnsw_index.rs:446-450 runs connect_new_links node by node).  Everything it computes is a function of
        // distances and of ONE link list at a time: heuristic(candidates) reads no graph state, put_links(v) writes
        // list (v, level), arrange_links(t, level, v) reads and writes list (t, level) only.  So the serial order only
        // matters per list: the selections run in parallel, then the events are grouped by (node, level) in their
        // serial order and the groups run in parallel -- the same lists as the node-by-node loop, bit for bit
        // (an earlier batch member can be a later member's neighbour, never the reverse, so a list's put comes first).
        struct Ev {
            uint64_t key;   // (list owner << 8) | level
            uint32_t seq;   // position in the serial order
            uint32_t v;     // new node
            int32_t sel;    // >= 0: put_links(owner = v, selections[sel]); < 0: arrange_links(owner, level, v)
        };
        std::vector<std::pair<uint32_t, uint32_t>> jobs;  // (batch member, slot)
        for (uint64_t i = 0; i < nb; i++)
            for (size_t sl = 0; sl < cands[i].size(); sl++) jobs.emplace_back((uint32_t)i, (uint32_t)sl);
        std::vector<std::vector<uint32_t>> selections(jobs.size());
        auto run_parallel = [&](size_t count, const std::function<void(size_t)> &fn) {
            if (nt == 1 || count < 2) {
                for (size_t j = 0; j < count; j++) fn(j);
                return;
            }
            std::atomic<size_t> next{0};
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++)
                th.emplace_back([&]() {
                    for (size_t j; (j = next.fetch_add(1)) < count;) fn(j);
                });
            for (auto &x : th) x.join();
        };
        run_parallel(jobs.size(), [&](size_t j) { selections[j] = heuristic(cands[jobs[j].first][jobs[j].second], h.m); });
        std::vector<Ev> evs;
        for (size_t j = 0; j < jobs.size(); j++) {  // jobs are in the serial order: node by node, top level first
            const uint64_t idx = first + jobs[j].first, level = h.vec_level[idx];
            const uint64_t l = std::min(level, enter_level) - jobs[j].second;
            evs.push_back({(idx << 8) | l, (uint32_t)evs.size(), (uint32_t)idx, (int32_t)j});
            for (uint32_t t : selections[j]) evs.push_back({(uint64_t(t) << 8) | l, (uint32_t)evs.size(), (uint32_t)idx, -1});
        }
        std::sort(evs.begin(), evs.end(), [](const Ev &a, const Ev &b) { return a.key != b.key ? a.key < b.key : a.seq < b.seq; });
        std::vector<size_t> gstart;
        for (size_t e = 0; e < evs.size(); e++)
            if (e == 0 || evs[e].key != evs[e - 1].key) gstart.push_back(e);
        gstart.push_back(evs.size());
        run_parallel(gstart.size() - 1, [&](size_t g) {
            for (size_t e = gstart[g]; e < gstart[g + 1]; e++) {
                const uint64_t owner = evs[e].key >> 8, l = evs[e].key & 0xff;
                if (evs[e].sel >= 0)
                    put_links(owner, l, selections[evs[e].sel]);
                else
                    arrange_links(owner, l, evs[e].v);
            }
        });
        if (gpu) {  // the lists this batch rewrote, for the device mirror of the graph
            std::vector<uint64_t> dirty;
            for (size_t g = 0; g + 1 < gstart.size(); g++) dirty.push_back(evs[gstart[g]].key);
            const auto ts = std::chrono::steady_clock::now();
            gpu->sync(dirty);
            t_sync += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
        }
        const auto t_end = std::chrono::steady_clock::now();
        t_search += std::chrono::duration<double>(t_mid - t_begin).count();
        t_connect += std::chrono::duration<double>(t_end - t_mid).count();
        for (uint64_t i = 0; i < nb; i++) {
            uint64_t idx = first + i;
            if (h.vec_level[idx] > h.enter_level) {
                h.enter_level = h.vec_level[idx];
                h.enter_point = idx;
            }
        }
    }
};

uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// rand_level (hnsw_index.rs:144-147) with u in (0,1) from the index's own stream
uint64_t rand_level(uint64_t &state, float inv_log_m) {
    float u = (float(uint32_t(splitmix64(state) >> 40)) + 0.5f) * (1.0f / 16777216.0f);
    return (uint64_t)std::floor(-std::log(u) * inv_log_m);
}

void hnsw_config(HNSWState &h, uint64_t M, uint64_t efc) {  // IndexBuilder::new (hnsw_index.rs:493-506)
    h.m = std::min<uint64_t>(M, 10000);
    h.max_m0 = h.m * 2;
    h.ef_construction = std::max(efc, h.max_m0);
    h.default_ef = h.ef_construction / 2;
    h.inv_log_m = 1.0f / std::log((float)h.m);
}

void reset_graph(HNSWState &h) {
    h.level0.clear();
    h.len0.clear();
    h.vec_level.clear();
    h.upper.clear();
    h.upper_len.clear();
    h.upper_off.assign(1, 0);
    h.has_enter = false;
    h.enter_point = h.enter_level = 0;
    h.dev_dirty = true;
}

}  // namespace

void hnsw_clear(Index &ix) {
    ix.hnsw.present = false;
    reset_graph(ix.hnsw);
    ix.hnsw.d_level0.release();
    ix.hnsw.d_upper.release();
}

// device mirror of the graph under construction + the GPU side of Builder::GpuAssist (defined behind the search kernels'
// launchers, below)
struct BuildDev;
static std::shared_ptr<BuildDev> hnsw_build_gpu_assist(Index &ix, Builder::GpuAssist &ga, const std::vector<uint64_t> &levels);
static std::atomic<int> g_hnsw_build_gpu{0};  // 0 auto (batches of >= 256 points, ef_construction <= 1024), 1 off
void hnsw_set_build_gpu(int v) { g_hnsw_build_gpu = v; }

// the builder proper, on host arrays only (no GPU call unless `prepare_gpu` installs the assist): what hnsw_build runs,
// and what the thread-sanitizer build of tests/cpp/tsan_host.cpp drives with 16 threads
static void hnsw_build_host(HNSWState &h, const float *rows, const float *sq, uint64_t n, uint64_t dim, int dist, uint64_t M,
                     uint64_t ef_construction, uint64_t seed, uint64_t batch, int nthreads,
                     const std::function<std::shared_ptr<void>(Builder &, Builder::GpuAssist &, const std::vector<uint64_t> &)> &prepare_gpu) {
    VDB_REQUIRE(M >= 2, "M must be >= 2");
    h.present = false;
    reset_graph(h);
    hnsw_config(h, M, ef_construction);
    h.rng_state = seed;
    Builder b(h, rows, sq, dim, dist);
    std::vector<uint64_t> levels(n);
    for (uint64_t i = 0; i < n; i++) levels[i] = rand_level(h.rng_state, h.inv_log_m);
    Builder::GpuAssist ga;
    std::shared_ptr<void> bdev;
    if (prepare_gpu) bdev = prepare_gpu(b, ga, levels);
    Scratch s;
    uint64_t cur = 0;
    while (cur < n) {
        uint64_t bs = 1;
        if (cur >= 1000) {  // next_batch_size (:391-397) with rayon's 4*threads replaced by `batch`
            bs = std::min<uint64_t>(std::max<uint64_t>(batch, 1), cur / h.m);
            if (bs < 1) bs = 1;
        }
        uint64_t next = std::min(n, cur + bs);
        b.add_batch(cur, next - cur, levels.data() + cur, nthreads, s);
        cur = next;
    }
    if (std::getenv("VDB_HNSW_PROF"))
        std::fprintf(stderr, "hnsw_build: n=%llu candidate phase %.1f s (%d threads; %.1f s of it GPU searches), linking phase %.1f s (%.1f s of it mirror updates)\n",
                     (unsigned long long)n, b.t_search, nthreads, b.t_gpu, b.t_connect, b.t_sync);
    h.present = true;
    h.dev_dirty = true;
}
// entry point of the host-only sanitizer build: the all-host builder over caller-owned arrays, graph out through `h`
void hnsw_build_host_only(HNSWState &h, const float *rows, const float *sq, uint64_t n, uint64_t dim, int dist, uint64_t M,
                          uint64_t ef_construction, uint64_t seed, uint64_t batch, int nthreads) {
    hnsw_build_host(h, rows, sq, n, dim, dist, M, ef_construction, seed, batch, nthreads, nullptr);
}

void hnsw_build(Index &ix, uint64_t M, uint64_t ef_construction, uint64_t seed, uint64_t batch, int nthreads) {
    VDB_REQUIRE(!ix.elem_u8, "PQ / HNSW / IVF are built over f32 tables (DynamicIndex, dynamic_index.rs:11-14): a VecSet<u8> index serves Flat search");
    VDB_REQUIRE(M >= 2, "M must be >= 2");
    const float *rows = ix.host_rows();
    hnsw_build_host(ix.hnsw, rows, ix.h_sq.data(), ix.n, ix.dim, ix.dist, M, ef_construction, seed, batch, nthreads,
                    [&](Builder &b, Builder::GpuAssist &ga, const std::vector<uint64_t> &levels) -> std::shared_ptr<void> {
                        if (g_hnsw_build_gpu != 1 && batch >= ga.min_batch && ix.hnsw.ef_construction <= 1024 && ix.n >= 4 * ga.min_batch) {
                            std::shared_ptr<BuildDev> bdev = hnsw_build_gpu_assist(ix, ga, levels);
                            b.gpu = &ga;
                            return bdev;
                        }
                        return nullptr;
                    });
}

void hnsw_attach(Index &ix, uint64_t M, uint64_t ef_construction, const uint32_t *level0, const uint64_t *len0,
                 const uint64_t *vec_level, const uint32_t *upper, const uint64_t *upper_len, int has_enter,
                 uint64_t enter_point, uint64_t enter_level) {
    VDB_REQUIRE(!ix.elem_u8, "PQ / HNSW / IVF are built over f32 tables (DynamicIndex, dynamic_index.rs:11-14): a VecSet<u8> index serves Flat search");
    VDB_REQUIRE(M >= 2, "M must be >= 2");
    HNSWState &h = ix.hnsw;
    h.present = false;
    reset_graph(h);
    hnsw_config(h, M, ef_construction);
    uint64_t n = ix.n;
    VDB_REQUIRE(n == 0 || (level0 && len0 && vec_level), "null graph arrays");
    h.level0.assign(level0, level0 + n * h.max_m0);
    h.len0.assign(len0, len0 + n);
    h.vec_level.assign(vec_level, vec_level + n);
    h.upper_off.assign(n + 1, 0);
    for (uint64_t i = 0; i < n; i++) h.upper_off[i + 1] = h.upper_off[i] + vec_level[i];
    uint64_t tot = h.upper_off[n];
    VDB_REQUIRE(tot == 0 || (upper && upper_len), "null upper-level arrays");
    h.upper.assign(upper, upper + tot * h.m);
    h.upper_len.assign(upper_len, upper_len + tot);
    // validate: every link must be a valid node, every length within its limit (get_links_len_checked :160-169)
    for (uint64_t v = 0; v < n; v++) {
        VDB_REQUIRE(h.len0[v] <= h.max_m0, "links_len exceeds limit");
        for (uint64_t j = 0; j < h.len0[v]; j++) VDB_REQUIRE(h.level0[v * h.max_m0 + j] < n, "link out of range");
    }
    for (uint64_t sidx = 0; sidx < tot; sidx++) {
        VDB_REQUIRE(h.upper_len[sidx] <= h.m, "links_len exceeds limit");
        for (uint64_t j = 0; j < h.upper_len[sidx]; j++) VDB_REQUIRE(h.upper[sidx * h.m + j] < n, "link out of range");
    }
    h.has_enter = has_enter != 0;
    VDB_REQUIRE(!h.has_enter || enter_point < n, "enter point out of range");
    VDB_REQUIRE(!h.has_enter || enter_level <= h.vec_level[enter_point], "enter level above the node's level");
    h.enter_point = enter_point;
    h.enter_level = enter_level;
    h.present = true;
    h.dev_dirty = true;
}

// DynamicIndex::add on the HNSW arm (dynamic_index.rs:47-52)
void hnsw_insert_rows(Index &ix, const float *rows, uint64_t count) {
    HNSWState &h = ix.hnsw;
    Scratch s;
    for (uint64_t i = 0; i < count; i++) {
        ix.add_rows(rows + i * ix.dim, 1, false);
        Builder b(h, ix.host_rows(), ix.h_sq.data(), ix.dim, ix.dist);  // both mirrors may have been reallocated
        b.add(ix.n - 1, rand_level(h.rng_state, h.inv_log_m), s);
    }
    h.dev_dirty = true;
}

static void hnsw_upload(Index &ix) {
    HNSWState &h = ix.hnsw;
    static std::mutex mu;  // read-side calls are re-entrant: the lazy upload must happen once
    std::lock_guard<std::mutex> lock(mu);
    if (!h.dev_dirty) return;
    ix.use_device();
    uint64_t n = ix.n;
    std::vector<uint32_t> len0(n), ulen(h.upper_len.size());
    for (uint64_t i = 0; i < n; i++) len0[i] = (uint32_t)h.len0[i];
    for (size_t i = 0; i < ulen.size(); i++) ulen[i] = (uint32_t)h.upper_len[i];
    h.d_level0.reserve(std::max<size_t>(h.level0.size(), 1) * 4);
    h.d_len0.reserve(std::max<size_t>(n, 1) * 4);
    h.d_upper.reserve(std::max<size_t>(h.upper.size(), 1) * 4);
    h.d_upper_len.reserve(std::max<size_t>(ulen.size(), 1) * 4);
    h.d_upper_off.reserve((n + 1) * 8);
    if (n) {
        VDB_HIP(hipMemcpy(h.d_level0.p, h.level0.data(), h.level0.size() * 4, hipMemcpyHostToDevice));
        VDB_HIP(hipMemcpy(h.d_len0.p, len0.data(), n * 4, hipMemcpyHostToDevice));
    }
    if (!h.upper.empty()) VDB_HIP(hipMemcpy(h.d_upper.p, h.upper.data(), h.upper.size() * 4, hipMemcpyHostToDevice));
    if (!ulen.empty()) VDB_HIP(hipMemcpy(h.d_upper_len.p, ulen.data(), ulen.size() * 4, hipMemcpyHostToDevice));
    VDB_HIP(hipMemcpy(h.d_upper_off.p, h.upper_off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    h.dev_dirty = false;
}

// defined in pq.hip
void pq_make_luts(Index &ix, Workspace &ws, const float *d_q, uint64_t nq);
void pq_resort_launch(const uint64_t *exact_keys, uint32_t ncand, uint32_t ldc, uint32_t nq, uint32_t k,
                      uint64_t *out, hipStream_t s);

template <int R, bool ADC>
static void hnsw_launch(const HnswDev &g, const float *d_q, const float *qsq, const float *lut, uint32_t lut_in_lds,
                        uint32_t ef, uint32_t *vis, uint64_t vwords, uint64_t *out, unsigned long long *stats,
                        uint32_t *err, uint32_t nq, size_t lds, hipStream_t s) {
    if (ADC && lut_in_lds) {
        func_max_lds(reinterpret_cast<const void *>(&k_hnsw_search<R, ADC, true>), int(160 * 1024));
        hipLaunchKernelGGL((k_hnsw_search<R, ADC, true>), dim3(nq), dim3(64), lds, s, g, d_q, qsq, lut, lut_in_lds, ef, vis, vwords, out,
                           stats, err);
        return;
    }
    func_max_lds(reinterpret_cast<const void *>(&k_hnsw_search<R, ADC, false>), int(160 * 1024));
    hipLaunchKernelGGL((k_hnsw_search<R, ADC, false>), dim3(nq), dim3(64), lds, s, g, d_q, qsq, lut, lut_in_lds, ef, vis,
                       vwords, out, stats, err);
}

static std::atomic<int> g_hnsw_dma{1};
void hnsw_set_dma(int v) { g_hnsw_dma = v; }
static std::atomic<int> g_hnsw_half{1};  // certified half-precision pre-pass of the exact walk: 1 auto (calls of >= 768 queries), 0 off, 2 always
void hnsw_set_half(int v) { g_hnsw_half = v; }
static std::atomic<uint32_t> g_hnsw_pool_cap{HNSW_POOL};
void hnsw_set_pool_cap(int v) { g_hnsw_pool_cap = v < 1 ? 1u : (v > (int)HNSW_POOL ? HNSW_POOL : (uint32_t)v); }
// Candidate pool entries of a walk.  Every expansion starts by dropping the pairs at or above the worst result, which
// leaves pairs that are also in the result list (< ef of them) or tie its worst distance with a smaller index; until the
// next expansion at most max_m0 pairs join.  ef + max_m0 + 64 therefore only overflows (-> k_hnsw_search_big) on
// near-duplicate data, and 2 KB instead of 16 KB of LDS per query is what lets 8 walks share a CU.
static uint32_t hnsw_pool_slots(uint32_t ef, uint32_t max_m0) {
    const uint64_t want = (uint64_t(ef) + max_m0 + 64 + 63) / 64 * 64;
    return (uint32_t)std::min<uint64_t>(std::min<uint64_t>(want, HNSW_POOL), g_hnsw_pool_cap);
}
// dynamic LDS of k_hnsw_search: pool | payload (query or ADC table) | row staging of the exact walk (mode 1: registers, 2: DMA)
static size_t hnsw_lds_bytes(uint32_t pool_cap, size_t payload, int dma_mode) {
    size_t lds = hnsw_lds_query_off(pool_cap) + payload;
    if (dma_mode) lds = ((lds + 511) & ~size_t(511)) + (dma_mode == 2 ? HNSW_DMA_BYTES : HNSW_REG_STAGE);
    return lds;
}

// ---------------------------------------------------------------------------------------------------
// GPU side of the builder's candidate phase (Builder::GpuAssist).  The device mirror of the graph is full-size from the
// start (offsets of the upper lists follow from the pre-drawn levels); after every batch only the lists that batch
// rewrote are sent over (a few thousand 128-B rows) and scattered into place.
// ---------------------------------------------------------------------------------------------------
__global__ void k_scatter_lists(uint32_t *__restrict__ dst, uint32_t *__restrict__ dst_len, const uint64_t *__restrict__ slots,
                                const uint32_t *__restrict__ src, const uint32_t *__restrict__ src_len, uint32_t row_words,
                                uint32_t count) {
    const uint32_t i = blockIdx.x;  // one list per block
    if (i >= count) return;
    const uint64_t slot = slots[i];
    for (uint32_t j = threadIdx.x; j < row_words; j += blockDim.x) dst[slot * row_words + j] = src[uint64_t(i) * row_words + j];
    if (threadIdx.x == 0) dst_len[slot] = src_len[i];
}
__global__ void k_iota_offset_keys(uint64_t *__restrict__ rows, uint32_t n, uint32_t ld, uint32_t first) {
    uint64_t *row = rows + uint64_t(blockIdx.x) * ld;
    for (uint32_t j = threadIdx.x; j < ld; j += blockDim.x) row[j] = j < n ? uint64_t(first + j) : PAIR_NONE;
}

struct BuildDev {
    DevBuf d_level0, d_len0, d_upper, d_upper_len, d_upper_off;  // the graph, sized for all n rows
    DevBuf d_vis, d_keys, d_flags, d_entry, d_cross_in, d_cross_out, d_stage_slots, d_stage_rows, d_stage_len;
    bool mirror_valid = false;
    bool use_half = false;  // the index holds the row-major fp16 image: the searches run the certified pre-pass
    std::vector<uint64_t> upper_off;  // [n + 1] from the pre-drawn levels
    hipStream_t stream = nullptr;
    ~BuildDev() {
        if (stream) (void)hipStreamDestroy(stream);
    }
};

static std::shared_ptr<BuildDev> hnsw_build_gpu_assist(Index &ix, Builder::GpuAssist &ga, const std::vector<uint64_t> &levels) {
    auto bd = std::make_shared<BuildDev>();
    HNSWState &h = ix.hnsw;
    const uint64_t n = ix.n;
    ix.use_device();
    VDB_HIP(hipStreamCreateWithFlags(&bd->stream, hipStreamNonBlocking));
    bd->upper_off.assign(n + 1, 0);
    for (uint64_t i = 0; i < n; i++) bd->upper_off[i + 1] = bd->upper_off[i] + levels[i];
    const uint64_t tot = bd->upper_off[n];
    bd->d_level0.reserve(std::max<uint64_t>(n * h.max_m0, 1) * 4);
    bd->d_len0.reserve(std::max<uint64_t>(n, 1) * 4);
    bd->d_upper.reserve(std::max<uint64_t>(tot * h.m, 1) * 4);
    bd->d_upper_len.reserve(std::max<uint64_t>(tot, 1) * 4);
    bd->d_upper_off.reserve((n + 1) * 8);
    VDB_HIP(hipMemcpy(bd->d_upper_off.p, bd->upper_off.data(), (n + 1) * 8, hipMemcpyHostToDevice));

    {  // the walk's half-precision pre-pass serves the builder's searches as well (all rows are in the index already)
        WsLease ws(ix);
        bd->use_half = g_hnsw_half && ix.ensure_rows_h(*ws);
    }
    Index *ixp = &ix;
    BuildDev *b = bd.get();
    ga.search = [ixp, b](uint64_t first, uint64_t nb, uint64_t enter_point, uint64_t enter_level, const std::vector<uint32_t> &entry0,
                         std::vector<uint64_t> &keys, uint32_t &cape, std::vector<uint64_t> &cross, uint32_t &ldx,
                         std::vector<uint8_t> &ok) {
        Index &ix = *ixp;
        HNSWState &h = ix.hnsw;
        ix.use_device();
        hipStream_t s = b->stream;
        const uint64_t n = ix.n, have = h.len0.size();  // rows whose lists exist on the host so far (incl. this batch, all empty)
        if (!b->mirror_valid) {  // first GPU batch: everything built so far, the rest zero (no links, length 0)
            VDB_HIP(hipMemsetAsync(b->d_len0.p, 0, std::max<uint64_t>(n, 1) * 4, s));
            VDB_HIP(hipMemsetAsync(b->d_upper_len.p, 0, std::max<uint64_t>(b->upper_off[n], 1) * 4, s));
            VDB_HIP(hipMemcpyAsync(b->d_level0.p, h.level0.data(), have * h.max_m0 * 4, hipMemcpyHostToDevice, s));
            std::vector<uint32_t> l0(have), ul(h.upper_len.size());
            for (uint64_t i = 0; i < have; i++) l0[i] = (uint32_t)h.len0[i];
            for (size_t i = 0; i < ul.size(); i++) ul[i] = (uint32_t)h.upper_len[i];
            VDB_HIP(hipMemcpyAsync(b->d_len0.p, l0.data(), have * 4, hipMemcpyHostToDevice, s));
            if (!h.upper.empty()) VDB_HIP(hipMemcpyAsync(b->d_upper.p, h.upper.data(), h.upper.size() * 4, hipMemcpyHostToDevice, s));
            if (!ul.empty()) VDB_HIP(hipMemcpyAsync(b->d_upper_len.p, ul.data(), ul.size() * 4, hipMemcpyHostToDevice, s));
            VDB_SYNC(s);
            b->mirror_valid = true;
        }
        const uint32_t efk = (uint32_t)std::min<uint64_t>(h.ef_construction, n + 1);
        cape = topk_capacity(efk);
        const bool dma = g_hnsw_dma && h.max_m0 <= 32 && ix.dim % 32 == 0;
        HnswDev g{};
        g.dma = dma ? (g_hnsw_dma == 2 ? 2 : 1) : 0;
        g.pool_cap = hnsw_pool_slots(efk, (uint32_t)h.max_m0);
        if (g.dma == 1 && b->use_half) {
            g.rows_h = ix.d_rows_h.as<uint16_t>();
            g.inv_sx = 1.0f / ix.half_sx();
            g.dx_abs = ix.half_dx_abs;
            g.dx_rel = ix.half_dx_rel;
        }
        const size_t lds = hnsw_lds_bytes(g.pool_cap, ix.dim * sizeof(float), g.dma);
        g.rows = ix.d_rows.as<float>();
        g.xsq = ix.d_sq.as<float>();
        g.level0 = b->d_level0.as<uint32_t>();
        g.len0 = b->d_len0.as<uint32_t>();
        g.upper = b->d_upper.as<uint32_t>();
        g.upper_len = b->d_upper_len.as<uint32_t>();
        g.upper_off = b->d_upper_off.as<uint64_t>();
        g.n = n;
        g.dim = (uint32_t)ix.dim;
        g.m = (uint32_t)h.m;
        g.max_m0 = (uint32_t)h.max_m0;
        g.enter_point = (uint32_t)enter_point;
        g.enter_level = (uint32_t)enter_level;
        g.cosine = ix.dist == 1 ? 1 : 0;
        const uint64_t vwords = (n + 31) / 32;
        constexpr uint64_t QB = 1024;
        b->d_vis.reserve(QB * vwords * 4);
        b->d_keys.reserve(nb * cape * 8);
        b->d_flags.reserve(64 + nb * 4);
        b->d_entry.reserve(nb * 4);
        VDB_HIP(hipMemcpyAsync(b->d_entry.p, entry0.data(), nb * 4, hipMemcpyHostToDevice, s));
        VDB_HIP(hipMemsetAsync(b->d_flags.p, 0, 64 + nb * 4, s));
        unsigned long long *stats = reinterpret_cast<unsigned long long *>(b->d_flags.p);
        uint32_t *err = reinterpret_cast<uint32_t *>(b->d_flags.as<uint8_t>() + 64);
        const float *d_q = ix.d_rows.as<float>() + first * ix.dim, *d_qsq = ix.d_sq.as<float>() + first;
        for (uint64_t q0 = 0; q0 < nb; q0 += QB) {
            const uint32_t nq = (uint32_t)std::min<uint64_t>(QB, nb - q0);
            VDB_HIP(hipMemsetAsync(b->d_vis.p, 0, uint64_t(nq) * vwords * 4, s));
            uint64_t *outk = b->d_keys.as<uint64_t>() + q0 * cape;
            g.entry0 = b->d_entry.as<uint32_t>() + q0;
#define HLB(R) hnsw_launch<R, false>(g, d_q + q0 * ix.dim, d_qsq + q0, nullptr, 0u, efk, b->d_vis.as<uint32_t>(), vwords, outk, stats, err + q0, nq, lds, s)
            switch (cape / 64) {
                case 1: HLB(1); break;
                case 2: HLB(2); break;
                case 4: HLB(4); break;
                case 8: HLB(8); break;
                case 16: HLB(16); break;
                default: throw Error(1, "hnsw build: unexpected result-list capacity");
            }
#undef HLB
        }
        // all-pairs cached-form distances inside the batch (the `rhs_idx < idx` offers of add_parallel, :431-437)
        ldx = (uint32_t)((nb + 63) & ~63ull);
        b->d_cross_in.reserve(nb * ldx * 8);
        b->d_cross_out.reserve(nb * ldx * 8);
        hipLaunchKernelGGL(k_iota_offset_keys, dim3((unsigned)nb), dim3(256), 0, s, b->d_cross_in.as<uint64_t>(), (uint32_t)nb, ldx,
                           (uint32_t)first);
        launch_rerank(ix.d_rows.as<float>(), (uint32_t)ix.dim, d_q, (uint32_t)nb, ix.dist == 0 ? MET_L2_CACHED : MET_COSINE,
                      ix.d_sq.as<float>(), d_qsq, b->d_cross_in.as<uint64_t>(), b->d_cross_out.as<uint64_t>(), (uint32_t)nb, ldx, s);
        keys.resize(nb * cape);
        cross.resize(nb * uint64_t(ldx));
        std::vector<uint32_t> e(nb);
        VDB_HIP(hipMemcpyAsync(keys.data(), b->d_keys.p, nb * cape * 8, hipMemcpyDeviceToHost, s));
        VDB_HIP(hipMemcpyAsync(cross.data(), b->d_cross_out.p, nb * uint64_t(ldx) * 8, hipMemcpyDeviceToHost, s));
        VDB_HIP(hipMemcpyAsync(e.data(), err, nb * 4, hipMemcpyDeviceToHost, s));
        VDB_SYNC(s);
        ok.resize(nb);
        for (uint64_t i = 0; i < nb; i++) ok[i] = e[i] ? 0 : 1;  // an overflowed LDS pool: that member is searched on the host
    };
    ga.sync = [ixp, b](const std::vector<uint64_t> &dirty) {
        if (!b->mirror_valid || dirty.empty()) return;
        Index &ix = *ixp;
        HNSWState &h = ix.hnsw;
        ix.use_device();
        hipStream_t s = b->stream;
        for (int upper = 0; upper < 2; upper++) {
            const uint32_t words = upper ? (uint32_t)h.m : (uint32_t)h.max_m0;
            std::vector<uint64_t> slots;
            std::vector<uint32_t> rows, lens;
            for (uint64_t key : dirty) {
                const uint64_t owner = key >> 8, level = key & 0xff;
                if ((level != 0) != (upper != 0)) continue;
                const uint64_t slot = upper ? h.upper_off[owner] + level - 1 : owner;
                const uint32_t *src = upper ? h.upper.data() + slot * h.m : h.level0.data() + owner * h.max_m0;
                slots.push_back(slot);
                rows.insert(rows.end(), src, src + words);
                lens.push_back((uint32_t)(upper ? h.upper_len[slot] : h.len0[owner]));
            }
            if (slots.empty()) continue;
            b->d_stage_slots.reserve(slots.size() * 8);
            b->d_stage_rows.reserve(rows.size() * 4);
            b->d_stage_len.reserve(lens.size() * 4);
            VDB_HIP(hipMemcpyAsync(b->d_stage_slots.p, slots.data(), slots.size() * 8, hipMemcpyHostToDevice, s));
            VDB_HIP(hipMemcpyAsync(b->d_stage_rows.p, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, s));
            VDB_HIP(hipMemcpyAsync(b->d_stage_len.p, lens.data(), lens.size() * 4, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_scatter_lists, dim3((unsigned)slots.size()), dim3(64), 0, s,
                               upper ? b->d_upper.as<uint32_t>() : b->d_level0.as<uint32_t>(),
                               upper ? b->d_upper_len.as<uint32_t>() : b->d_len0.as<uint32_t>(), b->d_stage_slots.as<uint64_t>(),
                               b->d_stage_rows.as<uint32_t>(), b->d_stage_len.as<uint32_t>(), words, (uint32_t)slots.size());
            VDB_SYNC(s);  // the staging vectors are reused by the other pass
        }
    };
    return bd;
}

void hnsw_knn_device(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t ef, bool use_pq,
                     uint64_t *d_idx, float *d_dist, uint64_t *d_cnt) {
    hipStream_t s = ws.stream;
    HNSWState &h = ix.hnsw;
    if (nq == 0) return;
    if (k == 0 || ix.n == 0 || !h.has_enter) {  // knn_with_ef on an empty index returns nothing (:625-627)
        VDB_HIP(hipMemsetAsync(d_cnt, 0, nq * sizeof(uint64_t), s));
        return;
    }
    VDB_REQUIRE(h.len0.size() == ix.n, "HNSW graph does not cover every row (rebuild or re-attach it)");
    hnsw_upload(ix);
    const uint64_t n = ix.n;
    // ef.max(k) :628.  A set of more than n pairs never fills, so n + 1 stands for every larger capacity (same walk, same
    // counters: the reference keeps expanding until the queue is empty)
    const uint64_t efk64 = std::min<uint64_t>(std::max(ef, k), n + 1);
    const uint32_t efk = (uint32_t)efk64;
    const bool big_all = efk > 1024;  // beyond the register-resident result list: every query takes the heap walk
    const uint32_t cape = big_all ? 64 : topk_capacity(efk);
    const uint32_t ksel = (uint32_t)std::min<uint64_t>(k, efk);
    const uint32_t capk = ksel <= 1024 ? topk_capacity(ksel) : 64;
    VDB_HIP(hipMemsetAsync(d_idx, 0, nq * k * sizeof(uint64_t), s));
    VDB_HIP(hipMemsetAsync(d_dist, 0, nq * k * sizeof(float), s));

    ws.qsq.reserve(nq * sizeof(float));
    launch_row_sqnorm(d_q, nq, (uint32_t)ix.dim, ws.qsq.as<float>(), s);
    PQState &pq = ix.pq;
    if (use_pq) VDB_REQUIRE(pq.present && pq.n_coded == n, "PQ table does not cover the rows of the index (rebuild it after add)");
    uint32_t lut_in_lds = 0;
    size_t lds_big = 0;
    if (use_pq) {
        pq_make_luts(ix, ws, d_q, nq);
        size_t lb = pq.m * pq.kc * sizeof(float);
        if (lb <= 32 * 1024) {
            lut_in_lds = 1;
            lds_big = lb;
        }
    } else {
        lds_big = ix.dim * sizeof(float);
    }
    const bool dma = !use_pq && g_hnsw_dma && h.max_m0 <= 32 && ix.dim % 32 == 0;
    HnswDev g{};
    g.dma = dma ? (g_hnsw_dma == 2 ? 2 : 1) : 0;  // 1: rows staged through registers, 2: through LDS by DMA (the round-1 form, kept for A/B)
    // (auto: calls of fewer than 768 queries are a fraction of one round of walks -- latency, not bytes, is what they wait for,
    // and the pre-pass is one more dependent gather per expansion: 2.23 vs 2.07 ms at 256 queries, 2.74 vs 2.89 ms at 1024)
    if (g.dma == 1 && (g_hnsw_half == 2 || (g_hnsw_half == 1 && nq >= 768)) && ix.ensure_rows_h(ws)) {
        g.rows_h = ix.d_rows_h.as<uint16_t>();
        g.inv_sx = 1.0f / ix.half_sx();
        g.dx_abs = ix.half_dx_abs;
        g.dx_rel = ix.half_dx_rel;
    }
    g.pool_cap = hnsw_pool_slots(efk, (uint32_t)h.max_m0);
    const size_t lds = hnsw_lds_bytes(g.pool_cap, lds_big, g.dma);
    g.rows = ix.d_rows.as<float>();
    g.xsq = ix.d_sq.as<float>();
    g.level0 = h.d_level0.as<uint32_t>();
    g.len0 = h.d_len0.as<uint32_t>();
    g.upper = h.d_upper.as<uint32_t>();
    g.upper_len = h.d_upper_len.as<uint32_t>();
    g.upper_off = h.d_upper_off.as<uint64_t>();
    g.n = n;
    g.dim = (uint32_t)ix.dim;
    g.m = (uint32_t)h.m;
    g.max_m0 = (uint32_t)h.max_m0;
    g.enter_point = (uint32_t)h.enter_point;
    g.enter_level = (uint32_t)h.enter_level;
    g.cosine = ix.dist == 1 ? 1 : 0;
    if (use_pq) {
        g.codes = pq.d_codes.as<uint8_t>();
        g.cent_cache = pq.d_cent_cache.as<float>();
        g.enc_dim = (uint32_t)pq.enc_dim;
        g.pq_m = (uint32_t)pq.m;
        g.pq_kc = (uint32_t)pq.kc;
        g.n_bits = (uint32_t)pq.n_bits;
    }
    const uint64_t vwords = (n + 31) / 32;
    // queries per launch: what bounds it is the visited bitmaps (QB * n/8 bytes, <= 2 GiB); a launch should hold several
    // times the 2048 walks the chip keeps resident, so that finished walks are replaced while the long ones run
    const uint64_t QB = std::min<uint64_t>(8192, std::max<uint64_t>(1024, (uint64_t(2) << 30) / (vwords * 4)));
    // flags: [0,16) work counters (n_dist, n_expanded), [64, 64 + 4 nq) per-query "candidate pool overflowed"
    ws.flags.reserve(64 + nq * sizeof(uint32_t));
    VDB_HIP(hipMemsetAsync(ws.flags.p, 0, 64 + nq * sizeof(uint32_t), s));
    unsigned long long *stats = reinterpret_cast<unsigned long long *>(ws.flags.p);
    uint32_t *err = reinterpret_cast<uint32_t *>(ws.flags.as<uint8_t>() + 64);
    if (!big_all) {
        ws.misc.reserve(QB * vwords * sizeof(uint32_t) + 64);
        ws.keys_a.reserve(nq * cape * sizeof(uint64_t));
        ws.keys_b.reserve(nq * cape * sizeof(uint64_t));
        ws.keys_c.reserve(nq * capk * sizeof(uint64_t));
        for (uint64_t q0 = 0; q0 < nq; q0 += QB) {
            uint32_t nb = (uint32_t)std::min<uint64_t>(QB, nq - q0);
            VDB_HIP(hipMemsetAsync(ws.misc.p, 0, uint64_t(nb) * vwords * sizeof(uint32_t), s));
            const float *lut = use_pq ? ws.lut.as<float>() + q0 * pq.m * pq.kc : nullptr;
            uint64_t *outk = ws.keys_a.as<uint64_t>() + q0 * cape;
            ix.prof_begin(ws, "hnsw", 0.0);
#define HL(R)                                                                                                          \
    if (use_pq)                                                                                                        \
        hnsw_launch<R, true>(g, d_q + q0 * ix.dim, ws.qsq.as<float>() + q0, lut, lut_in_lds, efk, ws.misc.as<uint32_t>(), \
                             vwords, outk, stats, err + q0, nb, lds, s);                                               \
    else                                                                                                               \
        hnsw_launch<R, false>(g, d_q + q0 * ix.dim, ws.qsq.as<float>() + q0, lut, lut_in_lds, efk, ws.misc.as<uint32_t>(), \
                              vwords, outk, stats, err + q0, nb, lds, s);
            switch (cape / 64) {
                case 1: HL(1); break;
                case 2: HL(2); break;
                case 4: HL(4); break;
                case 8: HL(8); break;
                case 16: HL(16); break;
                default: throw Error(1, "hnsw knn: unexpected result-list capacity");
            }
#undef HL
            ix.prof_end(ws);
        }
        if (use_pq) {
            // pq_resort with the cached-form distance (hnsw_index.rs:693-695)
            VDB_HIP(hipMemsetAsync(ws.keys_b.p, 0xff, nq * cape * sizeof(uint64_t), s));
            launch_rerank(ix.d_rows.as<float>(), (uint32_t)ix.dim, d_q, (uint32_t)nq, ix.dist == 0 ? MET_L2_CACHED : MET_COSINE,
                          ix.d_sq.as<float>(), ws.qsq.as<float>(), ws.keys_a.as<uint64_t>(), ws.keys_b.as<uint64_t>(), efk, cape, s);
            pq_resort_launch(ws.keys_b.as<uint64_t>(), efk, cape, (uint32_t)nq, ksel, ws.keys_c.as<uint64_t>(), s);
            launch_finalize(ws.keys_c.as<uint64_t>(), capk, (uint32_t)nq, ksel, (uint32_t)k, ix.id_offset, d_idx, d_dist, d_cnt, s);
        } else {
            // into_sorted_vec_limit(k) (:632)
            launch_finalize(ws.keys_a.as<uint64_t>(), cape, (uint32_t)nq, ksel, (uint32_t)k, ix.id_offset, d_idx, d_dist, d_cnt, s);
        }
    }
    // queries for the heap walk: all of them (ef > 1024), or those whose LDS candidate pool overflowed (graphs of
    // near-duplicates keep thousands of live candidates); the work counters of an overflowed walk are dropped first
    std::vector<uint32_t> redo;
    unsigned long long st[4] = {0, 0, 0, 0};
    auto read_stats = [&]() {
        unsigned char *hb = static_cast<unsigned char *>(ws.pinned(64 + nq * sizeof(uint32_t)));
        VDB_HIP(hipMemcpyAsync(hb, ws.flags.p, 64 + nq * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        VDB_SYNC(s);
        std::memcpy(st, hb, 32);
        return reinterpret_cast<const uint32_t *>(hb + 64);
    };
    if (big_all) {
        redo.resize(nq);
        for (uint64_t q = 0; q < nq; q++) redo[q] = (uint32_t)q;
    } else {
        const uint32_t *e = read_stats();
        for (uint64_t q = 0; q < nq; q++)
            if (e[q]) redo.push_back((uint32_t)q);
    }
    if (!redo.empty()) {
        h.heap_walk_queries += redo.size();
        const uint32_t res_ld = (efk + 63) & ~63u;
        // per slot: visited bitmap + candidate heap (n pairs) + result heap, its sorted copy and (PQ) the exact keys
        const uint64_t per_q = vwords * 4 + n * 8 + uint64_t(res_ld) * 24;
        const uint64_t QBB = std::max<uint64_t>(1, std::min<uint64_t>(256, (size_t(3) << 29) / per_q));
        const size_t tb = sort_rows_temp_bytes(QBB, res_ld);
        DevBuf b_vis, b_cand, b_res, b_sorted, b_exact, b_tmp, b_ql;  // rare path: allocated on demand
        b_vis.reserve(QBB * vwords * 4);
        b_cand.reserve(QBB * n * 8);
        b_res.reserve(QBB * res_ld * 8);
        b_sorted.reserve(QBB * res_ld * 8);
        if (use_pq) b_exact.reserve(QBB * res_ld * 8);
        b_tmp.reserve(tb);
        b_ql.reserve(QBB * 4);
        func_max_lds(reinterpret_cast<const void *>(&k_hnsw_search_big<true>), int(64 * 1024));
        func_max_lds(reinterpret_cast<const void *>(&k_hnsw_search_big<false>), int(64 * 1024));
        for (size_t c0 = 0; c0 < redo.size(); c0 += QBB) {
            const uint32_t nb = (uint32_t)std::min<size_t>(QBB, redo.size() - c0);
            VDB_HIP(hipMemcpyAsync(b_ql.p, redo.data() + c0, nb * 4, hipMemcpyHostToDevice, s));
            VDB_HIP(hipMemsetAsync(b_vis.p, 0, uint64_t(nb) * vwords * 4, s));
            ix.prof_begin(ws, "hnsw", 0.0);
            if (use_pq)
                hipLaunchKernelGGL((k_hnsw_search_big<true>), dim3(nb), dim3(64), lds_big, s, g, d_q, ws.qsq.as<float>(),
                                   ws.lut.as<float>(), lut_in_lds, efk, b_vis.as<uint32_t>(), vwords, b_ql.as<uint32_t>(),
                                   b_cand.as<uint64_t>(), n, b_res.as<uint64_t>(), res_ld, stats);
            else
                hipLaunchKernelGGL((k_hnsw_search_big<false>), dim3(nb), dim3(64), lds_big, s, g, d_q, ws.qsq.as<float>(),
                                   (const float *)nullptr, 0u, efk, b_vis.as<uint32_t>(), vwords, b_ql.as<uint32_t>(),
                                   b_cand.as<uint64_t>(), n, b_res.as<uint64_t>(), res_ld, stats);
            ix.prof_end(ws);
            launch_sort_rows(b_res.as<uint64_t>(), b_sorted.as<uint64_t>(), nb, res_ld, b_tmp.p, tb, s);
            // outputs per run of consecutive query ids (the whole chunk when every query takes this path)
            for (uint32_t j0 = 0; j0 < nb;) {
                uint32_t j1 = j0 + 1;
                while (j1 < nb && redo[c0 + j1] == redo[c0 + j1 - 1] + 1) j1++;
                const uint64_t q = redo[c0 + j0], run = j1 - j0;
                const uint64_t *srt = b_sorted.as<uint64_t>() + uint64_t(j0) * res_ld;
                if (use_pq) {
                    uint64_t *ex = b_exact.as<uint64_t>() + uint64_t(j0) * res_ld;
                    VDB_HIP(hipMemsetAsync(ex, 0xff, run * res_ld * 8, s));
                    launch_rerank(ix.d_rows.as<float>(), (uint32_t)ix.dim, d_q + q * ix.dim, (uint32_t)run,
                                  ix.dist == 0 ? MET_L2_CACHED : MET_COSINE, ix.d_sq.as<float>(), ws.qsq.as<float>() + q, srt, ex, efk,
                                  res_ld, s);
                    pq_resort_finalize(ix, ws, ex, efk, res_ld, run, ksel, k, ix.id_offset, d_idx + q * k, d_dist + q * k, d_cnt + q);
                } else {
                    launch_finalize(srt, res_ld, (uint32_t)run, ksel, (uint32_t)k, ix.id_offset, d_idx + q * k, d_dist + q * k,
                                    d_cnt + q, s);
                }
                j0 = j1;
            }
            VDB_SYNC(s);  // the next chunk reuses the buffers (and they are freed on return)
        }
        (void)read_stats();
    }
    h.last_n_dist = st[0];
    h.last_n_expanded = st[1];
    h.last_half_dropped = st[2];
#ifdef HNSW_STAMP
    {
        unsigned long long tk[4];
        VDB_HIP(hipMemcpy(tk, reinterpret_cast<const char *>(ws.flags.p) + 32, sizeof(tk), hipMemcpyDeviceToHost));
        const double per = st[1] ? 1.0 / double(st[1]) * 0.01 : 0.0;  // 100 MHz ticks -> us per expansion
        std::fprintf(stderr, "hnsw stamps: %llu distance evaluations, %llu through the half-precision pre-pass, %llu ruled out by it\n", st[0], st[3], st[2]);
        std::fprintf(stderr, "hnsw stamps (us per expansion): pop %.2f, links + visited %.2f, distances %.2f, inserts %.2f\n", tk[0] * per,
                     tk[1] * per, tk[2] * per, tk[3] * per);
#ifdef HNSW_STAMP2
        unsigned long long t2[8];
        VDB_HIP(hipMemcpyFromSymbol(t2, HIP_SYMBOL(g_hnsw_st2), sizeof(t2)));
        const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        VDB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_hnsw_st2), zero, sizeof(zero)));
        const double pc = t2[4] ? 0.01 / double(t2[4]) : 0.0;
        std::fprintf(stderr, "hnsw_exact_dists_regs (us per call, %llu calls): set-up + loads issued %.2f, first line there %.2f, fold of the lines %.2f, epilogue %.2f\n",
                     t2[4], t2[0] * pc, t2[1] * pc, t2[2] * pc, t2[3] * pc);
#endif
    }
#endif
    if (!ws.pending.empty()) {
        // algorithmic bytes are data-dependent: n_dist row fetches (+ cached norm) and n_expanded link rows,
        // known only now -- credited to the call's last launch record
        // (SURVEY 8d's figure, n_dist f32 rows, is what vdb_hnsw_last_stats lets a caller compute; the record holds the bytes
        // the walk asked for: with the pre-pass a scored row is dim*2 B of the fp16 image, plus dim*4 B unless it was ruled out)
        double row_bytes = use_pq ? double(pq.enc_dim) * double(st[0]) : (double(ix.dim) * sizeof(float)) * double(st[0] - st[2]) + sizeof(float) * double(st[0]);
        if (!use_pq) row_bytes += double(ix.dim) * sizeof(uint16_t) * double(st[3]);
        ws.pending.back().bytes += row_bytes + double(st[1]) * double(h.max_m0) * sizeof(uint32_t);
    }
}

}  // namespace vdb
