// pq.hip -- product quantisation on gfx950: lookup-table build, asymmetric-distance (ADC) scan with the
// per-query table pinned in LDS, reference-order re-sort, GPU encoder; plus the host-side PQ table
// management (attach / build / clear).  Reference: src/distance/pq_table.rs, src/distance/k_means.rs,
// FlatIndex::knn_pq (src/index_algorithm/flat_index.rs:84-104), ResultSet::pq_resort
// (src/index_algorithm/candidate_pair.rs:102-108).
//
// Bit-exactness: an ADC distance is sum_{i<m} lut[i*k + code_i] accumulated from 0.0 in ascending
// group order (pq_table.rs:254-292).  One thread owns one code row and adds in that order, so the sums
// equal the reference's bit for bit; the LUT entries themselves are strict-order folds.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

#include "pq_hnsw.hpp"

#pragma clang fp contract(off)

namespace vdb {

// ---------------------------------------------------------------------------------------------------
// create_lookup (pq_table.rs:195-224): lut[q][g*kc + c] = l2(qslice, cent) | dot(qslice, cent)
// ---------------------------------------------------------------------------------------------------
__global__ void k_pq_lut(const float *__restrict__ Q, uint32_t dim, const float *__restrict__ cent,
                         const uint64_t *__restrict__ gstart, uint32_t m, uint32_t kc, int cosine,
                         float *__restrict__ lut) {
    uint32_t q = blockIdx.y;
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * kc) return;
    uint32_t g = e / kc, c = e % kc;
    uint32_t s = (uint32_t)gstart[g], gd = (uint32_t)gstart[g + 1] - s;
    const float *v = Q + uint64_t(q) * dim + s;
    const float *cc = cent + uint64_t(kc) * s + uint64_t(c) * gd;
    float acc = 0.0f;
    if (cosine) {
        for (uint32_t j = 0; j < gd; j++) {
            float p = v[j] * cc[j];
            acc = acc + p;
        }
    } else {
        for (uint32_t j = 0; j < gd; j++) {
            float df = v[j] - cc[j];
            float sq = df * df;
            acc = acc + sq;
        }
    }
    lut[uint64_t(q) * m * kc + e] = acc;
}

// ---------------------------------------------------------------------------------------------------
// ADC scan (pq_table.rs:239-301 applied to every code row, flat_index.rs:98-101).
// BQ queries per pass share each code byte; their lookup tables sit in LDS (m*kc floats each;
// 20 KB for m=320, 4 bit).  16 consecutive floats of one group span 16 distinct banks, so a
// ds_read_b32 gather by code value is conflict-free whatever the codes are.
// LUT_IN_LDS = false: tables too large for LDS (8-bit codes with large m) are read through L1/L2.
// ---------------------------------------------------------------------------------------------------
// MODE 0 (dense): out[b*ld + j] for the visited rows (row blocks of blockDim.x rows, every `blk_step`-th block;
//                 j = dense position among the visited rows; +inf for rows >= n).  blk_step = 1 is the plain scan,
//                 blk_step > 1 the strided sample whose ef-th smallest value per query bounds the global one.
// MODE 1 (filter): all rows; pairs with adc <= tau[b] are parked in an LDS buffer and handed to cand[b][..] when
//                 the workgroup is done (one global atomic per query per workgroup) -- no dense N x B matrix, no
//                 separate top-ef pass over it.  ADC values are exact, so {adc <= tau} is a superset of the exact
//                 top-ef and the select that follows is the reference's (adc, idx) order.
struct AdcArgs {
    const uint8_t *codes;
    uint64_t n;
    uint32_t enc_dim, m;
    const float *lut;         // [BQ][m*kc]
    const float *cent_cache;  // [m*kc]
    const float *qsq;         // [BQ]
    uint32_t nq;              // valid queries of this pass (<= BQ)
    int cosine;
    uint32_t blk_step;
    float *out;               // MODE 0
    uint64_t ld;
    const float *tau;         // MODE 1: [BQ]
    uint64_t *cand;           //         [BQ][cap]
    uint32_t *cnt;            //         [BQ]
    uint32_t cap;
    uint32_t nq_total;        // queries of the whole launch; blockIdx.y selects the sub-batch of BQ queries
    int fast;                 // batched-lookup inner loop (tuning switch, vdb_set_param "pq_adc_fast")
};
static std::atomic<int> g_adc_fast{1};
void pq_set_adc_fast(int v) { g_adc_fast = v; }
static std::atomic<int> g_adc16_sample{0};  // threshold sample on the quantised tables (L2Sqr): 0 auto (on with the quantised scan), 1 off (f32 sample)
void pq_set_adc16_sample(int v) { g_adc16_sample = v; }
static std::atomic<int> g_adc8_sliced{0};  // 8-bit codes: 0 = sixteen queries per pass on sliced one-byte tables (k_pq_adc8x16), 1 = one query per pass on a byte table (k_pq_adc8), 2 = eight queries per pass on sliced 16-bit tables (k_pq_adc16x8)
void pq_set_adc8_sliced(int v) { g_adc8_sliced = v; }
static std::atomic<int> g_adc16{0};  // quantised first pass of the threshold-filter scan: 0 auto (4-bit, L2Sqr, 16-B code words), 1 off
void pq_set_adc16(int v) { g_adc16 = v; }
constexpr uint32_t ADC_WGBUF = 2048;  // LDS hit buffer entries per workgroup (MODE 1)

// One code row of the Gist1M-shaped table (4-bit codes, every nibble a group, 16-B code words) against 4 lookup tables
// held entry-major / query-minor in LDS: 8 independent ds_read_b128 per code word half, then the strict-order adds
// (two packed adds per lookup serve the 4 queries).  A separate, non-inlined function: inside k_pq_adc the register
// allocator shares 128 VGPRs with every other path of that kernel and spills the loaded entries to scratch.
template <bool COS>
__device__ __noinline__ float4 adc_row_fast(const uint4 *cw, uint32_t nwords, const char *lbase, const char *cbase,
                                            float &cdp_out) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f};
    float cdp = 0.0f;
    uint4 v = cw[0];
    for (uint32_t w = 0; w < nwords; w++) {
        const uint32_t words[4] = {v.x, v.y, v.z, v.w};
        if (w + 1 < nwords) v = cw[w + 1];  // next code word while this one is looked up
#pragma unroll
        for (int wi = 0; wi < 4; wi++) {
            float4 t[8];
            float cc[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {  // nibble j of the word = group 32w + 8wi + j (low nibble of a byte first)
                const uint32_t off = (w * 32 + 8 * wi + j) * 256 + (((words[wi] >> (4 * j)) & 0xf) << 4);
                t[j] = *reinterpret_cast<const float4 *>(lbase + off);
                if (COS) cc[j] = *reinterpret_cast<const float *>(cbase + (off >> 2));
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                s01 += (f32x2_t){t[j].x, t[j].y};
                s23 += (f32x2_t){t[j].z, t[j].w};
                if (COS) cdp = cdp + cc[j];
            }
        }
    }
    cdp_out = cdp;
    return make_float4(s01.x, s01.y, s23.x, s23.y);
}

template <int BQ, int NBITS, bool LUT_IN_LDS, int MODE>
__global__ __launch_bounds__(1024) void k_pq_adc(AdcArgs a) {
    {   // sub-batch blockIdx.y: BQ consecutive queries of the launch
        const uint32_t q0 = blockIdx.y * BQ;
        a.nq = a.nq_total - q0 < (uint32_t)BQ ? a.nq_total - q0 : (uint32_t)BQ;
        a.lut += uint64_t(q0) * a.m * (1u << NBITS);
        a.qsq += q0;
        if (MODE == 0) {
            a.out += uint64_t(q0) * a.ld;
        } else {
            a.tau += q0;
            a.cand += uint64_t(q0) * a.cap;
            a.cnt += q0;
        }
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr uint32_t KC = 1u << NBITS;
    const uint32_t m = a.m, enc_dim = a.enc_dim;
    const uint64_t n = a.n;
    const int cosine = a.cosine;
    const uint32_t lsz = m * KC;
    const float *lut[BQ];
    const float *ccache = a.cent_cache;
    uint32_t lds_floats = 0;
    if (LUT_IN_LDS) {
        // entry-major, query-minor: the BQ values of one (group, code) entry are adjacent, so one ds_read_b128
        // (BQ = 4) serves all queries of the pass -- half the LDS cycles of four ds_read_b32 gathers.  Distinct
        // codes of a group still land in distinct banks (16 entries x 16 B = all 64 banks).
        for (uint32_t i = threadIdx.x; i < lsz * BQ; i += blockDim.x) {
            uint32_t b = i / lsz, e = i % lsz;
            smem[e * BQ + b] = b < a.nq ? a.lut[i] : 0.0f;
        }
        if (cosine)
            for (uint32_t i = threadIdx.x; i < lsz; i += blockDim.x) smem[lsz * BQ + i] = a.cent_cache[i];
        ccache = smem + lsz * BQ;
        lds_floats = lsz * BQ + (cosine ? lsz : 0);
    } else {
#pragma unroll
        for (int b = 0; b < BQ; b++) lut[b] = a.lut + (b < (int)a.nq ? b : 0) * uint64_t(lsz);
    }
    // hit buffer behind the tables (MODE 1)
    uint64_t *hit_key = reinterpret_cast<uint64_t *>(smem + ((lds_floats + 3) & ~3u));
    uint32_t *hit_q = reinterpret_cast<uint32_t *>(hit_key + ADC_WGBUF);
    uint32_t *hit_n = hit_q + ADC_WGBUF;  // [0] entries, [1..BQ] per-query counts, [1+BQ..2BQ] bases
    if (MODE == 1 && threadIdx.x < 1 + 2 * BQ) hit_n[threadIdx.x] = 0;
    float tau[BQ];
#pragma unroll
    for (int b = 0; b < BQ; b++) tau[b] = (MODE == 1 && b < (int)a.nq) ? a.tau[b] : -INFINITY;
    __syncthreads();

    const uint64_t nblk = (n + blockDim.x - 1) / blockDim.x;
    for (uint64_t vb = blockIdx.x; vb * a.blk_step < nblk; vb += gridDim.x) {
        const uint64_t row = vb * a.blk_step * blockDim.x + threadIdx.x;
        const bool valid = row < n;
        const uint8_t *cr = a.codes + (valid ? row : n - 1) * enc_dim;
        float sum[BQ];
#pragma unroll
        for (int b = 0; b < BQ; b++) sum[b] = 0.0f;
        float cdp = 0.0f;
        auto push = [&](uint32_t i, uint32_t code) {
            if (i >= m) return;  // pq_table.rs:258-260
            uint32_t at = i * KC + code;
            if (LUT_IN_LDS) {
                if (BQ == 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(smem + at * 4);
                    sum[0] = sum[0] + v.x;
                    sum[1 % BQ] = sum[1 % BQ] + v.y;
                    sum[2 % BQ] = sum[2 % BQ] + v.z;
                    sum[3 % BQ] = sum[3 % BQ] + v.w;
                } else if (BQ == 2) {
                    const float2 v = *reinterpret_cast<const float2 *>(smem + at * 2);
                    sum[0] = sum[0] + v.x;
                    sum[1 % BQ] = sum[1 % BQ] + v.y;
                } else {
                    sum[0] = sum[0] + smem[at];
                }
            } else {
#pragma unroll
                for (int b = 0; b < BQ; b++) sum[b] = sum[b] + lut[b][at];
            }
            if (cosine) cdp = cdp + ccache[at];
        };
        if (LUT_IN_LDS && BQ == 4 && NBITS == 4 && a.fast && (enc_dim & 15) == 0 && m == 2 * enc_dim) {
            // Fast path (the Gist1M table: m = 320, 4 bit, 160-B code rows): every nibble is a group, so no bound checks,
            // and the 8 lookups of a code word are issued together -- addresses first, then 8 independent
            // ds_read_b128, then the strict-order adds.  (The generic path below waits for each lookup before it
            // issues the next and branches per nibble: measured 24.6 us per query and 1M rows against an LDS-gather
            // floor of ~8 us.)
            float4 r4;
            if (cosine)
                r4 = adc_row_fast<true>(reinterpret_cast<const uint4 *>(cr), enc_dim / 16, reinterpret_cast<const char *>(smem),
                                        reinterpret_cast<const char *>(ccache), cdp);
            else
                r4 = adc_row_fast<false>(reinterpret_cast<const uint4 *>(cr), enc_dim / 16, reinterpret_cast<const char *>(smem),
                                         nullptr, cdp);
            sum[0] = r4.x;
            sum[1 % BQ] = r4.y;
            sum[2 % BQ] = r4.z;
            sum[3 % BQ] = r4.w;
        } else if ((enc_dim & 15) == 0) {  // 16 code bytes per load
            const uint4 *cw = reinterpret_cast<const uint4 *>(cr);
            for (uint32_t w = 0; w < enc_dim / 16; w++) {
                const uint4 v = cw[w];
                const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int wi = 0; wi < 4; wi++) {
#pragma unroll
                    for (int byte = 0; byte < 4; byte++) {
                        uint32_t u = (words[wi] >> (8 * byte)) & 0xff;
                        uint32_t bi = w * 16 + wi * 4 + byte;
                        if (NBITS == 4) {
                            push(2 * bi, u & 0xf);  // low nibble = even group (pq_table.rs:274-280)
                            push(2 * bi + 1, u >> 4);
                        } else {
                            push(bi, u);
                        }
                    }
                }
            }
        } else if ((enc_dim & 3) == 0) {
            const uint32_t *cw = reinterpret_cast<const uint32_t *>(cr);
            for (uint32_t w = 0; w < enc_dim / 4; w++) {
                uint32_t word = cw[w];
#pragma unroll
                for (int byte = 0; byte < 4; byte++) {
                    uint32_t u = (word >> (8 * byte)) & 0xff;
                    uint32_t bi = w * 4 + byte;
                    if (NBITS == 4) {
                        push(2 * bi, u & 0xf);
                        push(2 * bi + 1, u >> 4);
                    } else {
                        push(bi, u);
                    }
                }
            }
        } else {
            for (uint32_t bi = 0; bi < enc_dim; bi++) {
                uint32_t u = cr[bi];
                if (NBITS == 4) {
                    push(2 * bi, u & 0xf);
                    push(2 * bi + 1, u >> 4);
                } else {
                    push(bi, u);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < BQ; b++) {
            if (b >= (int)a.nq) break;
            float d = sum[b];
            if (cosine) {  // pq_table.rs:294-299
                float norm0 = sqrtf(cdp);
                float norm1 = sqrtf(a.qsq[b]);
                float den = fmaxf(norm0 * norm1, 1e-10f);
                float r = sum[b] / den;
                d = 1.0f - r;
            }
            if (MODE == 0) {
                a.out[uint64_t(b) * a.ld + vb * blockDim.x + threadIdx.x] = valid ? d : INFINITY;
            } else if (valid && d <= tau[b]) {
                uint32_t pos = atomicAdd(hit_n, 1u);
                if (pos < ADC_WGBUF) {
                    hit_key[pos] = pair_key(d, uint32_t(row));
                    hit_q[pos] = b;
                } else {
                    atomicAdd(&a.cnt[b], a.cap + 1);  // mark the query as overflowed (-> dense path)
                }
            }
        }
    }
    if (MODE == 1) {
        __syncthreads();
        uint32_t total = hit_n[0];
        if (total > ADC_WGBUF) total = ADC_WGBUF;
        // ranks: each thread walks its strided entries (total <= 2048)
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
            uint32_t r_ = atomicAdd(&hit_n[1 + hit_q[i]], 1u);
            hit_q[i] |= r_ << 8;  // park the rank beside the query id (BQ <= 4, rank < 2048)
        }
        __syncthreads();
        if (threadIdx.x < BQ && hit_n[1 + threadIdx.x] > 0)
            hit_n[1 + BQ + threadIdx.x] = atomicAdd(&a.cnt[threadIdx.x], hit_n[1 + threadIdx.x]);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
            uint32_t q = hit_q[i] & 0xff, r_ = hit_q[i] >> 8;
            uint32_t slot = hit_n[1 + BQ + q] + r_;
            if (slot < a.cap) a.cand[uint64_t(q) * a.cap + slot] = hit_key[i];
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Quantised first pass of the ADC scan (4-bit codes, L2Sqr).
// The f32 scan above is bound by the LDS gather rate: one ds_read_b128 per (row, group) serves 4 queries (8 us per
// query and 1M rows at the LDS array's 256 B/clk/CU; measured 15.5).  The scan only has to find the rows whose ADC
// value is <= tau[q], so it can run on a coarser table as long as it never drops such a row:
//   lut16[q][g][c] = floor((lut[q][g][c] - mn[q][g]) / D[q]),   mn = min_c lut[q][g][.],   D[q] = sum_g range_g / 65000
// 16-bit entries, 8 queries side by side in one 16-B LDS entry -> one ds_read_b128 per (row, group) serves 8 queries,
// and because no 16-bit field can overflow (sum_g max_c lut16 <= 65000) the 8 sums are accumulated with plain 32-bit
// adds, two lookups per v_add3_u32.  For every row  M + D * S16 <= sum_g lut[g][code_g]  (real arithmetic; M = sum mn),
// and the reference's f32 left fold S of m non-negative terms satisfies S >= real * (1 - gamma_{m-1}), so
//   S <= tau   ==>   S16 <= T16 := floor((tau * (1 + 2 m 2^-24) - M (1 - 1e-12)) / D) + 2
// (the +2 and the 1e-12 cover the double-precision evaluation).  Rows with S16 <= T16 go to the candidate list of the
// query as bare row ids; k_pq_adc_exact then computes THEIR f32 ADC sums in group order with the f32 table in LDS and
// keeps those <= tau -- the same set, the same values, the same (adc, idx) order as the f32 scan produces.
// Queries whose table holds a non-finite or negative entry are flagged and take the f32 scan.
// ---------------------------------------------------------------------------------------------------
// Code rows are 160 B apart (m = 320): with one lane per row a 16-B code-word load of a wave touches 64 different
// 128-B lines, and the 1024 rows of a workgroup iteration (160 KB) do not fit the L1, so every line came from L2 up to 8
// times.  The scan therefore reads a word-major mirror of the codes: tile of 64 rows x word w x lane -> one wave load
// = 1 KB contiguous, every line fetched once.  Built once per table (the codes are immutable until the next build).
// Code rows that are not whole 16-B words (odd m, m not a multiple of 32: the DB's default m = ceil(dim / 3) gives 171 groups at
// dim 512, 342 at dim 1024) are padded with zero bytes up to the next word: the padded nibbles select entries of groups beyond
// m, whose tables are all zero in the quantised image (k_pq_quant16), so they add nothing to any sum.
__global__ __launch_bounds__(256) void k_pq_tile_codes(const uint8_t *__restrict__ codes, uint64_t n, uint32_t enc_dim, uint32_t nwords,
                                                       uint4 *__restrict__ tiled) {
    const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;  // output entry: ((tile * nwords) + w) * 64 + lane
    const uint64_t total = (n + 63) / 64 * 64 * nwords;
    if (i >= total) return;
    const uint64_t lane = i & 63, tw = i >> 6, tile = tw / nwords, w = tw - tile * nwords;
    const uint64_t row = tile * 64 + lane;
    uint32_t v[4] = {0u, 0u, 0u, 0u};
    if (row < n) {
        const uint8_t *src = codes + row * enc_dim;
        if ((enc_dim & 15u) == 0) {
            const uint4 x = reinterpret_cast<const uint4 *>(src)[w];
            v[0] = x.x, v[1] = x.y, v[2] = x.z, v[3] = x.w;
        } else {
            for (uint32_t b = 0; b < 16; b++) {
                const uint32_t at = uint32_t(w) * 16 + b;
                if (at < enc_dim) v[b >> 2] |= uint32_t(src[at]) << (8 * (b & 3));
            }
        }
    }
    tiled[i] = make_uint4(v[0], v[1], v[2], v[3]);
}

constexpr uint32_t ADC16_Q = 8;          // queries per workgroup pass
constexpr uint32_t ADC16_WGBUF = 2048;   // LDS hit buffer entries per workgroup

// one workgroup per query: table -> 16-bit image [q / NQ][g][c][q % NQ], offset M and step D.
// L2Sqr (NQ = 8): entries floor((lut - mn_g) / D), D = sum_g range_g / 65000: a LOWER bound of the sum.
// Cosine (NQ = 7): the ADC value is 1 - S / max(sqrt(C) |q|, 1e-10) with S = sum of dot-product entries (any sign) and
// C = sum of |centroid|^2 entries (pq_table.rs:262-299).  Slots 0..6 hold ceil((lut - mn_g) / D), D = sum range / 64000
// -- an UPPER bound of S --, slot 7 of every entry holds floor((cc - mnc_g) / DC): a lower bound of C, the same for all
// queries (written by the block of the group's first query).  A row survives when S_ub >= 0 and
// S_ub^2 >= rho^2 |q|^2 C_lb with rho = 1 - tau (less slack), see k_pq_adc16.
// m_pad = groups of the image (32 per 16-B code word); the entries of groups m .. m_pad - 1 are zero (padded code bytes).
__global__ __launch_bounds__(256) void k_pq_quant16(const float *__restrict__ lut, const float *__restrict__ cent_cache, uint32_t m, uint32_t m_pad,
                                                    uint32_t nq, int cosine, uint16_t *__restrict__ img, double *__restrict__ qM,
                                                    double *__restrict__ qD, double *__restrict__ qMC, double *__restrict__ qDC,
                                                    uint32_t *__restrict__ qflag) {
    const uint32_t q = blockIdx.x, t = threadIdx.x;
    const uint32_t NQ = cosine ? 7u : 8u;
    __shared__ double sR[256], sM[256], sA[256];
    __shared__ uint32_t sbad;
    if (t == 0) sbad = 0;
    __syncthreads();
    auto reduce3 = [&](double &a, double &b, double &c) {
        sR[t] = a;
        sM[t] = b;
        sA[t] = c;
        __syncthreads();
        for (uint32_t s = 128; s > 0; s >>= 1) {
            if (t < s) {
                sR[t] += sR[t + s];
                sM[t] += sM[t + s];
                sA[t] += sA[t + s];
            }
            __syncthreads();
        }
        a = sR[0];
        b = sM[0];
        c = sA[0];
        __syncthreads();
    };
    const float *lq = lut + uint64_t(q) * m * 16;
    double R = 0.0, M = 0.0, A = 0.0;  // sum of ranges, of minima, of the largest magnitudes (rounding slack of a mixed-sign sum)
    bool bad = false;
    for (uint32_t g = t; g < m; g += 256) {
        float mn = INFINITY, mx = -INFINITY, am = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const float v = lq[g * 16 + c];
            if (cosine ? !(fabsf(v) <= 3.0e38f) : (!(v >= 0.0f) || v > 3.0e38f)) bad = true;  // NaN, inf (L2Sqr: negative)
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
            am = fmaxf(am, fabsf(v));
        }
        R += double(mx) - double(mn);
        M += double(mn);
        A += double(am);
    }
    if (bad) atomicOr(&sbad, 1u);
    reduce3(R, M, A);
    const bool flag = sbad != 0 || !(R < 1.0e300) || !(fabs(M) < 1.0e300);
    const double D = (!flag && R > 0.0) ? R / (cosine ? 64000.0 : 65000.0) : 1.0;
    uint16_t *dst = img + uint64_t(q / NQ) * m_pad * 16 * 8 + (q % NQ);
    for (uint32_t i = m * 16 + t; i < m_pad * 16; i += 256) dst[i * 8] = 0;
    for (uint32_t g = t; g < m; g += 256) {
        float mn = INFINITY;
#pragma unroll
        for (int c = 0; c < 16; c++) mn = fminf(mn, lq[g * 16 + c]);
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const double y = (double(lq[g * 16 + c]) - double(mn)) / D;
            double x = flag ? 0.0 : (cosine ? ceil(y) : floor(y));
            x = x < 0.0 ? 0.0 : (x > 65000.0 ? 65000.0 : x);  // (unreachable clamps: sum of the maxima <= 64000 + m)
            dst[(g * 16 + c) * 8] = (uint16_t)x;
        }
    }
    double MC = 0.0, DC = 1.0;
    if (cosine) {  // the |centroid|^2 table: the same numbers in every block; the group's first query writes slot 7
        double RC = 0.0, Z = 0.0;
        MC = 0.0;
        for (uint32_t g = t; g < m; g += 256) {
            float mn = INFINITY, mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const float v = cent_cache[g * 16 + c];
                mn = fminf(mn, v);
                mx = fmaxf(mx, v);
            }
            RC += double(mx) - double(mn);
            MC += double(mn);
        }
        reduce3(RC, MC, Z);
        DC = RC > 0.0 && RC < 1.0e300 ? RC / 65000.0 : 1.0;
        if (q % NQ == 0) {
            uint16_t *dc = img + uint64_t(q / NQ) * m_pad * 16 * 8 + 7;
            for (uint32_t i = m * 16 + t; i < m_pad * 16; i += 256) dc[i * 8] = 0;
            for (uint32_t g = t; g < m; g += 256) {
                float mn = INFINITY;
#pragma unroll
                for (int c = 0; c < 16; c++) mn = fminf(mn, cent_cache[g * 16 + c]);
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    double x = floor((double(cent_cache[g * 16 + c]) - double(mn)) / DC);
                    x = x < 0.0 ? 0.0 : (x > 65000.0 ? 65000.0 : x);
                    dc[(g * 16 + c) * 8] = (uint16_t)x;
                }
            }
        }
    }
    if (t == 0) {
        // Cosine: the f32 left fold of mixed-sign terms errs by at most gamma_m * sum |t_g| <= m 2^-23 * A: part of the upper bound
        qM[q] = cosine ? M + 2.0 * double(m) * 0x1p-24 * A : M;
        qD[q] = D;
        qMC[q] = MC;
        qDC[q] = DC;
        qflag[q] = flag ? 1u : 0u;
    }
}

struct Adc16Args {
    const uint4 *codes_t;    // word-major mirror of the code rows (k_pq_tile_codes)
    uint64_t n;
    uint32_t enc_dim, m;     // m = groups of the IMAGE (32 per code word, >= the table's m); enc_dim = bytes of a code row
    const uint4 *img;        // [ceil(nq/8)][m*16] 16-B entries
    const double *qM, *qD;   // [nq]
    const double *qMC, *qDC; // [nq] Cosine: offset / step of the |centroid|^2 table (identical for all queries)
    const float *qsq;        // [nq] Cosine: |q|^2
    const uint32_t *qflag;   // [nq]
    const float *tau;        // [nq]
    uint32_t nq;
    uint64_t rows_per_wg;    // multiple of 64
    uint64_t *cand;          // [nq][cap] row ids (upper word 0)
    uint32_t *cnt;           // [nq]
    uint32_t cap;
    // SAMPLE mode (threshold sample on the quantised tables): every blk_step-th 1024-row block, dense sums out
    uint32_t blk_step;       // >= 1
    float *s16_out;          // [nq][ld_s] quantised sums of the sampled rows as floats (exact: <= 65000 + m), +inf past n
    uint64_t ld_s;
};

// NW = 16-B code words per row (enc_dim / 16) as a compile-time constant: the row loop is fully unrolled, every
// group's table offset is an instruction immediate and the address register of a lookup is ONE v_perm_b32 (byte b of
// the packed nibble offsets, plus bit 16 for the groups beyond the 16-bit immediate range).  NW = 0: runtime count,
// one extra add per lookup.  (Shift + mask + base add per lookup made the loop VALU-bound: 154 vector instructions
// per 32 lookups against 128 LDS cycles.)
// Early exit of rows that are out (L2Sqr: the running sums only grow; once all eight are above their thresholds in every lane of
// a wave, the wave's 64 rows cannot qualify).  Exact, but on the bench corpus (1M low-rank gist-like rows, 4-bit m = 320, ef =
// 100) it LOSES: 6.04 instead of 5.45 ms per 1000 queries -- the smallest of a wave's 512 (row, query) sums crosses its
// threshold too late for the skipped lookups to pay for one check per code word.  Measurement switch (make EXTRA=-DADC16_ABANDON=1).
#ifndef ADC16_ABANDON
#define ADC16_ABANDON 0
#endif
// SAMPLE = true (L2Sqr only): the same row sums over a strided block sample, written densely -- the threshold sample of the
// scan at the scan's own rate (the f32 kernel needed 0.36 ms for 17 408 rows x 1000 queries, a quarter of this kernel's rate
// per lookup).  The threshold only steers how many rows the scan keeps (the count is CHECKED afterwards and a short list is
// redone densely), so it may come from quantised sums: tau = M + D (s* + m/2), s* = the r-th smallest sampled sum, m/2 = the
// expected loss of the m floors.
template <int NW, bool COS, bool SAMPLE = false>
__global__ __launch_bounds__(1024) void k_pq_adc16(Adc16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    constexpr uint32_t NQ = COS ? 7 : 8;  // query slots of an entry (Cosine: slot 7 = the |centroid|^2 table)
    const uint32_t tid = threadIdx.x;
    const uint32_t m = a.m;
    const uint32_t q0 = blockIdx.y * NQ;
    uint4 *tab = reinterpret_cast<uint4 *>(smem16);                   // [m*16] entries of 8 x u16
    int32_t *thr = reinterpret_cast<int32_t *>(tab + m * 16);         // L2Sqr: [8] T16 per slot (-1: unused / flagged); Cosine: [8][4] floats
    uint32_t *hit_row = reinterpret_cast<uint32_t *>(thr + 32);       // [ADC16_WGBUF]
    uint32_t *hit_q = hit_row + ADC16_WGBUF;                          // [ADC16_WGBUF] slot | rank << 8
    uint32_t *hit_n = hit_q + ADC16_WGBUF;                            // [0] entries, [1..8] per-slot counts, [9..16] bases
    // the lookups address the table with absolute LDS offsets from 0: the dynamic segment is the kernel's only LDS
    // object, so it starts there; if a toolchain ever places it elsewhere the queries go to the f32 scan
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem16 != 0u) {
        if (tid < NQ && q0 + tid < a.nq) atomicAdd(&a.cnt[q0 + tid], a.cap + 1);
        return;
    }
    {
        const uint4 *src = a.img + uint64_t(blockIdx.y) * m * 16;
        for (uint32_t i = tid; i < m * 16; i += 1024) tab[i] = src[i];
        if (tid < 1 + 2 * ADC16_Q) hit_n[tid] = 0;
        if (!COS && !SAMPLE && tid < NQ) {
            int32_t T = -1;
            const uint32_t q = q0 + tid;
            if (q < a.nq && a.qflag[q] == 0) {
                const double tau = double(a.tau[q]);
                const double M = a.qM[q], D = a.qD[q];
                const double x = floor((tau * (1.0 + 2.0 * double(m) * 0x1p-24) - M * (1.0 - 1e-12)) / D) + 2.0;
                // tau = +inf / NaN (fewer sampled rows than the rank, NaN rows): everything passes -> the candidate list
                // overflows and the query takes the f32 scan
                T = !(x < 70000.0) ? 70000 : (x < 0.0 ? -1 : (int32_t)x);
            }
            thr[tid] = T;
        }
        if (COS && tid < 8) {
            // per query slot: S_ub = Ms + Ds * s16 (constants rounded up), test S_ub >= 0 && S_ub^2 (1 + 1e-6) >= K * C_lb with
            // K = rho^2 |q|^2 (1 - 1e-5), rho = (1 - tau) - 1e-6 (2 - tau): d_f32 <= tau implies S_f32 / den_f32 >= rho.
            // rho <= 0 (tau >= 1: rows of the opposite half-space qualify), a tiny |q| (the 1e-10 clamp of the reference can
            // act) or a flagged table: Ms = -inf, the query takes the f32 scan (too few hits).  Slot 7: the C table's constants.
            float *f = reinterpret_cast<float *>(thr) + tid * 4;
            float Ms = -INFINITY, Ds = 0.0f, K = INFINITY, tq = -1.0f;  // (unused / unsupported slot: neither test can pass)
            const uint32_t q = q0 + tid;
            if (tid < NQ && q < a.nq && a.qflag[q] == 0) {
                const double tau = double(a.tau[q]), qs = double(a.qsq[q]);
                const double rho = (1.0 - tau) - 1e-6 * (2.0 + fabs(tau));
                if (rho > 0.0 && rho <= 2.0 && qs > 1e-30 && qs < 1e30) {
                    Ms = float(a.qM[q] + fabs(a.qM[q]) * 1e-6 + 1e-30);
                    Ds = float(a.qD[q] * (1.0 + 1e-6));
                    K = float(rho * rho * qs * (1.0 - 1e-5));
                    tq = float(4e-20 / qs);  // C below this: |centroid sum| |q| may be under the reference's 1e-10 clamp
                }
            }
            if (tid == 7 && a.nq > 0) {  // C_lb = (MC + DC * c16) (1 - 4 m 2^-24 - 1e-6): the f32 fold of C and this evaluation round
                const double g1 = 1.0 - 4.0 * double(m) * 0x1p-24 - 1e-6;
                Ms = float(fmax(a.qMC[0], 0.0) * g1);
                Ds = float(a.qDC[0] * g1);
                K = 0.0f;
                tq = 0.0f;
            }
            f[0] = Ms;
            f[1] = Ds;
            f[2] = K;
            f[3] = tq;
        }
    }
    __syncthreads();
    int32_t T[8];
    float cMs[8], cDs[8], cK[8], cT[8];
#pragma unroll
    for (int b = 0; b < 8; b++) {
        T[b] = thr[b];
        if (COS) {
            const float *f = reinterpret_cast<const float *>(thr) + b * 4;
            cMs[b] = f[0];
            cDs[b] = f[1];
            cK[b] = f[2];
            cT[b] = f[3];
        }
    }
    // (ADC16_ABANDON) Tp = T + 1 per 16-bit half, 0 for an unused slot (always "above"), 65535 when T does not fit (never abandoned);
    // checked after every code word of the second half: one v_pk_max_u16 + compare per register
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    uint32_t Tp[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int32_t t0 = T[2 * i] + 1, t1 = T[2 * i + 1] + 1;
        Tp[i] = uint32_t(t0 > 65535 ? 65535 : t0) | (uint32_t(t1 > 65535 ? 65535 : t1) << 16);
    }
    // scan: this workgroup's contiguous share of the rows; sample: sampled blocks blockIdx.x, + gridDim.x, ... (block j of the
    // sample = rows [j * blk_step * 1024, + 1024))
    const uint64_t n_sb = SAMPLE ? ((a.n + 1023) / 1024 + a.blk_step - 1) / a.blk_step : 0;
    const uint64_t r_begin = SAMPLE ? uint64_t(blockIdx.x) * a.blk_step * 1024 : uint64_t(blockIdx.x) * a.rows_per_wg;
    const uint64_t r_end = SAMPLE ? (blockIdx.x < n_sb ? (n_sb - 1) * a.blk_step * 1024 + 1 : 0)
                                  : (r_begin + a.rows_per_wg < a.n ? r_begin + a.rows_per_wg : a.n);
    const uint64_t r_inc = SAMPLE ? uint64_t(gridDim.x) * a.blk_step * 1024 : 1024;
    const uint32_t nwords = NW ? (uint32_t)NW : a.m / 32;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) uint4 lds_u4;
#endif
    for (uint64_t rb = r_begin; rb < r_end; rb += r_inc) {
        const uint64_t row = rb + tid;
        const bool valid = SAMPLE ? row < a.n : row < r_end;
        // word w of row (rb + tid): tile = row / 64 (uniform in the wave: rb is a multiple of 64), lane = row % 64; rows
        // past n inside the last tile are zero padding
        // Lanes past the workgroup's share (a share is shorter than the 1024 lanes when the table has fewer than 1024 rows per
        // CU) must not form addresses from their row number: the last workgroup's reach up to 1023 rows past the table, i.e.
        // past the mirror -- found by tools/fuzz_pq.py as a memory-access fault on a 74 205-row table whose mirror ended on
        // a page boundary (configuration #727, seed 4242; the over-read is as old as the kernel).  They re-read a valid row.
        const uint64_t lrow = valid ? row : (SAMPLE ? a.n - 1 : r_end - 1);
        const uint4 *cw = a.codes_t + (lrow >> 6) * nwords * 64 + (lrow & 63);
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        // one 16-B code word = 32 groups.  Byte b of a 32-bit word: low nibble = group 8wi + 2b, high nibble the next one;
        // their byte offsets inside the group's 256-B block (code * 16) sit packed in the bytes of e4 / o4.
        // A step = the 8 groups of one 32-bit code word: byte b holds group 2b (low nibble) and 2b+1 (high nibble); their
        // byte offsets inside a group's 256-B block (code * 16) sit packed in the bytes of e4 / o4.  gbyte = byte offset
        // of the step's first group block, folded into the instructions' immediates; the table starts at LDS address 0
        // (checked at kernel entry), so a lookup's address register is the permute alone.
        auto issue = [&](uint32_t word, uint32_t gbyte, uint32_t hi /* 0 or 0x04: adds bit 16 through the permute */, uint4 *E, uint4 *O) {
            const uint32_t e4 = (word & 0x0f0f0f0fu) << 4, o4 = word & 0xf0f0f0f0u;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t sel = 0x0c000c00u | (hi << 16) | (uint32_t)b | (hi ? 0u : 0x000c0000u);
                const uint32_t oe = __builtin_amdgcn_perm(1u, e4, sel), oo = __builtin_amdgcn_perm(1u, o4, sel);
#if defined(__HIP_DEVICE_COMPILE__)
                E[b] = *reinterpret_cast<const lds_u4 *>(oe + gbyte + (2 * b) * 256);
                O[b] = *reinterpret_cast<const lds_u4 *>(oo + gbyte + (2 * b + 1) * 256);
#else
                E[b] = O[b] = make_uint4(oe, oo, gbyte, 0);  // (host pass of the single-source compile: never runs)
#endif
            }
        };
        auto consume = [&](const uint4 *E, const uint4 *O) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                a0 = a0 + E[b].x + O[b].x;
                a1 = a1 + E[b].y + O[b].y;
                a2 = a2 + E[b].z + O[b].z;
                a3 = a3 + E[b].w + O[b].w;
            }
            // pin the four running sums here: left alone, the optimiser re-associates the unrolled row into four
            // separate 320-term chains, finishes one and parks the other three components of every table entry
            // in scratch (measured: 1.7 - 7.8 KB of spills per lane)
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        };
        if (NW) {
            // software pipeline over the 4 NW steps of the row: the lookups of step i+1 are issued before the sums of
            // step i are taken, so 8 - 16 ds_read_b128 per wave are in flight while the adds run
            uint4 cv[NW ? NW : 1];
#pragma unroll
            for (int w = 0; w < NW; w++) cv[w] = cw[w * 64];
            uint4 E[2][4], O[2][4];
            issue(cv[0].x, 0u, 0u, E[0], O[0]);
            bool done = false;  // wave-uniform: every row of the wave is out (a `break` would keep the loop from unrolling)
#pragma unroll
            for (int st = 0; st < 4 * NW; st++) {
                if (done) continue;
                if (st + 1 < 4 * NW) {
                    const int w = (st + 1) >> 2, wi = (st + 1) & 3;
                    const uint32_t word = wi == 0 ? cv[w].x : (wi == 1 ? cv[w].y : (wi == 2 ? cv[w].z : cv[w].w));
                    // groups 32w .. 32w+31 at byte offset w * 8192: beyond 65535 the immediate cannot hold it -> bit 16
                    // rides in the address register (sel byte 2 = 4 picks the 0x01 of the constant operand)
                    issue(word, (w < 8 ? w : w - 8) * 8192 + wi * 2048, w < 8 ? 0u : 0x04u, E[(st + 1) & 1], O[(st + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks these reads below the adds of step st)
                }
                consume(E[st & 1], O[st & 1]);
                if (!COS && ADC16_ABANDON && (st & 3) == 3 && st + 1 < 4 * NW && st >= 2 * NW - 1) {
                    auto above = [](uint32_t acc, uint32_t tp) {
                        return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, acc), __builtin_bit_cast(u16x2, tp))) == acc;
                    };
                    const bool out = !valid || (above(a0, Tp[0]) && above(a1, Tp[1]) && above(a2, Tp[2]) && above(a3, Tp[3]));
                    done = __ballot(!out) == 0;  // the threshold test below then sees sums that are already above
                }
            }
        } else {
            uint4 v = cw[0];
            for (uint32_t w = 0; w < nwords; w++) {
                const uint4 cur = v;
                if (w + 1 < nwords) v = cw[(w + 1) * 64];  // next code word while this one is looked up
                const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                for (int wi = 0; wi < 4; wi++) {
                    uint4 E[4], O[4];
                    issue(words[wi], w * 8192 + wi * 2048, 0u, E, O);
                    consume(E, O);
                }
            }
        }
        const int32_t s[8] = {int32_t(a0 & 0xffffu), int32_t(a0 >> 16), int32_t(a1 & 0xffffu), int32_t(a1 >> 16),
                              int32_t(a2 & 0xffffu), int32_t(a2 >> 16), int32_t(a3 & 0xffffu), int32_t(a3 >> 16)};
        if (SAMPLE) {
            const uint64_t col = rb / (uint64_t(a.blk_step) * 1024) * 1024 + tid;  // position in the sample
#pragma unroll
            for (int b = 0; b < (int)NQ; b++)
                if (q0 + b < a.nq) a.s16_out[uint64_t(q0 + b) * a.ld_s + col] = valid ? float(s[b]) : INFINITY;
            continue;
        }
        float c_lb = 0.0f;
        if (COS) c_lb = fmaxf(cMs[7] + cDs[7] * float(s[7]), 0.0f);
#pragma unroll
        for (int b = 0; b < (int)NQ; b++) {
            bool pass;
            if (COS) {
                const float sub = cMs[b] + cDs[b] * float(s[b]);
                // (second term: even the lower bound of C cannot rule out the reference's 1e-10 clamp -> the exact stage decides)
                pass = (sub >= 0.0f && sub * sub * 1.000001f >= cK[b] * c_lb) || c_lb <= cT[b];
            } else {
                pass = s[b] <= T[b];
            }
            if (valid && pass) {
                const uint32_t pos = atomicAdd(hit_n, 1u);
                if (pos < ADC16_WGBUF) {
                    hit_row[pos] = uint32_t(row);
                    hit_q[pos] = b;
                } else {
                    atomicAdd(&a.cnt[q0 + b], a.cap + 1);  // mark the query as overflowed (-> f32 scan)
                }
            }
        }
    }
    if (SAMPLE) return;
    __syncthreads();
    uint32_t total = hit_n[0];
    if (total > ADC16_WGBUF) total = ADC16_WGBUF;
    for (uint32_t i = tid; i < total; i += 1024) {
        const uint32_t r_ = atomicAdd(&hit_n[1 + hit_q[i]], 1u);
        hit_q[i] |= r_ << 8;
    }
    __syncthreads();
    if (tid < NQ && hit_n[1 + tid] > 0) hit_n[1 + ADC16_Q + tid] = atomicAdd(&a.cnt[q0 + tid], hit_n[1 + tid]);
    __syncthreads();
    for (uint32_t i = tid; i < total; i += 1024) {
        const uint32_t b = hit_q[i] & 0xffu, r_ = hit_q[i] >> 8;
        const uint32_t slot = hit_n[1 + ADC16_Q + b] + r_;
        if (slot < a.cap) a.cand[uint64_t(q0 + b) * a.cap + slot] = hit_row[i];
    }
}

// exact f32 ADC sums (strict group order, pq_table.rs:254-292) of the candidates of every query: row ids in, pair keys
// out (PAIR_NONE for sums above tau); valid[q] counts the pairs kept.  One workgroup per query, its f32 table in LDS.
template <bool COS>
__global__ __launch_bounds__(256) void k_pq_adc_exact(const uint8_t *__restrict__ codes, uint32_t enc_dim, uint32_t m,
                                                      const float *__restrict__ lut, const float *__restrict__ cent_cache,
                                                      const float *__restrict__ qsq, const float *__restrict__ tau,
                                                      uint64_t *__restrict__ cand, const uint32_t *__restrict__ cnt,
                                                      uint32_t cap, uint32_t *__restrict__ valid) {
    extern __shared__ __attribute__((aligned(16))) float slut[];  // the query's table; Cosine: the |centroid|^2 table behind it
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t total = cnt[q];
    if (total > cap) return;  // overflowed list: the query is redone by the f32 scan
    const float *lq = lut + uint64_t(q) * m * 16;
    for (uint32_t i = tid; i < m * 16; i += 256) slut[i] = lq[i];
    const float *scc = slut + m * 16;
    if (COS)
        for (uint32_t i = tid; i < m * 16; i += 256) slut[m * 16 + i] = cent_cache[i];
    __syncthreads();
    const float t = tau[q];
    uint64_t *cq = cand + uint64_t(q) * cap;
    uint32_t kept = 0;
    for (uint32_t i = tid; i < total; i += 256) {
        const uint32_t row = uint32_t(cq[i]);
        const uint4 *cw = reinterpret_cast<const uint4 *>(codes + uint64_t(row) * enc_dim);
        float sum = 0.0f, cdp = 0.0f;
        const bool words = (enc_dim & 15u) == 0 && m == 2 * enc_dim;  // whole 16-B words AND every nibble a real group
        if (!words) {  // otherwise byte by byte, same group order (an odd m leaves the last high nibble without a table:
            // tools/fuzz_pq.py, Cosine, m = 31 / 63 -- the word loop below would add whatever sits behind the query's table)
            const uint8_t *cb = codes + uint64_t(row) * enc_dim;
            for (uint32_t g = 0; g < m; g += 2) {
                const uint32_t byte = cb[g >> 1];
                const uint32_t a0 = g * 16 + (byte & 0xf), a1 = (g + 1) * 16 + (byte >> 4);
                sum = sum + slut[a0];
                if (COS) cdp = cdp + scc[a0];
                if (g + 1 < m) {
                    sum = sum + slut[a1];
                    if (COS) cdp = cdp + scc[a1];
                }
            }
        }
        // code words in batches of WB, the next batch in flight while this one is looked up: every load is executed by every lane
        // of the branch (clamped word index) -- the earlier form fetched ONE word ahead under a
        // condition, i.e. a branch per word and a full round trip to the code rows (L2 / Infinity Cache) per 16 bytes: 10 per candidate
        // at m = 320, 131 us per 1000 queries of which the 320 table lookups and adds of a candidate are ~1.5 us
        constexpr uint32_t WB = 5;
        const uint32_t nw = words ? enc_dim / 16 : 0u, lastw = nw ? nw - 1 : 0u;
        uint4 cur[WB], nxt[WB];
        if (words) {  // (uniform; rows of other shapes are not 16-B aligned and were summed above)
#pragma unroll
        for (uint32_t j = 0; j < WB; j++) cur[j] = cw[j < lastw ? j : lastw];
        for (uint32_t w0 = 0; w0 < nw; w0 += WB) {
#pragma unroll
            for (uint32_t j = 0; j < WB; j++) {
                const uint32_t wn = w0 + WB + j;
                nxt[j] = cw[wn < lastw ? wn : lastw];
            }
#pragma unroll
            for (uint32_t jw = 0; jw < WB; jw++) {
                const uint32_t w = w0 + jw;
                if (w >= nw) break;  // uniform; no load behind it
                const uint32_t wd[4] = {cur[jw].x, cur[jw].y, cur[jw].z, cur[jw].w};
#pragma unroll
                for (int wi = 0; wi < 4; wi++) {
                    float e[8], c[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t at = (w * 32 + 8 * wi + j) * 16 + ((wd[wi] >> (4 * j)) & 0xf);
                        e[j] = slut[at];
                        if (COS) c[j] = scc[at];
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        sum = sum + e[j];
                        if (COS) cdp = cdp + c[j];
                    }
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < WB; j++) cur[j] = nxt[j];
        }
        }
        if (COS) {  // pq_table.rs:294-299, the operation order of k_pq_adc
            float norm0 = sqrtf(cdp);
            float norm1 = sqrtf(qsq[q]);
            float den = fmaxf(norm0 * norm1, 1e-10f);
            float r = sum / den;
            sum = 1.0f - r;
        }
        const bool keep = sum <= t;
        cq[i] = keep ? pair_key(sum, row) : PAIR_NONE;
        kept += keep ? 1u : 0u;
    }
    if (kept) atomicAdd(&valid[q], kept);
}

// ---------------------------------------------------------------------------------------------------
// 8-bit codes (n_bits = 8, pq_table.rs:142-145: 256 centroids per group) on a quantised pass of their own (round 3).
// A query's f32 table is m x 256 entries -- 327 KB at m = 320: no form of it that serves several queries fits LDS, and the
// f32 scan reads it through L1 / L2 (0.37 ms per query per 1M rows).  One query's table at ONE BYTE per entry does fit
// (m x 256 B = 80 KB):  lut8[g][c] = min(255, floor((lut[g][c] - mn_g) / D)),  D = sum_g range_g / (128 m).
// The minimum with 255 only lowers an entry, so M + D * S8 <= sum_g lut[g][code_g] still holds (M = sum mn_g) and the
// superset threshold of the 16-bit pass carries over:  S <= tau  ==>  S8 <= T8 := floor((tau (1 + 2 m 2^-24) - M (1 - 1e-12)) / D) + 2.
// All lanes of a wave look up the SAME group at a time: its 256 one-byte entries are exactly one word per LDS bank, so a
// wave's 64 lookups never conflict.  The scan is bound by the code bytes -- m bytes per row and query, one query per pass
// -- i.e. by HBM: 320 MB per query at m = 320.  Candidates get their exact f32 sums in group order (k_pq_adc_exact8: the
// query's f32 table read through L2) and the pairs above tau are dropped, as on the 16-bit pass.  L2Sqr only.
__global__ __launch_bounds__(256) void k_pq_quant8(const float *__restrict__ lut, uint32_t m, uint32_t m_pad, uint32_t nq, uint8_t *__restrict__ img,
                                                   double *__restrict__ qM, double *__restrict__ qD, uint32_t *__restrict__ qflag) {
    const uint32_t q = blockIdx.x, t = threadIdx.x;
    __shared__ double sR[256], sM[256];
    __shared__ float smn[1024];  // group minima (m <= 1024: the table of a larger m does not fit LDS anyway)
    __shared__ uint32_t sbad;
    if (t == 0) sbad = 0;
    __syncthreads();
    const float *lq = lut + uint64_t(q) * m * 256;
    double R = 0.0, M = 0.0;
    bool bad = false;
    // a wave per group, 4 entries per lane: coalesced reads, wave-level min / max
    for (uint32_t g = t >> 6; g < m; g += 4) {
        const float4 v = reinterpret_cast<const float4 *>(lq + g * 256)[t & 63];
        const float e[4] = {v.x, v.y, v.z, v.w};
        float mn = INFINITY, mx = -INFINITY;
        bool b = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!(e[j] >= 0.0f) || e[j] > 3.0e38f) b = true;  // NaN, negative, inf: the query takes the f32 scan
            mn = fminf(mn, e[j]);
            mx = fmaxf(mx, e[j]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, off));
            mx = fmaxf(mx, __shfl_xor(mx, off));
        }
        if (__ballot(b) != 0) bad = true;
        if ((t & 63) == 0) {
            smn[g] = mn;
            R += double(mx) - double(mn);
            M += double(mn);
        }
    }
    if (bad) atomicOr(&sbad, 1u);
    sR[t] = R;
    sM[t] = M;
    __syncthreads();
    for (uint32_t st = 128; st > 0; st >>= 1) {
        if (t < st) {
            sR[t] += sR[t + st];
            sM[t] += sM[t + st];
        }
        __syncthreads();
    }
    R = sR[0];
    M = sM[0];
    const bool flag = sbad != 0 || !(R < 1.0e300) || !(fabs(M) < 1.0e300);
    const double D = (!flag && R > 0.0) ? R / (128.0 * double(m)) : 1.0;
    uint8_t *dst = img + uint64_t(q) * m_pad * 256;
    for (uint32_t i = m * 256 + t; i < m_pad * 256; i += 256) dst[i] = 0;  // groups of the padded code bytes
    for (uint32_t i = t; i < m * 256; i += 256) {
        const float mn = smn[i >> 8];
        double x = flag ? 0.0 : floor((double(lq[i]) - double(mn)) / D);
        x = x < 0.0 ? 0.0 : (x > 255.0 ? 255.0 : x);
        dst[i] = (uint8_t)x;
    }
    if (t == 0) {
        qM[q] = M;
        qD[q] = D;
        qflag[q] = flag ? 1u : 0u;
    }
}

struct Adc8Args {
    const uint4 *codes_t;   // word-major mirror of the (zero-padded) code rows
    uint64_t n;
    uint32_t nwords, m;     // 16-B code words per row; groups of the table (the image holds 16 * nwords of them)
    const uint8_t *img;     // [nq][16 * nwords * 256]
    const double *qM, *qD;
    const uint32_t *qflag;
    const float *tau;
    uint32_t nq;
    uint64_t rows_per_wg;   // multiple of 64
    uint64_t *cand;         // [nq][cap] row ids
    uint32_t *cnt;          // [nq]
    uint32_t cap;
    uint32_t blk_step;      // SAMPLE: every blk_step-th 1024-row block
    float *s8_out;          // SAMPLE: [nq][ld_s] quantised sums as floats, +inf past n
    uint64_t ld_s;
};
constexpr uint32_t ADC8_WGBUF = 4096;  // LDS hit buffer entries (one query per workgroup)

template <bool SAMPLE>
__global__ __launch_bounds__(1024) void k_pq_adc8(Adc8Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    const uint32_t tid = threadIdx.x, q = blockIdx.y;
    const uint32_t tab_bytes = a.nwords * 16 * 256;
    uint8_t *tab = smem8;                                                  // [16 nwords][256]
    uint32_t *hit_row = reinterpret_cast<uint32_t *>(smem8 + tab_bytes);   // [ADC8_WGBUF]
    uint32_t *hit_n = hit_row + ADC8_WGBUF;                                // [0] entries, [1] threshold, [2] base of the flush
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.img + uint64_t(q) * tab_bytes);
        uint4 *dst = reinterpret_cast<uint4 *>(tab);
        for (uint32_t i = tid; i < tab_bytes / 16; i += 1024) dst[i] = src[i];
        if (tid == 0) {
            hit_n[0] = 0;
            int32_t T = -1;
            if (!SAMPLE && a.qflag[q] == 0) {
                const double x = floor((double(a.tau[q]) * (1.0 + 2.0 * double(a.m) * 0x1p-24) - a.qM[q] * (1.0 - 1e-12)) / a.qD[q]) + 2.0;
                // tau = +inf / NaN: everything passes -> the list overflows and the query takes the f32 scan
                T = !(x < 2.0e9) ? 2000000000 : (x < 0.0 ? -1 : (int32_t)x);
            }
            hit_n[1] = (uint32_t)T;
        }
    }
    __syncthreads();
    const int32_t T = (int32_t)hit_n[1];
    const uint64_t n_sb = SAMPLE ? ((a.n + 1023) / 1024 + a.blk_step - 1) / a.blk_step : 0;
    const uint64_t r_begin = SAMPLE ? uint64_t(blockIdx.x) * a.blk_step * 1024 : uint64_t(blockIdx.x) * a.rows_per_wg;
    const uint64_t r_end = SAMPLE ? (blockIdx.x < n_sb ? (n_sb - 1) * a.blk_step * 1024 + 1 : 0)
                                  : (r_begin + a.rows_per_wg < a.n ? r_begin + a.rows_per_wg : a.n);
    const uint64_t r_inc = SAMPLE ? uint64_t(gridDim.x) * a.blk_step * 1024 : 1024;
    const uint32_t nwords = a.nwords;
    for (uint64_t rb = r_begin; rb < r_end; rb += r_inc) {
        const uint64_t row = rb + tid;
        const bool valid = SAMPLE ? row < a.n : row < r_end;
        const uint64_t lrow = valid ? row : (SAMPLE ? a.n - 1 : r_end - 1);  // (see k_pq_adc16: idle lanes re-read a valid row)
        const uint4 *cw = a.codes_t + (lrow >> 6) * nwords * 64 + (lrow & 63);
        uint32_t s0 = 0, s1 = 0;  // two running sums: the lookups of a word are independent, the adds need not be one chain
        uint4 v = cw[0];
        for (uint32_t w = 0; w < nwords; w++) {
            const uint4 cur = v;
            if (w + 1 < nwords) v = cw[(w + 1) * 64];  // next code word while this one is looked up
            const uint8_t *tw = tab + w * 4096;       // 16 groups x 256 entries
            const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int wi = 0; wi < 4; wi++) {
                const uint32_t c = words[wi];
                const uint32_t e0 = tw[(4 * wi + 0) * 256 + (c & 0xffu)], e1 = tw[(4 * wi + 1) * 256 + ((c >> 8) & 0xffu)];
                const uint32_t e2 = tw[(4 * wi + 2) * 256 + ((c >> 16) & 0xffu)], e3 = tw[(4 * wi + 3) * 256 + (c >> 24)];
                s0 += e0 + e2;
                s1 += e1 + e3;
            }
        }
        const uint32_t S8 = s0 + s1;
        if (SAMPLE) {
            const uint64_t col = rb / (uint64_t(a.blk_step) * 1024) * 1024 + tid;
            a.s8_out[uint64_t(q) * a.ld_s + col] = valid ? float(S8) : INFINITY;
            continue;
        }
        if (valid && (int32_t)S8 <= T) {
            const uint32_t pos = atomicAdd(&hit_n[0], 1u);
            if (pos < ADC8_WGBUF)
                hit_row[pos] = uint32_t(row);
            else
                atomicAdd(&a.cnt[q], a.cap + 1);  // mark the query as overflowed (-> f32 scan)
        }
    }
    if (SAMPLE) return;
    __syncthreads();
    uint32_t total = hit_n[0];
    if (total > ADC8_WGBUF) total = ADC8_WGBUF;
    if (tid == 0 && total > 0) hit_n[2] = atomicAdd(&a.cnt[q], total);
    __syncthreads();
    const uint32_t base = hit_n[2];
    for (uint32_t i = tid; i < total; i += 1024) {
        const uint32_t slot = base + i;
        if (slot < a.cap) a.cand[uint64_t(q) * a.cap + slot] = hit_row[i];
    }
}

// ---------------------------------------------------------------------------------------------------
// 8-bit codes, EIGHT queries per pass (round 4).  k_pq_adc8 above reads the whole code mirror once per QUERY (one byte table per query
// fills the LDS): 320 GB of code bytes per 1000 queries at 1M rows, m = 320 -- 44 ms.  The 4-bit scan's scheme (k_pq_adc16: 16-bit
// entries, 8 queries side by side in a 16-B LDS entry, one ds_read_b128 per (row, group) for all eight) needs 256 entries x 16 B = 4 KB
// per group here, 1.3 MB for the table of 8 queries: it does not fit.  So the table is cut into SLICES of 32 groups (128 KB, two 16-B
// code words of a row): a workgroup keeps 4 rows per thread (4096 rows), and for every slice loads the slice of the image into LDS and
// adds the slice's 32 lookups to the rows' eight packed 16-bit sums, which stay in registers across the slices.  The image of a query
// group is re-read from L2 by every workgroup (1.3 MB against the 1.3 MB of code bytes of its 4096 rows), the code mirror once per 8
// queries.  Same quantisation and the same superset proof as k_pq_adc16 (entries floor((lut - mn_g) / D), D = sum of ranges / 65000:
// M + D S16 <= the reference's f32 sum up to its gamma_m, so S <= tau implies S16 <= T16); the candidates' exact f32 sums, the
// (adc, idx) order and the re-rank are k_pq_adc_exact8's and the merge's, unchanged.  L2Sqr tables (as the one-byte scan).
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t ADC16X8_RPT = 4;       // rows per thread
constexpr uint32_t ADC16X8_GS = 32;       // groups per slice (two 16-B code words)
constexpr uint32_t ADC16X8_WGBUF = 2048;  // LDS hit buffer entries per workgroup

// per query: the minimum of every group's 256 table entries (mn[q][g]), M = their sum, D = sum of the ranges / 65000
__global__ __launch_bounds__(256) void k_pq_quant16x8_stats(const float *__restrict__ lut, uint32_t m, uint32_t nq, float *__restrict__ mn_out,
                                                            double *__restrict__ qM, double *__restrict__ qD, uint32_t *__restrict__ qflag,
                                                            double divisor) {
    const uint32_t q = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    __shared__ double sR[4], sM[4];
    __shared__ uint32_t sbad;
    if (t == 0) sbad = 0;
    __syncthreads();
    const float *lq = lut + uint64_t(q) * m * 256;
    double R = 0.0, M = 0.0;
    bool bad = false;
    for (uint32_t g = wave; g < m; g += 4) {  // a wave per group: 4 entries per lane
        const float4 e = reinterpret_cast<const float4 *>(lq + uint64_t(g) * 256)[lane];
        const float v[4] = {e.x, e.y, e.z, e.w};
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!(v[j] >= 0.0f) || v[j] > 3.0e38f) bad = true;  // NaN, negative, inf: the query takes the f32 scan
            mn = fminf(mn, v[j]);
            mx = fmaxf(mx, v[j]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, off));
            mx = fmaxf(mx, __shfl_xor(mx, off));
        }
        if (lane == 0) {
            mn_out[uint64_t(q) * m + g] = mn;
            R += double(mx) - double(mn);
            M += double(mn);
        }
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(&sbad, 1u);
    if (lane == 0) {
        sR[wave] = R;
        sM[wave] = M;
    }
    __syncthreads();
    if (t == 0) {
        R = (sR[0] + sR[1]) + (sR[2] + sR[3]);
        M = (sM[0] + sM[1]) + (sM[2] + sM[3]);
        const bool flag = sbad != 0 || !(R < 1.0e300) || !(fabs(M) < 1.0e300);
        qM[q] = M;
        qD[q] = (!flag && R > 0.0) ? R / divisor : 1.0;  // (65000 for 16-bit entries; k_pq_adc8x16: see there)
        qflag[q] = flag ? 1u : 0u;
    }
}
// the image: img[(grp * m_pad + g) * 256 + c] = 8 x u16, slot b = floor((lut[8 grp + b][g][c] - mn) / D) (0 for g >= m, for query slots
// past nq and for flagged queries: padded code bytes select entry 0 of an all-zero group).  One thread per 16-B entry.
__global__ __launch_bounds__(256) void k_pq_quant16x8_img(const float *__restrict__ lut, const float *__restrict__ mn, const double *__restrict__ qD,
                                                          const uint32_t *__restrict__ qflag, uint32_t m, uint32_t m_pad, uint32_t nq,
                                                          uint4 *__restrict__ img) {
    const uint32_t grp = blockIdx.y, g = blockIdx.x, c = threadIdx.x;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if (g < m) {
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint32_t q = grp * 8 + b;
            if (q < nq && qflag[q] == 0) {
                const double y = (double(lut[(uint64_t(q) * m + g) * 256 + c]) - double(mn[uint64_t(q) * m + g])) / qD[q];
                double x = floor(y);
                x = x < 0.0 ? 0.0 : (x > 65000.0 ? 65000.0 : x);  // (unreachable clamps: the sum of the maxima is <= 65000)
                w[b >> 1] |= uint32_t(x) << (16 * (b & 1));
            }
        }
    }
    img[(uint64_t(grp) * m_pad + g) * 256 + c] = make_uint4(w[0], w[1], w[2], w[3]);
}

struct Adc16x8Args {
    const uint4 *codes_t;   // word-major mirror of the (zero-padded) code rows
    uint64_t n;
    uint32_t nwords, m;     // 16-B code words per row; groups of the table
    uint32_t nslices;       // slices of 32 groups of the image (m_pad = 32 nslices)
    const uint4 *img;       // [ngrp][m_pad * 256] 16-B entries
    const double *qM, *qD;  // [nq] of the 16-bit quantisation
    const uint32_t *qflag;
    const float *tau;
    uint32_t nq;
    uint64_t rows_per_wg;   // multiple of 64
    uint64_t *cand;         // [nq][cap] row ids
    uint32_t *cnt;          // [nq]
    uint32_t cap;
    // k_pq_adc8x16<SAMPLE>: the quantised sums of every blk_step-th block of 1024 rows, as floats (+inf past the table), [nq][ld_s]
    uint32_t blk_step = 0;
    uint64_t n_sb = 0;      // sampled blocks
    float *s8_out = nullptr;
    uint64_t ld_s = 0;
};

__global__ __launch_bounds__(1024) void k_pq_adc16x8(Adc16x8Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem168[];
    constexpr uint32_t RPT = ADC16X8_RPT, GS = ADC16X8_GS;
    uint4 *tab = reinterpret_cast<uint4 *>(smem168);                                   // [GS][256] entries of 8 x u16
    uint32_t *hit_row = reinterpret_cast<uint32_t *>(smem168 + size_t(GS) * 256 * 16);  // [WGBUF]
    uint32_t *hit_q = hit_row + ADC16X8_WGBUF;                                         // [WGBUF] slot | rank << 8
    uint32_t *hit_n = hit_q + ADC16X8_WGBUF;                                           // [0] entries, [1..8] per-slot counts, [9..16] bases
    int32_t *thr = reinterpret_cast<int32_t *>(hit_n + 20);                            // [8] T16 per slot (-1: unused / flagged)
    const uint32_t tid = threadIdx.x, q0 = blockIdx.y * 8, m = a.m, nwords = a.nwords;
    if (tid < 17) hit_n[tid] = 0;
    if (tid < 8) {
        int32_t T = -1;
        const uint32_t q = q0 + tid;
        if (q < a.nq && a.qflag[q] == 0) {
            const double x = floor((double(a.tau[q]) * (1.0 + 2.0 * double(m) * 0x1p-24) - a.qM[q] * (1.0 - 1e-12)) / a.qD[q]) + 2.0;
            // tau = +inf / NaN: everything passes -> the candidate list overflows and the query takes the f32 scan
            T = !(x < 70000.0) ? 70000 : (x < 0.0 ? -1 : (int32_t)x);
        }
        thr[tid] = T;
    }
    __syncthreads();
    int32_t T[8];
#pragma unroll
    for (int b = 0; b < 8; b++) T[b] = thr[b];
    const uint4 *img = a.img + uint64_t(blockIdx.y) * a.nslices * GS * 256;
    const uint64_t r_begin = uint64_t(blockIdx.x) * a.rows_per_wg;
    const uint64_t r_end = r_begin + a.rows_per_wg < a.n ? r_begin + a.rows_per_wg : a.n;
    for (uint64_t rb = r_begin; rb < r_end; rb += uint64_t(RPT) * 1024) {
        uint32_t acc[RPT][4];
        const uint4 *cw[RPT];
        bool valid[RPT];
#pragma unroll
        for (int j = 0; j < (int)RPT; j++) {
            acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0u;
            const uint64_t row = rb + uint64_t(j) * 1024 + tid;
            valid[j] = row < r_end;
            const uint64_t lrow = valid[j] ? row : r_end - 1;  // (idle lanes re-read a valid row: see k_pq_adc16)
            cw[j] = a.codes_t + (lrow >> 6) * nwords * 64 + (lrow & 63);
        }
        for (uint32_t sl = 0; sl < a.nslices; sl++) {
            __syncthreads();  // the previous slice's readers are done
            {
                const uint4 *src = img + uint64_t(sl) * GS * 256;
#pragma unroll
                for (int i = 0; i < 8; i++) tab[i * 1024 + tid] = src[i * 1024 + tid];
            }
            // the slice's two code words of the thread's rows (zero words past the mirror's last: entry 0 of an all-zero group)
            uint4 c0[RPT], c1[RPT];
            const uint32_t w0 = 2 * sl, w1 = 2 * sl + 1;
#pragma unroll
            for (int j = 0; j < (int)RPT; j++) {
                c0[j] = w0 < nwords ? cw[j][w0 * 64] : make_uint4(0u, 0u, 0u, 0u);
                c1[j] = w1 < nwords ? cw[j][w1 * 64] : make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < (int)RPT; j++) {
                const uint32_t words[8] = {c0[j].x, c0[j].y, c0[j].z, c0[j].w, c1[j].x, c1[j].y, c1[j].z, c1[j].w};
                uint32_t a0 = acc[j][0], a1 = acc[j][1], a2 = acc[j][2], a3 = acc[j][3];
                // software pipeline over the 8 words: the four lookups of word k + 1 are issued before the sums of word k are taken
                uint4 E[2][4];
                auto issue = [&](int k, uint4 *e) {
#pragma unroll
                    for (int b = 0; b < 4; b++) e[b] = tab[(4 * k + b) * 256 + ((words[k] >> (8 * b)) & 0xffu)];
                };
                issue(0, E[0]);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (k + 1 < 8) {
                        issue(k + 1, E[(k + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks these reads below the adds of word k)
                    }
                    const uint4 *e = E[k & 1];
                    a0 = a0 + e[0].x + e[1].x;
                    a1 = a1 + e[0].y + e[1].y;
                    a2 = a2 + e[0].z + e[1].z;
                    a3 = a3 + e[0].w + e[1].w;
                    a0 = a0 + e[2].x + e[3].x;
                    a1 = a1 + e[2].y + e[3].y;
                    a2 = a2 + e[2].z + e[3].z;
                    a3 = a3 + e[2].w + e[3].w;
                    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));  // (keeps the four running sums four: see k_pq_adc16)
                }
                acc[j][0] = a0;
                acc[j][1] = a1;
                acc[j][2] = a2;
                acc[j][3] = a3;
            }
        }
#pragma unroll
        for (int j = 0; j < (int)RPT; j++) {
            const int32_t s[8] = {int32_t(acc[j][0] & 0xffffu), int32_t(acc[j][0] >> 16), int32_t(acc[j][1] & 0xffffu), int32_t(acc[j][1] >> 16),
                                  int32_t(acc[j][2] & 0xffffu), int32_t(acc[j][2] >> 16), int32_t(acc[j][3] & 0xffffu), int32_t(acc[j][3] >> 16)};
            const uint32_t row = uint32_t(rb + uint64_t(j) * 1024 + tid);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                if (valid[j] && s[b] <= T[b]) {
                    const uint32_t pos = atomicAdd(hit_n, 1u);
                    if (pos < ADC16X8_WGBUF) {
                        hit_row[pos] = row;
                        hit_q[pos] = b;
                    } else {
                        atomicAdd(&a.cnt[q0 + b], a.cap + 1);  // mark the query as overflowed (-> f32 scan)
                    }
                }
            }
        }
        // hand the block's hits to the per-query candidate lists (one global atomic per query), leave an empty buffer
        __syncthreads();
        uint32_t total = hit_n[0];
        if (total > ADC16X8_WGBUF) total = ADC16X8_WGBUF;
        for (uint32_t i = tid; i < total; i += 1024) {
            const uint32_t r_ = atomicAdd(&hit_n[1 + hit_q[i]], 1u);
            hit_q[i] |= r_ << 8;
        }
        __syncthreads();
        if (tid < 8 && hit_n[1 + tid] > 0) hit_n[9 + tid] = atomicAdd(&a.cnt[q0 + tid], hit_n[1 + tid]);
        __syncthreads();
        for (uint32_t i = tid; i < total; i += 1024) {
            const uint32_t b = hit_q[i] & 0xffu, r_ = hit_q[i] >> 8;
            const uint32_t slot = hit_n[9 + b] + r_;
            if (slot < a.cap) a.cand[uint64_t(q0 + b) * a.cap + slot] = hit_row[i];
        }
        __syncthreads();
        if (tid < 17) hit_n[tid] = 0;
    }
}

// ---------------------------------------------------------------------------------------------------
// 8-bit codes, SIXTEEN queries per pass on one-byte entries (round 4, second step).  k_pq_adc16x8 is bound by the LDS: a group's 256
// entries of 16 B span 4 KB, the 16 lanes of a ds_read_b128 quarter hit random 4-bank groups (expected worst load ~4.3 against the 2 of
// a conflict-free read), 0.27 cycles per lane lookup.  The conflicts are in the data (the codes); what can change is how many queries
// one 16-B lookup serves: here an entry holds ONE BYTE for each of 16 queries,
//     e8[g][c] = min(255, floor((lut[g][c] - mn_g) / D)),   D = sum_g range_g / min(65000, 512 m)
// so the same LDS time serves twice the queries and the code mirror is read once per 16 of them.  The minimum with 255 only LOWERS an
// entry (M + D S8 <= the f32 sum still holds, as in k_pq_adc8) and it only touches entries far up a group's range -- lookups of rows
// that are far from the query in that group; D itself is the 16-bit pass's (R / 65000 from m = 127 on), so rows near the threshold
// keep the resolution they had.  The sums of a row stay below 2^16 without saturation or wrap: sum_g e8 <= sum_g range_g / D <=
// 65000.  A dword of an entry holds slots 4d, 4d+2, 4d+1, 4d+3 (bytes 0 .. 3): `x & 0x00ff00ff` is the pair of 16-bit addends of slots
// (4d, 4d+1) and one v_perm_b32 that of (4d+2, 4d+3); eight packed 16-bit pairs per row, two rows per thread (2048 rows per workgroup),
// slices of 16 groups (see the kernel).  Same superset rule, same exact stage and order.  L2Sqr tables.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t ADC8X16_RPT = 2;       // rows per thread
constexpr uint32_t ADC8X16_GS = 16;       // groups per slice (one 16-B code word): 16 x 256 x 16 B = 64 KB, two blocks in LDS
constexpr uint32_t ADC8X16_WGBUF = 2048;  // LDS hit buffer entries per workgroup

// the image: img[(grp * m_pad + g) * 256 + c] = 16 bytes, slot b of dword b / 4 at byte {0, 2, 1, 3}[b % 4] (0 for g >= m, for query
// slots past nq and for flagged queries).  One thread per 16-B entry.
__global__ __launch_bounds__(256) void k_pq_quant8x16_img(const float *__restrict__ lut, const float *__restrict__ mn, const double *__restrict__ qD,
                                                          const uint32_t *__restrict__ qflag, uint32_t m, uint32_t m_pad, uint32_t nq,
                                                          uint4 *__restrict__ img) {
    const uint32_t grp = blockIdx.y, g = blockIdx.x, c = threadIdx.x;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if (g < m) {
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const uint32_t q = grp * 16 + b;
            if (q < nq && qflag[q] == 0) {
                const double y = (double(lut[(uint64_t(q) * m + g) * 256 + c]) - double(mn[uint64_t(q) * m + g])) / qD[q];
                double x = floor(y);
                x = x < 0.0 ? 0.0 : (x > 255.0 ? 255.0 : x);
                const int j = b & 3, pos = j == 1 ? 2 : (j == 2 ? 1 : j);
                w[b >> 2] |= uint32_t(x) << (8 * pos);
            }
        }
    }
    img[(uint64_t(grp) * m_pad + g) * 256 + c] = make_uint4(w[0], w[1], w[2], w[3]);
}

// 16-B piece per lane from global memory straight into LDS (no register destination): lane l's piece lands at lds_dst + 16 l
__device__ __forceinline__ void pq_glds16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// The slices are 16 groups (one code word, 64 KB) in TWO LDS blocks: while a slice is scanned the next one is on its way into the other
// block by LDS-DMA and the rows' next code words into registers -- with one 128-KB block the workgroup stood still for every load (all
// waves at the barrier: ~5 k of ~23 k cycles per slice).
// SAMPLE: the threshold sample on the same image (it was a pass of its own on one-byte tables of ONE query per workgroup, k_pq_adc8<true> +
// k_pq_quant8: 0.97 ms per 1000 queries): a workgroup scores RPT sampled blocks of 1024 rows for its 16 queries and writes the sums.
template <bool SAMPLE>
__global__ __launch_bounds__(1024) void k_pq_adc8x16(Adc16x8Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem816[];
    constexpr uint32_t RPT = ADC8X16_RPT, GS = ADC8X16_GS, BLK = GS * 256 * 16;        // 64 KB per block
    uint32_t *hit_row = reinterpret_cast<uint32_t *>(smem816 + 2 * size_t(BLK));       // [WGBUF]
    uint32_t *hit_q = hit_row + ADC8X16_WGBUF;                                         // [WGBUF] slot | rank << 8
    uint32_t *hit_n = hit_q + ADC8X16_WGBUF;                                           // [0] entries, [1..16] per-slot counts, [17..32] bases
    int32_t *thr = reinterpret_cast<int32_t *>(hit_n + 36);                            // [16] T8 per slot (-1: unused / flagged)
    const uint32_t tid = threadIdx.x, q0 = blockIdx.y * 16, m = a.m, nwords = a.nwords, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the lookups form LDS byte addresses themselves: the table has to start at LDS address 0 (it does: the kernel has no static LDS);
    // were it ever not so, every query of the group is marked as overflowed and answered by the f32 scan
    typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const v4u_t *lds_u4;
    typedef __attribute__((address_space(3))) unsigned char *lds_b;
    if ((uint32_t)(uintptr_t)(lds_b)smem816 != 0u) {
        if (tid < 16 && q0 + tid < a.nq && blockIdx.x == 0) atomicAdd(&a.cnt[q0 + tid], a.cap + 1);
        return;
    }
    if (tid < 33) hit_n[tid] = 0;
    if (!SAMPLE && tid < 16) {
        int32_t T = -1;
        const uint32_t q = q0 + tid;
        if (q < a.nq && a.qflag[q] == 0) {
            const double x = floor((double(a.tau[q]) * (1.0 + 2.0 * double(m) * 0x1p-24) - a.qM[q] * (1.0 - 1e-12)) / a.qD[q]) + 2.0;
            // tau = +inf / NaN: everything passes -> the candidate list overflows and the query takes the f32 scan
            T = !(x < 70000.0) ? 70000 : (x < 0.0 ? -1 : (int32_t)x);
        }
        thr[tid] = T;
    }
    const uint4 *img = a.img + uint64_t(blockIdx.y) * a.nslices * GS * 256;
    // SAMPLE: ONE pass; thread tid's j-th row is row tid of sampled block RPT blockIdx.x + j (block b = rows [b blk_step 1024, + 1024))
    const uint64_t r_begin = SAMPLE ? 0 : uint64_t(blockIdx.x) * a.rows_per_wg;
    const uint64_t r_end = SAMPLE ? 1 : (r_begin + a.rows_per_wg < a.n ? r_begin + a.rows_per_wg : a.n);
    uint32_t four = 4u, blk1 = BLK;
    asm volatile("" : "+s"(four), "+s"(blk1));
    // slice sl -> block bk: 4 DMA instructions per wave (4096 entries of 16 B, wave w moves entries 1024 i + 64 w .. + 63)
    auto dma = [&](uint32_t sl, uint32_t bk) {
        const uint4 *src = img + uint64_t(sl) * GS * 256 + tid;
#pragma unroll
        for (int i = 0; i < 4; i++) pq_glds16(src + i * 1024, bk * BLK + (i * 1024 + wave * 64) * 16);
    };
    for (uint64_t rb = r_begin; rb < r_end; rb += uint64_t(RPT) * 1024) {
        uint32_t acc[RPT][8];
        const uint4 *cw[RPT];
        bool valid[RPT];
        uint4 cnext[RPT];
        __syncthreads();  // (thresholds / counters written; the previous block's readers are done)
        dma(0, 0);
#pragma unroll
        for (int j = 0; j < (int)RPT; j++) {
#pragma unroll
            for (int d = 0; d < 8; d++) acc[j][d] = 0u;
            const uint64_t sblk = uint64_t(blockIdx.x) * RPT + j;
            const uint64_t row = SAMPLE ? sblk * a.blk_step * 1024 + tid : rb + uint64_t(j) * 1024 + tid;
            valid[j] = SAMPLE ? (sblk < a.n_sb && row < a.n) : row < r_end;
            const uint64_t lrow = valid[j] ? row : (SAMPLE ? a.n - 1 : r_end - 1);  // (idle lanes re-read a valid row: see k_pq_adc16)
            cw[j] = a.codes_t + (lrow >> 6) * nwords * 64 + (lrow & 63);
            cnext[j] = cw[j][0];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        auto slice = [&](uint32_t sl, auto bkc) {
            constexpr uint32_t BK = decltype(bkc)::value;
            uint32_t words[RPT][4];
#pragma unroll
            for (int j = 0; j < (int)RPT; j++) {
                words[j][0] = cnext[j].x;
                words[j][1] = cnext[j].y;
                words[j][2] = cnext[j].z;
                words[j][3] = cnext[j].w;
            }
            if (sl + 1 < a.nslices) {  // the next slice: its entries into the other block, the rows' next code word into registers
                dma(sl + 1, BK ^ 1u);
#pragma unroll
                for (int j = 0; j < (int)RPT; j++) cnext[j] = sl + 1 < nwords ? cw[j][uint64_t(sl + 1) * 64] : make_uint4(0u, 0u, 0u, 0u);
            }
            // entry address = code byte x 16 + group x 4096 (+ 64 KB in block 1): the byte comes out of its word shifted in ONE
            // instruction (SDWA source select), the group part is the immediate offset of the ds_read
            uint4 E[2][4];
            auto issue = [&](int t, uint4 *e) {
                const int j = t >> 2, k = t & 3;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    uint32_t off;
                    if (b == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "s"(four), "v"(words[j][k]));
                    if (b == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(off) : "s"(four), "v"(words[j][k]));
                    if (b == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(off) : "s"(four), "v"(words[j][k]));
                    if (b == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(off) : "s"(four), "v"(words[j][k]));
                    if (BK) off += blk1;
                    const v4u_t ev = *(lds_u4)(uintptr_t)(off + (4 * k + b) * 4096);
                    e[b] = make_uint4(ev.x, ev.y, ev.z, ev.w);
                }
            };
            issue(0, E[0]);
#pragma unroll
            for (int t = 0; t < 4 * (int)RPT; t++) {
                if (t + 1 < 4 * (int)RPT) {
                    issue(t + 1, E[(t + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);  // (the reads of step t + 1 stay above the sums of step t)
                }
                const uint4 *e = E[t & 1];
                const int j = t >> 2;
                const uint32_t x[4][4] = {{e[0].x, e[1].x, e[2].x, e[3].x}, {e[0].y, e[1].y, e[2].y, e[3].y},
                                          {e[0].z, e[1].z, e[2].z, e[3].z}, {e[0].w, e[1].w, e[2].w, e[3].w}};
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    constexpr uint32_t MK = 0x00ff00ffu, SEL = 0x0c030c01u;  // bytes 0, 2 | bytes 1, 3 as 16-bit pairs
                    uint32_t A = acc[j][2 * d], B = acc[j][2 * d + 1];
                    A = A + (x[d][0] & MK) + (x[d][1] & MK);
                    A = A + (x[d][2] & MK) + (x[d][3] & MK);
                    B = B + __builtin_amdgcn_perm(0u, x[d][0], SEL) + __builtin_amdgcn_perm(0u, x[d][1], SEL);
                    B = B + __builtin_amdgcn_perm(0u, x[d][2], SEL) + __builtin_amdgcn_perm(0u, x[d][3], SEL);
                    acc[j][2 * d] = A;
                    acc[j][2 * d + 1] = B;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next slice has landed (DMA is not counted by the compiler)
            __syncthreads();                                    // ... for every wave, and this slice's readers are done
        };
        for (uint32_t sl = 0; sl < a.nslices; sl += 2) {
            slice(sl, std::integral_constant<uint32_t, 0>{});
            if (sl + 1 < a.nslices) slice(sl + 1, std::integral_constant<uint32_t, 1>{});
        }
        if constexpr (SAMPLE) {
#pragma unroll
            for (int j = 0; j < (int)RPT; j++) {
                const uint64_t sblk = uint64_t(blockIdx.x) * RPT + j;
                if (sblk >= a.n_sb) continue;  // uniform
#pragma unroll
                for (int b = 0; b < 16; b++) {
                    const uint32_t sb = (b & 1) ? (acc[j][b >> 1] >> 16) : (acc[j][b >> 1] & 0xffffu);
                    if (q0 + b < a.nq) a.s8_out[uint64_t(q0 + b) * a.ld_s + sblk * 1024 + tid] = valid[j] ? float(sb) : INFINITY;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < (int)RPT; j++) {
            const uint32_t row = uint32_t(rb + uint64_t(j) * 1024 + tid);
#pragma unroll
            for (int b = 0; b < 16; b++) {
                const int32_t sb = int32_t((b & 1) ? (acc[j][b >> 1] >> 16) : (acc[j][b >> 1] & 0xffffu));
                if (valid[j] && sb <= thr[b]) {
                    const uint32_t pos = atomicAdd(hit_n, 1u);
                    if (pos < ADC8X16_WGBUF) {
                        hit_row[pos] = row;
                        hit_q[pos] = b;
                    } else {
                        atomicAdd(&a.cnt[q0 + b], a.cap + 1);  // mark the query as overflowed (-> f32 scan)
                    }
                }
            }
        }
        // hand the block's hits to the per-query candidate lists (one global atomic per query), leave an empty buffer
        __syncthreads();
        uint32_t total = hit_n[0];
        if (total > ADC8X16_WGBUF) total = ADC8X16_WGBUF;
        for (uint32_t i = tid; i < total; i += 1024) {
            const uint32_t r_ = atomicAdd(&hit_n[1 + hit_q[i]], 1u);
            hit_q[i] |= r_ << 8;
        }
        __syncthreads();
        if (tid < 16 && hit_n[1 + tid] > 0) hit_n[17 + tid] = atomicAdd(&a.cnt[q0 + tid], hit_n[1 + tid]);
        __syncthreads();
        for (uint32_t i = tid; i < total; i += 1024) {
            const uint32_t b = hit_q[i] & 0xffu, r_ = hit_q[i] >> 8;
            const uint32_t slot = hit_n[17 + b] + r_;
            if (slot < a.cap) a.cand[uint64_t(q0 + b) * a.cap + slot] = hit_row[i];
        }
        __syncthreads();
        if (tid < 33) hit_n[tid] = 0;
    }
}

// exact f32 ADC sums of the candidates of an 8-bit table, strict group order (pq_table.rs:254-292).  Row ids in, pair keys out
// (PAIR_NONE above tau).  One workgroup per query, a candidate per thread: the query's f32 table (327 KB at m = 320) passes through LDS in
// slices of 32 groups, loaded with coalesced 16-B reads, and every thread adds its candidate's 32 entries of the slice to its running sum
// -- the order of the adds is the group order.  (Reading the table in place cost a 64-B sector per 4-B entry, from L2 / the Infinity
// Cache once some thirty queries' tables were in use per XCD: 1.97 ms per 1000 queries for ~1000 candidates each; code bytes one at a
// time from the row-major codes were 320 more loads per candidate.)  The code bytes come as 16-B words from the word-major mirror.
constexpr uint32_t EXACT8_GS = 32;  // groups per slice: 32 KB of LDS
constexpr uint32_t EXACT8_CPT = 4;  // candidates per thread and pass over the table
// ROWMAJOR: the 32 code bytes of a slice are two 16-B words of the row-major code row (enc_dim % 16 == 0: aligned); otherwise they come
// from the word-major mirror, whose 16-B words of one row lie 1 KB apart (a 64-B sector each: 4x the bytes)
template <bool ROWMAJOR>
__global__ __launch_bounds__(1024) void k_pq_adc_exact8(const uint8_t *__restrict__ codes, uint32_t enc_dim, const uint4 *__restrict__ codes_t, uint32_t nwords,
                                                        uint32_t m, const float *__restrict__ lut, const float *__restrict__ tau, uint64_t *__restrict__ cand,
                                                        const uint32_t *__restrict__ cnt, uint32_t cap, uint32_t *__restrict__ valid) {
    __shared__ __attribute__((aligned(16))) float tab[EXACT8_GS * 256];
    constexpr int CPT = (int)EXACT8_CPT;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t total = cnt[q];
    if (total > cap || total == 0) return;  // overflowed list: the query is redone by the f32 scan
    const float *lq = lut + uint64_t(q) * m * 256;
    const float t = tau[q];
    uint64_t *cq = cand + uint64_t(q) * cap;
    uint32_t kept = 0;
    for (uint32_t base = 0; base < total; base += CPT * 1024) {
        uint32_t row[CPT];
        const uint4 *cw[CPT];
        float sum[CPT];
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            const uint32_t i = base + c * 1024 + tid;
            row[c] = uint32_t(cq[i < total ? i : base]);
            cw[c] = ROWMAJOR ? reinterpret_cast<const uint4 *>(codes + uint64_t(row[c]) * enc_dim)
                             : codes_t + uint64_t(row[c] >> 6) * nwords * 64 + (row[c] & 63);
            sum[c] = 0.0f;
        }
        const uint32_t nlive = total - base < CPT * 1024 ? total - base : CPT * 1024;  // candidates of this pass
        for (uint32_t g0 = 0; g0 < m; g0 += EXACT8_GS) {
            const uint32_t ng = m - g0 < EXACT8_GS ? m - g0 : EXACT8_GS;
            __syncthreads();  // the previous slice's readers are done
            {
                const float4 *src = reinterpret_cast<const float4 *>(lq + uint64_t(g0) * 256);
                for (uint32_t e = tid; e < ng * 64; e += 1024) reinterpret_cast<float4 *>(tab)[e] = src[e];
            }
            const uint32_t w0 = g0 / 16;
            uint4 c0[CPT], c1[CPT];
#pragma unroll
            for (int c = 0; c < CPT; c++) {
                if (uint32_t(c) * 1024 < nlive) {  // (uniform: whole candidate ranks past the list are skipped)
                    c0[c] = cw[c][ROWMAJOR ? uint64_t(w0) : uint64_t(w0) * 64];
                    c1[c] = w0 + 1 < nwords ? cw[c][ROWMAJOR ? uint64_t(w0 + 1) : uint64_t(w0 + 1) * 64] : make_uint4(0u, 0u, 0u, 0u);
                }
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CPT; c++) {
                if (uint32_t(c) * 1024 >= nlive) continue;
                const uint32_t cc[8] = {c0[c].x, c0[c].y, c0[c].z, c0[c].w, c1[c].x, c1[c].y, c1[c].z, c1[c].w};
                float sm = sum[c];
                if (ng == EXACT8_GS) {
                    float e[EXACT8_GS];
#pragma unroll
                    for (int j = 0; j < (int)EXACT8_GS; j++) e[j] = tab[j * 256 + ((cc[j >> 2] >> (8 * (j & 3))) & 0xffu)];
#pragma unroll
                    for (int j = 0; j < (int)EXACT8_GS; j++) sm = sm + e[j];
                } else {
#pragma unroll
                    for (int j = 0; j < (int)EXACT8_GS; j++)
                        if ((uint32_t)j < ng) sm = sm + tab[j * 256 + ((cc[j >> 2] >> (8 * (j & 3))) & 0xffu)];
                }
                sum[c] = sm;
            }
        }
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            const uint32_t i = base + c * 1024 + tid;
            const bool live = i < total, keep = live && sum[c] <= t;
            if (live) cq[i] = keep ? pair_key(sum[c], row[c]) : PAIR_NONE;
            kept += keep ? 1u : 0u;
        }
    }
    if (kept) atomicAdd(&valid[q], kept);
}

// ---------------------------------------------------------------------------------------------------
// ResultSet::pq_resort (candidate_pair.rs:102-108): replay `add` over the candidates in ADC order.
// `add` admits a pair when the set is not full, or when its DISTANCE is strictly smaller than the worst
// distance (candidate_pair.rs:61-74) -- not the lexicographic test -- so ties at the cut keep the
// earlier-in-ADC-order pair.  One wave per query keeps the set as a sorted register list.
// ---------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(64) void k_pq_resort(const uint64_t *__restrict__ exact_keys, uint32_t ncand,
                                                  uint32_t ldc, uint32_t k, uint64_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t q = blockIdx.x;
    uint64_t v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = PAIR_NONE;
    uint64_t tau = PAIR_NONE;
    // the replay is serial, the loads are not: one coalesced load fetches the next 64 keys (lane = position), the
    // offers then take them lane by lane (ncand runs into the thousands for IVF probe lists)
    for (uint32_t j0 = 0; j0 < ncand; j0 += 64) {
      const uint64_t mine = j0 + lane < ncand ? exact_keys[uint64_t(q) * ldc + j0 + lane] : PAIR_NONE;
      if (__ballot(mine != PAIR_NONE) == 0) continue;
      // only lanes whose distance beats the current worst can be admitted (tau only tightens): visit those, in offer order
      // (a serial pass over all 64 lanes cost 1.07 ms per 1000 queries x ~4000 IVF candidates)
      uint64_t todo = __ballot(mine != PAIR_NONE && uint32_t(mine >> 32) < uint32_t(tau >> 32));
      while (todo) {
        const uint32_t jj = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        uint64_t e = __shfl(mine, jj);  // wave-uniform
        if (uint32_t(e >> 32) >= uint32_t(tau >> 32)) continue;  // full and not strictly closer (tau moved since the ballot)
        bool placed = false;
#pragma unroll
        for (int r = 0; r < R; r++) {
            uint64_t cur = v[r];
            uint64_t mask = placed ? ~0ull : __ballot(cur > e);
            if (mask != 0) {
                uint32_t pos = placed ? 0u : (uint32_t)__builtin_ctzll(mask);
                uint64_t carry = __shfl(cur, 63);
                uint64_t up = __shfl_up(cur, 1);
                v[r] = lane < pos ? cur : (lane == pos ? e : up);
                e = carry;
                placed = true;
            }
        }
        uint32_t p = k - 1;
        uint64_t t = PAIR_NONE;
#pragma unroll
        for (int r = 0; r < R; r++)
            if ((p >> 6) == (uint32_t)r) t = __shfl(v[r], p & 63);
        tau = t;
      }
    }
#pragma unroll
    for (int r = 0; r < R; r++) out[uint64_t(q) * (64 * R) + r * 64 + lane] = v[r];
}

// ---------------------------------------------------------------------------------------------------
// pq_encode (pq_table.rs:66-91) + find_nearest_base (k_means.rs:40-57): per group the centroid with the
// smallest (distance, index) under the CandidatePair order; one thread per row.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pair_less(float da, uint32_t ia, float db, uint32_t ib) {
    uint32_t oa = f32_orderable(da), ob = f32_orderable(db);
    return oa != ob ? oa < ob : ia < ib;
}

__global__ __launch_bounds__(256) void k_pq_encode(const float *__restrict__ X, uint64_t n, uint32_t dim,
                                                   const float *__restrict__ cent,
                                                   const float *__restrict__ cent_sq,
                                                   const uint64_t *__restrict__ gstart, uint32_t m, uint32_t kc,
                                                   uint32_t n_bits, int cosine, uint32_t enc_dim,
                                                   uint8_t *__restrict__ codes) {
    uint64_t row = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row >= n) return;
    const float *x = X + row * dim;
    uint8_t *dst = codes + row * enc_dim;
    uint32_t pending = 0;
    for (uint32_t g = 0; g < m; g++) {
        uint32_t s = (uint32_t)gstart[g], gd = (uint32_t)gstart[g + 1] - s;
        const float *v = x + s;
        const float *cg = cent + uint64_t(kc) * s;
        float vnorm = 0.0f;
        if (cosine) {  // cosine_distance recomputes vec_norm(a) per pair (distance/mod.rs:60-64); same value every time
            float a = 0.0f;
            for (uint32_t j = 0; j < gd; j++) {
                float p = v[j] * v[j];
                a = a + p;
            }
            vnorm = sqrtf(a);
        }
        uint32_t best = 0;
        float bd = 0.0f;
        for (uint32_t c = 0; c < kc; c++) {
            const float *cc = cg + uint64_t(c) * gd;
            float acc = 0.0f;
            float d;
            if (cosine) {
                for (uint32_t j = 0; j < gd; j++) {
                    float p = v[j] * cc[j];
                    acc = acc + p;
                }
                float den = fmaxf(vnorm * sqrtf(cent_sq[g * kc + c]), 1e-10f);
                float r = acc / den;
                d = 1.0f - r;
            } else {
                for (uint32_t j = 0; j < gd; j++) {
                    float df = v[j] - cc[j];
                    float sq = df * df;
                    acc = acc + sq;
                }
                d = acc;
            }
            if (c == 0 || pair_less(d, c, bd, best)) {
                bd = d;
                best = c;
            }
        }
        if (n_bits == 4) {
            if ((g & 1) == 0) {
                pending = best;
                if (g == m - 1) dst[g / 2] = (uint8_t)pending;  // odd m: last byte holds one code
            } else {
                dst[g / 2] = (uint8_t)(pending | (best << 4));
            }
        } else {
            dst[g] = (uint8_t)best;
        }
    }
}

// ===================================================================================================
// host side
// ===================================================================================================

// pq_groups (pq_table.rs:38-53)
static std::vector<uint64_t> pq_groups(uint64_t dim, uint64_t m) {
    std::vector<uint64_t> g{0};
    uint64_t cur = 0;
    while (cur < dim) {
        uint64_t rem = m - (g.size() - 1);
        uint64_t gs = (dim - cur + rem - 1) / rem;
        cur += gs;
        g.push_back(cur);
    }
    return g;
}

static float host_dot(const float *a, const float *b, size_t n) {
    float acc = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float p = a[i] * b[i];
        acc = acc + p;
    }
    return acc;
}
static float host_l2(const float *a, const float *b, size_t n) {
    float acc = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float df = a[i] - b[i];
        float sq = df * df;
        acc = acc + sq;
    }
    return acc;
}
static float host_dist(int dist, const float *a, const float *b, size_t n) {
    if (dist == 0) return host_l2(a, b, n);
    float na = std::sqrt(host_dot(a, a, n)), nb = std::sqrt(host_dot(b, b, n));
    float den = std::fmax(na * nb, 1e-10f);
    return 1.0f - host_dot(a, b, n) / den;
}

void pq_clear(Index &ix) {
    ix.pq.present = false;
    ix.pq.n_coded = 0;
    ix.pq.codes_t_valid = false;
    ix.pq.d_codes.release();
    ix.pq.d_codes_t.release();
}

static void pq_install(Index &ix, uint64_t n_bits, uint64_t m, const float *centroids) {
    VDB_REQUIRE(n_bits == 4 || n_bits == 8, "n_bits must be 4 or 8 in PQTable.");  // pq_table.rs:142-145
    VDB_REQUIRE(m > 0 && m <= ix.dim, "m must be in 1..=dim");
    PQState &pq = ix.pq;
    pq.present = false;
    pq.n_bits = n_bits;
    pq.m = m;
    pq.kc = 1ull << n_bits;
    pq.enc_dim = n_bits == 4 ? (m + 1) / 2 : m;
    pq.gstart = pq_groups(ix.dim, m);
    VDB_REQUIRE(pq.gstart.size() == m + 1, "pq_groups produced fewer groups than m");
    pq.h_centroids.assign(centroids, centroids + pq.kc * ix.dim);
    pq.h_cent_cache.assign(m * pq.kc, 0.0f);
    // dot(c,c) is needed by the cosine ADC (pq_table.rs:160-165) and by the cosine encoder
    for (uint64_t g = 0; g < m; g++) {
        uint64_t s = pq.gstart[g], gd = pq.gstart[g + 1] - s;
        for (uint64_t c = 0; c < pq.kc; c++) {
            const float *cc = pq.h_centroids.data() + pq.kc * s + c * gd;
            pq.h_cent_cache[g * pq.kc + c] = host_dot(cc, cc, gd);
        }
    }
    ix.use_device();
    pq.d_centroids.reserve(pq.h_centroids.size() * sizeof(float));
    pq.d_cent_cache.reserve(pq.h_cent_cache.size() * sizeof(float));
    pq.d_gstart.reserve(pq.gstart.size() * sizeof(uint64_t));
    VDB_HIP(hipMemcpy(pq.d_centroids.p, pq.h_centroids.data(), pq.h_centroids.size() * sizeof(float), hipMemcpyHostToDevice));
    VDB_HIP(hipMemcpy(pq.d_cent_cache.p, pq.h_cent_cache.data(), pq.h_cent_cache.size() * sizeof(float), hipMemcpyHostToDevice));
    VDB_HIP(hipMemcpy(pq.d_gstart.p, pq.gstart.data(), pq.gstart.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    pq.d_codes.reserve(std::max<uint64_t>(ix.n, 1) * pq.enc_dim);
}

// word-major mirror of the codes for the quantised scans (rows padded to whole 16-B code words)
static void pq_tile_codes(Index &ix) {
    PQState &pq = ix.pq;
    pq.codes_t_valid = false;
    if ((pq.n_bits != 4 && pq.n_bits != 8) || ix.n == 0) return;
    const uint32_t nwords = (uint32_t)((pq.enc_dim + 15) / 16);
    // tables whose quantised image (4-bit: 16-bit entries of 8 queries; 8-bit: one-byte entries of one query) would not fit LDS never take the quantised scan
    if (size_t(nwords) * (pq.n_bits == 4 ? 32 * 256 : 16 * 256) > 150 * 1024) return;
    const uint64_t total = (ix.n + 63) / 64 * 64 * nwords;
    pq.d_codes_t.reserve(total * sizeof(uint4));
    WsLease ws(ix);
    hipLaunchKernelGGL(k_pq_tile_codes, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ws->stream, pq.d_codes.as<uint8_t>(), ix.n,
                       (uint32_t)pq.enc_dim, nwords, pq.d_codes_t.as<uint4>());
    VDB_SYNC(ws->stream);
    pq.codes_t_valid = true;
}

static void pq_encode_all(Index &ix) {
    PQState &pq = ix.pq;
    if (ix.n == 0) return;
    WsLease ws(ix);
    hipLaunchKernelGGL(k_pq_encode, dim3((unsigned)((ix.n + 255) / 256)), dim3(256), 0, ws->stream,
                       ix.d_rows.as<float>(), ix.n, (uint32_t)ix.dim, pq.d_centroids.as<float>(),
                       pq.d_cent_cache.as<float>(), pq.d_gstart.as<uint64_t>(), (uint32_t)pq.m, (uint32_t)pq.kc,
                       (uint32_t)pq.n_bits, ix.dist == 1 ? 1 : 0, (uint32_t)pq.enc_dim, pq.d_codes.as<uint8_t>());
    VDB_SYNC(ws->stream);
}

void pq_attach(Index &ix, uint64_t n_bits, uint64_t m, const float *centroids, const uint8_t *codes) {
    VDB_REQUIRE(!ix.elem_u8, "PQ / HNSW / IVF are built over f32 tables (DynamicIndex, dynamic_index.rs:11-14): a VecSet<u8> index serves Flat search");
    pq_install(ix, n_bits, m, centroids);
    if (codes) {
        if (ix.n) VDB_HIP(hipMemcpy(ix.pq.d_codes.p, codes, ix.n * ix.pq.enc_dim, hipMemcpyHostToDevice));
    } else {
        pq_encode_all(ix);
    }
    pq_tile_codes(ix);
    ix.pq.n_coded = ix.n;
    ix.pq.present = true;
}

// ---- k-means (k_means.rs:61-162), host side, one RNG stream per group so groups can run in parallel ----
static uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float uniform01(uint64_t &s) { return (float(uint32_t(splitmix64(s) >> 40)) + 0.5f) * (1.0f / 16777216.0f); }

static size_t nearest_centroid(const float *v, const float *cents, size_t k, size_t gd, int dist) {
    size_t best = 0;
    float bd = 0;
    for (size_t c = 0; c < k; c++) {
        float d = host_dist(dist, v, cents + c * gd, gd);
        uint32_t od = f32_orderable(d), ob = f32_orderable(bd);
        if (c == 0 || od < ob) {  // equal distance keeps the smaller index
            bd = d;
            best = c;
        }
    }
    return best;
}

static void kmeans_group(const float *train, size_t nt, size_t dim, size_t c0, size_t c1, size_t k, size_t max_iter,
                         float tol, int dist, uint64_t seed, float *cent, const KMeansAssignFn &assign_fn = nullptr) {
    size_t gd = c1 - c0;
    std::vector<float> sel(nt * gd);
    for (size_t i = 0; i < nt; i++) std::memcpy(&sel[i * gd], train + i * dim + c0, gd * sizeof(float));
    uint64_t rng = seed;
    // k-means++ seeding (k_means.rs:61-87): weights are running-min distances
    size_t first = splitmix64(rng) % nt;
    std::memcpy(cent, &sel[first * gd], gd * sizeof(float));
    std::vector<float> w(nt, INFINITY);
    for (size_t idx = 1; idx < k; idx++) {
        const float *prev = cent + (idx - 1) * gd;
        bool bad = false;
        float total = 0;
        for (size_t i = 0; i < nt; i++) {
            float d = host_dist(dist, prev, &sel[i * gd], gd);
            if (std::isnan(w[i]))
                w[i] = d;
            else if (!std::isnan(d) && d < w[i])
                w[i] = d;
            if (!(w[i] >= 0.0f) || std::isinf(w[i])) bad = true;
            total += w[i];
        }
        size_t c;
        if (bad || !(total > 0.0f) || std::isinf(total)) {
            c = splitmix64(rng) % nt;  // WeightedIndex error -> uniform (k_means.rs:80-82)
        } else {
            float u = uniform01(rng) * total, cum = 0;
            c = nt - 1;
            for (size_t i = 0; i < nt; i++) {
                cum += w[i];
                if (cum > u) {
                    c = i;
                    break;
                }
            }
        }
        std::memcpy(cent + idx * gd, &sel[c * gd], gd * sizeof(float));
    }
    // Lloyd (k_means.rs:95-162)
    std::vector<float> sums(k * gd);
    std::vector<size_t> cnt(k);
    std::vector<uint32_t> asg(assign_fn ? nt : 0);
    for (size_t it = 0; it < max_iter; it++) {
        std::fill(sums.begin(), sums.end(), 0.0f);
        std::fill(cnt.begin(), cnt.end(), 0);
        // assignment step (k_means.rs:117-120): on the GPU when the caller provides it (same (distance, index) minimum,
        // same strict-order distances, so the same clusters as the host loop); the sums stay in row order
        if (assign_fn) assign_fn(cent, asg.data());
        for (size_t i = 0; i < nt; i++) {
            size_t c = assign_fn ? asg[i] : nearest_centroid(&sel[i * gd], cent, k, gd, dist);
            cnt[c]++;
            for (size_t j = 0; j < gd; j++) sums[c * gd + j] += sel[i * gd + j];
        }
        float max_diff = -INFINITY;
        for (size_t c = 0; c < k; c++) {
            if (cnt[c] == 0)
                std::memcpy(&sums[c * gd], cent + c * gd, gd * sizeof(float));
            else
                for (size_t j = 0; j < gd; j++) sums[c * gd + j] /= float(cnt[c]);
            float d = host_l2(cent + c * gd, &sums[c * gd], gd);
            if (!std::isnan(d) && d > max_diff) max_diff = d;
        }
        std::memcpy(cent, sums.data(), k * gd * sizeof(float));
        if (max_diff < tol) break;
    }
}

// PQTable::from_vec_set (pq_table.rs:141-191): sample, per-group k-means on the host, encode on the GPU
void host_kmeans(const float *train, size_t nt, size_t dim, size_t c0, size_t c1, size_t k, size_t max_iter, float tol,
                 int dist, uint64_t seed, float *cent, const KMeansAssignFn &assign_fn) {
    kmeans_group(train, nt, dim, c0, c1, k, max_iter, tol, dist, seed, cent, assign_fn);
}
uint64_t host_splitmix64(uint64_t &s) { return splitmix64(s); }

void pq_build(Index &ix, uint64_t n_bits, uint64_t m, uint64_t train_n, uint64_t max_iter, float tol,
              uint64_t seed) {
    VDB_REQUIRE(n_bits == 4 || n_bits == 8, "n_bits must be 4 or 8 in PQTable.");
    VDB_REQUIRE(!ix.elem_u8, "PQ / HNSW / IVF are built over f32 tables (DynamicIndex, dynamic_index.rs:11-14): a VecSet<u8> index serves Flat search");
    VDB_REQUIRE(m > 0 && m <= ix.dim, "m must be in 1..=dim");
    VDB_REQUIRE(ix.n > 0, "Cannot build PQ table for an empty table");  // metadata_vec_table.rs:120-122
    const float *rows = ix.host_rows();
    size_t n = ix.n, dim = ix.dim, kc = 1ull << n_bits;
    std::vector<float> sample;
    const float *train = rows;
    size_t nt = n;
    uint64_t rng = seed;
    if (train_n && train_n < n) {  // VecSet::random_sample (vec_set.rs:154-163)
        std::vector<size_t> perm(n);
        for (size_t i = 0; i < n; i++) perm[i] = i;
        sample.resize(train_n * dim);
        for (size_t i = 0; i < train_n; i++) {
            size_t j = i + splitmix64(rng) % (n - i);
            std::swap(perm[i], perm[j]);
            std::memcpy(&sample[i * dim], rows + perm[i] * dim, dim * sizeof(float));
        }
        train = sample.data();
        nt = train_n;
    }
    auto gs = pq_groups(dim, m);
    std::vector<float> cent(kc * dim);
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    unsigned nth = (unsigned)std::min<uint64_t>(hw, m);
    auto par_groups = [&](const std::function<void(uint64_t)> &fn) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nth; t++)
            th.emplace_back([&, t]() {
                for (uint64_t g = t; g < m; g += nth) fn(g);
            });
        for (auto &t : th) t.join();
    };
    // (1) k-means++ seeding per group on the host (k_means.rs:61-87: a sequential weighted draw per centroid)
    par_groups([&](uint64_t g) {
        uint64_t gseed = seed ^ (0xD1B54A32D192ED03ull * (g + 1));
        kmeans_group(train, nt, dim, gs[g], gs[g + 1], kc, /*max_iter=*/0, tol, ix.dist, gseed, cent.data() + kc * gs[g]);
    });
    // (2) Lloyd (k_means.rs:95-162) for ALL groups together: the assignment step (:117-120) -- nt x m x kc strict-order
    // sub-vector distances per iteration -- is find_nearest_base of every training row against the current centroids,
    // i.e. exactly the encoder: one k_pq_encode launch on the training rows per iteration.  The update (sums in row
    // order, empty clusters keep their centroid, max shift against tol) stays on the host, per group in parallel; a
    // group that has converged is frozen, as its own loop would have stopped.
    {
        ix.use_device();
        const uint64_t enc = n_bits == 4 ? (m + 1) / 2 : m;
        DevBuf d_train, d_cent, d_csq, d_gs, d_codes;
        d_train.reserve(nt * dim * sizeof(float));
        d_cent.reserve(kc * dim * sizeof(float));
        d_csq.reserve(m * kc * sizeof(float));
        d_gs.reserve((m + 1) * sizeof(uint64_t));
        d_codes.reserve(nt * enc);
        VDB_HIP(hipMemcpy(d_train.p, train, nt * dim * sizeof(float), hipMemcpyHostToDevice));
        VDB_HIP(hipMemcpy(d_gs.p, gs.data(), (m + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
        std::vector<uint8_t> codes(nt * enc);
        std::vector<float> csq(m * kc);
        std::vector<char> active(m, 1);
        WsLease ws(ix);
        for (uint64_t it = 0; it < max_iter; it++) {
            for (uint64_t g = 0; g < m; g++)
                for (uint64_t c = 0; c < kc; c++) {
                    const float *cc = cent.data() + kc * gs[g] + c * (gs[g + 1] - gs[g]);
                    csq[g * kc + c] = host_dot(cc, cc, gs[g + 1] - gs[g]);
                }
            VDB_HIP(hipMemcpyAsync(d_cent.p, cent.data(), kc * dim * sizeof(float), hipMemcpyHostToDevice, ws->stream));
            VDB_HIP(hipMemcpyAsync(d_csq.p, csq.data(), m * kc * sizeof(float), hipMemcpyHostToDevice, ws->stream));
            hipLaunchKernelGGL(k_pq_encode, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, ws->stream, d_train.as<float>(), (uint64_t)nt,
                               (uint32_t)dim, d_cent.as<float>(), d_csq.as<float>(), d_gs.as<uint64_t>(), (uint32_t)m, (uint32_t)kc,
                               (uint32_t)n_bits, ix.dist == 1 ? 1 : 0, (uint32_t)enc, d_codes.as<uint8_t>());
            VDB_HIP(hipMemcpyAsync(codes.data(), d_codes.p, nt * enc, hipMemcpyDeviceToHost, ws->stream));
            VDB_SYNC(ws->stream);
            par_groups([&](uint64_t g) {
                if (!active[g]) return;
                const size_t c0 = gs[g], gd = gs[g + 1] - gs[g];
                float *cg = cent.data() + kc * c0;
                std::vector<float> sums(kc * gd, 0.0f);
                std::vector<size_t> cnt(kc, 0);
                for (size_t i = 0; i < nt; i++) {
                    const size_t c = n_bits == 4 ? ((codes[i * enc + g / 2] >> (4 * (g & 1))) & 0xf) : codes[i * enc + g];
                    cnt[c]++;
                    const float *v = train + i * dim + c0;
                    for (size_t j = 0; j < gd; j++) sums[c * gd + j] += v[j];
                }
                float max_diff = -INFINITY;
                for (size_t c = 0; c < kc; c++) {
                    if (cnt[c] == 0)
                        std::memcpy(&sums[c * gd], cg + c * gd, gd * sizeof(float));
                    else
                        for (size_t j = 0; j < gd; j++) sums[c * gd + j] /= float(cnt[c]);
                    const float d = host_l2(cg + c * gd, &sums[c * gd], gd);
                    if (!std::isnan(d) && d > max_diff) max_diff = d;
                }
                std::memcpy(cg, sums.data(), kc * gd * sizeof(float));
                if (max_diff < tol) active[g] = 0;
            });
            if (std::none_of(active.begin(), active.end(), [](char a) { return a != 0; })) break;
        }
    }
    pq_install(ix, n_bits, m, cent.data());
    pq_encode_all(ix);
    pq_tile_codes(ix);
    ix.pq.n_coded = ix.n;
    ix.pq.present = true;
}

template <int NW, bool COS>
static void adc16_launch_nw(const Adc16Args &a, dim3 grid, size_t lds, hipStream_t s) {
    func_max_lds(reinterpret_cast<const void *>(&k_pq_adc16<NW, COS>), int(160 * 1024));
    hipLaunchKernelGGL((k_pq_adc16<NW, COS>), grid, dim3(1024), lds, s, a);
}
template <int NW>
static void adc16_sample_nw(const Adc16Args &a, dim3 grid, size_t lds, hipStream_t s) {
    func_max_lds(reinterpret_cast<const void *>(&k_pq_adc16<NW, false, true>), int(160 * 1024));
    hipLaunchKernelGGL((k_pq_adc16<NW, false, true>), grid, dim3(1024), lds, s, a);
}
static void adc16_sample_launch(const Adc16Args &a, uint32_t nwords, dim3 grid, size_t lds, hipStream_t s) {
    switch (nwords) {
        case 4: adc16_sample_nw<4>(a, grid, lds, s); break;
        case 6: adc16_sample_nw<6>(a, grid, lds, s); break;
        case 11: adc16_sample_nw<11>(a, grid, lds, s); break;
        case 8: adc16_sample_nw<8>(a, grid, lds, s); break;
        case 10: adc16_sample_nw<10>(a, grid, lds, s); break;
        case 16: adc16_sample_nw<16>(a, grid, lds, s); break;
        default: adc16_sample_nw<0>(a, grid, lds, s); break;
    }
}
// tau[q] = M + D (s* + m/2) from the r-th smallest quantised sample sum s* (in tau[q] on entry); a flagged table (not
// quantisable) gets +inf: its list overflows and the query takes the f32 scan, as without the sample
__global__ void k_pq_tau_from16(float *__restrict__ tau, const double *__restrict__ qM, const double *__restrict__ qD,
                                const uint32_t *__restrict__ qflag, uint32_t m, uint32_t nq) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const float s = tau[q];
    tau[q] = (qflag[q] != 0 || !(s < 1.0e9f)) ? INFINITY : float(qM[q] + qD[q] * (double(s) + 0.5 * double(m)));
}
template <bool COS>
static void adc16_launch_c(const Adc16Args &a, uint32_t nwords, dim3 grid, size_t lds, hipStream_t s) {
    switch (nwords) {  // 32 * nwords groups: 128 / 171 -> 192 (dim 512 / 3) / 256 / 320 (Gist1M, dim / 3) / 342 -> 352 (dim 1024 / 3) / 512
        case 4: adc16_launch_nw<4, COS>(a, grid, lds, s); break;
        case 6: adc16_launch_nw<6, COS>(a, grid, lds, s); break;
        case 11: adc16_launch_nw<11, COS>(a, grid, lds, s); break;
        case 8: adc16_launch_nw<8, COS>(a, grid, lds, s); break;
        case 10: adc16_launch_nw<10, COS>(a, grid, lds, s); break;
        case 16: adc16_launch_nw<16, COS>(a, grid, lds, s); break;
        default: adc16_launch_nw<0, COS>(a, grid, lds, s); break;
    }
}
static void adc16_launch(const Adc16Args &a, bool cosine, uint32_t nwords, dim3 grid, size_t lds, hipStream_t s) {
    if (cosine)
        adc16_launch_c<true>(a, nwords, grid, lds, s);
    else
        adc16_launch_c<false>(a, nwords, grid, lds, s);
}

// ---- FlatIndex::knn_pq (flat_index.rs:84-104) --------------------------------------------------------
template <int BQ, int NBITS, int MODE>
static void adc_launch_m(Index &ix, Workspace &ws, AdcArgs a) {
    PQState &pq = ix.pq;
    size_t lsz = pq.m * pq.kc * sizeof(float);
    size_t tables = lsz * BQ + (ix.dist == 1 ? lsz : 0);
    bool in_lds = tables <= 120 * 1024;
    size_t lds = (in_lds ? ((tables + 15) & ~size_t(15)) : 0) + (MODE == 1 ? ADC_WGBUF * 12 + (1 + 2 * BQ) * 4 + 16 : 0);
    // LUTs in LDS: one 1024-thread workgroup per CU (16 waves keep the LDS gather pipe busy); else 8 x 256
    uint32_t nt = in_lds ? 1024 : 256;
    uint64_t nblk = ((ix.n + nt - 1) / nt + a.blk_step - 1) / a.blk_step;
    uint32_t grid = (uint32_t)std::min<uint64_t>(nblk, uint64_t(ix.num_cu) * (in_lds ? 1 : 8));
    if (grid == 0 || a.nq_total == 0) return;
    dim3 g(grid, (a.nq_total + BQ - 1) / BQ);
    if (in_lds) {
        func_max_lds(reinterpret_cast<const void *>(&k_pq_adc<BQ, NBITS, true, MODE>), int(160 * 1024));
        hipLaunchKernelGGL((k_pq_adc<BQ, NBITS, true, MODE>), g, dim3(nt), lds, ws.stream, a);
    } else {
        hipLaunchKernelGGL((k_pq_adc<BQ, NBITS, false, MODE>), g, dim3(nt), lds, ws.stream, a);
    }
}

template <int MODE>
static void adc_launch(Index &ix, Workspace &ws, uint32_t BQ, const AdcArgs &a) {
    if (ix.pq.n_bits == 4) {
        if (BQ == 4) adc_launch_m<4, 4, MODE>(ix, ws, a);
        else if (BQ == 2) adc_launch_m<2, 4, MODE>(ix, ws, a);
        else adc_launch_m<1, 4, MODE>(ix, ws, a);
    } else {
        if (BQ == 4) adc_launch_m<4, 8, MODE>(ix, ws, a);
        else if (BQ == 2) adc_launch_m<2, 8, MODE>(ix, ws, a);
        else adc_launch_m<1, 8, MODE>(ix, ws, a);
    }
}

void pq_make_luts(Index &ix, Workspace &ws, const float *d_q, uint64_t nq) {
    PQState &pq = ix.pq;
    uint32_t lsz = (uint32_t)(pq.m * pq.kc);
    ws.lut.reserve(nq * lsz * sizeof(float));
    hipLaunchKernelGGL(k_pq_lut, dim3((lsz + 255) / 256, (unsigned)nq), dim3(256), 0, ws.stream, d_q,
                       (uint32_t)ix.dim, pq.d_centroids.as<float>(), pq.d_gstart.as<uint64_t>(), (uint32_t)pq.m,
                       (uint32_t)pq.kc, ix.dist == 1 ? 1 : 0, ws.lut.as<float>());
}

void pq_resort_launch(const uint64_t *exact_keys, uint32_t ncand, uint32_t ldc, uint32_t nq, uint32_t k,
                      uint64_t *out, hipStream_t s) {
    uint32_t cap = topk_capacity(k);
    switch (cap / 64) {
        case 1: hipLaunchKernelGGL((k_pq_resort<1>), dim3(nq), dim3(64), 0, s, exact_keys, ncand, ldc, k, out); break;
        case 2: hipLaunchKernelGGL((k_pq_resort<2>), dim3(nq), dim3(64), 0, s, exact_keys, ncand, ldc, k, out); break;
        case 4: hipLaunchKernelGGL((k_pq_resort<4>), dim3(nq), dim3(64), 0, s, exact_keys, ncand, ldc, k, out); break;
        case 8: hipLaunchKernelGGL((k_pq_resort<8>), dim3(nq), dim3(64), 0, s, exact_keys, ncand, ldc, k, out); break;
        case 16: hipLaunchKernelGGL((k_pq_resort<16>), dim3(nq), dim3(64), 0, s, exact_keys, ncand, ldc, k, out); break;
        default: throw Error(1, "pq_resort: k must be <= 1024");
    }
}

// row stride of the per-query shortlists: the register-resident select writes topk_capacity(efk) entries (efk <= 1024);
// beyond that the any-size path writes efk rounded up to 64
static uint32_t pq_list_ld(uint32_t efk) { return efk <= 1024 ? topk_capacity(efk) : ((efk + 63) & ~63u); }

// any ef: every ADC value of every row, a full (adc, idx) sort per query, the first efk pairs (flat_index.rs:96-100
// with ResultSet::new(ef.max(k)) of any size) -> ws.keys_a [nq][pq_list_ld(efk)]
static void pq_adc_shortlist_sorted(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint32_t efk) {
    hipStream_t s = ws.stream;
    PQState &pq = ix.pq;
    const uint64_t n = ix.n;
    ws.qsq.reserve(nq * sizeof(float));
    launch_row_sqnorm(d_q, nq, (uint32_t)ix.dim, ws.qsq.as<float>(), s);
    pq_make_luts(ix, ws, d_q, nq);
    const uint32_t lsz = (uint32_t)(pq.m * pq.kc);
    const size_t lbytes = size_t(lsz) * sizeof(float), cbytes = ix.dist == 1 ? lbytes : 0;
    const uint32_t BQ = lbytes * 4 + cbytes <= 120 * 1024 ? 4 : (lbytes * 2 + cbytes <= 120 * 1024 ? 2 : 1);
    const uint32_t nt = lbytes * BQ + cbytes <= 120 * 1024 ? 1024 : 256;
    const uint64_t ld = (n + nt + 63) & ~63ull;
    const uint32_t ldo = pq_list_ld(efk);
    const uint64_t GQ = std::max<uint64_t>(BQ, std::min<uint64_t>(64, (size_t(1) << 30) / (ld * 20)) / BQ * BQ);
    const size_t tb = sort_rows_temp_bytes(GQ, ld);
    ws.keys_a.reserve(nq * ldo * sizeof(uint64_t));
    ws.keys_b.reserve(nq * ldo * sizeof(uint64_t));
    ws.dense.reserve(GQ * ld * sizeof(float));
    ws.lists.reserve(2 * GQ * ld * sizeof(uint64_t) + tb);
    uint64_t *k_in = ws.lists.as<uint64_t>(), *k_out = k_in + GQ * ld;
    void *temp = k_out + GQ * ld;
    for (uint64_t g0 = 0; g0 < nq; g0 += GQ) {
        const uint64_t gn = std::min<uint64_t>(GQ, nq - g0);
        AdcArgs a{};
        a.codes = pq.d_codes.as<uint8_t>();
        a.n = n;
        a.enc_dim = (uint32_t)pq.enc_dim;
        a.m = (uint32_t)pq.m;
        a.cent_cache = pq.d_cent_cache.as<float>();
        a.cosine = ix.dist == 1 ? 1 : 0;
        a.blk_step = 1;
        a.fast = g_adc_fast;
        a.nq_total = (uint32_t)gn;
        a.lut = ws.lut.as<float>() + g0 * lsz;
        a.qsq = ws.qsq.as<float>() + g0;
        a.out = ws.dense.as<float>();
        a.ld = ld;
        ix.prof_begin(ws, "pq_adc", double((gn + BQ - 1) / BQ) * double(n) * pq.enc_dim);
        adc_launch<0>(ix, ws, BQ, a);
        ix.prof_end(ws);
        launch_pair_keys_rows(ws.dense.as<float>(), ld, n, (uint32_t)gn, k_in, ld, s);
        launch_sort_rows(k_in, k_out, gn, ld, temp, tb, s);
        launch_copy_prefix(k_out, ld, ws.keys_a.as<uint64_t>() + g0 * ldo, ldo, efk, (uint32_t)gn, s);
    }
}

// ADC scan of the whole code table (pq_table.rs:239-301 through flat_index.rs:96-100): per query the efk smallest
// (ADC distance, row) pairs, sorted by the CandidatePair order, land in ws.keys_a [nq][topk_capacity(efk)];
// ws.qsq holds the query norms afterwards.
static void pq_adc_shortlist(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint32_t efk) {
    hipStream_t s = ws.stream;
    PQState &pq = ix.pq;
    VDB_REQUIRE(pq.present && pq.n_coded == ix.n, "PQ table does not cover the rows of the index (rebuild it after add)");
    if (efk > 1024) {  // beyond the register-resident select
        pq_adc_shortlist_sorted(ix, ws, d_q, nq, efk);
        return;
    }
    const uint64_t n = ix.n;
    ws.qsq.reserve(nq * sizeof(float));
    launch_row_sqnorm(d_q, nq, (uint32_t)ix.dim, ws.qsq.as<float>(), s);
    pq_make_luts(ix, ws, d_q, nq);

    const uint32_t cape = topk_capacity(efk);
    const uint32_t lsz = (uint32_t)(pq.m * pq.kc);
    // queries per pass: as many lookup tables as fit beside each other in LDS (<= 4)
    size_t lbytes = size_t(lsz) * sizeof(float);
    const size_t cbytes = ix.dist == 1 ? lbytes : 0;
    uint32_t BQ = lbytes * 4 + cbytes <= 120 * 1024 ? 4 : (lbytes * 2 + cbytes <= 120 * 1024 ? 2 : 1);
    const uint32_t nt = lbytes * BQ + cbytes <= 120 * 1024 ? 1024 : 256;  // rows per row block (= workgroup size)
    ws.keys_a.reserve(nq * cape * sizeof(uint64_t));
    ws.keys_b.reserve(nq * cape * sizeof(uint64_t));

    AdcArgs base{};
    base.codes = pq.d_codes.as<uint8_t>();
    base.n = n;
    base.enc_dim = (uint32_t)pq.enc_dim;
    base.m = (uint32_t)pq.m;
    base.cent_cache = pq.d_cent_cache.as<float>();
    base.cosine = ix.dist == 1 ? 1 : 0;
    base.blk_step = 1;
    base.fast = g_adc_fast;

    // dense path for one group of queries: every ADC distance, then the wave select (also the overflow fallback)
    auto dense_group = [&](uint64_t g0, uint64_t gn) {
        const uint64_t ld = (n + nt + 63) & ~63ull;
        const uint32_t nl = topk_num_lists(n);
        ws.dense.reserve(gn * ld * sizeof(float));
        ws.lists.reserve(gn * nl * cape * sizeof(uint64_t));
        {
            AdcArgs a = base;
            a.nq_total = (uint32_t)gn;
            a.lut = ws.lut.as<float>() + g0 * lsz;
            a.qsq = ws.qsq.as<float>() + g0;
            a.out = ws.dense.as<float>();
            a.ld = ld;
            ix.prof_begin(ws, "pq_adc", double((gn + BQ - 1) / BQ) * double(n) * pq.enc_dim);
            adc_launch<0>(ix, ws, BQ, a);
            ix.prof_end(ws);
        }
        launch_topk_dense(ws.dense.as<float>(), ld, n, (uint32_t)gn, efk, ws.lists.as<uint64_t>(), s);
        launch_topk_merge(ws.lists.as<uint64_t>(), nl, cape, (uint32_t)gn, efk, ws.keys_a.as<uint64_t>() + g0 * cape, s);
    };

    // ---- 8-bit codes: sixteen queries per pass on one-byte table entries (k_pq_adc8x16; pq_adc8_sliced = 1 / 2: the earlier forms) ------
    {
        const uint32_t nw8 = (uint32_t)((pq.enc_dim + 15) / 16), m8 = 16 * nw8;
        const size_t lds8 = size_t(m8) * 256 + ADC8_WGBUF * 4 + 16;
        const bool q8 = g_adc16 != 1 && pq.codes_t_valid && pq.n_bits == 8 && ix.dist == 0 && pq.m <= 1024 && lds8 <= 150 * 1024 && n >= 65536;
        const uint64_t nblk8 = (n + 1023) / 1024;
        uint32_t step8 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(64, nblk8 / 16)), rank8 = efk;
        {
            const uint32_t r = efk / 8 < 8 ? 8 : efk / 8;
            const uint64_t target = std::max<uint64_t>(1024, 4ull * efk);
            const uint64_t st = std::min<uint64_t>(nblk8 / 16, target / r);
            const double thr = r < 16 ? r / 8.0 : (r < 32 ? r / 4.0 : r / 3.0);
            if (r < efk && st >= step8 && double(efk) <= thr * double(st) * 1.1) {
                rank8 = r;
                step8 = (uint32_t)st;
            }
        }
        const uint64_t n_sb8 = (nblk8 + step8 - 1) / step8, n_s8 = n_sb8 * 1024;
        if (q8 && n_s8 >= 2 * uint64_t(efk) && n_s8 >= 64ull * rank8) {
            const uint64_t ld_s = (n_s8 + 63) & ~63ull;
            const uint32_t nl_s = topk_num_lists(n_s8);
            // (lists of ~1000 expected candidates reach 4000 on a query in a thousand: 8192 entries keep those off the f32 scan)
            const uint32_t cap = (uint32_t)std::min<uint64_t>(65536, std::max<uint64_t>(8192, 4ull * rank8 * step8));
            const uint64_t GQ = std::max<uint64_t>(64, std::min<uint64_t>(2048, (size_t(256) << 20) / (ld_s * sizeof(float))));
            const uint64_t gq_max = std::min<uint64_t>(GQ, nq);
            ws.dense.reserve(gq_max * ld_s * sizeof(float));
            ws.lists.reserve(std::max<size_t>(gq_max * nl_s * cape, gq_max * size_t(cap)) * sizeof(uint64_t));
            ws.misc.reserve(nq * (sizeof(float) + 2 * sizeof(uint32_t)));
            float *d_tau = ws.misc.as<float>();
            uint32_t *d_hits = reinterpret_cast<uint32_t *>(d_tau + nq), *d_valid = d_hits + nq;
            VDB_HIP(hipMemsetAsync(d_hits, 0, 2 * nq * sizeof(uint32_t), s));
            ws.qfrag_g.reserve(nq * size_t(m8) * 256);
            ws.qaux.reserve(nq * (2 * sizeof(double) + sizeof(uint32_t)));
            double *d_qM = ws.qaux.as<double>(), *d_qD = d_qM + nq;
            uint32_t *d_qflag = reinterpret_cast<uint32_t *>(d_qD + nq);
            const size_t lds816 = 2 * size_t(ADC8X16_GS) * 256 * 16 + size_t(ADC8X16_WGBUF) * 8 + (36 + 16) * 4 + 16;
            const bool use816 = g_adc8_sliced == 0 && lds816 <= 158 * 1024;  // sixteen queries per pass, sample included (k_pq_adc8x16)
            if (!use816)
                hipLaunchKernelGGL(k_pq_quant8, dim3((unsigned)nq), dim3(256), 0, s, ws.lut.as<float>(), (uint32_t)pq.m, m8, (uint32_t)nq,
                                   ws.qfrag_g.as<uint8_t>(), d_qM, d_qD, d_qflag);
            func_max_lds(reinterpret_cast<const void *>(&k_pq_adc8<false>), int(160 * 1024));
            func_max_lds(reinterpret_cast<const void *>(&k_pq_adc8<true>), int(160 * 1024));
            for (uint64_t g0 = 0; g0 < nq; g0 += GQ) {
                const uint64_t gn = std::min<uint64_t>(GQ, nq - g0);
                Adc8Args a{};
                a.codes_t = pq.d_codes_t.as<uint4>();
                a.n = n;
                a.nwords = nw8;
                a.m = (uint32_t)pq.m;
                a.img = ws.qfrag_g.as<uint8_t>() + g0 * size_t(m8) * 256;
                a.qM = d_qM + g0;
                a.qD = d_qD + g0;
                a.qflag = d_qflag + g0;
                a.tau = d_tau + g0;
                a.nq = (uint32_t)gn;
                a.cand = ws.lists.as<uint64_t>();
                a.cnt = d_hits + g0;
                a.cap = cap;
                a.blk_step = step8;
                a.s8_out = ws.dense.as<float>();
                a.ld_s = ld_s;
                // the image of the sixteen-query scan, needed by its sample already
                const uint32_t nsl8 = (uint32_t)((pq.m + ADC8X16_GS - 1) / ADC8X16_GS), m_pad8 = nsl8 * ADC8X16_GS;
                const uint32_t ngrp816 = (uint32_t)((gn + 15) / 16);
                double *qM16 = nullptr, *qD16 = nullptr;
                uint32_t *qf16 = nullptr;
                Adc16x8Args b816{};
                if (use816) {
                    ws.pq_img16.reserve(size_t(ngrp816) * m_pad8 * 256 * sizeof(uint4));
                    ws.pq_aux16.reserve(gn * pq.m * sizeof(float) + gn * (2 * sizeof(double) + sizeof(uint32_t)) + 64);
                    qM16 = ws.pq_aux16.as<double>();
                    qD16 = qM16 + gn;
                    qf16 = reinterpret_cast<uint32_t *>(qD16 + gn);
                    float *mn16 = reinterpret_cast<float *>(qf16 + ((gn + 3) & ~uint64_t(3)));
                    const float *lutg = ws.lut.as<float>() + g0 * lsz;
                    hipLaunchKernelGGL(k_pq_quant16x8_stats, dim3((unsigned)gn), dim3(256), 0, s, lutg, (uint32_t)pq.m, (uint32_t)gn, mn16, qM16, qD16, qf16,
                                       std::min(65000.0, 512.0 * double(pq.m)));
                    hipLaunchKernelGGL(k_pq_quant8x16_img, dim3(m_pad8, ngrp816), dim3(256), 0, s, lutg, mn16, qD16, qf16, (uint32_t)pq.m, m_pad8,
                                       (uint32_t)gn, ws.pq_img16.as<uint4>());
                    b816.codes_t = pq.d_codes_t.as<uint4>();
                    b816.n = n;
                    b816.nwords = nw8;
                    b816.m = (uint32_t)pq.m;
                    b816.nslices = nsl8;
                    b816.img = ws.pq_img16.as<uint4>();
                    b816.qM = qM16;
                    b816.qD = qD16;
                    b816.qflag = qf16;
                    b816.tau = d_tau + g0;
                    b816.nq = (uint32_t)gn;
                    b816.rows_per_wg = uint64_t(ADC8X16_RPT) * 1024;  // one block of rows per workgroup: every slice of the image is loaded once
                    b816.cand = ws.lists.as<uint64_t>();
                    b816.cnt = d_hits + g0;
                    b816.cap = cap;
                    b816.blk_step = step8;
                    b816.n_sb = n_sb8;
                    b816.s8_out = ws.dense.as<float>();
                    b816.ld_s = ld_s;
                    func_max_lds(reinterpret_cast<const void *>(&k_pq_adc8x16<true>), int(160 * 1024));
                    hipLaunchKernelGGL(k_pq_adc8x16<true>, dim3((unsigned)((n_sb8 + ADC8X16_RPT - 1) / ADC8X16_RPT), ngrp816), dim3(1024), lds816, s, b816);
                } else {
                    // every workgroup loads its query's 80-KB table: few workgroups per query, enough of them to fill the chip
                    const uint32_t gx = (uint32_t)std::min<uint64_t>(n_sb8, std::max<uint64_t>(1, (2ull * ix.num_cu + gn - 1) / gn));
                    hipLaunchKernelGGL(k_pq_adc8<true>, dim3(gx, (unsigned)gn), dim3(1024), lds8, s, a);
                }
                if (n_s8 <= select_tau_max_n()) {
                    launch_select_tau(ws.dense.as<float>(), ld_s, (uint32_t)n_s8, (uint32_t)gn, (uint32_t)gn, rank8, d_tau + g0, s);
                } else {
                    launch_topk_dense(ws.dense.as<float>(), ld_s, n_s8, (uint32_t)gn, rank8, ws.lists.as<uint64_t>(), s);
                    launch_topk_merge(ws.lists.as<uint64_t>(), nl_s, cape, (uint32_t)gn, rank8, ws.keys_a.as<uint64_t>() + g0 * cape, s);
                    launch_extract_tau(ws.keys_a.as<uint64_t>() + g0 * cape, cape, (uint32_t)gn, rank8, d_tau + g0, s);
                }
                if (use816)  // (the sample's sums are in the scan's own quantisation)
                    hipLaunchKernelGGL(k_pq_tau_from16, dim3((unsigned)((gn + 255) / 256)), dim3(256), 0, s, d_tau + g0, qM16, qD16, qf16, (uint32_t)pq.m,
                                       (uint32_t)gn);
                else
                    hipLaunchKernelGGL(k_pq_tau_from16, dim3((unsigned)((gn + 255) / 256)), dim3(256), 0, s, d_tau + g0, d_qM + g0, d_qD + g0, d_qflag + g0,
                                       (uint32_t)pq.m, (uint32_t)gn);
                const uint32_t nsl = (uint32_t)((pq.m + ADC16X8_GS - 1) / ADC16X8_GS), m_pad16 = nsl * ADC16X8_GS;
                const size_t lds168 = size_t(ADC16X8_GS) * 256 * 16 + size_t(ADC16X8_WGBUF) * 8 + (20 + 8) * 4 + 16;
                if (use816) {
                    // sixteen queries per pass over the code mirror: one-byte entries in slices of 16 groups (k_pq_adc8x16)
                    func_max_lds(reinterpret_cast<const void *>(&k_pq_adc8x16<false>), int(160 * 1024));
                    ix.prof_begin(ws, "pq_adc", double(ngrp816) * double(n) * pq.enc_dim);
                    hipLaunchKernelGGL(k_pq_adc8x16<false>, dim3((unsigned)((n + b816.rows_per_wg - 1) / b816.rows_per_wg), ngrp816), dim3(1024), lds816, s, b816);
                    ix.prof_end(ws);
                } else if (g_adc8_sliced != 1 && lds168 <= 158 * 1024) {
                    // eight queries per pass over the code mirror: 16-bit tables in slices of 32 groups (k_pq_adc16x8)
                    const uint32_t ngrp = (uint32_t)((gn + 7) / 8);
                    ws.pq_img16.reserve(size_t(ngrp) * m_pad16 * 256 * sizeof(uint4));
                    ws.pq_aux16.reserve(gn * pq.m * sizeof(float) + gn * (2 * sizeof(double) + sizeof(uint32_t)) + 64);
                    double *qM16 = ws.pq_aux16.as<double>(), *qD16 = qM16 + gn;
                    uint32_t *qf16 = reinterpret_cast<uint32_t *>(qD16 + gn);
                    float *mn16 = reinterpret_cast<float *>(qf16 + ((gn + 3) & ~uint64_t(3)));
                    const float *lutg = ws.lut.as<float>() + g0 * lsz;
                    hipLaunchKernelGGL(k_pq_quant16x8_stats, dim3((unsigned)gn), dim3(256), 0, s, lutg, (uint32_t)pq.m, (uint32_t)gn, mn16, qM16, qD16, qf16, 65000.0);
                    hipLaunchKernelGGL(k_pq_quant16x8_img, dim3(m_pad16, ngrp), dim3(256), 0, s, lutg, mn16, qD16, qf16, (uint32_t)pq.m, m_pad16,
                                       (uint32_t)gn, ws.pq_img16.as<uint4>());
                    Adc16x8Args b{};
                    b.codes_t = pq.d_codes_t.as<uint4>();
                    b.n = n;
                    b.nwords = nw8;
                    b.m = (uint32_t)pq.m;
                    b.nslices = nsl;
                    b.img = ws.pq_img16.as<uint4>();
                    b.qM = qM16;
                    b.qD = qD16;
                    b.qflag = qf16;
                    b.tau = d_tau + g0;
                    b.nq = (uint32_t)gn;
                    b.rows_per_wg = uint64_t(ADC16X8_RPT) * 1024;  // one block of rows per workgroup: every slice of the image is loaded once
                    b.cand = ws.lists.as<uint64_t>();
                    b.cnt = d_hits + g0;
                    b.cap = cap;
                    func_max_lds(reinterpret_cast<const void *>(&k_pq_adc16x8), int(160 * 1024));
                    ix.prof_begin(ws, "pq_adc", double(ngrp) * double(n) * pq.enc_dim);
                    hipLaunchKernelGGL(k_pq_adc16x8, dim3((unsigned)((n + b.rows_per_wg - 1) / b.rows_per_wg), ngrp), dim3(1024), lds168, s, b);
                    ix.prof_end(ws);
                } else {
                    const uint32_t nwg = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(ix.num_cu, (4ull * ix.num_cu + gn - 1) / gn));
                    a.rows_per_wg = ((n + nwg - 1) / nwg + 63) / 64 * 64;
                    const uint32_t nwg_eff = (uint32_t)((n + a.rows_per_wg - 1) / a.rows_per_wg);
                    ix.prof_begin(ws, "pq_adc", double(gn) * double(n) * pq.enc_dim);
                    hipLaunchKernelGGL(k_pq_adc8<false>, dim3(nwg_eff, (unsigned)gn), dim3(1024), lds8, s, a);
                    ix.prof_end(ws);
                }
                pq.adc16_queries += gn;
                if (pq.enc_dim % 16 == 0)
                    hipLaunchKernelGGL(k_pq_adc_exact8<true>, dim3((unsigned)gn), dim3(1024), 0, s, pq.d_codes.as<uint8_t>(), (uint32_t)pq.enc_dim,
                                       pq.d_codes_t.as<uint4>(), nw8, (uint32_t)pq.m, ws.lut.as<float>() + g0 * lsz, d_tau + g0, ws.lists.as<uint64_t>(),
                                       d_hits + g0, cap, d_valid + g0);
                else
                    hipLaunchKernelGGL(k_pq_adc_exact8<false>, dim3((unsigned)gn), dim3(1024), 0, s, pq.d_codes.as<uint8_t>(), (uint32_t)pq.enc_dim,
                                       pq.d_codes_t.as<uint4>(), nw8, (uint32_t)pq.m, ws.lut.as<float>() + g0 * lsz, d_tau + g0, ws.lists.as<uint64_t>(),
                                       d_hits + g0, cap, d_valid + g0);
                launch_topk_merge_counted(ws.lists.as<uint64_t>(), cap, d_hits + g0, (uint32_t)gn, efk, ws.keys_a.as<uint64_t>() + g0 * cape, s);
            }
            uint32_t *hv = static_cast<uint32_t *>(ws.pinned(2 * nq * sizeof(uint32_t)));
            VDB_HIP(hipMemcpyAsync(hv, d_hits, 2 * nq * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            VDB_SYNC(s);
            const uint64_t need = std::min<uint64_t>(efk, n);
            uint64_t hsum = 0, hmax = 0, n_over = 0, n_short = 0;
            for (uint64_t q = 0; q < nq; q++) {
                if (hv[q] > cap)
                    n_over++;
                else {
                    hsum += hv[q];
                    hmax = std::max<uint64_t>(hmax, hv[q]);
                    if (hv[nq + q] < need) n_short++;
                }
            }
            pq.q8_overflow += n_over;
            pq.q8_short += n_short;
            pq.q8_hits_sum += hsum;
            if (hmax > pq.q8_hits_max) pq.q8_hits_max = hmax;
            for (uint64_t q = 0; q < nq; q++)
                if (hv[q] > cap || hv[nq + q] < need) dense_group(q, 1);  // overflowed / short / unquantisable: the f32 scan answers
            return;
        }
    }

    const uint64_t nblk = (n + nt - 1) / nt;
    // Threshold sample: every step-th row block, tau = the s_rank-th smallest sampled ADC value.  s_rank = efk guarantees
    // >= efk hits (the sample is a subset) at efk * step expected ones; when the shard is large enough a thinner rank
    // r = max(8, efk / 8) with step ~ max(1024, 4 efk) / r gives ~1000 hits instead of ~6400 (every hit costs the scan's
    // epilogue an LDS append and the counted merge an insertion) and the hit count is CHECKED: hits / step is
    // Gamma(r)-distributed, the rule below keeps P[hits < efk] under ~1e-5 (same rule as mfma_sample_plan).
    uint32_t step = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(64, nblk / 16));
    uint32_t s_rank = efk;
    {
        const uint32_t r = efk / 8 < 8 ? 8 : efk / 8;
        const uint64_t target = std::max<uint64_t>(1024, 4ull * efk);
        const uint64_t st = std::min<uint64_t>(nblk / 16, target / r);
        const double thr = r < 16 ? r / 8.0 : (r < 32 ? r / 4.0 : r / 3.0);
        if (r < efk && st >= step && double(efk) <= thr * double(st) * 1.1) {
            s_rank = r;
            step = (uint32_t)st;
        }
    }
    const uint64_t n_s = (nblk + step - 1) / step * nt;  // sampled rows (incl. padding past n)
    const bool fused = n_s >= 2 * uint64_t(efk) && n >= 65536 && n_s >= 64ull * s_rank;
    if (!fused) {
        const uint64_t GQ = std::max<uint64_t>(BQ, std::min<uint64_t>(64, (size_t(512) << 20) / ((n + nt + 64) * sizeof(float))) / BQ * BQ);
        for (uint64_t g0 = 0; g0 < nq; g0 += GQ) dense_group(g0, std::min<uint64_t>(GQ, nq - g0));
        return;
    }
    // fused path: tau[q] from a strided row-block sample (exact f32 ADC values), then one filtered scan of all rows.
    // q16: the scan runs on the 16-bit quantised tables, 8 queries per pass (k_pq_adc16), and the exact f32 sums are
    // computed for its candidates only (k_pq_adc_exact); otherwise the f32 scan itself filters (k_pq_adc MODE 1).
    const uint32_t nw16 = (uint32_t)((pq.enc_dim + 15) / 16), m16 = 32 * nw16;  // code words per (padded) row, groups of the image
    const size_t lds16 = size_t(m16) * 256 + 128 + ADC16_WGBUF * 8 + (1 + 2 * ADC16_Q) * 4;
    const bool q16 = g_adc16 != 1 && pq.codes_t_valid && pq.n_bits == 4 && lds16 <= 150 * 1024 && nt == 1024 && (ix.dist == 0 ? BQ == 4 : true);
    const bool cos16 = q16 && ix.dist == 1;
    const uint32_t NQ16 = cos16 ? 7 : 8;  // queries per pass of the quantised scan (Cosine: slot 7 of an entry is the |centroid|^2 table)
    const uint64_t ld_s = (n_s + 63) & ~63ull;
    const uint32_t nl_s = topk_num_lists(n_s);
    const uint32_t cap = (uint32_t)std::min<uint64_t>(65536, std::max<uint64_t>(4096, 4ull * s_rank * step));
    // queries per round: bounded by the sample matrix (GQ x ld_s floats) and the candidate lists (GQ x cap keys)
    // (q16: a multiple of 448 = 7 * 64, so that a round starts on a whole image group for either metric)
    const uint64_t GQ = q16 ? std::max<uint64_t>(448, std::min<uint64_t>(1792, (size_t(256) << 20) / (ld_s * sizeof(float))) / 448 * 448) : 64;
    const uint64_t gq_max = std::min<uint64_t>(GQ, nq);
    ws.dense.reserve(gq_max * ld_s * sizeof(float));
    ws.lists.reserve(std::max<size_t>(gq_max * nl_s * cape, gq_max * size_t(cap)) * sizeof(uint64_t));
    ws.misc.reserve(nq * (sizeof(float) + 2 * sizeof(uint32_t)));
    float *d_tau = ws.misc.as<float>();
    uint32_t *d_hits = reinterpret_cast<uint32_t *>(d_tau + nq);
    uint32_t *d_valid = d_hits + nq;
    VDB_HIP(hipMemsetAsync(d_hits, 0, 2 * nq * sizeof(uint32_t), s));
    double *d_qM = nullptr, *d_qD = nullptr, *d_qMC = nullptr, *d_qDC = nullptr;
    uint32_t *d_qflag = nullptr;
    if (q16) {
        const uint64_t ngrp = (nq + NQ16 - 1) / NQ16;
        ws.qfrag_g.reserve(ngrp * size_t(m16) * 256);
        ws.qaux.reserve(nq * (4 * sizeof(double) + sizeof(uint32_t)));
        d_qM = ws.qaux.as<double>();
        d_qD = d_qM + nq;
        d_qMC = d_qD + nq;
        d_qDC = d_qMC + nq;
        d_qflag = reinterpret_cast<uint32_t *>(d_qDC + nq);
        // slots of the last image group beyond nq keep whatever the buffer held: their thresholds never hit
        hipLaunchKernelGGL(k_pq_quant16, dim3((unsigned)nq), dim3(256), 0, s, ws.lut.as<float>(), pq.d_cent_cache.as<float>(),
                           (uint32_t)pq.m, m16, (uint32_t)nq, cos16 ? 1 : 0, ws.qfrag_g.as<uint16_t>(), d_qM, d_qD, d_qMC, d_qDC, d_qflag);
    }
    for (uint64_t g0 = 0; g0 < nq; g0 += GQ) {
        const uint64_t gn = std::min<uint64_t>(GQ, nq - g0);
        const bool sample16 = q16 && !cos16 && g_adc16_sample != 1;
        if (sample16) {  // the sample at the scan's rate, on the scan's tables: quantised sums, tau from their s_rank-th smallest
            Adc16Args a{};
            a.codes_t = pq.d_codes_t.as<uint4>();
            a.n = n;
            a.enc_dim = (uint32_t)pq.enc_dim;
            a.m = m16;
            a.img = reinterpret_cast<const uint4 *>(ws.qfrag_g.as<uint8_t>() + (g0 / NQ16) * size_t(m16) * 256);
            a.nq = (uint32_t)gn;
            a.blk_step = step;
            a.s16_out = ws.dense.as<float>();
            a.ld_s = ld_s;
            const uint32_t ngrp = (uint32_t)((gn + NQ16 - 1) / NQ16);
            const uint64_t n_sb = (nblk + step - 1) / step;
            // one table image per workgroup: as few workgroups per query group as still fill the chip
            const uint32_t gx = (uint32_t)std::min<uint64_t>(n_sb, std::max<uint64_t>(1, (uint64_t(ix.num_cu) + ngrp - 1) / ngrp));
            adc16_sample_launch(a, nw16, dim3(gx, ngrp), lds16, s);
        } else {
            AdcArgs a = base;
            a.nq_total = (uint32_t)gn;
            a.lut = ws.lut.as<float>() + g0 * lsz;
            a.qsq = ws.qsq.as<float>() + g0;
            a.blk_step = step;
            a.out = ws.dense.as<float>();
            a.ld = ld_s;
            adc_launch<0>(ix, ws, BQ, a);
        }
        if (n_s <= select_tau_max_n()) {  // tau = efk-th smallest sampled ADC value: a selection, not a sort
            launch_select_tau(ws.dense.as<float>(), ld_s, (uint32_t)n_s, (uint32_t)gn, (uint32_t)gn, s_rank, d_tau + g0, s);
        } else {
            launch_topk_dense(ws.dense.as<float>(), ld_s, n_s, (uint32_t)gn, s_rank, ws.lists.as<uint64_t>(), s);
            launch_topk_merge(ws.lists.as<uint64_t>(), nl_s, cape, (uint32_t)gn, s_rank, ws.keys_a.as<uint64_t>() + g0 * cape, s);
            launch_extract_tau(ws.keys_a.as<uint64_t>() + g0 * cape, cape, (uint32_t)gn, s_rank, d_tau + g0, s);
        }
        if (sample16)
            hipLaunchKernelGGL(k_pq_tau_from16, dim3((unsigned)((gn + 255) / 256)), dim3(256), 0, s, d_tau + g0, d_qM + g0, d_qD + g0,
                               d_qflag + g0, (uint32_t)pq.m, (uint32_t)gn);
        uint64_t *d_cand = ws.lists.as<uint64_t>();  // the sample lists are dead now
        if (q16) {
            Adc16Args a{};
            a.codes_t = pq.d_codes_t.as<uint4>();
            a.n = n;
            a.enc_dim = (uint32_t)pq.enc_dim;
            a.m = m16;
            a.img = reinterpret_cast<const uint4 *>(ws.qfrag_g.as<uint8_t>() + (g0 / NQ16) * size_t(m16) * 256);  // GQ % 56 == 0
            a.qM = d_qM + g0;
            a.qD = d_qD + g0;
            a.qMC = d_qMC + g0;
            a.qDC = d_qDC + g0;
            a.qsq = ws.qsq.as<float>() + g0;
            a.qflag = d_qflag + g0;
            a.tau = d_tau + g0;
            a.nq = (uint32_t)gn;
            a.rows_per_wg = ((n + ix.num_cu - 1) / ix.num_cu + 63) / 64 * 64;
            a.cand = d_cand;
            a.cnt = d_hits + g0;
            a.cap = cap;
            const uint32_t nwg = (uint32_t)((n + a.rows_per_wg - 1) / a.rows_per_wg);
            const uint32_t ngrp = (uint32_t)((gn + NQ16 - 1) / NQ16);
            ix.prof_begin(ws, "pq_adc", double(ngrp) * double(n) * pq.enc_dim);
            adc16_launch(a, cos16, nw16, dim3(nwg, ngrp), lds16, s);
            pq.adc16_queries += gn;
            ix.prof_end(ws);
            if (cos16)
                hipLaunchKernelGGL(k_pq_adc_exact<true>, dim3((unsigned)gn), dim3(256), 2 * lsz * sizeof(float), s, pq.d_codes.as<uint8_t>(),
                                   (uint32_t)pq.enc_dim, (uint32_t)pq.m, ws.lut.as<float>() + g0 * lsz, pq.d_cent_cache.as<float>(),
                                   ws.qsq.as<float>() + g0, d_tau + g0, d_cand, d_hits + g0, cap, d_valid + g0);
            else
                hipLaunchKernelGGL(k_pq_adc_exact<false>, dim3((unsigned)gn), dim3(256), lsz * sizeof(float), s, pq.d_codes.as<uint8_t>(),
                                   (uint32_t)pq.enc_dim, (uint32_t)pq.m, ws.lut.as<float>() + g0 * lsz, (const float *)nullptr,
                                   (const float *)nullptr, d_tau + g0, d_cand, d_hits + g0, cap, d_valid + g0);
        } else {
            AdcArgs a = base;
            a.nq_total = (uint32_t)gn;
            a.lut = ws.lut.as<float>() + g0 * lsz;
            a.qsq = ws.qsq.as<float>() + g0;
            a.tau = d_tau + g0;
            a.cand = d_cand;
            a.cnt = d_hits + g0;
            a.cap = cap;
            ix.prof_begin(ws, "pq_adc", double((gn + BQ - 1) / BQ) * double(n) * pq.enc_dim);
            adc_launch<1>(ix, ws, BQ, a);
            ix.prof_end(ws);
        }
        launch_topk_merge_counted(d_cand, cap, d_hits + g0, (uint32_t)gn, efk, ws.keys_a.as<uint64_t>() + g0 * cape, s);
    }
    // queries whose candidate list overflowed (or whose workgroup buffer filled, or whose table cannot be quantised) are
    // redone densely; so are those with fewer than efk rows at or below tau (thinned sample)
    uint32_t *hv = static_cast<uint32_t *>(ws.pinned(2 * nq * sizeof(uint32_t)));
    VDB_HIP(hipMemcpyAsync(hv, d_hits, 2 * nq * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    VDB_SYNC(s);
    const uint64_t need = std::min<uint64_t>(efk, n);
    std::vector<uint64_t> redo;
    for (uint64_t q = 0; q < nq; q++)
        if (hv[q] > cap || (q16 ? hv[nq + q] : hv[q]) < need) redo.push_back(q);
    for (uint64_t q : redo) dense_group(q, 1);
}

// PQTable::create_lookup (pq_table.rs:195-224) and the ADC adapter (pq_table.rs:239-301) over every code row, exported
// as they are computed by the search kernels (k_pq_lut, k_pq_adc dense mode): lut [nq][m*kc], qcache [nq] (0 for L2Sqr,
// |q| for Cosine: PQLookupTable::dist_cache), adc [nq][n].  Host outputs; used by the direct parity tests of a11 / a12.
void pq_export_lookup(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, float *h_lut, float *h_qcache) {
    hipStream_t s = ws.stream;
    PQState &pq = ix.pq;
    VDB_REQUIRE(pq.present, "no PQ table");
    if (nq == 0) return;
    const uint64_t lsz = pq.m * pq.kc;
    ws.qsq.reserve(nq * sizeof(float));
    launch_row_sqnorm(d_q, nq, (uint32_t)ix.dim, ws.qsq.as<float>(), s);
    pq_make_luts(ix, ws, d_q, nq);
    if (h_lut) VDB_HIP(hipMemcpyAsync(h_lut, ws.lut.p, nq * lsz * sizeof(float), hipMemcpyDeviceToHost, s));
    std::vector<float> qs(nq);
    VDB_HIP(hipMemcpyAsync(qs.data(), ws.qsq.p, nq * sizeof(float), hipMemcpyDeviceToHost, s));
    VDB_SYNC(s);
    if (h_qcache)
        for (uint64_t q = 0; q < nq; q++) h_qcache[q] = ix.dist == 1 ? std::sqrt(qs[q]) : 0.0f;
}

void pq_export_adc_all(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, float *h_out) {
    hipStream_t s = ws.stream;
    PQState &pq = ix.pq;
    VDB_REQUIRE(pq.present && pq.n_coded == ix.n, "PQ table does not cover the rows of the index (rebuild it after add)");
    const uint64_t n = ix.n;
    if (nq == 0 || n == 0) return;
    ws.qsq.reserve(nq * sizeof(float));
    launch_row_sqnorm(d_q, nq, (uint32_t)ix.dim, ws.qsq.as<float>(), s);
    pq_make_luts(ix, ws, d_q, nq);
    const uint32_t lsz = (uint32_t)(pq.m * pq.kc);
    const size_t lbytes = size_t(lsz) * sizeof(float), cbytes = ix.dist == 1 ? lbytes : 0;
    const uint32_t BQ = lbytes * 4 + cbytes <= 120 * 1024 ? 4 : (lbytes * 2 + cbytes <= 120 * 1024 ? 2 : 1);
    const uint32_t nt = lbytes * BQ + cbytes <= 120 * 1024 ? 1024 : 256;
    const uint64_t ld = (n + nt + 63) & ~63ull;
    const uint64_t GQ = std::max<uint64_t>(BQ, std::min<uint64_t>(64, (size_t(512) << 20) / (ld * sizeof(float))) / BQ * BQ);
    ws.dense.reserve(GQ * ld * sizeof(float));
    for (uint64_t g0 = 0; g0 < nq; g0 += GQ) {
        const uint64_t gn = std::min<uint64_t>(GQ, nq - g0);
        AdcArgs a{};
        a.codes = pq.d_codes.as<uint8_t>();
        a.n = n;
        a.enc_dim = (uint32_t)pq.enc_dim;
        a.m = (uint32_t)pq.m;
        a.cent_cache = pq.d_cent_cache.as<float>();
        a.cosine = ix.dist == 1 ? 1 : 0;
        a.blk_step = 1;
        a.fast = g_adc_fast;
        a.nq_total = (uint32_t)gn;
        a.lut = ws.lut.as<float>() + g0 * lsz;
        a.qsq = ws.qsq.as<float>() + g0;
        a.out = ws.dense.as<float>();
        a.ld = ld;
        adc_launch<0>(ix, ws, BQ, a);
        VDB_HIP(hipMemcpy2DAsync(h_out + g0 * n, n * sizeof(float), ws.dense.p, ld * sizeof(float), n * sizeof(float), gn,
                                 hipMemcpyDeviceToHost, s));
        VDB_SYNC(s);
    }
}

// exact distances of the ADC shortlist, in ADC order (the operand order of pq_resort, candidate_pair.rs:102-108):
// ws.keys_a -> ws.keys_b
static void pq_exact_of_shortlist(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint32_t efk) {
    hipStream_t s = ws.stream;
    const uint32_t cape = pq_list_ld(efk);
    VDB_HIP(hipMemsetAsync(ws.keys_b.p, 0xff, nq * cape * sizeof(uint64_t), s));
    launch_rerank(ix.d_rows.as<float>(), (uint32_t)ix.dim, d_q, (uint32_t)nq, ix.dist == 0 ? MET_L2_DIRECT : MET_COSINE,
                  ix.d_sq.as<float>(), ws.qsq.as<float>(), ws.keys_a.as<uint64_t>(), ws.keys_b.as<uint64_t>(), efk,
                  cape, s);
}

// ResultSet::pq_resort (candidate_pair.rs:102-108) of the exact keys [nq][ldc] (ADC order, first ncand per row) with set
// capacity ksel, then the outputs.  ksel <= 1024: the wave-resident replay; beyond: the heap replay + a row sort.
void pq_resort_finalize(Index &ix, Workspace &ws, const uint64_t *exact_keys, uint32_t ncand, uint32_t ldc, uint64_t nq,
                               uint32_t ksel, uint64_t k, uint64_t id_offset, uint64_t *d_idx, float *d_dist, uint64_t *d_cnt) {
    hipStream_t s = ws.stream;
    if (ksel <= 1024) {
        const uint32_t capk = topk_capacity(ksel);
        ws.keys_c.reserve(nq * capk * sizeof(uint64_t));
        pq_resort_launch(exact_keys, ncand, ldc, (uint32_t)nq, ksel, ws.keys_c.as<uint64_t>(), s);
        launch_finalize(ws.keys_c.as<uint64_t>(), capk, (uint32_t)nq, ksel, (uint32_t)k, id_offset, d_idx, d_dist, d_cnt, s);
        return;
    }
    const uint32_t ldk = (ksel + 63) & ~63u;
    const size_t tb = sort_rows_temp_bytes(nq, ldk);
    ws.keys_c.reserve(nq * ldk * sizeof(uint64_t));
    ws.lists.reserve(nq * ldk * sizeof(uint64_t) + tb);
    uint64_t *heap = ws.lists.as<uint64_t>();
    launch_resort_big(exact_keys, ncand, ldc, (uint32_t)nq, ksel, heap, ldk, s);
    launch_sort_rows(heap, ws.keys_c.as<uint64_t>(), nq, ldk, heap + nq * ldk, tb, s);
    launch_finalize(ws.keys_c.as<uint64_t>(), ldk, (uint32_t)nq, ksel, (uint32_t)k, id_offset, d_idx, d_dist, d_cnt, s);
}

void flat_knn_pq_device(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t ef,
                        uint64_t *d_idx, float *d_dist, uint64_t *d_cnt) {
    hipStream_t s = ws.stream;
    if (nq == 0) return;
    if (k == 0 || ix.n == 0) {
        VDB_HIP(hipMemsetAsync(d_cnt, 0, nq * sizeof(uint64_t), s));
        return;
    }
    const uint64_t n = ix.n;
    const uint64_t efk64 = std::min<uint64_t>(std::max(ef, k), n);  // ResultSet::new(ef.max(k)) flat_index.rs:96
    const uint64_t ksel64 = std::min<uint64_t>(k, n);
    VDB_REQUIRE(efk64 < (1ull << 31), "knn_pq: ef too large");
    const uint32_t efk = (uint32_t)efk64, ksel = (uint32_t)ksel64;
    if (efk > 1024) {  // any-size path: bound the per-call buffers (efk keys per query, three times)
        const uint64_t qs = std::max<uint64_t>(1, (size_t(1) << 30) / (uint64_t(pq_list_ld(efk)) * 32));
        if (nq > qs) {
            for (uint64_t q0 = 0; q0 < nq; q0 += qs)
                flat_knn_pq_device(ix, ws, d_q + q0 * ix.dim, std::min(qs, nq - q0), k, ef, d_idx + q0 * k, d_dist + q0 * k, d_cnt + q0);
            return;
        }
    }
    if (k > ksel) {
        VDB_HIP(hipMemsetAsync(d_idx, 0, nq * k * sizeof(uint64_t), s));
        VDB_HIP(hipMemsetAsync(d_dist, 0, nq * k * sizeof(float), s));
    }
    pq_adc_shortlist(ix, ws, d_q, nq, efk);
    pq_exact_of_shortlist(ix, ws, d_q, nq, efk);
    pq_resort_finalize(ix, ws, ws.keys_b.as<uint64_t>(), efk, pq_list_ld(efk), nq, ksel, k, ix.id_offset, d_idx, d_dist, d_cnt);
}

// ---------------------------------------------------------------------------------------------------
// Row-sharded PQ-Flat.  pq_resort replays the candidates in GLOBAL (ADC, id) order, so a shard cannot
// finish the re-sort alone.  Each shard exports its own ADC top-max(ef,k) as two key rows per query, both carrying
// the GLOBAL row id in the low word: the ADC key (sorted ascending) and, at the same position, the exact-distance
// key.  After one all-gather every rank merges the S sorted rows by ADC key, keeps the first max(ef,k), and replays
// ResultSet::add over the exact keys in that order -- the same operand sequence the unsharded scan produces,
// because the global ADC top-ef is contained in the union of the per-shard ADC top-ef lists.
// ---------------------------------------------------------------------------------------------------
__global__ void k_pq_export_keys(const uint64_t *__restrict__ adc, const uint64_t *__restrict__ exact, uint32_t ld,
                                 uint32_t efk_local, uint32_t efk_out, uint64_t id_offset,
                                 uint64_t *__restrict__ out_adc, uint64_t *__restrict__ out_exact) {
    const uint32_t q = blockIdx.x;
    for (uint32_t j = threadIdx.x; j < efk_out; j += blockDim.x) {
        uint64_t a = PAIR_NONE, e = PAIR_NONE;
        if (j < efk_local) {
            a = adc[uint64_t(q) * ld + j];
            e = exact[uint64_t(q) * ld + j];
            if (a != PAIR_NONE) {
                a += id_offset;  // local row < 2^32 - id_offset (checked by the launcher): no carry into the distance
                e += id_offset;
            } else {
                e = PAIR_NONE;
            }
        }
        out_adc[uint64_t(q) * efk_out + j] = a;
        out_exact[uint64_t(q) * efk_out + j] = e;
    }
}

void flat_knn_pq_shard_device(Index &ix, Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t ef,
                              uint64_t *d_adc_keys, uint64_t *d_exact_keys) {
    hipStream_t s = ws.stream;
    if (nq == 0) return;
    const uint64_t efg = std::max(ef, k);
    VDB_REQUIRE(efg >= 1 && efg < (1ull << 31), "knn_pq shard: max(ef, k) must be in 1..2^31");
    VDB_REQUIRE(ix.id_offset + ix.n <= (1ull << 32), "knn_pq shard: global row ids must fit 32 bits");
    if (ix.n == 0) {
        VDB_HIP(hipMemsetAsync(d_adc_keys, 0xff, nq * efg * sizeof(uint64_t), s));
        VDB_HIP(hipMemsetAsync(d_exact_keys, 0xff, nq * efg * sizeof(uint64_t), s));
        return;
    }
    const uint32_t efk = (uint32_t)std::min<uint64_t>(efg, ix.n);
    pq_adc_shortlist(ix, ws, d_q, nq, efk);
    pq_exact_of_shortlist(ix, ws, d_q, nq, efk);
    hipLaunchKernelGGL(k_pq_export_keys, dim3((unsigned)nq), dim3(256), 0, s, ws.keys_a.as<uint64_t>(),
                       ws.keys_b.as<uint64_t>(), pq_list_ld(efk), efk, (uint32_t)efg, ix.id_offset, d_adc_keys,
                       d_exact_keys);
}

// merge of S per-shard rows: an entry's position in the merged order is the number of entries with a smaller ADC key
// (keys are unique: global ids), found by one binary search per shard row.
__global__ __launch_bounds__(256) void k_pq_shard_merge(const uint64_t *__restrict__ adc,
                                                        const uint64_t *__restrict__ exact, uint32_t n_shards,
                                                        uint32_t nq, uint32_t efg, uint64_t *__restrict__ merged) {
    const uint32_t q = blockIdx.x;
    for (uint32_t j = threadIdx.x; j < efg; j += blockDim.x) merged[uint64_t(q) * efg + j] = PAIR_NONE;
    __syncthreads();
    const uint32_t total = n_shards * efg;
    for (uint32_t e = threadIdx.x; e < total; e += blockDim.x) {
        const uint32_t sh = e / efg, j = e - sh * efg;
        const uint64_t key = adc[(uint64_t(sh) * nq + q) * efg + j];
        if (key == PAIR_NONE) continue;
        uint32_t rank = j;  // entries of the own (sorted) row before this one
        for (uint32_t t = 0; t < n_shards && rank < efg; t++) {
            if (t == sh) continue;
            const uint64_t *row = adc + (uint64_t(t) * nq + q) * efg;
            uint32_t lo = 0, hi = efg;
            while (lo < hi) {  // first position with row[pos] >= key (PAIR_NONE pads the tail)
                uint32_t mid = (lo + hi) >> 1;
                if (row[mid] < key) lo = mid + 1;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < efg) merged[uint64_t(q) * efg + rank] = exact[(uint64_t(sh) * nq + q) * efg + j];
    }
}

void pq_merge_resort_device(Index &ix, Workspace &ws, const uint64_t *d_adc, const uint64_t *d_exact,
                            uint64_t n_shards, uint64_t nq, uint64_t efg, uint64_t k, uint64_t *d_idx, float *d_dist,
                            uint64_t *d_cnt) {
    hipStream_t s = ws.stream;
    if (nq == 0) return;
    if (k == 0) {
        VDB_HIP(hipMemsetAsync(d_cnt, 0, nq * sizeof(uint64_t), s));
        return;
    }
    VDB_REQUIRE(efg >= k && efg < (1ull << 31), "pq merge: need k <= max(ef, k) < 2^31");
    ws.keys_b.reserve(nq * efg * sizeof(uint64_t));
    hipLaunchKernelGGL(k_pq_shard_merge, dim3((unsigned)nq), dim3(256), 0, s, d_adc, d_exact, (uint32_t)n_shards,
                       (uint32_t)nq, (uint32_t)efg, ws.keys_b.as<uint64_t>());
    VDB_HIP(hipMemsetAsync(d_idx, 0, nq * k * sizeof(uint64_t), s));
    VDB_HIP(hipMemsetAsync(d_dist, 0, nq * k * sizeof(float), s));
    pq_resort_finalize(ix, ws, ws.keys_b.as<uint64_t>(), (uint32_t)efg, (uint32_t)efg, nq, (uint32_t)k, k, 0, d_idx, d_dist, d_cnt);
}

}  // namespace vdb
