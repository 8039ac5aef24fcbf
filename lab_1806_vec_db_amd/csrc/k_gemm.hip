// k_gemm.hip -- Flat brute force for LARGE query batches: 128 queries per HBM pass (second form).
//
// k_flat_mfma (k_mfma.hip) keeps the whole Q image of 32 queries in LDS and streams the corpus past it; every
// workgroup therefore ingests 4 B of X per (row, column) for only 32 queries.  Measured on MI355X that per-CU
// ingest, not HBM, becomes the limit once several batches share a pass through L2 (two batches: 0.69 ms per pass,
// four: 1.15 ms; a CU takes in at most ~50-70 GB/s from L2).  This kernel turns the loop nest around so that a
// byte of X entering a CU is used for 128 queries:
//
//   * a wave owns a UNIT of TW 16-row tiles and keeps the TW x 8 accumulator tiles (rows x 128 queries) in
//     registers for the whole contraction;
//   * the query group's B-operand image is cut into K-chunks of KC 32-column blocks (KC x 16 KB); the workgroup
//     double-buffers the chunks in LDS (global -> registers -> LDS while the previous chunk is being multiplied,
//     one barrier per chunk).  The chunks are re-read from L2 once per unit step (480 KB per 8 x TW x 16 rows),
//     i.e. +33 % on-chip traffic at TW = 3 against 4x fewer X bytes per query;
//   * X still never touches LDS: each fragment is read once, by one wave, straight from the fragment-ordered
//     mirror into a register ring that runs KC-1 k-blocks ahead across chunk, unit and group boundaries
//     (unconditional loads -> counted vmcnt, as in k_flat_mfma);
//   * same split-bf16 arithmetic (xh*qh + xh*ql + xl*qh), same epilogue: keys <= tau[q] are parked in an LDS hit
//     buffer and handed to the per-query candidate lists once per group.
//
// HBM traffic: one corpus pass (N x d x 4 B) per 128 queries.  The keys are approximate ranking keys exactly as in
// k_flat_mfma; exactness comes from k_rerank + k_certify downstream (index.hip).
#include <type_traits>
#include <atomic>

#include "common.hpp"
#include "kernels.hpp"

namespace vdb {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr uint32_t GEMM_NH = 8;             // 16-query halves per group
constexpr uint32_t GEMM_BQ = 16 * GEMM_NH;  // 128 queries per pass
constexpr uint32_t GEMM_WGBUF = 3072;       // LDS hit buffer entries per workgroup and group
constexpr uint32_t GEMM_STAGE = 64;         // per wave and unit: lanes whose (tile, half) key quartet passed the threshold
static std::atomic<int> g_gemm_tw{3};
void gemm_set_tw(int v) { g_gemm_tw = v == 2 ? 2 : 3; }
static std::atomic<int> g_gemm_nt{0};  // 0 auto (by mirror size), 1 never, 2 always
void gemm_set_nt(int v) { g_gemm_nt = v; }
static std::atomic<int> g_gemm_zigzag{0};  // 0 auto, 1 never, 2 always
void gemm_set_zigzag(int v) { g_gemm_zigzag = v; }
static std::atomic<uint64_t> g_gemm_block_rows{0};  // 0 auto; n: scan in blocks of n rows (rounded to whole workgroup steps), one launch each
void gemm_set_block_rows(uint64_t v) { g_gemm_block_rows = v; }
static std::atomic<int> g_gemm_stagger{0};  // measured: no effect (the epilogue cost is per CU, not a chip-wide HBM gap), kept as a switch
void gemm_set_stagger(int v) { g_gemm_stagger = v; }
uint32_t gemm_group() { return GEMM_BQ; }

struct GemmArgs {
    const uint4 *XT;     // fragment-ordered split-bf16 mirror (k_tile_rows); GEMM_F16: the scaled fp16 mirror (k_tile_rows_h)
    const uint4 *qfrag;  // [ngroups][KB][8 halves][hi|lo][64] (k_mfma_pack_queries with NH = 8); GEMM_F16: [..][sub 0|1][64]
    const float *qmul;   // GEMM_F16: [ngroups*128] 1 / (row scale * query scale), a power of two: S = acc * qmul exactly
    const float *xsq;
    uint64_t n;
    uint32_t KB, n_units, ngroups;
    uint32_t unit_step;  // GEMM_SAMPLE: every unit_step-th unit is scored (n_units counts the sampled ones); else 1
    float *out;          // GEMM_SAMPLE: dense keys out[q*ld + v*16*TW + row in unit] (+inf past n), v = sampled ordinal
    uint64_t ld;
    const float *tau;  // [ngroups*128]
    uint64_t *cand;    // [ngroups*128][cap]
    uint32_t *cnt;     // [ngroups*128]
    uint32_t cap;
    int cosine;
    uint32_t debug;
    uint32_t stagger;  // GEMM_FILTER: one unit step in 10-ns ticks (0: start all workgroups together), see the kernel
    uint32_t nt;       // GEMM_FILTER: non-temporal X loads (host-side choice, see GEMM_NT_BIT)
    uint32_t zigzag;   // GEMM_FILTER: odd query groups walk their full steps in reverse (the rows the previous pass read last are still in the Infinity Cache)
    uint32_t row_base; // GEMM_FILTER: first row of the block this launch scans (XT, xsq and n are the block's; ids are global)
    uint32_t coop;        // GEMM_FILTER: > 1 = cooperative sets of that many workgroups per XCD (k_gemm8.hip: the same scheme)
    uint32_t coop_block;  // ... unit steps between two hand-overs of the workgroup's hit buffer
    uint32_t *sync;       // ... 128 zeroed words: one arrival counter per set (cnt + ngroups * 128)
};

enum { GEMM_FILTER = 0, GEMM_SAMPLE = 1 };
// measurement builds only (make EXTRA=-DGEMM_ABLATE=n): 1 = the epilogue looks at one (tile, half) pair only,
// 2 = only the first half's MFMAs are issued, 4 = no chunk barrier; loads and LDS traffic stay.  Results are wrong by design.
#ifndef GEMM_ABLATE
#define GEMM_ABLATE 0
#endif
// Cache policy of the X stream (template bit GEMM_NT_BIT of the precision parameter, filter pass only).  A mirror far
// larger than the 256 MB Infinity Cache is read exactly once per pass and nothing of it survives to the next pass:
// non-temporal loads leave L2 / the Infinity Cache to the Q chunks and stream faster (1M x 960: 2.69 -> 2.58 ms fp16,
// 5.43 -> 5.24 ms split-bf16, old and new library on one box).  A shard that FITS the Infinity Cache (125k x 960 fp16 =
// 240 MB) is served from it in passes 2..8 -- there non-temporal loads cost 0.32 -> 0.38 ms.  launch_flat_gemm_filter
// picks by the mirror's size.
constexpr int GEMM_NT_BIT = 2;
// Arithmetic of the contraction.  GEMM_BF16X3: x*q ~ xh*qh + xh*ql + xl*qh on 32-column k-blocks, 4 B per element
// in the mirror.  GEMM_F16: x*q ~ fp16(sx*x) * fp16(sq*q) on 64-column k-blocks (two 16x16x32 f16 MFMAs), 2 B per
// element: a third of the matrix work and half of the HBM bytes per row.  The data movement is the same in both: a
// k-block of a tile is two 1-KB fragments ([hi|lo] or [columns 0-31 | 32-63]).  The coarser rounding of GEMM_F16 is
// measured when the mirror / the query image is written (k_row_split_err, k_query_prep_h) and enters the
// certification bound (k_flat_finish); queries it cannot certify are redone with GEMM_BF16X3 (index.hip).
enum { GEMM_BF16X3 = 0, GEMM_F16 = 1 };

// GEMM_SAMPLE: blockIdx.y = query group, one step per workgroup over the sampled units, keys written densely -- the
// threshold sample of the same queries with the same arithmetic as the filter pass (the small-batch kernel's sample
// mode re-reads the sampled rows once per 32 queries; this one once per 128: 80 -> ~25 us at a 125k-row shard).
template <int TW, int KC, int MODE, int PV>
__global__ __launch_bounds__(512, 1) void k_flat_gemm(GemmArgs a) {
    constexpr int PREC = PV & 1;
    constexpr bool XNT = (PV & GEMM_NT_BIT) != 0;
    constexpr int NT = 512, NW = 8, NH = GEMM_NH, R = KC;
    constexpr uint32_t CHUNK = KC * NH * 128;  // uint4 per Q chunk
    constexpr int QST = CHUNK / NT;            // staged uint4 per thread
    static_assert(CHUNK % NT == 0, "chunk must split evenly over the workgroup");
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];  // [2][CHUNK] Q chunks, then the hit buffer
    uint64_t *hit_key = reinterpret_cast<uint64_t *>(smem + 2 * CHUNK);
    uint32_t *hit_q = reinterpret_cast<uint32_t *>(hit_key + GEMM_WGBUF);
    uint32_t *hit_n = hit_q + GEMM_WGBUF;  // [0] entries, [1..128] per-query counts, [129..256] per-query bases
    float *tau_s = reinterpret_cast<float *>(hit_n + 4 + 2 * GEMM_BQ);  // [128] thresholds of the current group (16-B aligned from here on)
    float *qm_s = tau_s + GEMM_BQ;                                      // [128] GEMM_F16: scale undo per query
    float *xs_s = qm_s + GEMM_BQ;                                       // [8 waves][64] row norms of the wave's current unit
    float *stage_s = xs_s + 8 * 64;  // [8 waves][GEMM_STAGE] float4 keys, then [8][GEMM_STAGE] first rows, then [8][GEMM_STAGE] queries

    // wave-uniform values are made visibly uniform (readfirstlane) so that addresses are SGPR base + 32-bit lane offset
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * NW, gw = blockIdx.x * NW + wave;
    const uint32_t KB = a.KB, nchunk = KB / KC;
    const uint64_t n = a.n;

    // ---- X stream: k-block k of a unit is consumed from ring slot k % R; as soon as its MFMAs are issued the slot is
    // refilled with k-block k + R (R = KC: the same position of the next chunk).  In the unit's last chunk the refills
    // run into the wave's NEXT unit (next step, or step 0 again for the next query group), so the ring never drains
    // and R k-blocks (18 KB per wave at TW = 3) stay in flight across the chunk barrier and the unit's epilogue; the
    // position where that happens is static (last chunk), which keeps the chunk body free of branches.
    static_assert(R == KC, "a refill targets the same slot of the next chunk");
    // Units are dealt round-robin: step s < S0 gives wave gw unit s*nwaves + gw.  The n_units % nwaves units that are
    // left do not fill another step of the whole chip; they go to a window of workgroups that ROTATES with the query
    // group, so that over the launch every workgroup gets about the same number of steps (1M rows, TW = 3: 10.17
    // units per wave; 11 steps everywhere would idle 7.5 % of the launch) -- workgroups move on to the next group
    // without a grid-wide barrier, only the per-workgroup totals matter.  With fewer units than waves (S0 = 0) every
    // workgroup runs the single partial step.
    const uint32_t S0 = a.n_units / nwaves, rem = a.n_units - S0 * nwaves;
    const uint32_t rem_wg = (rem + NW - 1) / NW;  // workgroups in the window
    // slot = this workgroup's position relative to the window's start; the window moves by rem_wg per group
    const uint32_t rw = S0 == 0 ? 0u : rem_wg;
    // Cooperative sets (a.coop = S, 256 workgroups; derived and measured in k_gemm8.hip): the 32 workgroups of an XCD (workgroup b runs
    // on XCD b % 8) form 32 / S slices of S members that take S different query groups and walk the SAME units at the same time, so
    // a unit comes from HBM once per S groups and from the XCD's L2 for the other members.  Here the chunk barriers need the same
    // number of unit steps in every wave: all run wave 0's count, a wave past its last unit scores the clamped unit masked.
    const uint32_t coopS = MODE == GEMM_FILTER ? a.coop : 0u;
    const bool coop = coopS > 1;
    const uint32_t c_li = blockIdx.x >> 3, c_member = coop ? c_li % coopS : 0u, c_slices = coop ? 32u / coopS : 1u;
    const uint32_t c_stride = c_slices * 64u, c_base0 = ((coop ? c_li / coopS : 0u) * 8u) * 8u + (blockIdx.x & 7u), c_base = c_base0 + wave * 8u;
    const uint32_t c_steps_max = c_base0 < a.n_units ? (a.n_units - c_base0 + c_stride - 1) / c_stride : 0u;
    const uint32_t gstep = coop ? coopS : 1u;
    auto adv = [&](uint32_t slot) -> uint32_t {
        if (coop) return slot;
        return slot >= rw ? slot - rw : slot + gridDim.x - rw;
    };
    auto steps_of = [&](uint32_t slot) -> uint32_t {
        if (coop) return c_steps_max;
        if (rem == 0) return S0;
        if (S0 == 0) return 1;
        return S0 + (slot < rem_wg ? 1u : 0u);
    };
    // Odd query groups may walk the full steps backwards (a.zigzag): a pass then starts on the rows the previous pass
    // read last -- up to an Infinity Cache worth of the mirror is still resident -- instead of on the rows it evicted first.
    auto unit_of = [&](uint32_t slot, uint32_t st, uint32_t g) -> uint32_t {  // may be >= n_units (idle wave of the window's tail)
        if (coop) return c_base + st * c_stride;
        const uint32_t sw = (MODE == GEMM_FILTER && a.zigzag && (g & 1)) ? S0 - 1 - st : st;
        return st < S0 ? sw * nwaves + gw : S0 * nwaves + slot * NW + wave;
    };
    auto unit_ptr = [&](uint32_t u) -> const char * {  // wave-uniform
        if (u >= a.n_units) u = a.n_units - 1;  // idle waves re-read the last unit (L2 hits, results masked)
        return reinterpret_cast<const char *>(a.XT) + uint64_t(u) * a.unit_step * TW * KB * 2048;
    };
    // Set rendezvous (k_gemm8.hip: why): the members of a set have to START together -- a member that comes tens of microseconds late (its CU was
    // still held by another call's exact stage: pipelined or concurrent callers) finds nothing of the others' rows in the L2 any
    // more and never catches up, and eight members streaming apart with allocating loads are slower than the plain form (measured:
    // 3 steps in flight at 1M rows 2.33 ms per step against 1.03 alone).  One counter per set behind the hit counters (zeroed by
    // the query preparation); the wait is bounded, so two such grids can never hold each other up for good.
    if (coop && a.sync) {
        if (threadIdx.x == 0) {
            uint32_t *ctr = a.sync + (blockIdx.x & 7u) * 16u + c_li / coopS;
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = __builtin_readcyclecounter();
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < coopS && __builtin_readcyclecounter() - t0 < 250000ull)  // ~0.1 ms (k_gemm8.hip: why)
                __builtin_amdgcn_s_sleep(16);
        }
        __syncthreads();
    }
    // Every workgroup does the same work at the same rate, so without help all 256 CUs reach their unit epilogues -- where a
    // workgroup issues no loads for ~2.5 us -- at the same moments and HBM idles chip-wide once per unit step.  A one-off
    // start delay of 0..15/16 of a unit step, different for neighbouring workgroups, spreads those gaps over the period.
    if (MODE == GEMM_FILTER && a.stagger) {
        const uint64_t t0 = wall_clock64();
        const uint64_t wait = (uint64_t((blockIdx.x * 7u) & 15u) * a.stagger) >> 4;
        while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    uint32_t slot_cur = blockIdx.x, slot_nxt = adv(slot_cur);  // of the current and of the next query group
    const uint32_t g0 = MODE == GEMM_SAMPLE ? blockIdx.y : 0;
    const char *cp_cur = unit_ptr(unit_of(slot_cur, 0, g0)),
               *cp_nxt = unit_ptr(1 < steps_of(slot_cur) ? unit_of(slot_cur, 1, g0) : unit_of(slot_nxt, 0, g0 + 1));
    uint32_t voff[TW];  // this lane's byte offset inside a unit, per tile
#pragma unroll
    for (int t = 0; t < TW; t++) voff[t] = lane * 16 + t * KB * 2048;
    uint4 ring[R][TW][2];
    auto fetch_at = [&](uint4(&dst)[TW][2], const char *base, uint32_t kb) {
        const char *sb = base + kb * 2048;  // scalar
#pragma unroll
        for (int t = 0; t < TW; t++) {
            if constexpr (XNT) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                dst[t][0] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(sb + voff[t])));
                dst[t][1] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(sb + voff[t] + 1024)));
            } else {
                dst[t][0] = *reinterpret_cast<const uint4 *>(sb + voff[t]);
                dst[t][1] = *reinterpret_cast<const uint4 *>(sb + voff[t] + 1024);
            }
        }
    };
#pragma unroll
    for (int p = 0; p < R; p++) fetch_at(ring[p], cp_cur, p);

    // ---- Q chunk staging: QP uint4 per thread and k-block, global -> registers at the top of the k-block (BEFORE
    // its X loads, so that waiting for them later is a counted vmcnt that leaves the X loads in flight), registers ->
    // the other LDS buffer at the bottom ----
    constexpr int QP = QST / KC;
    static_assert(QST % KC == 0, "staging must split evenly over the k-blocks of a chunk");
    {
        const uint4 *src = a.qfrag + (MODE == GEMM_SAMPLE ? uint64_t(blockIdx.y) : uint64_t(c_member)) * nchunk * CHUNK;
#pragma unroll
        for (int j = 0; j < QST; j++) smem[j * NT + threadIdx.x] = src[j * NT + threadIdx.x];
    }
    uint32_t buf = 0;
    __syncthreads();

    const uint32_t g_begin = MODE == GEMM_SAMPLE ? blockIdx.y : 0, g_end = MODE == GEMM_SAMPLE ? blockIdx.y + 1 : a.ngroups;
    for (uint32_t grp = g_begin + c_member; grp < g_end; grp += gstep) {
        const uint4 *qgrp = a.qfrag + uint64_t(grp) * nchunk * CHUNK;
        if (MODE == GEMM_FILTER && threadIdx.x < 1 + 2 * GEMM_BQ) hit_n[threadIdx.x] = 0;  // ordered before the first append by the chunk barriers
        if (MODE == GEMM_FILTER && threadIdx.x < GEMM_BQ)  // thresholds live in LDS, not in 8 registers per lane
            tau_s[threadIdx.x] = (a.debug & 1) ? -INFINITY : a.tau[grp * GEMM_BQ + threadIdx.x];
        if (PREC == GEMM_F16 && threadIdx.x < GEMM_BQ) qm_s[threadIdx.x] = a.qmul[grp * GEMM_BQ + threadIdx.x];
        const uint32_t steps = steps_of(slot_cur);
        const uint32_t blk = coop && a.coop_block ? a.coop_block : 0xFFFFFFFFu;  // unit steps between two hand-overs
        for (uint32_t b0 = 0;;) {  // blocks of unit steps [b0, b1) (one block unless cooperative), a hand-over after each
        const uint32_t b1 = blk >= steps - b0 ? steps : b0 + blk;
        for (uint32_t st = b0; st < b1; st++) {
            const uint32_t u_raw = unit_of(slot_cur, st, grp);
            const uint32_t u = u_raw < a.n_units ? u_raw : a.n_units - 1;
            f32x4 acc[TW][NH];
#pragma unroll
            for (int t = 0; t < TW; t++)
#pragma unroll
                for (int h = 0; h < NH; h++) acc[t][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // row norms of this unit: one dword per lane, staged global -> VGPR -> LDS with the Q chunk (every chunk
            // re-stages the same 16*TW values: no branch in the chunk body).  As scalar loads in the epilogue they cost
            // several dependent round trips per unit with all 8 waves waiting (measured: 0.58 of 3.1 ms).
            // (scalar base + 32-bit lane offset, like the X loads: a 64-bit per-lane pointer held across the unit gets spilled,
            // and its reload waits behind the whole X ring)
            const char *xs_base = reinterpret_cast<const char *>(a.xsq + uint64_t(__builtin_amdgcn_readfirstlane(u)) * a.unit_step * (16 * TW));
            uint32_t xs_off = (lane < 16 * TW ? lane : 0) * 4;
            asm volatile("" : "+v"(xs_off));  // per unit on purpose: hoisted out of the unit loop it becomes base + offset as a 64-bit VGPR pair
            for (uint32_t c = 0; c < nchunk; c++) {
                // next chunk in consumption order: same group until its last step is done
                const uint4 *nxt = (c + 1 < nchunk ? qgrp + uint64_t(c + 1) * CHUNK
                                    : (st + 1 < steps ? qgrp : (grp + gstep < a.ngroups ? qgrp + uint64_t(gstep) * nchunk * CHUNK : a.qfrag)));
                const uint32_t tid16 = threadIdx.x * 16;
                uint4 *qdst = smem + (buf ^ 1) * CHUNK + threadIdx.x;
                const uint4 *qcur = smem + buf * CHUNK + lane;
                const bool last_c = c + 1 == nchunk;
                bf16x8 qh_n = __builtin_bit_cast(bf16x8, qcur[0]);  // [hi|lo] (GEMM_F16: [columns 0-31 | 32-63]) of half 0
                bf16x8 ql_n = __builtin_bit_cast(bf16x8, qcur[64]);
                float xs_stage = 0.0f;
#pragma unroll
                for (int p = 0; p < KC; p++) {
                    static_assert(QP == 2, "two staged uint4 per thread and k-block");  // scalars: an array here stays in scratch
                    const uint4 qs0 = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(nxt + (p * QP + 0) * NT) + tid16);
                    const uint4 qs1 = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(nxt + (p * QP + 1) * NT) + tid16);
                    if (p == 0) xs_stage = *reinterpret_cast<const float *>(xs_base + xs_off);
                    // the machine scheduler otherwise sinks these loads next to their uses (measured: vmcnt(0) before every
                    // staging ds_write, i.e. the whole X ring drained once per k-block) and pulls the B-fragment reads
                    // back to just before their MFMAs; pin the issue order instead
                    __builtin_amdgcn_sched_barrier(0);
                    bf16x8 xh[TW], xl[TW];
#pragma unroll
                    for (int t = 0; t < TW; t++) {
                        xh[t] = __builtin_bit_cast(bf16x8, ring[p][t][0]);
                        xl[t] = __builtin_bit_cast(bf16x8, ring[p][t][1]);
                    }
#pragma unroll
                    for (int h = 0; h < NH; h++) {
                        const bf16x8 qh = qh_n, ql = ql_n;
                        if (!(p == KC - 1 && h == NH - 1)) {  // B fragments one step ahead of their MFMAs
                            const int pn = h + 1 < NH ? p : p + 1, hn = h + 1 < NH ? h + 1 : 0;
                            qh_n = __builtin_bit_cast(bf16x8, qcur[((pn * NH + hn) * 2 + 0) * 64]);
                            ql_n = __builtin_bit_cast(bf16x8, qcur[((pn * NH + hn) * 2 + 1) * 64]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if ((GEMM_ABLATE & 2) && h > 0) {
                            acc[0][h][0] += __builtin_bit_cast(f32x4, qh)[0] + __builtin_bit_cast(f32x4, ql)[0];  // keep the reads
                        } else if constexpr (PREC == GEMM_BF16X3) {
#pragma unroll
                            for (int t = 0; t < TW; t++)
                                acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[t], qh, acc[t][h], 0, 0, 0);
#pragma unroll
                            for (int t = 0; t < TW; t++)
                                acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[t], ql, acc[t][h], 0, 0, 0);
#pragma unroll
                            for (int t = 0; t < TW; t++)
                                acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[t], qh, acc[t][h], 0, 0, 0);
                        } else {  // the two fragments are columns 0-31 and 32-63 of the 64-column k-block
#pragma unroll
                            for (int t = 0; t < TW; t++)
                                acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xh[t]), __builtin_bit_cast(f16x8, qh), acc[t][h], 0, 0, 0);
#pragma unroll
                            for (int t = 0; t < TW; t++)
                                acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xl[t]), __builtin_bit_cast(f16x8, ql), acc[t][h], 0, 0, 0);
                            if (GEMM_ABLATE & 8) {  // measurement: the matrix work and B-fragment reads of 256 queries per pass
                                typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
                                const volatile u32x4v *qv = reinterpret_cast<const volatile u32x4v *>(qcur);
                                u32x4v e0, e1;
                                if (GEMM_ABLATE & 16) {  // (16: the matrix work only, B fragments reused -- what wider row units would do)
                                    e0 = __builtin_bit_cast(u32x4v, qh);
                                    e1 = __builtin_bit_cast(u32x4v, ql);
                                } else {
                                    e0 = qv[((p * NH + h) * 2 + 0) * 64];
                                    e1 = qv[((p * NH + h) * 2 + 1) * 64];
                                }
#pragma unroll
                                for (int t = 0; t < TW; t++)
                                    acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xh[t]), __builtin_bit_cast(f16x8, e0), acc[t][h], 0, 0, 0);
#pragma unroll
                                for (int t = 0; t < TW; t++)
                                    acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xl[t]), __builtin_bit_cast(f16x8, e1), acc[t][h], 0, 0, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // slot p is free: refill it (before the staging writes, the chunk barrier and a possible epilogue)
                    fetch_at(ring[p], last_c ? cp_nxt : cp_cur, last_c ? uint32_t(p) : (c + 1) * KC + p);
                    __builtin_amdgcn_sched_barrier(0);
                    qdst[(p * QP + 0) * NT] = qs0;
                    qdst[(p * QP + 1) * NT] = qs1;
                    if (p == 0) xs_s[wave * 64 + lane] = xs_stage;
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!(GEMM_ABLATE & 4)) __syncthreads();  // (4: measurement only -- the Q chunks race)
                buf ^= 1;
            }
            cp_cur = cp_nxt;
            {   // cp_cur now points at the unit after (grp, st); cp_nxt at the one after that (past the last group the
                // sequence just continues: those loads are never used)
                uint32_t un;
                if (st + 2 < steps)
                    un = unit_of(slot_cur, st + 2, grp);
                else if (st + 1 < steps)
                    un = unit_of(slot_nxt, 0, grp + 1);
                else
                    un = 1 < steps_of(slot_nxt) ? unit_of(slot_nxt, 1, grp + 1) : unit_of(adv(slot_nxt), 0, grp + 2);
                cp_nxt = unit_ptr(un);
            }
            // ---- epilogue: lane holds rows 4*g4..4*g4+3 of each tile for query r of each half ----
            // All 8 waves reach their epilogues together, so these cycles are not hidden behind a partner wave's MFMAs
            // and everything on the common path is kept short: key = c + (acc * qm) * m with c = |x|^2, m = -2 (L2Sqr)
            // or c = 0, m = -1/|x| (Cosine; the products with powers of two are exact, so one rounding per key in either
            // form, the same in the sample and the filter instantiation); one v_min3 + v_min per (tile, half); the
            // thresholds / scale factors of the next half are read from LDS while this one is tested.  A lane whose
            // smallest key passes only STAGES its four keys (one ds_write_b128 + one b64); the per-key tests, the slot
            // reservation in the workgroup's hit buffer and the pair keys are done once per unit, for all staged records
            // in parallel, by one shared piece of code (24 inlined copies of that path cost an instruction-cache miss
            // per use: 0.35 of 3.0 ms).
            const uint64_t row0 = uint64_t(u_raw) * a.unit_step * (16 * TW);  // the unclamped unit: idle waves are past n
            uint32_t stage_n = 0;                                             // wave-uniform
            // Lane constants of the epilogue are recomputed here from an opaque lane id: carried across the main loop they
            // are spilled, and a scratch reload at this point waits (vmcnt is in order) for the whole X ring to land --
            // a full drain of the prefetch once per unit (measured: the epilogue cost its full duration, 0.3 of 2.8 ms).
            uint32_t lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(lane_e));
            const uint32_t r = lane_e & 15, g4 = lane_e >> 4, lane = lane_e;  // (shadows the kernel-wide lane)
            float4 *stage_k = reinterpret_cast<float4 *>(stage_s) + wave * GEMM_STAGE;
            uint32_t *stage_r = reinterpret_cast<uint32_t *>(reinterpret_cast<float4 *>(stage_s) + 8 * GEMM_STAGE) + wave * GEMM_STAGE;  // first row
            uint32_t *stage_q = stage_r + 8 * GEMM_STAGE;                                                                             // query in group
            float tau_n = MODE == GEMM_FILTER ? tau_s[r] : 0.0f, qm_n = PREC == GEMM_F16 ? qm_s[r] : 1.0f;
#pragma unroll
            for (int t = 0; t < TW; t++) {
                const float4 x4 = *reinterpret_cast<const float4 *>(&xs_s[wave * 64 + t * 16 + 4 * g4]);
                float cv[4] = {x4.x, x4.y, x4.z, x4.w}, mv[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {  // zero-norm rows under Cosine: S = 0 and the reference distance is exactly 1 -> key 0
                    const float inv = cv[e] > 0.0f ? __frsqrt_rn(cv[e]) : 0.0f;
                    mv[e] = a.cosine ? -inv : -2.0f;
                    cv[e] = a.cosine ? 0.0f : cv[e];
                }
                const f32x2 c01 = {cv[0], cv[1]}, c23 = {cv[2], cv[3]}, m01 = {mv[0], mv[1]}, m23 = {mv[2], mv[3]};
                const uint32_t rb32 = uint32_t(row0) + t * 16 + 4 * g4;  // rows < 2^32 (gemm_args)
                // two instantiations of the pair loop (the metric is launch-uniform): L2Sqr needs one packed fma per key pair
                // (m = -2 qm per query), Cosine two packed multiplies
                auto tile_pairs = [&](auto cos_tag) {
                constexpr bool COS = decltype(cos_tag)::value;
                // the stage holds GEMM_STAGE lane records; when the next pair's passing lanes do not fit, the loop over the
                // halves stops there, the stage is drained and the loop resumes at that pair (hub rows -- small norms under
                // L2Sqr -- pass for most queries of a group at once)
                int h_from = -1;  // wave-uniform; >= 0: resume at this pair after a drain
                for (;;) {
                int h_stop = NH;
                if (h_from >= 0) {  // the stopped pair's prefetch has already moved tau_n / qm_n on (also when it was pair 0)
                    if (MODE == GEMM_FILTER) tau_n = tau_s[h_from * 16 + r];
                    if (PREC == GEMM_F16) qm_n = qm_s[h_from * 16 + r];
                }
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    if (MODE == GEMM_FILTER && h < h_from) continue;
                    if ((GEMM_ABLATE & 1) && (t > 0 || h > 0)) {
                        if (acc[t][h][0] + acc[t][h][1] + acc[t][h][2] + acc[t][h][3] == 1.2345f) atomicAdd(hit_n, 1u);
                        continue;
                    }
                    const float tau_h = tau_n, qm = qm_n;
                    {
                        const int hn = (h + 1) % NH;  // the next pair's query (h = 0 again for the next tile)
                        if (MODE == GEMM_FILTER) tau_n = tau_s[hn * 16 + r];
                        if (PREC == GEMM_F16) qm_n = qm_s[hn * 16 + r];
                    }
                    f32x2 a01 = {acc[t][h][0], acc[t][h][1]}, a23 = {acc[t][h][2], acc[t][h][3]};
                    f32x2 k01, k23;
                    if constexpr (COS) {  // -(S qm) / |x|: the power-of-two scale is exact, one rounding
                        if (PREC == GEMM_F16) {
                            a01 *= qm;
                            a23 *= qm;
                        }
                        k01 = a01 * m01;
                        k23 = a23 * m23;
                    } else {  // |x|^2 - 2 qm S as one fma: the product with the power of two is exact, one rounding
                        const float mq = PREC == GEMM_F16 ? -2.0f * qm : -2.0f;
                        const f32x2 mq2 = {mq, mq};
                        k01 = __builtin_elementwise_fma(a01, mq2, c01);
                        k23 = __builtin_elementwise_fma(a23, mq2, c23);
                    }
                    if (MODE == GEMM_SAMPLE) {
                        if (u_raw < a.n_units) {  // wave-uniform: waves past the last sampled unit write nothing
                            float4 kv;
                            kv.x = rb32 + 0 < n ? k01.x : INFINITY;
                            kv.y = rb32 + 1 < n ? k01.y : INFINITY;
                            kv.z = rb32 + 2 < n ? k23.x : INFINITY;
                            kv.w = rb32 + 3 < n ? k23.y : INFINITY;
                            const uint64_t col = uint64_t(u_raw) * (16 * TW) + t * 16 + 4 * g4;  // dense position in the sample
                            *reinterpret_cast<float4 *>(a.out + (uint64_t(grp) * GEMM_BQ + h * 16 + r) * a.ld + col) = kv;
                        }
                        continue;
                    }
                    float kmin3, kmin;  // NaN keys never pass: v_min returns the other operand, the per-key tests are ordered
                    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(kmin3) : "v"(k01.x), "v"(k01.y), "v"(k23.x));
                    asm("v_min_f32 %0, %1, %2" : "=v"(kmin) : "v"(kmin3), "v"(k23.y));
                    const bool pass = kmin <= tau_h;
                    const uint64_t pm = __ballot(pass);
                    if (pm) {  // rare: ~1000 rows per query in total (mfma_sample_plan)
                        const uint32_t np = __builtin_popcountll(pm);
                        if (__builtin_amdgcn_readfirstlane(stage_n + np > GEMM_STAGE)) {  // wave-uniform
                            h_stop = h;
                            break;
                        }
                        if (pass) {
                            const uint32_t slot = stage_n + __builtin_amdgcn_mbcnt_hi(uint32_t(pm >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(pm), 0u));
                            stage_k[slot] = make_float4(k01.x, k01.y, k23.x, k23.y);
                            stage_r[slot] = rb32;
                            stage_q[slot] = h * 16 + r;
                        }
                        stage_n += np;
                    }
                }
                // drained once per unit (after the last tile) and whenever the stage fills up
                if (MODE == GEMM_FILTER && stage_n && (h_stop != NH || t == TW - 1)) {  // wave-uniform
                    const uint32_t cnt = stage_n < GEMM_STAGE ? stage_n : GEMM_STAGE;
                    for (uint32_t i = lane; i < cnt; i += 64) {
                        const float4 kv = stage_k[i];
                        const uint2 mt = make_uint2(stage_r[i], stage_q[i]);
                        const float tq = tau_s[mt.y];
                        const bool p0 = kv.x <= tq && mt.x + 0 < n, p1 = kv.y <= tq && mt.x + 1 < n;
                        const bool p2 = kv.z <= tq && mt.x + 2 < n, p3 = kv.w <= tq && mt.x + 3 < n;
                        const uint32_t mine = uint32_t(p0) + uint32_t(p1) + uint32_t(p2) + uint32_t(p3);
                        if (mine) {
                            uint32_t pos = atomicAdd(hit_n, mine);
                            // every reserved slot below GEMM_WGBUF is written (the hand-off reads min(total, GEMM_WGBUF) slots)
    #define VDB_PARK(P, KEY, E)                                   \
        if (P) {                                                  \
            if (pos < GEMM_WGBUF) {                               \
                hit_key[pos] = pair_key(KEY, a.row_base + mt.x + E); \
                hit_q[pos] = mt.y;                                \
            }                                                     \
            pos++;                                                \
        }
                            VDB_PARK(p0, kv.x, 0)
                            VDB_PARK(p1, kv.y, 1)
                            VDB_PARK(p2, kv.z, 2)
                            VDB_PARK(p3, kv.w, 3)
    #undef VDB_PARK
                            if (pos > GEMM_WGBUF)  // buffer full: mark the query as overflowed (-> redone by the caller)
                                atomicAdd(&a.cnt[grp * GEMM_BQ + mt.y], a.cap + 1);
                        }
                    }
                    stage_n = 0;
                }
                if (MODE != GEMM_FILTER || h_stop == NH) break;
                h_from = h_stop;
                }
                };
                if (a.cosine)
                    tile_pairs(std::true_type{});
                else
                    tile_pairs(std::false_type{});
            }
        }
        // ---- group end: hand the parked hits to the per-query candidate lists (one global atomic per query) ----
        __syncthreads();
        if (MODE == GEMM_FILTER) {
            uint32_t total = hit_n[0];
            if (total > GEMM_WGBUF) total = GEMM_WGBUF;
            constexpr uint32_t NJ = (GEMM_WGBUF + NT - 1) / NT;
            uint32_t rank[NJ];
#pragma unroll
            for (uint32_t j = 0; j < NJ; j++) {
                uint32_t i = j * NT + threadIdx.x;
                rank[j] = i < total ? atomicAdd(&hit_n[1 + hit_q[i]], 1u) : 0u;
            }
            __syncthreads();
            if (threadIdx.x < GEMM_BQ && hit_n[1 + threadIdx.x] > 0)
                hit_n[1 + GEMM_BQ + threadIdx.x] = atomicAdd(&a.cnt[grp * GEMM_BQ + threadIdx.x], hit_n[1 + threadIdx.x]);
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < NJ; j++) {
                uint32_t i = j * NT + threadIdx.x;
                if (i < total) {
                    uint32_t q = hit_q[i];
                    uint32_t slot = hit_n[1 + GEMM_BQ + q] + rank[j];
                    if (slot < a.cap) a.cand[(uint64_t(grp) * GEMM_BQ + q) * a.cap + slot] = hit_key[i];
                }
            }
            __syncthreads();
            if (coop) {  // the next block parks into an empty buffer
                if (threadIdx.x < 1 + 2 * GEMM_BQ) hit_n[threadIdx.x] = 0;
                __syncthreads();
            }
        }
        b0 = b1;
        if (b0 >= steps) break;
        }
        slot_cur = slot_nxt;
        slot_nxt = adv(slot_nxt);
    }
}

template <int TW, int KC, int MODE, int PREC>
static void flat_gemm_launch1(const GemmArgs &a0, int num_cu, hipStream_t s);
static std::atomic<int> g_gemm_coop{0};  // 0 auto (cooperative sets when the shape allows), 1 off
void gemm_set_coop(int v) { g_gemm_coop = v; }
static std::atomic<uint32_t> g_gemm_last_coop{0};  // set size of the most recent filter launch (0: no sets)
uint32_t gemm_last_coop() { return g_gemm_last_coop; }
template <int TW, int KC, int MODE, int PREC>
static void flat_gemm_launch(const GemmArgs &a0, int num_cu, hipStream_t s) {
    GemmArgs a = a0;
    if constexpr (MODE == GEMM_FILTER) {
        // cooperative sets (k_gemm8.hip): the chip's 8 x 32 CUs, a set size that divides the group count, whole-table launches, every
        // wave of a slice with units to score; blocks sized so that ~1000 hits per query fill less than half the hit buffer
        if (g_gemm_coop != 1 && num_cu == 256 && a.row_base == 0 && !a.stagger) {
            const uint32_t S = a.ngroups % 8 == 0 ? 8u : (a.ngroups % 4 == 0 ? 4u : (a.ngroups % 2 == 0 ? 2u : 1u));
            const uint64_t units = ((a.n + 15) / 16 + TW - 1) / TW;
            if (S > 1 && units >= 2048) {
                a.coop = S;
                g_gemm_last_coop = S;
                a.zigzag = 0;
                const double per_step = 8.0 * 16.0 * TW * 128.0 * 1024.0 / double(a.n);
                const double b = double(GEMM_WGBUF) * 0.45 / per_step;
                a.coop_block = b < 1.0 ? 1u : (b > 4096.0 ? 4096u : uint32_t(b));
                flat_gemm_launch1<TW, KC, MODE, PREC>(a, num_cu, s);  // default loads: the members meet in the L2
                return;
            }
        }
        g_gemm_last_coop = 0;
        if (a.nt) {
            flat_gemm_launch1<TW, KC, MODE, PREC | GEMM_NT_BIT>(a, num_cu, s);
            return;
        }
    }
    flat_gemm_launch1<TW, KC, MODE, PREC>(a, num_cu, s);
}
template <int TW, int KC, int MODE, int PREC>
static void flat_gemm_launch1(const GemmArgs &a0, int num_cu, hipStream_t s) {
    GemmArgs a = a0;
    const uint64_t n_tiles = (a.n + 15) / 16;
    const uint32_t units_all = (uint32_t)((n_tiles + TW - 1) / TW);
    uint32_t grid;
    if (MODE == GEMM_SAMPLE) {
        a.n_units = (units_all + a.unit_step - 1) / a.unit_step;  // sampled units: ordinal v scores unit v * unit_step
        grid = (a.n_units + 7) / 8;                               // one step: a wave per sampled unit
    } else {
        a.unit_step = 1;
        a.n_units = units_all;
        grid = (uint32_t)num_cu;
        const uint32_t need = (a.n_units + 7) / 8;
        if (need < grid) grid = need;
    }
    if (grid == 0 || a.ngroups == 0) return;
    const size_t lds = size_t(2) * KC * GEMM_NH * 128 * sizeof(uint4) + size_t(GEMM_WGBUF) * 12 + (4 + 4 * GEMM_BQ + 8 * 64) * 4 + size_t(8) * GEMM_STAGE * 24 + 16;
    func_max_lds(reinterpret_cast<const void *>(&k_flat_gemm<TW, KC, MODE, PREC>), int(160 * 1024));
    hipLaunchKernelGGL((k_flat_gemm<TW, KC, MODE, PREC>), dim3(grid, MODE == GEMM_SAMPLE ? a.ngroups : 1), dim3(512), lds, s, a);
    VDB_HIP(hipGetLastError());
}

template <int MODE>
static void flat_gemm_dispatch(const GemmArgs &a, int num_cu, hipStream_t s) {
    if (a.qmul) {  // GEMM_F16: KB counts 64-column k-blocks; chunks of 3, 2 or 1 of them
        VDB_REQUIRE(a.KB % 3 == 0 || a.KB % 2 == 0, "flat_gemm: fp16 mirror needs a k-block count divisible by 2 or 3");
        if (a.KB % 3 == 0) {
            if (g_gemm_tw == 2)
                flat_gemm_launch<2, 3, MODE, GEMM_F16>(a, num_cu, s);
            else
                flat_gemm_launch<3, 3, MODE, GEMM_F16>(a, num_cu, s);
        } else {
            if (g_gemm_tw == 2)
                flat_gemm_launch<2, 2, MODE, GEMM_F16>(a, num_cu, s);
            else
                flat_gemm_launch<3, 2, MODE, GEMM_F16>(a, num_cu, s);
        }
        return;
    }
    if (a.KB % 3 == 0) {
        if (g_gemm_tw == 2)
            flat_gemm_launch<2, 3, MODE, GEMM_BF16X3>(a, num_cu, s);
        else
            flat_gemm_launch<3, 3, MODE, GEMM_BF16X3>(a, num_cu, s);
    } else {  // KB is even (columns padded to a multiple of 64)
        if (g_gemm_tw == 2)
            flat_gemm_launch<2, 2, MODE, GEMM_BF16X3>(a, num_cu, s);
        else
            flat_gemm_launch<3, 2, MODE, GEMM_BF16X3>(a, num_cu, s);
    }
}
// GEMM_F16 serves dims whose 64-column k-block count splits into chunks of 3 or 2 (every dim except those with an odd
// count that is not a multiple of 3: 64*5, 64*7, ... use the split-bf16 kernel only)
bool gemm_f16_supported(uint32_t dim) {
    const uint32_t kb = mfma_dim_pad(dim) / 64;
    return kb >= 2 && (kb % 3 == 0 || kb % 2 == 0) && mfma_dim_pad(dim) <= 2048;
}

static GemmArgs gemm_args(const float *XT, uint64_t n, uint32_t dim, const float *qfrag, const float *qmul, uint32_t ngroups,
                          const float *xsq, int cosine) {
    VDB_REQUIRE(n < (1ull << 32), "flat_gemm: too many rows for one shard");
    GemmArgs a{};
    a.XT = reinterpret_cast<const uint4 *>(XT);
    a.qfrag = reinterpret_cast<const uint4 *>(qfrag);
    a.xsq = xsq;
    a.n = n;
    a.qmul = qmul;
    a.KB = mfma_dim_pad(dim) / (qmul ? 64 : 32);
    a.ngroups = ngroups;
    a.cosine = cosine;
    a.unit_step = 1;
    return a;
}

// rows past n up to a whole unit are read from the mirror (zero tiles) and from xsq (padding): see Index::add_rows
void launch_flat_gemm_filter(const float *XT, uint64_t n, uint32_t dim, const float *qfrag, const float *qmul, uint32_t ngroups,
                             const float *xsq, int cosine, const float *tau, uint64_t *cand, uint32_t *cnt,
                             uint32_t cap, int debug, int num_cu, hipStream_t s) {
    if (n == 0 || ngroups == 0) return;
    GemmArgs a = gemm_args(XT, n, dim, qfrag, qmul, ngroups, xsq, cosine);
    a.tau = tau;
    a.cand = cand;
    a.cnt = cnt;
    a.sync = cnt + uint64_t(ngroups) * GEMM_BQ;  // (Index::flat_knn_enqueue: the rendezvous words follow the padded counters, zeroed)
    a.cap = cap;
    a.debug = (uint32_t)debug;
    {   // mirror bytes of this shard against the Infinity Cache (256 MB): stream past it, or let it serve passes 2..
        const double mirror_bytes = double((n + 15) / 16 * 16) * mfma_dim_pad(dim) * (qmul ? 2 : 4);
        a.nt = g_gemm_nt == 2 || (g_gemm_nt == 0 && mirror_bytes > 384.0 * 1024 * 1024) ? 1u : 0u;
        a.zigzag = g_gemm_zigzag == 2 ? 1u : 0u;
    }
    if (g_gemm_stagger) {  // one unit step of a workgroup at ~6 TB/s spread over the CUs, in 10-ns ticks (wall_clock64 runs at 100 MHz)
        const double unit_bytes = 8.0 * g_gemm_tw * 16 * mfma_dim_pad(dim) * (qmul ? 2 : 4);
        a.stagger = (uint32_t)(unit_bytes / (6.0e12 / num_cu) * 1e8 * g_gemm_stagger);
    }
    // Row-blocked scan (measurement switch, flat_gemm_block_rows): one launch per block of rows walking ALL query groups, so
    // that a block comes from HBM once and from the Infinity Cache for the other groups.  Measured on mirrors of 1-4
    // Infinity-Cache sizes (250k / 400k / 1M rows x 960, blocks of ~100k rows): 0.716 -> 0.738, 1.074 -> 1.040,
    // 2.50 -> 2.59 ms per 1000 queries -- a block leaves a wave barely more than one unit per group, and the kernel, not
    // HBM, is the limit anyway (a cache-resident shard reads at the same 6.1 TB/s).  Off unless requested.
    uint64_t block_rows = g_gemm_block_rows;
    const uint64_t unit_rows = 16ull * g_gemm_tw;
    block_rows = block_rows / (unit_rows * 8) * (unit_rows * 8);  // whole steps of a workgroup
    if (block_rows == 0 || block_rows >= n) {
        flat_gemm_dispatch<GEMM_FILTER>(a, num_cu, s);
        return;
    }
    for (uint64_t r0 = 0; r0 < n; r0 += block_rows) {
        GemmArgs b = a;
        b.row_base = (uint32_t)r0;
        b.n = std::min<uint64_t>(block_rows, n - r0);
        b.XT = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.XT) + (r0 / 16) * uint64_t(a.KB) * 2048);
        b.xsq = a.xsq + r0;
        b.nt = 0;  // the block is meant to stay in the Infinity Cache
        flat_gemm_dispatch<GEMM_FILTER>(b, num_cu, s);
    }
}

// rows of the threshold sample: every unit_step-th unit of 16*TW rows (the unit size follows flat_gemm_tw)
uint64_t gemm_sample_rows(uint64_t n, uint32_t unit_step) {
    const uint32_t tw = (uint32_t)g_gemm_tw;
    const uint64_t units = ((n + 15) / 16 + tw - 1) / tw;
    return (units + unit_step - 1) / unit_step * (16 * tw);
}
// dense keys of the sample for every query of every group: out[q*ld + j], j < gemm_sample_rows(n, unit_step), +inf past n
void launch_flat_gemm_sample(const float *XT, uint64_t n, uint32_t dim, const float *qfrag, const float *qmul, uint32_t ngroups,
                             const float *xsq, int cosine, uint32_t unit_step, float *out, uint64_t ld, int num_cu,
                             hipStream_t s) {
    if (n == 0 || ngroups == 0) return;
    VDB_REQUIRE(unit_step >= 1 && (ld & 3) == 0 && ld >= gemm_sample_rows(n, unit_step), "flat_gemm: ld must cover the sample");
    VDB_REQUIRE(ngroups <= 65535, "flat_gemm: too many query groups");
    GemmArgs a = gemm_args(XT, n, dim, qfrag, qmul, ngroups, xsq, cosine);
    a.unit_step = unit_step;
    a.out = out;
    a.ld = ld;
    flat_gemm_dispatch<GEMM_SAMPLE>(a, num_cu, s);
}

}  // namespace vdb
