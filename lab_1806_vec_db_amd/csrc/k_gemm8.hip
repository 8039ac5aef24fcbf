// k_gemm8.hip -- the Flat shortlist pass on 8-bit operands: 128 queries per HBM pass over a 1-B/element mirror.
//
// Same loop nest and data movement as k_flat_gemm (k_gemm.hip: a wave owns a unit of TW 16-row tiles with its TW x 8
// accumulator tiles in registers, the X stream runs HBM -> register ring, the query group's B image is cut into K-chunks
// double-buffered in LDS), with v_mfma_i32_16x16x64_i8 as the product: one 1-KB fragment per (tile, 64-column k-block),
// half the bytes per row of the fp16 pass and the same number of matrix instructions per byte.
//
// What the keys are (k_i8.hip writes the operands and derives this): rows and queries are CENTRED on one vector mu of the
// index (L2Sqr is translation invariant) and rounded to int8 with one scale per row / per query; the integer sums I(r,q)
// are exact, and
//     key(r, q) = C_r + M_r * (s_q * I(r, q))          M_r = -2 s_r
// with C_r = |x_r - mu|^2 minus the row's share of the rounding error (Cauchy-Schwarz on the measured residuals, split
// between row and query by the AM-GM inequality so that it stays a sum of a row term and a query term) is a LOWER BOUND of
// D(r, q) - O_q for a per-query offset O_q -- not an estimate with an error bar.  A row whose key exceeds the threshold
// cannot be closer than threshold + O_q; the exact stage (k_flat_tail_lb, k_exact.hip) walks the shortlist in key order
// and stops as soon as the k-th exact distance is below the next key's bound.  Cosine uses the same kernel on unit rows.
#include <atomic>
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

namespace vdb {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr uint32_t G8_NH = 8;            // 16-query halves per group
constexpr uint32_t G8_BQ = 16 * G8_NH;   // 128 queries per pass (= gemm_group())
constexpr uint32_t G8_WGBUF = 3072;      // LDS hit buffer entries per workgroup and group
constexpr uint32_t G8_WGBUF_RES = 1536;  // ... of the resident form (expected ~512 at 1000 hits per query; a full buffer flags the query)
constexpr uint32_t G8_STAGE = 64;        // per wave and unit: lanes whose (tile, half) key quartet passed the threshold
constexpr uint32_t G8_TW = 3;            // 16-row tiles per unit (the mirror is padded to whole units of 12 tiles: index.hip)

struct Gemm8Args {
    const uint4 *XT;      // [tile][kb64][lane] 16 int8: row 16*tile + (lane & 15), columns 64*kb + 16*(lane >> 4) + j
    const uint4 *qfrag;   // [group][kb64][half][lane] 16 int8 of query 16*half + (lane & 15), same columns
    const float *qscale;  // [ngroups*128] s_q
    const float2 *rowc;   // [rows padded to whole units] {C_r, M_r}; rows >= n: {+inf, 0}
    uint64_t n;
    uint32_t KB, n_units, ngroups;
    uint32_t unit_step;   // G8_SAMPLE: every unit_step-th unit is scored (n_units counts the sampled ones); else 1
    float *out;           // G8_SAMPLE: dense keys out[q*ld + v*16*TW + row in unit] (+inf past n); unit_min: out[q*ld + v] = the unit's smallest key
    uint32_t unit_min;    // G8_SAMPLE: 1 = one value per (query, sampled unit) instead of one per row (see launch_flat_gemm8_sample)
    uint64_t ld;
    const float *tau;     // [ngroups*128]
    uint64_t *cand;       // [ngroups*128][cap]
    uint32_t *cnt;        // [ngroups*128]
    uint32_t cap;
    uint32_t nt;          // non-temporal X loads (mirror beyond the Infinity Cache)
    uint32_t debug;       // bit 0: thresholds of -inf (nothing passes: the no-hit detection downstream, tests)
    uint32_t coop;        // resident filter form: > 1 = the workgroups of an XCD in sets of `coop` that share one row stream (see the kernel)
    uint32_t coop_block;  // ... units a wave scores between two hand-overs of the workgroup's hit buffer (plain resident form: 0 = one hand-over per group)
    uint32_t hits_expected;  // hits per query the thresholds were made for (sizes the hand-over blocks; 0 = the sample plan's ~1000)
    uint32_t *sync;       // ... 128 zeroed words: one arrival counter per set (cnt + ngroups * 128)
};

enum { G8_FILTER = 0, G8_SAMPLE = 1 };
// measurement builds only (make EXTRA=-DG8_ABLATE=n): 4 = no Q staging and no chunk barrier (every chunk multiplies with the group's
// first chunk: what the per-chunk rendezvous of the 8 waves costs), 1 = the epilogue looks at one (tile, half) pair only, 2 = only the first
// half's MFMAs are issued; loads, LDS traffic and barriers stay.  Results are wrong by design.
#ifndef G8_ABLATE
#define G8_ABLATE 0
#endif

// BURST: the next Q chunk's KC staged pieces are all loaded at the top of the chunk and written at its bottom, so that the
// wait for them leaves every refill of the chunk in flight (vmcnt is in order: waiting for a staging load that was issued
// AFTER a ring refill forces that refill to have landed; with one piece per k-block the ring is effectively one k-block
// deep -- 1.5 us of matrix work in the fp16 kernel, only half that here).  Costs 4 (KC - 1) registers.
// RES (round 3, last sessions): the whole query group's image (KB x 8 KB, 120 KB at dim 960) stays in LDS for the group's pass
// instead of being staged chunk by chunk through two 24-KB buffers.  What that removes is the workgroup barrier per chunk: with it
// the 8 waves consumed their ring slots in lockstep, so every chunk waited for the SLOWEST of the workgroup's 72 loads in flight
// (tools/inflight_probe.cpp: a wave that only waits for its own oldest load streams 7.0 TB/s non-temporally with 48 KB per CU in
// flight; the chunked kernel got 5.3 - 5.7; the ablation build -DG8_ABLATE=4, no staging and no chunk barrier: -11 % kernel time).
// Each wave now runs on its own counted waits, the ring is KC k-blocks deep without the staging registers, and a barrier is
// left per group (Q image load, hit hand-over).  The hit buffer shrinks to G8_WGBUF_RES entries to make room.
template <int KC, int MODE, bool XNT, bool BURST, bool RES>
__global__ __launch_bounds__(512, 1) void k_flat_gemm8(Gemm8Args a) {
    constexpr int TW = G8_TW, NT = 512, NW = 8, NH = G8_NH, R = KC;
    constexpr uint32_t CHUNK = KC * NH * 64;  // uint4 per Q chunk (KC k-blocks of 8 KB)
    constexpr int QST = CHUNK / NT;           // staged uint4 per thread and chunk: one per k-block
    static_assert(CHUNK % NT == 0 && QST == KC, "one staged uint4 per thread and k-block");
    static_assert(!(RES && BURST), "the resident form stages nothing");
    constexpr uint32_t WGBUF = RES ? G8_WGBUF_RES : G8_WGBUF;
    extern __shared__ __attribute__((aligned(16))) uint4 smem8[];  // [2][CHUNK] Q chunks (RES: the group's [KB / KC][CHUNK]), then the hit buffer
    uint64_t *hit_key = reinterpret_cast<uint64_t *>(smem8 + (RES ? (a.KB / KC) * CHUNK : 2 * CHUNK));
    uint32_t *hit_q = reinterpret_cast<uint32_t *>(hit_key + WGBUF);
    uint32_t *hit_n = hit_q + WGBUF;  // [0] entries, [1..128] per-query counts, [129..256] per-query bases
    float *tau_s = reinterpret_cast<float *>(hit_n + 4 + 2 * G8_BQ);  // [128] thresholds of the current group
    float *qs_s = tau_s + G8_BQ;                                      // [128] query scales
    float *c_s = qs_s + G8_BQ;                                        // [8 waves][64] C_r of the wave's current unit
    float *m_s = c_s + 8 * 64;                                        // [8 waves][64] M_r
    float *stage_s = m_s + 8 * 64;  // [8 waves][G8_STAGE] float4 keys, then [8][G8_STAGE] first rows, then [8][G8_STAGE] queries

    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * NW, gw = blockIdx.x * NW + wave;
    const uint32_t KB = a.KB, nchunk = KB / KC;
    const uint64_t n = a.n;

    // units are dealt exactly as in k_flat_gemm: round-robin full steps, the remainder to a window of workgroups that rotates
    // with the query group
    const uint32_t S0 = a.n_units / nwaves, rem = a.n_units - S0 * nwaves;
    const uint32_t rem_wg = (rem + NW - 1) / NW;
    const uint32_t rw = S0 == 0 ? 0u : rem_wg;
    // COOPERATIVE form (RES filter, 256 workgroups = 8 XCDs x 32 CUs, a.coop = S in {2, 4, 8} dividing the group count): workgroup b runs
    // on XCD b % 8 (round-robin dispatch; tools/l2share_probe.cpp: 256 of 256), and the 32 workgroups of an XCD form 32 / S slices of S
    // members.  The members of a slice take S DIFFERENT query groups and walk the SAME units in the same order at the same time, so a
    // unit comes from HBM once per S groups and from the XCD's 4-MB L2 for the other members (the probe: 8 x 0.96 GB read in 0.42 ms
    // = 18 TB/s where 8 passes from HBM take 1.10 ms).  A member that runs ahead misses and slows down, one that lags hits: the set
    // holds together by itself.  A workgroup handles ngroups / S groups (one image load and hand-over chain each) over 1 / (8 x 32 / S)
    // of the units, i.e. S x the hits per group: the hit buffer is handed over every a.coop_block units.  Nothing here is needed for
    // correctness -- on a chip that dispatches differently the members only stop sharing.
    const uint32_t coopS = (RES && MODE == G8_FILTER) ? a.coop : 0u;
    const bool coop = coopS > 1;
    const uint32_t c_li = blockIdx.x >> 3, c_member = coop ? c_li % coopS : 0u, c_slices = coop ? (gridDim.x >> 3) / coopS : 1u;  // (256 workgroups: 32 per XCD)
    const uint32_t c_stride = c_slices * 64u, c_base0 = ((coop ? c_li / coopS : 0u) * 8u) * 8u + (blockIdx.x & 7u), c_base = c_base0 + wave * 8u;
    const uint32_t c_steps = c_base < a.n_units ? (a.n_units - c_base + c_stride - 1) / c_stride : 0u;       // this wave's units per group
    const uint32_t c_steps_max = c_base0 < a.n_units ? (a.n_units - c_base0 + c_stride - 1) / c_stride : 0u;  // wave 0's: the most of the workgroup
    auto adv = [&](uint32_t slot) -> uint32_t {
        if (coop) return slot;
        return slot >= rw ? slot - rw : slot + gridDim.x - rw;
    };
    auto steps_of = [&](uint32_t slot) -> uint32_t {
        if (coop) return c_steps;
        if (rem == 0) return S0;
        if (S0 == 0) return 1;
        return S0 + (slot < rem_wg ? 1u : 0u);
    };
    auto unit_of = [&](uint32_t slot, uint32_t st) -> uint32_t {  // may be >= n_units (idle wave of the window's tail)
        if (coop) return c_base + st * c_stride;
        return st < S0 ? st * nwaves + gw : S0 * nwaves + slot * NW + wave;
    };
    auto unit_ptr = [&](uint32_t u) -> const char * {  // wave-uniform
        if (u >= a.n_units) u = a.n_units - 1;  // idle waves re-read the last unit (L2 hits, results masked)
        return reinterpret_cast<const char *>(a.XT) + uint64_t(u) * a.unit_step * TW * KB * 1024;
    };
    // Set rendezvous: the members of a set have to START together -- a member that comes tens of microseconds late (its CU was
    // still held by another call's exact stage: pipelined or concurrent callers) finds nothing of the others' rows in the L2 any
    // more and never catches up, and eight members streaming apart with allocating loads are slower than the plain form (measured:
    // 3 steps in flight at 1M rows 2.33 ms per step against 1.03 alone).  One counter per set behind the hit counters (zeroed by
    // the query preparation); the wait is bounded, so two such grids can never hold each other up for good.
    if (coop && a.sync) {
        if (threadIdx.x == 0) {
            uint32_t *ctr = a.sync + (blockIdx.x & 7u) * 16u + c_li / coopS;
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = __builtin_readcyclecounter();
            // (bounded at the lateness that matters: ~0.1 ms -- a member later than that is held by ANOTHER grid (a second index's pass, a
            // walk), and waiting longer only adds its delay to this call; the set then runs unshared, nothing else changes)
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < coopS && __builtin_readcyclecounter() - t0 < 250000ull)
                __builtin_amdgcn_s_sleep(16);
        }
        __syncthreads();
    }
    uint32_t slot_cur = blockIdx.x, slot_nxt = adv(slot_cur);
    const char *cp_cur = unit_ptr(unit_of(slot_cur, 0)),
               *cp_nxt = unit_ptr(1 < steps_of(slot_cur) ? unit_of(slot_cur, 1) : unit_of(slot_nxt, 0));
    uint32_t voff[TW];
#pragma unroll
    for (int t = 0; t < TW; t++) voff[t] = lane * 16 + t * KB * 1024;
    uint4 ring[R][TW];
    auto fetch_at = [&](uint4(&dst)[TW], const char *base, uint32_t kb) {
        const char *sb = base + kb * 1024;  // scalar
#pragma unroll
        for (int t = 0; t < TW; t++) {
            if constexpr (XNT) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                dst[t] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(sb + voff[t])));
            } else {
                dst[t] = *reinterpret_cast<const uint4 *>(sb + voff[t]);
            }
        }
    };
    // (issue order pinned: the waits inside the loop are computed from the merge of this order and the steady state's; left
    // to the scheduler the prologue goes tile-major and every chunk then starts by draining most of the ring)
#pragma unroll
    for (int p = 0; p < R; p++) {
        fetch_at(ring[p], cp_cur, p);
        __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (!RES) {
        const uint4 *src = a.qfrag + (MODE == G8_SAMPLE ? uint64_t(blockIdx.y) * nchunk * CHUNK : 0);
#pragma unroll
        for (int j = 0; j < QST; j++) smem8[j * NT + threadIdx.x] = src[j * NT + threadIdx.x];
    }
    uint32_t buf = 0;
    if constexpr (!RES) __syncthreads();

    const uint32_t g_begin = MODE == G8_SAMPLE ? blockIdx.y : 0, g_end = MODE == G8_SAMPLE ? blockIdx.y + 1 : a.ngroups;
    for (uint32_t grp = g_begin + c_member; grp < g_end; grp += coop ? coopS : 1u) {
        const uint4 *qgrp = a.qfrag + uint64_t(grp) * nchunk * CHUNK;
        if constexpr (RES) {  // (the previous group's readers are past the hand-over's last barrier)
            for (uint32_t c = 0; c < nchunk; c++) {
                uint4 t[QST];
#pragma unroll
                for (int j = 0; j < QST; j++) t[j] = qgrp[(c * QST + j) * NT + threadIdx.x];
#pragma unroll
                for (int j = 0; j < QST; j++) smem8[(c * QST + j) * NT + threadIdx.x] = t[j];
            }
        }
        if (MODE == G8_FILTER && threadIdx.x < 1 + 2 * G8_BQ) hit_n[threadIdx.x] = 0;  // ordered before the first append by the chunk barriers (RES: the one below)
        if (MODE == G8_FILTER && threadIdx.x < G8_BQ) tau_s[threadIdx.x] = (a.debug & 1) ? -INFINITY : a.tau[grp * G8_BQ + threadIdx.x];
        if (threadIdx.x < G8_BQ) qs_s[threadIdx.x] = a.qscale[grp * G8_BQ + threadIdx.x];
        if constexpr (RES) __syncthreads();
        // RES: the B fragments run through a ring of FOUR registers sets, three (tile-triple) steps ahead of their MFMAs and across
        // chunk and unit boundaries (a unit is KB x 8 steps, a multiple of 4, so the ring keeps its places; the last chunk of a unit
        // prefetches the first fragments of the image again, for the next unit).  One step ahead (rounds 3: ds_read, then
        // lgkmcnt(1) in front of the next triple) left 48 matrix cycles for an LDS round trip of ~100+ with eight waves reading.
        i32x4 qr0 = {0, 0, 0, 0}, qr1 = qr0, qr2 = qr0, qr3 = qr0;
        if constexpr (RES) {
            const uint4 *q0p = smem8 + lane;
            qr0 = __builtin_bit_cast(i32x4, q0p[0]);
            qr1 = __builtin_bit_cast(i32x4, q0p[64]);
            qr2 = __builtin_bit_cast(i32x4, q0p[128]);
        }
        const uint32_t steps = steps_of(slot_cur);
        const uint32_t steps_all = coop ? c_steps_max : steps;                       // workgroup-uniform
        const uint32_t blk = RES && a.coop_block ? a.coop_block : 0xFFFFFFFFu;       // units between two hand-overs
        for (uint32_t b0 = 0;;) {  // blocks of units [b0, b1) (one block unless cooperative), a hand-over after each
        const uint32_t b1 = blk >= steps_all - b0 ? steps_all : b0 + blk;  // workgroup-uniform (b0 <= steps_all)
        const uint32_t st_end = b1 < steps ? b1 : steps;                   // this wave's share of the block
        for (uint32_t st = b0; st < st_end; st++) {
            const uint32_t u_raw = unit_of(slot_cur, st);
            const uint32_t u = u_raw < a.n_units ? u_raw : a.n_units - 1;
            i32x4 acc[TW][NH];
#pragma unroll
            for (int t = 0; t < TW; t++)
#pragma unroll
                for (int h = 0; h < NH; h++) acc[t][h] = (i32x4){0, 0, 0, 0};
            // row constants of this unit: one float2 per lane (lanes < 16 TW), staged global -> VGPR -> LDS with the Q chunk
            const char *rc_base = reinterpret_cast<const char *>(a.rowc + uint64_t(__builtin_amdgcn_readfirstlane(u)) * a.unit_step * (16 * TW));
            uint32_t rc_off = (lane < 16 * TW ? lane : 0) * 8;
            asm volatile("" : "+v"(rc_off));  // per unit on purpose (see k_flat_gemm)
            for (uint32_t c = 0; c < nchunk; c++) {
                const uint4 *nxt = (c + 1 < nchunk ? qgrp + uint64_t(c + 1) * CHUNK
                                    : (st + 1 < steps ? qgrp : (grp + 1 < a.ngroups ? qgrp + uint64_t(nchunk) * CHUNK : a.qfrag)));
                const uint32_t tid16 = threadIdx.x * 16;
                uint4 *qdst = smem8 + (buf ^ 1) * CHUNK + threadIdx.x;
                const uint4 *qcur = smem8 + (RES ? c : buf) * CHUNK + lane;
                const bool last_c = c + 1 == nchunk;
                i32x4 q_n = {0, 0, 0, 0};
                if constexpr (!RES) q_n = __builtin_bit_cast(i32x4, qcur[0]);
                const uint4 *qnx = smem8 + (last_c ? 0u : (c + 1) * CHUNK) + lane;  // RES: the fragments that follow this chunk's
                float2 rc_stage = make_float2(0.0f, 0.0f);
                // staged pieces as named scalars (a local array stays in scratch: k_flat_gemm); piece p lives in qs<p>
                uint4 qs0, qs1, qs2, qs3, qs4;
                auto stage_ref = [&](auto P) -> uint4 & {
                    constexpr int p = decltype(P)::value;
                    if constexpr (p == 0) return qs0;
                    else if constexpr (p == 1) return qs1;
                    else if constexpr (p == 2) return qs2;
                    else if constexpr (p == 3) return qs3;
                    else return qs4;
                };
                auto stage_load = [&](auto P) {
                    constexpr int p = decltype(P)::value;
                    stage_ref(P) = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(nxt + p * NT) + tid16);
                };
                auto kblock = [&](auto P) {
                    constexpr int p = decltype(P)::value;
                    if constexpr (!BURST && !RES) {
                        stage_load(P);
                        if (p == 0) rc_stage = *reinterpret_cast<const float2 *>(rc_base + rc_off);
                    }
                    if constexpr (RES) {  // the unit's row constants: every chunk fetches and parks them (no branch around a load: the
                        // wait counts stay exact); the epilogue reads what the last chunk parked
                        if (p == 0) rc_stage = *reinterpret_cast<const float2 *>(rc_base + rc_off);
                    }
                    __builtin_amdgcn_sched_barrier(0);  // pin the issue order (k_flat_gemm: why)
                    i32x4 xv[TW];
#pragma unroll
                    for (int t = 0; t < TW; t++) xv[t] = __builtin_bit_cast(i32x4, ring[p][t]);
#pragma unroll
                    for (int h = 0; h < NH; h++) {
                        i32x4 qv;
                        if constexpr (RES) {
                            const int sidx = p * NH + h, f = sidx + 3;  // this step's ring place is sidx & 3; fragment f goes to (sidx + 3) & 3
                            qv = (sidx & 3) == 0 ? qr0 : ((sidx & 3) == 1 ? qr1 : ((sidx & 3) == 2 ? qr2 : qr3));
                            const i32x4 ld = __builtin_bit_cast(i32x4, f < KC * NH ? qcur[f * 64] : qnx[(f - KC * NH) * 64]);
                            if ((f & 3) == 0) qr0 = ld;
                            else if ((f & 3) == 1) qr1 = ld;
                            else if ((f & 3) == 2) qr2 = ld;
                            else qr3 = ld;
                        } else {
                            qv = q_n;
                            if (!(p == KC - 1 && h == NH - 1)) {  // B fragments one step ahead of their MFMAs
                                const int pn = h + 1 < NH ? p : p + 1, hn = h + 1 < NH ? h + 1 : 0;
                                q_n = __builtin_bit_cast(i32x4, qcur[(pn * NH + hn) * 64]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if ((G8_ABLATE & 2) && h > 0) {
                            acc[0][h][0] += qv[0];  // keep the read
                        } else {
#pragma unroll
                            for (int t = 0; t < TW; t++) acc[t][h] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xv[t], qv, acc[t][h], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // slot p is free: refill it (before the staging writes, the chunk barrier and a possible epilogue)
                    fetch_at(ring[p], last_c ? cp_nxt : cp_cur, last_c ? uint32_t(p) : (c + 1) * KC + p);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!BURST && !RES) {
                        qdst[p * NT] = stage_ref(P);
                        if (p == 0) {
                            c_s[wave * 64 + lane] = rc_stage.x;
                            m_s[wave * 64 + lane] = rc_stage.y;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                using I0 = std::integral_constant<int, 0>;
                using I1 = std::integral_constant<int, 1>;
                using I2 = std::integral_constant<int, 2>;
                using I3 = std::integral_constant<int, 3>;
                using I4 = std::integral_constant<int, 4>;
                if constexpr (BURST && !(G8_ABLATE & 4)) {
                    stage_load(I0{});
                    stage_load(I1{});
                    if constexpr (KC > 2) stage_load(I2{});
                    if constexpr (KC > 3) stage_load(I3{});
                    if constexpr (KC > 4) stage_load(I4{});
                    rc_stage = *reinterpret_cast<const float2 *>(rc_base + rc_off);
                }
                kblock(I0{});
                kblock(I1{});
                if constexpr (KC > 2) kblock(I2{});
                if constexpr (KC > 3) kblock(I3{});
                if constexpr (KC > 4) kblock(I4{});
                if constexpr (BURST && !(G8_ABLATE & 4)) {
                    qdst[0 * NT] = qs0;
                    qdst[1 * NT] = qs1;
                    if constexpr (KC > 2) qdst[2 * NT] = qs2;
                    if constexpr (KC > 3) qdst[3 * NT] = qs3;
                    if constexpr (KC > 4) qdst[4 * NT] = qs4;
                    c_s[wave * 64 + lane] = rc_stage.x;
                    m_s[wave * 64 + lane] = rc_stage.y;
                }
                if constexpr (RES) {  // this wave's own [64] slice: written and read by the same wave, LDS is in order per wave
                    c_s[wave * 64 + lane] = rc_stage.x;
                    m_s[wave * 64 + lane] = rc_stage.y;
                } else if constexpr (!(G8_ABLATE & 4)) {
                    __syncthreads();
                    buf ^= 1;
                }
            }
            cp_cur = cp_nxt;
            {
                uint32_t un;
                if (st + 2 < steps)
                    un = unit_of(slot_cur, st + 2);
                else if (st + 1 < steps)
                    un = unit_of(slot_nxt, 0);
                else
                    un = 1 < steps_of(slot_nxt) ? unit_of(slot_nxt, 1) : unit_of(adv(slot_nxt), 0);
                cp_nxt = unit_ptr(un);
            }
            // ---- epilogue: lane holds rows 4*g4..4*g4+3 of each tile for query r of each half ----
            const uint64_t row0 = uint64_t(u_raw) * a.unit_step * (16 * TW);  // the unclamped unit: idle waves are past n
            uint32_t stage_n = 0;                                             // wave-uniform
            uint32_t lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(lane_e));  // lane constants recomputed here: carried across the main loop they are spilled
            const uint32_t r = lane_e & 15, g4 = lane_e >> 4, lane = lane_e;  // (shadows the kernel-wide lane)
            float4 *stage_k = reinterpret_cast<float4 *>(stage_s) + wave * G8_STAGE;
            uint32_t *stage_r = reinterpret_cast<uint32_t *>(reinterpret_cast<float4 *>(stage_s) + 8 * G8_STAGE) + wave * G8_STAGE;
            uint32_t *stage_q = stage_r + 8 * G8_STAGE;
            // the lane's query of every half: thresholds and scales for the whole unit, fetched in one batch (per pair, one pair ahead,
            // the ~100-cycle LDS round trip was not covered by the ~45 cycles of a pair's arithmetic)
            float tauv[NH], qsv[NH];
#pragma unroll
            for (int h = 0; h < NH; h++) {
                tauv[h] = MODE == G8_FILTER ? tau_s[h * 16 + r] : 0.0f;
                qsv[h] = qs_s[h * 16 + r];
            }
            // the wave's stage -> the workgroup's hit buffer (per key: exact test against the query's threshold, rows past n dropped)
            auto drain = [&]() {
                const uint32_t cnt = stage_n < G8_STAGE ? stage_n : G8_STAGE;
                for (uint32_t i = lane; i < cnt; i += 64) {
                    const float4 kv = stage_k[i];
                    const uint2 mt = make_uint2(stage_r[i], stage_q[i]);
                    const float tq = tau_s[mt.y];
                    const bool p0 = kv.x <= tq && mt.x + 0 < n, p1 = kv.y <= tq && mt.x + 1 < n;
                    const bool p2 = kv.z <= tq && mt.x + 2 < n, p3 = kv.w <= tq && mt.x + 3 < n;
                    const uint32_t mine = uint32_t(p0) + uint32_t(p1) + uint32_t(p2) + uint32_t(p3);
                    if (mine) {
                        uint32_t pos = atomicAdd(hit_n, mine);
                        // every reserved slot below WGBUF is written (the hand-off reads min(total, WGBUF) slots); a key that finds the
                        // buffer full goes straight to its query's candidate list (one global atomic per key: the slow path of hit
                        // densities above ~3 % of the rows -- small tables with long lists, the second attempt of k_redo.hip -- where
                        // rounds 3 flagged the query as overflowed and sent it to the next tier)
#ifdef G8_NOSPILL  /* measurement builds: the full buffer flags the query (rounds 3) */
#define G8_SPILL(KEY, E) atomicAdd(&a.cnt[grp * G8_BQ + mt.y], a.cap + 1);
#else
#define G8_SPILL(KEY, E)                                                                  \
    const uint32_t gslot = atomicAdd(&a.cnt[grp * G8_BQ + mt.y], 1u);                     \
    if (gslot < a.cap) a.cand[(uint64_t(grp) * G8_BQ + mt.y) * a.cap + gslot] = pair_key(KEY, mt.x + E);
#endif
#define VDB_PARK8(P, KEY, E)                                                                          \
    if (P) {                                                                                          \
        if (pos < WGBUF) {                                                                            \
            hit_key[pos] = pair_key(KEY, mt.x + E);                                                   \
            hit_q[pos] = mt.y;                                                                        \
        } else {                                                                                      \
            G8_SPILL(KEY, E)                                                                          \
        }                                                                                             \
        pos++;                                                                                        \
    }
                        VDB_PARK8(p0, kv.x, 0)
                        VDB_PARK8(p1, kv.y, 1)
                        VDB_PARK8(p2, kv.z, 2)
                        VDB_PARK8(p3, kv.w, 3)
#undef VDB_PARK8
#undef G8_SPILL
                    }
                }
                stage_n = 0;
            };
            float umin[NH];  // (G8_SAMPLE with unit_min)
#pragma unroll
            for (int h = 0; h < NH; h++) umin[h] = INFINITY;
#pragma unroll
            for (int t = 0; t < TW; t++) {
                const float4 c4 = *reinterpret_cast<const float4 *>(&c_s[wave * 64 + t * 16 + 4 * g4]);
                const float4 m4 = *reinterpret_cast<const float4 *>(&m_s[wave * 64 + t * 16 + 4 * g4]);
                const f32x2 c01 = {c4.x, c4.y}, c23 = {c4.z, c4.w}, m01 = {m4.x, m4.y}, m23 = {m4.z, m4.w};
                const uint32_t rb32 = uint32_t(row0) + t * 16 + 4 * g4;  // rows < 2^32
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    if ((G8_ABLATE & 1) && (t > 0 || h > 0)) {
                        if (acc[t][h][0] + acc[t][h][1] + acc[t][h][2] + acc[t][h][3] == 12345) atomicAdd(hit_n, 1u);
                        continue;
                    }
                    const float tau_h = tauv[h], sq = qsv[h];
                    // I is an exact integer (|I| <= 127^2 dim); the conversion is exact up to 2^24, one rounding beyond.
                    // key = C + M * (s_q * I): two roundings, covered by the certification's rounding term
                    const f32x2 sq2 = {sq, sq};
                    f32x2 p01 = {float(acc[t][h][0]), float(acc[t][h][1])}, p23 = {float(acc[t][h][2]), float(acc[t][h][3])};
                    p01 *= sq2;
                    p23 *= sq2;
                    const f32x2 k01 = __builtin_elementwise_fma(p01, m01, c01), k23 = __builtin_elementwise_fma(p23, m23, c23);
                    if (MODE == G8_SAMPLE && a.unit_min) {  // (wave-uniform) the running minimum of this lane's query of half h over the unit's rows
                        float m4;
                        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m4) : "v"(rb32 + 0 < n ? k01.x : INFINITY), "v"(rb32 + 1 < n ? k01.y : INFINITY),
                            "v"(rb32 + 2 < n ? k23.x : INFINITY));
                        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(umin[h]) : "v"(m4), "v"(rb32 + 3 < n ? k23.y : INFINITY), "v"(umin[h]));
                        continue;
                    }
                    if (MODE == G8_SAMPLE) {
                        if (u_raw < a.n_units) {  // wave-uniform: waves past the last sampled unit write nothing
                            float4 kv;
                            kv.x = rb32 + 0 < n ? k01.x : INFINITY;
                            kv.y = rb32 + 1 < n ? k01.y : INFINITY;
                            kv.z = rb32 + 2 < n ? k23.x : INFINITY;
                            kv.w = rb32 + 3 < n ? k23.y : INFINITY;
                            const uint64_t col = uint64_t(u_raw) * (16 * TW) + t * 16 + 4 * g4;  // dense position in the sample
                            *reinterpret_cast<float4 *>(a.out + (uint64_t(grp) * G8_BQ + h * 16 + r) * a.ld + col) = kv;
                        }
                        continue;
                    }
                    float kmin3, kmin;  // NaN keys never pass: v_min returns the other operand, the per-key tests are ordered
                    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(kmin3) : "v"(k01.x), "v"(k01.y), "v"(k23.x));
                    asm("v_min_f32 %0, %1, %2" : "=v"(kmin) : "v"(kmin3), "v"(k23.y));
                    const bool pass = kmin <= tau_h;
                    const uint64_t pm = __ballot(pass);
                    if (pm) {  // wave-uniform (a quarter of the pairs at ~1000 hits per query and 1M rows)
                        const uint32_t np = __builtin_popcountll(pm);
                        // the stage is drained BEFORE a pair that would not fit (np <= 64 = G8_STAGE always fits an empty one): no
                        // resume state -- the earlier break-and-redo form kept its loop state in vector registers and cost 14 % of
                        // the kernel (ablation -DG8_ABLATE=1 on the resident form: 1.30 -> 1.12 ms)
                        if (stage_n + np > G8_STAGE) drain();
                        if (pass) {
                            const uint32_t slot = stage_n + __builtin_amdgcn_mbcnt_hi(uint32_t(pm >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(pm), 0u));
                            stage_k[slot] = make_float4(k01.x, k01.y, k23.x, k23.y);
                            stage_r[slot] = rb32;
                            stage_q[slot] = h * 16 + r;
                        }
                        stage_n += np;
                    }
                }
            }
            if (MODE == G8_SAMPLE && a.unit_min && u_raw < a.n_units) {
                // the four lanes r, r + 16, r + 32, r + 48 hold the same query's rows 4 g4 .. 4 g4 + 3 of every tile: one minimum per query
                // (v_min ignores a NaN operand: NaN keys never set a threshold, as they never pass one)
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    float m = umin[h];
                    m = fminf(m, __shfl_xor(m, 16));
                    m = fminf(m, __shfl_xor(m, 32));
                    if (g4 == 0) a.out[(uint64_t(grp) * G8_BQ + h * 16 + r) * a.ld + u_raw] = m;
                }
            }
            if (MODE == G8_FILTER && stage_n) drain();  // once per unit
        }
        // ---- group end (cooperative form: end of a block of units): hand the parked hits to the per-query candidate lists (one
        // global atomic per query) ----
        __syncthreads();
        if (MODE == G8_FILTER) {
            uint32_t total = hit_n[0];
            if (total > WGBUF) total = WGBUF;
            constexpr uint32_t NJ = (WGBUF + NT - 1) / NT;
            uint32_t rank[NJ];
#pragma unroll
            for (uint32_t j = 0; j < NJ; j++) {
                uint32_t i = j * NT + threadIdx.x;
                rank[j] = i < total ? atomicAdd(&hit_n[1 + hit_q[i]], 1u) : 0u;
            }
            __syncthreads();
            if (threadIdx.x < G8_BQ && hit_n[1 + threadIdx.x] > 0)
                hit_n[1 + G8_BQ + threadIdx.x] = atomicAdd(&a.cnt[grp * G8_BQ + threadIdx.x], hit_n[1 + threadIdx.x]);
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < NJ; j++) {
                uint32_t i = j * NT + threadIdx.x;
                if (i < total) {
                    uint32_t q = hit_q[i];
                    uint32_t slot = hit_n[1 + G8_BQ + q] + rank[j];
                    if (slot < a.cap) a.cand[(uint64_t(grp) * G8_BQ + q) * a.cap + slot] = hit_key[i];
                }
            }
            __syncthreads();
            if (blk != 0xFFFFFFFFu) {  // the next block parks into an empty buffer
                if (threadIdx.x < 1 + 2 * G8_BQ) hit_n[threadIdx.x] = 0;
                __syncthreads();
            }
        }
        b0 = b1;
        if (b0 >= steps_all) break;  // (a workgroup without units hands over once, an empty buffer)
        }
        slot_cur = slot_nxt;
        slot_nxt = adv(slot_nxt);
    }
}

static std::atomic<int> g_gemm8_nt{0};  // 0 auto (by mirror size), 1 never, 2 always
void gemm8_set_nt(int v) { g_gemm8_nt = v; }

// dims whose 64-column k-block count splits into chunks of 5, 3 or 2 k-blocks
bool gemm8_supported(uint32_t dim) {
    const uint32_t pad = mfma_dim_pad(dim);
    const uint32_t kb = pad / 64;
    return kb >= 2 && pad <= 2048 && (kb % 5 == 0 || kb % 3 == 0 || kb % 2 == 0);
}

// LDS of the resident form: the group's whole image + the smaller hit buffer; it exists for dimensions whose image fits
static size_t gemm8_res_lds(uint32_t KB) {
    return size_t(KB) * G8_NH * 64 * sizeof(uint4) + size_t(G8_WGBUF_RES) * 12 + (4 + 4 * G8_BQ + 2 * 8 * 64) * 4 + size_t(8) * G8_STAGE * 24 + 16;
}
template <int KC, int MODE, bool XNT, bool BURST, bool RES>
static void flat_gemm8_launch1(const Gemm8Args &a0, int num_cu, hipStream_t s) {
    Gemm8Args a = a0;
    const uint64_t n_tiles = (a.n + 15) / 16;
    const uint32_t units_all = (uint32_t)((n_tiles + G8_TW - 1) / G8_TW);
    uint32_t grid;
    if (MODE == G8_SAMPLE) {
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
    const size_t lds = RES ? gemm8_res_lds(a.KB)
                           : size_t(2) * KC * G8_NH * 64 * sizeof(uint4) + size_t(G8_WGBUF) * 12 + (4 + 4 * G8_BQ + 2 * 8 * 64) * 4 +
                                 size_t(8) * G8_STAGE * 24 + 16;
    func_max_lds(reinterpret_cast<const void *>(&k_flat_gemm8<KC, MODE, XNT, BURST, RES>), int(160 * 1024));
    hipLaunchKernelGGL((k_flat_gemm8<KC, MODE, XNT, BURST, RES>), dim3(grid, MODE == G8_SAMPLE ? a.ngroups : 1), dim3(512), lds, s, a);
    VDB_HIP(hipGetLastError());
}
static std::atomic<int> g_gemm8_kc{0};  // 0 auto; 5 / 3 / 2: chunk length when the k-block count allows it
void gemm8_set_kc(int v) { g_gemm8_kc = v; }
static std::atomic<int> g_gemm8_burst{0};  // 0 auto (burst for KC <= 3), 1 per k-block, 2 burst
void gemm8_set_burst(int v) { g_gemm8_burst = v; }
static std::atomic<int> g_gemm8_res{0};  // 0 auto (resident image when it fits the LDS), 1 off (chunked staging), 2 as 0
void gemm8_set_res(int v) { g_gemm8_res = v; }

static std::atomic<int> g_gemm8_sample_res{0};  // SAMPLE mode: 0 = as the filter, 1 = chunked staging
void gemm8_set_sample_res(int v) { g_gemm8_sample_res = v; }

static std::atomic<int> g_gemm8_coop{0};  // 0 auto (cooperative sets when the shape allows), 1 off
void gemm8_set_coop(int v) { g_gemm8_coop = v; }
static std::atomic<int> g_gemm8_grid{0};  // measurement switch: workgroups of the cooperative filter (0 = 256; a multiple of 64: whole sets on every XCD)
void gemm8_set_grid(int v) { g_gemm8_grid = (v >= 64 && v <= 256 && v % 64 == 0) ? v : 0; }
static std::atomic<uint32_t> g_gemm8_last_coop{0};  // set size of the most recent filter launch (0: no sets)
uint32_t gemm8_last_coop() { return g_gemm8_last_coop; }

template <int KC, int MODE>
static void flat_gemm8_launch(const Gemm8Args &a0, int num_cu, hipStream_t s) {
    Gemm8Args a = a0;
    const bool nt = MODE == G8_FILTER && a.nt;
    if (g_gemm8_res != 1 && !(MODE == G8_SAMPLE && g_gemm8_sample_res == 1) && gemm8_res_lds(a.KB) <= size_t(160) * 1024) {
        // cooperative sets (see the kernel): the chip's 8 x 32 CUs, a set size that divides the group count, every wave of a slice with
        // units to score; hand-overs sized for ~1000 hits per query (the sample plan's target): a block of B units parks about
        // 8 waves x B x 48 rows x 128 queries x 1024 / n hits, kept near half the buffer (0.55: two units per block on a 125k-row shard)
        if (MODE == G8_FILTER && g_gemm8_coop != 1 && num_cu == 256) {
            const uint32_t S = a.ngroups % 8 == 0 ? 8u : (a.ngroups % 4 == 0 ? 4u : (a.ngroups % 2 == 0 ? 2u : 1u));
            const uint64_t units = ((a.n + 15) / 16 + G8_TW - 1) / G8_TW;
            if (S > 1 && units >= 2048) {
                a.coop = S;
                g_gemm8_last_coop = S;
                // (0.55 of the buffer while a full buffer cost the query its answer from this tier; entries beyond it now spill to the
                // candidate lists directly, so the blocks are sized for 0.8 and the sample plan's own expectation: 125k-row shard 2 -> 6
                // units per block)
                const double per_unit = 8.0 * 48.0 * 128.0 * double(a.hits_expected ? a.hits_expected : 1024u) / double(a.n);
                const double b = double(G8_WGBUF_RES) * 0.8 / per_unit;
                a.coop_block = b < 1.0 ? 1u : (b > 4096.0 ? 4096u : uint32_t(b));
                flat_gemm8_launch1<KC, MODE, false, false, true>(a, g_gemm8_grid ? (int)g_gemm8_grid : num_cu, s);  // default loads: the members meet in the L2
                return;
            }
        }
        if (MODE == G8_FILTER) g_gemm8_last_coop = 0;
        if (MODE == G8_FILTER && a.hits_expected > 2048) {
            // thresholds made for long hit lists (the second attempt, k_redo.hip): the plain form hands its buffer over in blocks too
            // (a workgroup's share of a group's hits: 128 queries x hits / workgroups, against 1536 entries)
            const double per_unit = 8.0 * 48.0 * 128.0 * double(a.hits_expected) / double(a.n);
            const double b = double(G8_WGBUF_RES) * 0.55 / per_unit;
            a.coop_block = b < 1.0 ? 1u : (b > 4096.0 ? 4096u : uint32_t(b));
        }
        if (nt)
            flat_gemm8_launch1<KC, MODE, true, false, true>(a, num_cu, s);
        else
            flat_gemm8_launch1<KC, MODE, false, false, true>(a, num_cu, s);
        return;
    }
    const int bm = g_gemm8_burst;
    const bool burst = KC <= 3 && bm != 1;  // (KC = 5: the 16 extra registers do not fit)
    if (nt) {
        if (burst) {
            if constexpr (KC <= 3) flat_gemm8_launch1<KC, MODE, true, true, false>(a, num_cu, s);
        } else {
            flat_gemm8_launch1<KC, MODE, true, false, false>(a, num_cu, s);
        }
    } else {
        if (burst) {
            if constexpr (KC <= 3) flat_gemm8_launch1<KC, MODE, false, true, false>(a, num_cu, s);
        } else {
            flat_gemm8_launch1<KC, MODE, false, false, false>(a, num_cu, s);
        }
    }
}
template <int MODE>
static void flat_gemm8_dispatch(const Gemm8Args &a, int num_cu, hipStream_t s) {
    const int want = g_gemm8_kc;
    int kc = 0;
    if (want == 5 || want == 3 || want == 2)
        if (a.KB % uint32_t(want) == 0) kc = want;
    const bool res = g_gemm8_res != 1 && !(MODE == G8_SAMPLE && g_gemm8_sample_res == 1) && gemm8_res_lds(a.KB) <= size_t(160) * 1024;
    // (resident form: the chunk length is only the depth of the X ring -- the deepest that divides the k-block count)
    if (kc == 0) kc = res ? (a.KB % 5 == 0 ? 5 : (a.KB % 3 == 0 ? 3 : 2)) : (a.KB % 3 == 0 ? 3 : (a.KB % 5 == 0 ? 5 : 2));
    VDB_REQUIRE(a.KB % uint32_t(kc) == 0, "flat_gemm8: k-block count must be divisible by 5, 3 or 2");
    if (kc == 5)
        flat_gemm8_launch<5, MODE>(a, num_cu, s);
    else if (kc == 3)
        flat_gemm8_launch<3, MODE>(a, num_cu, s);
    else
        flat_gemm8_launch<2, MODE>(a, num_cu, s);
}

static Gemm8Args gemm8_args(const void *XT, uint64_t n, uint32_t dim, const void *qfrag, const float *qscale, uint32_t ngroups,
                            const float *rowc) {
    VDB_REQUIRE(n < (1ull << 32), "flat_gemm8: too many rows for one shard");
    VDB_REQUIRE(gemm8_supported(dim), "flat_gemm8: unsupported dimension");
    Gemm8Args a{};
    a.XT = reinterpret_cast<const uint4 *>(XT);
    a.qfrag = reinterpret_cast<const uint4 *>(qfrag);
    a.qscale = qscale;
    a.rowc = reinterpret_cast<const float2 *>(rowc);
    a.n = n;
    a.KB = mfma_dim_pad(dim) / 64;
    a.ngroups = ngroups;
    a.unit_step = 1;
    return a;
}

// rows past n up to a whole unit are read from the mirror (zero tiles) and from rowc ({+inf, 0}): see Index::i8_refresh
void launch_flat_gemm8_filter(const void *XT, uint64_t n, uint32_t dim, const void *qfrag, const float *qscale, uint32_t ngroups,
                              const float *rowc, const float *tau, uint64_t *cand, uint32_t *cnt, uint32_t cap, int debug, int num_cu,
                              hipStream_t s, uint32_t hits_expected) {
    if (n == 0 || ngroups == 0) return;
    Gemm8Args a = gemm8_args(XT, n, dim, qfrag, qscale, ngroups, rowc);
    a.debug = (uint32_t)debug;
    a.tau = tau;
    a.cand = cand;
    a.cnt = cnt;
    a.sync = cnt + uint64_t(ngroups) * G8_BQ;  // (Index::flat_knn_enqueue: the rendezvous words follow the padded counters)
    a.cap = cap;
    a.hits_expected = hits_expected;
    const double mirror_bytes = double((n + 15) / 16 * 16) * mfma_dim_pad(dim);
    a.nt = g_gemm8_nt == 2 || (g_gemm8_nt == 0 && mirror_bytes > 384.0 * 1024 * 1024) ? 1u : 0u;
    flat_gemm8_dispatch<G8_FILTER>(a, num_cu, s);
}

uint64_t gemm8_sample_rows(uint64_t n, uint32_t unit_step) {
    const uint64_t units = ((n + 15) / 16 + G8_TW - 1) / G8_TW;
    return (units + unit_step - 1) / unit_step * (16 * G8_TW);
}
// dense keys of the sample for every query of every group: out[q*ld + j], j < gemm8_sample_rows(n, unit_step), +inf past n
// unit_min: one value per (query, sampled unit) -- the smallest key of the unit's 48 rows -- instead of one per row: out[q*ld + v],
// v < gemm8_sample_units(n, unit_step).  The r-th smallest of these minima is >= the r-th smallest sampled key (so a threshold taken from
// it lets at least as many rows through: the sample plan's guarantee stands) and equals it unless two of the r smallest rows share a
// unit (r (r - 1) / 2 in n_units: the caller uses it when the units are many); the selection then reads 48 x fewer values.
uint64_t gemm8_sample_units(uint64_t n, uint32_t unit_step) {
    const uint64_t units = ((n + 15) / 16 + G8_TW - 1) / G8_TW;
    return (units + unit_step - 1) / unit_step;
}
void launch_flat_gemm8_sample(const void *XT, uint64_t n, uint32_t dim, const void *qfrag, const float *qscale, uint32_t ngroups,
                              const float *rowc, uint32_t unit_step, float *out, uint64_t ld, int num_cu, hipStream_t s, int unit_min) {
    if (n == 0 || ngroups == 0) return;
    VDB_REQUIRE(unit_step >= 1 && (ld & 3) == 0 && ld >= (unit_min ? gemm8_sample_units(n, unit_step) : gemm8_sample_rows(n, unit_step)),
                "flat_gemm8: ld must cover the sample");
    VDB_REQUIRE(ngroups <= 65535, "flat_gemm8: too many query groups");
    Gemm8Args a = gemm8_args(XT, n, dim, qfrag, qscale, ngroups, rowc);
    a.unit_step = unit_step;
    a.out = out;
    a.ld = ld;
    a.unit_min = unit_min ? 1u : 0u;
    flat_gemm8_dispatch<G8_SAMPLE>(a, num_cu, s);
}

}  // namespace vdb
