// k_mfma.hip -- batched Flat brute force on the gfx950 matrix cores.
//
// FlatIndex::knn (flat_index.rs:48-57) is called once per query in the reference.  Batched over
// B = 32 queries the distance stage is a dense contraction S = X * Q^T, streamed once over the corpus
// per 32 queries: 3.84 GB of HBM reads per pass for Gist1M instead of 32 x 3.84 GB.
// This kernel produces APPROXIMATE ranking keys  key(r,b) = |x_r|^2 - 2 * S(r,b)  (the expanded form of
// distance/mod.rs:54-57 without the per-query constant); the keys only build a shortlist.  The exact,
// reference-order distances of the shortlist are recomputed by k_rerank (k_exact.hip) and the shortlist
// is certified (k_certify) with an error bound that covers this kernel's arithmetic; what leaves the
// library is bit-identical to the reference's strict f32 fold.
//
// Arithmetic: split-bf16.  Every f32 value v is stored as hi = bf16(v), lo = bf16(v - hi)
// (|v - hi - lo| <= 2^-17 |v|) and x*q is taken as xh*qh + xh*ql + xl*qh on
// v_mfma_f32_16x16x32_bf16 with f32 accumulation: three bf16 MFMAs at 16x the f32 MFMA rate, i.e. 3/16
// of the matrix-pipe time (and energy) of the exact-f32 MFMA form.  Measured on MI355X the f32-MFMA
// variant of this kernel sat at 4.5-4.9 TB/s with the matrix pipe ~50 % busy, while a pure read of the
// same stream reaches 6.25 TB/s; the bf16x3 form leaves the kernel bound by HBM alone.  The dropped
// term xl*ql and the representation error are <= 3*2^-17 |x_j q_j| per product -- accounted for in E
// (k_certify), far below the gaps between neighbours that matter for a shortlist.
//
// Layout: the VecSet has a fragment-ordered mirror in HBM (same bytes as the rows, 4 B per element):
//   XT[tile of 16 rows][k-block of 32 columns][hi|lo][lane 0..63] -> 16 B = 8 bf16 = columns
//   32*kb + 8*(lane>>4) + 0..7 of row 16*tile + (lane&15)
// which is exactly the A operand of v_mfma_f32_16x16x32_bf16 (lane l: A[row l&15][k = 8*(l>>4)+j]).  One
// wave load is one contiguous 1 KB; a wave's 32-row item is one contiguous 120 KB stream.  (Loading the
// fragments straight from row-major rows makes the address unit issue 64 separate 16-B accesses per wave
// load -- measured TA ~90 % busy -- because quad-adjacent lanes sit in different rows.)
// Q lives in LDS as the matching B-operand image [kb][query half][hi|lo][lane] (120 KB for d = 960):
// every B fragment is one lane-linear, conflict-free ds_read_b128.  X never touches LDS: it is read
// once, by exactly one wave, straight into VGPRs through a register ring (GEMV regime).
#include <algorithm>
#include <atomic>

#include "common.hpp"
#include "kernels.hpp"

namespace vdb {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// the mirror and the Q images are zero-padded to a multiple of 64 columns, so every dim <= 1024 qualifies
uint32_t mfma_dim_pad(uint32_t dim) { return (dim + 63) & ~63u; }
// queries per workgroup batch: the Q image (batch x dim_pad x 4 B) must fit 128 KB of LDS: 32 queries up to
// dim 1024, 16 queries (one MFMA query half) up to dim 2048, beyond that the exact scan serves the index
uint32_t mfma_batch(uint32_t dim) {
    uint32_t p = mfma_dim_pad(dim);
    return dim == 0 ? 0 : (p <= 1024 ? 32 : (p <= 2048 ? 16 : 0));
}
size_t mfma_qfrag_floats(uint32_t dim) { return size_t(mfma_dim_pad(dim)) * mfma_batch(dim); }
bool mfma_supported(uint32_t dim) { return mfma_batch(dim) != 0; }
constexpr int MFMA_RT = 2;              // 16-row tiles per wave item
constexpr uint32_t MFMA_WGBUF = 3072;   // per-workgroup LDS hit buffer entries (MODE_FILTER): 36 KB beside the 120 KB Q image
uint64_t mfma_row_pad() { return 64; }  // rows of padding the kernel may touch past n (xsq reads)
static std::atomic<int> g_mfma_variant{0};
void mfma_set_variant(int v) { g_mfma_variant = v; }
static std::atomic<uint32_t> g_mfma_share{2};  // query batches per HBM pass (XCD-shared passes, see k_flat_mfma)
void mfma_set_share(int v) { g_mfma_share = v < 1 ? 1 : (v > 8 ? 8 : (uint32_t)v); }
uint32_t mfma_share() { return g_mfma_share; }
size_t mfma_sync_words(uint32_t nbatch, int num_cu) { return size_t(num_cu) * (nbatch + 1) * (1 + 8); }

// L2-coherent scalar read: SMEM is tracked by lgkmcnt, so polling a counter does not touch the vector-memory ring
__device__ __forceinline__ uint32_t sload_glc(const uint32_t *p) {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// hi/lo split of 8 consecutive f32 -> two packed bf16x8 (v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN)
__device__ __forceinline__ void split8(const float4 &a, const float4 &b, uint4 &hi, uint4 &lo) {
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned short h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        __bf16 hb = (__bf16)v[i];
        float hf = (float)hb;
        __bf16 lb = (__bf16)(v[i] - hf);
        h[i] = __builtin_bit_cast(unsigned short, hb);
        l[i] = __builtin_bit_cast(unsigned short, lb);
    }
    hi = make_uint4(h[0] | (uint32_t(h[1]) << 16), h[2] | (uint32_t(h[3]) << 16), h[4] | (uint32_t(h[5]) << 16),
                    h[6] | (uint32_t(h[7]) << 16));
    lo = make_uint4(l[0] | (uint32_t(l[1]) << 16), l[2] | (uint32_t(l[3]) << 16), l[4] | (uint32_t(l[5]) << 16),
                    l[6] | (uint32_t(l[7]) << 16));
}

// 8 consecutive columns [col, col+8) of a row of `dim` floats, zero past the row end (the padding columns)
__device__ __forceinline__ void load8_padded(const float *row, uint32_t dim, uint32_t col, float4 &a, float4 &b) {
    if ((dim & 3) == 0 && col + 8 <= dim) {
        a = *reinterpret_cast<const float4 *>(row + col);
        b = *reinterpret_cast<const float4 *>(row + col + 4);
        return;
    }
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = col + i < dim ? row[col + i] : 0.0f;
    a = make_float4(v[0], v[1], v[2], v[3]);
    b = make_float4(v[4], v[5], v[6], v[7]);
}

// ---------------------------------------------------------------------------------------------------
// rows [16*tile0, 16*tile1) of the row-major VecSet -> fragment-ordered split-bf16 mirror (rows >= n: zero)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_rows(const float *__restrict__ X, uint64_t n, uint32_t dim,
                                                   uint64_t tile0, uint64_t tile1, uint4 *__restrict__ T) {
    const uint32_t KB = ((dim + 63) & ~63u) / 32;
    uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;  // (tile, kb, lane)
    uint64_t total = (tile1 - tile0) * KB * 64;
    if (i >= total) return;
    uint32_t l = uint32_t(i & 63);
    uint64_t tk = i >> 6;
    uint32_t kb = uint32_t(tk % KB);
    uint64_t tile = tile0 + tk / KB;
    uint64_t row = tile * 16 + (l & 15);
    uint32_t col = kb * 32 + 8 * (l >> 4);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (row < n) load8_padded(X + row * dim, dim, col, a, b);
    uint4 hi, lo;
    split8(a, b, hi, lo);
    T[((tile * KB + kb) * 2 + 0) * 64 + l] = hi;
    T[((tile * KB + kb) * 2 + 1) * 64 + l] = lo;
}

void launch_tile_rows(const float *X, uint64_t n, uint32_t dim, uint64_t tile0, uint64_t tile1, float *T,
                      hipStream_t s) {
    if (tile1 <= tile0) return;
    uint64_t total = (tile1 - tile0) * (mfma_dim_pad(dim) / 32) * 64;
    hipLaunchKernelGGL(k_tile_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, n, dim, tile0, tile1,
                       reinterpret_cast<uint4 *>(T));
}

// Q [nq][dim] -> per batch of 32 queries a B-operand image [kb][half][hi|lo][lane] (queries >= nq: zero)
__global__ void k_mfma_pack_queries(const float *__restrict__ Q, uint32_t nq, uint32_t dim, uint32_t NH,
                                    uint4 *__restrict__ qfrag) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // (kb, half, lane)
    uint32_t KB = ((dim + 63) & ~63u) / 32;
    if (i >= KB * NH * 64) return;
    qfrag += uint64_t(blockIdx.y) * KB * NH * 128;
    uint32_t l = i & 63, h = (i >> 6) % NH, kb = (i >> 6) / NH;
    uint32_t q = blockIdx.y * 16 * NH + h * 16 + (l & 15);
    uint32_t c = kb * 32 + 8 * (l >> 4);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (q < nq) load8_padded(Q + size_t(q) * dim, dim, c, a, b);
    uint4 hi, lo;
    split8(a, b, hi, lo);
    qfrag[((kb * NH + h) * 2 + 0) * 64 + l] = hi;
    qfrag[((kb * NH + h) * 2 + 1) * 64 + l] = lo;
}

// images for batches of 16*NH queries (NH = 8: the 128-query groups of k_gemm.hip)
void launch_mfma_pack_queries_nh(const float *Q, uint32_t nq, uint32_t nq_cover, uint32_t dim, uint32_t NH, float *qfrag,
                                 hipStream_t s) {
    const uint32_t bq = 16 * NH;
    uint32_t total = (mfma_dim_pad(dim) / 32) * NH * 64;
    uint32_t nbatch = (std::max(nq, nq_cover) + bq - 1) / bq;  // queries in [nq, nq_cover) get zero images
    if (nbatch == 0) return;
    hipLaunchKernelGGL(k_mfma_pack_queries, dim3((total + 255) / 256, nbatch), dim3(256), 0, s, Q, nq, dim, NH,
                       reinterpret_cast<uint4 *>(qfrag));
}

void launch_mfma_pack_queries(const float *Q, uint32_t nq, uint32_t nq_cover, uint32_t dim, float *qfrag, hipStream_t s) {
    const uint32_t bq = mfma_batch(dim);
    VDB_REQUIRE(bq != 0, "flat_mfma: unsupported dim");
    launch_mfma_pack_queries_nh(Q, nq, nq_cover, dim, bq / 16, qfrag, s);
}

// ---------------------------------------------------------------------------------------------------
// 2 row tiles of 16 rows per wave item, ring of R = PD+1 k-blocks, 512 threads per (persistent) workgroup
// ---------------------------------------------------------------------------------------------------
// MODE_SAMPLE: visit every `item_step`-th item and write its keys densely (out[q*ld + j*32 + row_in_item],
//              +inf for rows >= n): a strided sample whose k'-th smallest key per query is an upper bound tau
//              of the k'-th smallest key over all rows.  blockIdx.y = query batch.
// MODE_FILTER: for every query batch in turn, visit every item and collect pair_key(key,row) of the keys
//              <= tau[q]; nothing else is written.  (Writing all B x N keys instead costs 18 % of the kernel
//              -- measured 0.768 ms vs 0.629 ms per pass -- although it is 3 % of the bytes, because the stores
//              are 64-B pieces scattered over 32 rows of the key matrix.)  ONE launch walks all batches: a
//              workgroup that has finished batch b starts batch b+1 at once (no grid-wide ramp-down / ramp-up
//              per 32 queries), and the next batch's Q image is prefetched into registers while the
//              workgroup's other waves finish, so the LDS refill is a register->LDS copy.
enum { MODE_SAMPLE = 0, MODE_FILTER = 1 };

struct MfmaArgs {
    const uint4 *XT;
    const uint4 *qfrag;  // [nbatch][KB*256]
    const float *xsq;
    uint64_t n;
    uint32_t dim, n_items, item_step, nbatch;
    // MODE_SAMPLE
    float *out;          // [nbatch*32][ld]
    uint64_t ld;
    // MODE_FILTER
    const float *tau;    // [nbatch*32]
    uint64_t *cand;      // [nbatch*32][cap]
    uint32_t *cnt;       // [nbatch*32]
    uint32_t cap;
    uint32_t debug;
    uint32_t *sync;      // MODE_FILTER, share > 1: arrival counters [groups][passes], zeroed before the launch
    uint32_t *isync;     // MODE_FILTER, share > 1: per-item arrival counters [groups*8 waves][passes], zeroed likewise
    uint32_t wgbuf;      // MODE_FILTER: entries of the per-workgroup LDS hit buffer (<= MFMA_WGBUF, set by the launcher)
    uint32_t share;      // MODE_FILTER: workgroups per XCD group that ride one HBM pass with different query batches
    int cosine;          // keys for DistanceAlgorithm::Cosine: -S/|x| (ranks like 1 - S/(|x||q|) for a fixed query)
};

template <int PD, int MODE, int NH>
__global__ __launch_bounds__(512, 1) void k_flat_mfma(MfmaArgs a) {
    constexpr int NT = 512, RT = MFMA_RT;
    constexpr uint32_t BQ = 16 * NH;  // queries per workgroup batch
    constexpr uint32_t NW = NT / 64;
    const uint4 *__restrict__ XT = a.XT;
    const float *__restrict__ xsq = a.xsq;
    const uint64_t n = a.n;
    const uint32_t dim = a.dim;
    const uint32_t n_visit = (a.n_items + a.item_step - 1) / a.item_step;  // items one pass visits
    extern __shared__ __attribute__((aligned(16))) uint4 qs[];  // [KB][2 halves][hi|lo][64], then the hit buffer
    const uint32_t KB = dim / 32;
    const uint32_t qn = KB * NH * 128;  // uint4 in one Q image
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per-workgroup hit buffer (MODE_FILTER): pairs that pass the threshold are parked in LDS and handed to the
    // global per-query candidate lists once per batch, when the workgroup has finished streaming.  Measured
    // alternatives: a global slot atomic per hit inside the item loop forces s_waitcnt vmcnt(0) (drains the
    // prefetch ring: 0.667 ms/pass instead of 0.636); per-wave buffers flushed at the end serialise ~2000
    // atomics per query counter in the kernel's tail (0.667 ms); flushing per 8 hits is worse (0.712 ms).
    // Here each workgroup reserves its range with ONE atomic per query (256 per counter per pass).
    uint64_t *hit_key = reinterpret_cast<uint64_t *>(qs + qn);
    uint32_t *hit_q = reinterpret_cast<uint32_t *>(hit_key + a.wgbuf);
    uint32_t *hit_n = hit_q + a.wgbuf;  // [0] entries, [1..32] per-query counts, [33..64] per-query bases

    const uint32_t r = lane & 15, g = lane >> 4;
    // XCD-shared passes (MODE_FILTER, share = S > 1): measured with tools/stream_probe.hip, S workgroups on the SAME
    // XCD that read the same stream at the same time are served once from HBM and S times from that XCD's L2
    // (S = 2: 0.643 ms for what one reader streams in 0.62 ms; on different XCDs it costs 2x).  Workgroups are
    // dealt round-robin over the 8 XCDs (block b -> XCD b % 8; placement affects speed only, never results), so
    // the S members of a group are the blocks with equal b % 8 and equal (b / 8) / S.  All members walk the same
    // items; member m serves query batches m, m + S, m + 2S, ...  One HBM pass then serves 32 * S queries.
    uint32_t S_ = 1, member = 0, group = blockIdx.x, n_groups = gridDim.x;
    if (MODE == MODE_FILTER && a.share > 1) {
        S_ = a.share;
        const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        member = slot % S_;
        group = (slot / S_) * 8 + xcd;
        n_groups = gridDim.x / S_;
    }
    const uint32_t stride = n_groups * NW;
    const uint32_t first = group * NW + wave;
    const uint32_t b_begin = MODE == MODE_SAMPLE ? blockIdx.y : member;
    const uint32_t b_end = MODE == MODE_SAMPLE ? blockIdx.y + 1 : a.nbatch;
    const uint32_t b_step = S_;

    // Q image staging through registers (<= 128 KB / 512 threads = 16 uint4 per thread)
    constexpr int QREG = 16;
    uint4 qreg[QREG];
    auto q_load = [&](uint32_t b) {
        const uint4 *src = a.qfrag + uint64_t(b) * qn;
#pragma unroll
        for (int j = 0; j < QREG; j++) {
            uint32_t i = j * NT + threadIdx.x;
            qreg[j] = i < qn ? src[i] : make_uint4(0, 0, 0, 0);
        }
    };
    auto q_store = [&]() {
#pragma unroll
        for (int j = 0; j < QREG; j++) {
            uint32_t i = j * NT + threadIdx.x;
            if (i < qn) qs[i] = qreg[j];
        }
    };

    constexpr int R = PD + 1;
    if (b_begin < b_end) q_load(b_begin);
    for (uint32_t b = b_begin; b < b_end; b += b_step) {
        q_store();
        if (MODE == MODE_FILTER && threadIdx.x < 80) hit_n[threadIdx.x] = 0;
        if (MODE == MODE_FILTER && S_ > 1 && a.sync && threadIdx.x == 0) {
            // rendezvous of the group's members before each shared pass: L2 sharing only works while they read
            // the same lines within a few microseconds of each other, and without it their start times drift
            // apart batch after batch.  Bounded spin: a member that is not resident (or already finished)
            // only costs speed, never correctness.
            uint32_t *ctr = a.sync + uint64_t(group) * ((a.nbatch + S_ - 1) / S_) + (b / S_);
            uint32_t want = b + S_ <= a.nbatch || a.nbatch % S_ == 0 ? S_ : a.nbatch % S_;  // members with a batch in this pass
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int spin = 0; spin < 4000; spin++) {
                if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
        float tau[NH];
#pragma unroll
        for (int h = 0; h < NH; h++) {
            tau[h] = MODE == MODE_FILTER ? a.tau[b * BQ + h * 16 + r] : 0.0f;
            if (MODE == MODE_FILTER && (a.debug & 1)) tau[h] = -INFINITY;  // timing experiment: no hits
        }

        // The X stream is one continuous sequence of (item, k-block) pairs per wave; a cursor runs PD k-blocks
        // ahead of the MFMAs and never stops at an item boundary, so every load is unconditional (the compiler
        // counts vmcnt instead of draining it) and HBM latency is covered across items.  Past the wave's last
        // item the cursor re-reads that item (L2 hits, results unused).
        uint32_t c_item = first, c_kb = 0;
        const uint4 *cp;  // this lane's 16 B of (item, tile 0, kb 0, hi); tile t is KB*128 uint4 further
        auto set_ptrs = [&](uint32_t v) {  // v = ordinal among the visited items
            if (v >= n_visit) v = n_visit - 1;
            cp = XT + uint64_t(v) * a.item_step * RT * KB * 128 + lane;
        };
        set_ptrs(c_item);
        // Register ring of R = PD+1 slots: k-block i is consumed from slot i%R while the loads for k-block
        // i+PD land in slot (i-1)%R, whose MFMAs were issued one step earlier.  With the loop unrolled by R
        // every slot keeps its registers across the back edge (no rotation copies, no vmcnt(0) drain).
        uint4 ring[R][RT][2];
        auto fetch = [&](uint4(&dst)[RT][2]) {
#pragma unroll
            for (int t = 0; t < RT; t++) {
                dst[t][0] = cp[((uint64_t(t) * KB + c_kb) * 2 + 0) * 64];
                dst[t][1] = cp[((uint64_t(t) * KB + c_kb) * 2 + 1) * 64];
            }
            c_kb++;
            if (c_kb == KB) {
                c_kb = 0;
                c_item += stride;
                set_ptrs(c_item);
            }
        };
#pragma unroll
        for (int p = 0; p < PD; p++) fetch(ring[p]);

        uint32_t pass_want = 1, items_started = 0;
        uint32_t *ictr = nullptr;
        if (MODE == MODE_FILTER && S_ > 1 && a.isync) {
            const uint32_t npass = (a.nbatch + S_ - 1) / S_;
            pass_want = b + S_ <= a.nbatch || a.nbatch % S_ == 0 ? S_ : a.nbatch % S_;
            ictr = a.isync + (uint64_t(__builtin_amdgcn_readfirstlane(group * NW + wave)) * npass + (b / S_));
        }
        for (uint32_t item = first; item < n_visit; item += stride) {
            const uint64_t row0 = uint64_t(item) * a.item_step * (16 * RT);
            if (MODE == MODE_FILTER && S_ > 1 && ictr && pass_want > 1) {
                // keep the group's members within a few microseconds of each other item by item: a line stays in
                // the XCD's L2 for ~5 us at this fill rate, and only a partner that arrives inside that window
                // is served from L2 (measured hit rate without this: 41.8 % of 50 %).  Arrival = fire-and-forget
                // atomic; the poll is a scalar (lgkmcnt) read, so the vector prefetch ring is never drained.
                items_started++;
                if (lane == 0) __hip_atomic_fetch_add(ictr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t target = items_started * pass_want;
                for (int spin = 0; spin < 48; spin++) {  // bounded: a missing partner costs speed, never results
                    if (sload_glc(ictr) >= target) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            // |x|^2 of the item's 32 rows through SCALAR loads (the address is wave-uniform): SMEM is tracked by
            // lgkmcnt, so reading it in the epilogue does not drain the vector-memory prefetch ring.  (As vector
            // loads the compiler sinks them next to their use and emits s_waitcnt vmcnt(1) there: one full HBM
            // round trip of the ring per item.)  constant address space + uniform address => s_load_dwordx8.
            typedef const __attribute__((address_space(4))) float *cfloat_p;
            cfloat_p xs_item = (cfloat_p)(xsq + uint64_t(__builtin_amdgcn_readfirstlane(item)) * a.item_step * (16 * RT));
            f32x4 acc[RT][NH];
#pragma unroll
            for (int t = 0; t < RT; t++)
#pragma unroll
                for (int h = 0; h < NH; h++) acc[t][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (uint32_t k0 = 0; k0 < KB; k0 += R) {
#pragma unroll
                for (int p = 0; p < R; p++) {
                    fetch(ring[(p + R - 1) % R]);
                    const uint32_t kb = k0 + p;
                    bf16x8 qh[NH], ql[NH];
#pragma unroll
                    for (int h = 0; h < NH; h++) {
                        qh[h] = __builtin_bit_cast(bf16x8, qs[((kb * NH + h) * 2 + 0) * 64 + lane]);
                        ql[h] = __builtin_bit_cast(bf16x8, qs[((kb * NH + h) * 2 + 1) * 64 + lane]);
                    }
#pragma unroll
                    for (int t = 0; t < RT; t++) {
                        const bf16x8 xh = __builtin_bit_cast(bf16x8, ring[p][t][0]);
                        const bf16x8 xl = __builtin_bit_cast(bf16x8, ring[p][t][1]);
#pragma unroll
                        for (int h = 0; h < NH; h++) {
                            acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, qh[h], acc[t][h], 0, 0, 0);
                            acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, ql[h], acc[t][h], 0, 0, 0);
                            acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, qh[h], acc[t][h], 0, 0, 0);
                        }
                    }
                }
            }
            // epilogue: lane holds rows rb..rb+3 of each row tile for queries r and 16+r
            float xsv[16 * RT];  // all norms of the item first (wide s_load), then per-lane selects on values
#pragma unroll
            for (int i = 0; i < 16 * RT; i++) xsv[i] = xs_item[i];
#pragma unroll
            for (int t = 0; t < RT; t++) {
                const uint64_t rb = row0 + t * 16 + 4 * g;
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    float key[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float x0 = xsv[t * 16 + 0 + e], x1 = xsv[t * 16 + 4 + e];
                        const float x2 = xsv[t * 16 + 8 + e], x3 = xsv[t * 16 + 12 + e];
                        const float xv = g == 0 ? x0 : (g == 1 ? x1 : (g == 2 ? x2 : x3));
                        if (a.cosine)  // zero-norm rows: S = 0 and the reference distance is exactly 1 -> key 0
                            key[e] = -acc[t][h][e] * (xv > 0.0f ? __frsqrt_rn(xv) : 0.0f);
                        else
                            key[e] = xv - 2.0f * acc[t][h][e];
                    }
                    if (MODE == MODE_SAMPLE) {
                        float4 kv;
                        kv.x = rb + 0 < n ? key[0] : INFINITY;
                        kv.y = rb + 1 < n ? key[1] : INFINITY;
                        kv.z = rb + 2 < n ? key[2] : INFINITY;
                        kv.w = rb + 3 < n ? key[3] : INFINITY;
                        const uint64_t col = uint64_t(item) * (16 * RT) + t * 16 + 4 * g;  // dense position in the sample
                        *reinterpret_cast<float4 *>(a.out + (uint64_t(b) * BQ + h * 16 + r) * a.ld + col) = kv;
                    } else {
                        const uint32_t q = h * 16 + r;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            if (key[e] <= tau[h] && rb + e < n) {  // rare: ~k' * sample step hits per query in total
                                uint32_t pos = atomicAdd(hit_n, 1u);  // LDS atomic: waits on lgkmcnt only
                                if (pos < a.wgbuf) {
                                    hit_key[pos] = pair_key(key[e], uint32_t(rb + e));
                                    hit_q[pos] = q;
                                } else {  // buffer full: mark the query as overflowed (-> exact fallback)
                                    atomicAdd(&a.cnt[b * BQ + q], a.cap + 1);
                                }
                            }
                        }
                    }
                }
            }
        }
        if (b + b_step < b_end) q_load(b + b_step);  // lands while the other waves finish and the hits are flushed
        __syncthreads();                   // every wave has left the item loop: qs and the hit buffer are quiescent
        if (MODE == MODE_FILTER) {
            uint32_t total = hit_n[0];
            if (total > a.wgbuf) total = a.wgbuf;
            constexpr uint32_t NJ = (MFMA_WGBUF + NT - 1) / NT;
            uint32_t rank[NJ];
#pragma unroll
            for (uint32_t j = 0; j < NJ; j++) {
                uint32_t i = j * NT + threadIdx.x;
                rank[j] = i < total ? atomicAdd(&hit_n[1 + hit_q[i]], 1u) : 0u;  // rank inside (workgroup, query)
            }
            __syncthreads();
            if (threadIdx.x < BQ && hit_n[1 + threadIdx.x] > 0)  // reserve this workgroup's range: one atomic per query
                hit_n[33 + threadIdx.x] = atomicAdd(&a.cnt[b * BQ + threadIdx.x], hit_n[1 + threadIdx.x]);
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < NJ; j++) {
                uint32_t i = j * NT + threadIdx.x;
                if (i < total) {
                    uint32_t q = hit_q[i];
                    uint32_t slot = hit_n[33 + q] + rank[j];
                    if (slot < a.cap) a.cand[(uint64_t(b) * BQ + q) * a.cap + slot] = hit_key[i];
                }
            }
            __syncthreads();
        }
    }
}

template <int PD, int MODE, int NH>
static void flat_mfma_launch(const MfmaArgs &a, int num_cu, hipStream_t s) {
    // LDS: Q image + hit buffer (12 B per entry) + counters; the hit buffer takes what the Q image leaves of 160 KB
    const size_t q_bytes = size_t(a.dim) * 16 * NH * sizeof(float), fixed = 80 * 4 + 16;
    VDB_REQUIRE(q_bytes + fixed + 64 * 12 <= 160 * 1024, "flat mfma: Q image does not fit LDS");
    uint32_t wgbuf = (uint32_t)std::min<size_t>(MFMA_WGBUF, ((160 * 1024 - q_bytes - fixed) / 12) & ~size_t(63));
    size_t lds = q_bytes + size_t(wgbuf) * 12 + fixed;
    func_max_lds(reinterpret_cast<const void *>(&k_flat_mfma<PD, MODE, NH>), int(160 * 1024));
    constexpr uint32_t NW = 8;
    uint32_t n_visit = (a.n_items + a.item_step - 1) / a.item_step;
    uint32_t grid = (uint32_t)num_cu;
    uint32_t need = (n_visit + NW - 1) / NW;
    MfmaArgs b = a;
    b.wgbuf = wgbuf;
    if (MODE == MODE_FILTER && b.share > 1) {
        // sharing needs whole XCD groups: grid a multiple of 8*share, every group with work, >= share batches
        uint32_t unit = 8 * b.share;
        grid = grid / unit * unit;
        if (grid == 0 || need < grid / b.share || b.nbatch < b.share) {
            b.share = 1;
            grid = (uint32_t)num_cu;
        }
    }
    if (b.share <= 1 && need < grid) grid = need;
    if (MODE == MODE_SAMPLE && b.nbatch > 0) {
        // every workgroup first loads its batch's whole Q image (up to 128 KB): with one item per wave that load IS the
        // pass (measured 157 us for 32 batches x 33 workgroups at 125k rows).  One chip-full of workgroups in total,
        // each walking several sampled items, reads each image ~8 times instead of ~33.
        uint32_t per_batch = ((uint32_t)num_cu + b.nbatch - 1) / b.nbatch;
        if (per_batch < grid) grid = per_batch;
    }
    if (grid == 0 || b.nbatch == 0) return;
    dim3 g(grid, MODE == MODE_SAMPLE ? b.nbatch : 1);
    hipLaunchKernelGGL((k_flat_mfma<PD, MODE, NH>), g, dim3(512), lds, s, b);
    VDB_HIP(hipGetLastError());
}

template <int MODE, int NH>
static void flat_mfma_dispatch_nh(const MfmaArgs &a, int num_cu, hipStream_t s) {
    uint32_t KB = a.dim / 32;  // the ring size R = PD+1 must divide KB
    if (KB % 6 == 0 && (g_mfma_variant & 15) == 2)
        flat_mfma_launch<2, MODE, NH>(a, num_cu, s);
    else if (KB % 6 == 0)
        flat_mfma_launch<5, MODE, NH>(a, num_cu, s);
    else if (KB % 4 == 0)
        flat_mfma_launch<3, MODE, NH>(a, num_cu, s);
    else
        flat_mfma_launch<1, MODE, NH>(a, num_cu, s);
}
template <int MODE>
static void flat_mfma_dispatch(const MfmaArgs &a, int num_cu, hipStream_t s) {
    if (a.dim <= 1024)
        flat_mfma_dispatch_nh<MODE, 2>(a, num_cu, s);
    else
        flat_mfma_dispatch_nh<MODE, 1>(a, num_cu, s);
}

uint32_t mfma_num_items(uint64_t n) { return (uint32_t)((n + 16 * MFMA_RT - 1) / (16 * MFMA_RT)); }
// Threshold sample: every `step`-th item (32 rows) is scored and tau = the `rank`-th smallest sampled key.  Expected
// rows with key <= tau over the whole shard: step * rank, aimed at ~1024 per query (candidate lists hold 8192; every
// hit costs the filter kernel's epilogue a divergent LDS append: 2048 hits per query were 0.37 of 3.0 ms).
//  - rank = k': tau is an upper bound of the k'-th smallest key of ALL rows (the sample is a subset), so the filter
//    pass is guaranteed to return at least k' rows;
//  - when the shard is large enough, rank r = max(8, k'/8) on a thinner sample (>= 64 items): less sample-pass and
//    selection work for the same expected hit count (target = max(1024, 4 k')).  The k'-hit guarantee becomes a probability: hits / step is
//    Gamma(r)-distributed, P[hits < k'] = P[Gamma(r) < k'/step] (r = 8, k'/step = 0.5: 6e-8), so it is CHECKED: a
//    query with fewer than k' hits is redone like any other uncertified query.
static std::atomic<int> g_sample_thin{1};
void mfma_set_sample_thin(int v) { g_sample_thin = v; }
void mfma_sample_plan(uint64_t n, uint32_t kprime, uint32_t *step_out, uint32_t *rank_out, uint32_t target_floor) {
    const uint32_t items = mfma_num_items(n), target = std::max<uint32_t>(target_floor, 4 * kprime);
    uint32_t rank = kprime < 1 ? 1 : kprime;
    uint32_t step = items / 256;  // >= 256 sampled items (8192 rows) when the shard has them
    step = step < 1 ? 1 : step;
    const uint32_t cap = target / rank < 1 ? 1 : target / rank;
    if (step > cap) step = cap;
    if (g_sample_thin) {  // thinner sample (>= 64 items), lower rank: same expected hits, less sample work
        for (uint32_t div = 8; div >= 2; div /= 2) {
            const uint32_t r = kprime / div < 8 ? 8 : kprime / div;
            uint32_t st = items / 64;
            if (st > target / r) st = target / r;
            // P[hits < k'] = P[Gamma(r) < k'/st] must be negligible: k'/st <= r/8 (r < 16: 1e-5), r/4 (r < 32: 1e-6), r/3
            const double thr = r < 16 ? r / 8.0 : (r < 32 ? r / 4.0 : r / 3.0);
            if (r < rank && st >= 2 * step && double(kprime) <= thr * st * 1.1) {
                rank = r;
                step = st;
                break;
            }
        }
    }
    *step_out = step;
    *rank_out = rank;
}
uint64_t mfma_sample_rows(uint64_t n, uint32_t step) {
    uint32_t items = mfma_num_items(n);
    return uint64_t((items + step - 1) / step) * 16 * MFMA_RT;
}

static MfmaArgs mfma_args(const float *XT, uint64_t n, uint32_t dim, const float *qfrag, uint32_t nbatch,
                          const float *xsq, int cosine) {
    VDB_REQUIRE(mfma_supported(dim), "flat_mfma: dim (padded to a multiple of 64) must be <= 2048");
    VDB_REQUIRE(n < (1ull << 32), "flat_mfma: too many rows for one shard");
    MfmaArgs a{};
    a.XT = reinterpret_cast<const uint4 *>(XT);
    a.qfrag = reinterpret_cast<const uint4 *>(qfrag);
    a.xsq = xsq;
    a.n = n;
    a.dim = mfma_dim_pad(dim);  // the kernel only sees the padded mirror / images
    a.n_items = mfma_num_items(n);
    a.nbatch = nbatch;
    a.cosine = cosine;
    return a;
}

void launch_flat_mfma_sample(const float *XT, uint64_t n, uint32_t dim, const float *qfrag, uint32_t nbatch,
                             const float *xsq, int cosine, uint32_t step, float *out, uint64_t ld, int num_cu,
                             hipStream_t s) {
    if (n == 0 || nbatch == 0) return;
    MfmaArgs a = mfma_args(XT, n, dim, qfrag, nbatch, xsq, cosine);
    VDB_REQUIRE(step >= 1 && (ld & 3) == 0 && ld >= mfma_sample_rows(n, step), "flat_mfma: ld must cover the sample");
    VDB_REQUIRE(nbatch <= 65535, "flat_mfma: too many query batches");
    a.item_step = step;
    a.out = out;
    a.ld = ld;
    flat_mfma_dispatch<MODE_SAMPLE>(a, num_cu, s);
}

void launch_flat_mfma_filter(const float *XT, uint64_t n, uint32_t dim, const float *qfrag, uint32_t nbatch,
                             const float *xsq, int cosine, const float *tau, uint64_t *cand, uint32_t *cnt,
                             uint32_t cap, uint32_t *sync, int num_cu, hipStream_t s) {
    if (n == 0 || nbatch == 0) return;
    MfmaArgs a = mfma_args(XT, n, dim, qfrag, nbatch, xsq, cosine);
    a.item_step = 1;
    a.tau = tau;
    a.cand = cand;
    a.cnt = cnt;
    a.cap = cap;
    a.debug = g_mfma_variant >= 16 ? (g_mfma_variant >> 4) : 0;
    a.share = g_mfma_share;
    a.sync = sync;
    a.isync = (g_mfma_variant & 32) ? nullptr : sync + size_t(num_cu) * (nbatch + 1);
    flat_mfma_dispatch<MODE_FILTER>(a, num_cu, s);
}

}  // namespace vdb
