// k_small.hip -- FlatIndex::knn (flat_index.rs:48-57) on a SMALL table for a FEW queries in ONE launch: the shape of
// db.search() (pyo3/mod.rs:199-214 -> one query against a table of a few thousand rows).  The general exact path costs
// five launches (query norm, scan, two select levels, finalize) and a lane that walks a whole 3 840-B row behind
// uncoalesced loads; on a 1000-row table that was 0.14 ms per call against 0.024 ms for the 16-core CPU.
//
// Here a workgroup owns R = 16 / 32 / 64 rows (the host picks R so that a table of up to 16 384 rows spreads over as many CUs as
// it has workgroups to give) and walks them in chunks of 8192 / R columns:
//   * all 256 threads fetch the chunk with whole 128-B lines, two chunks ahead in registers (64 KB in flight per workgroup);
//   * the thread that holds a piece turns it into PRODUCTS -- (x - q)^2 or x * q, separately rounded exactly as
//     distance/mod.rs:72-77 rounds them, any thread may do that -- and parks them in LDS transposed (row stride +4 floats:
//     conflict-free ds_read_b128 for the fold);
//   * lane r of wave 0 is left with nothing but the reference's strict left fold over row r's products: 4 dependent adds per
//     ds_read_b128, ~1.9 us for 960 columns -- the floor of a strict fold;
//   * wave 0 sorts its (distance, index) pair keys across the lanes, writes the first min(k, R) of them (later ones cannot be among
//     the k smallest; stored rank-major), and the LAST workgroup to arrive (one counter per query) selects the k smallest of all
//     lists and writes the ids, distances and count -- to device memory or straight into the caller's pinned host block
//     (the lists cross XCDs as agent-scope atomic stores / loads: no L2 write-back or invalidation);
// Cosine: |q|^2 is the strict fold of q_j * q_j (distance/mod.rs:60-69 recomputes both norms per pair; the row norms are the
// cached folds of d_sq), done once per workgroup by one lane of wave 3 over products all threads prepared.
#include <algorithm>

#include "common.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace vdb {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int FS_PIECES = 8;       // 16-B pieces per thread and chunk: 256 threads x 8 = R rows x (2048 / R) pieces
constexpr int FS_TILE_FLOATS = 8192;  // products per chunk

template <int R>
struct SmallGeom {
    static constexpr int CW = FS_TILE_FLOATS / R;  // columns per chunk: 512 / 256 / 128
    static constexpr int LDT = CW + 4;             // row stride in LDS (floats): 4 mod 64 -> 16 lanes hit 16 distinct bank quads
    static constexpr int PPR = CW / 4;             // pieces per row and chunk
};

template <int R, bool COS>
__global__ __launch_bounds__(256) void k_flat_small(FlatSmallArgs a) {
    using G = SmallGeom<R>;
    extern __shared__ __attribute__((aligned(16))) float fs_dyn[];  // [dim_pad] query, then (COS) [dim_pad] q*q products
    __shared__ __attribute__((aligned(16))) float tile[2][R * G::LDT];
    __shared__ uint64_t sbest[4][64];
    __shared__ float s_qsq;
    __shared__ uint32_t s_last;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t qi = blockIdx.y, wg = blockIdx.x, nwg = gridDim.x;
    const uint32_t dim = a.dim, dim_pad = (dim + 3) & ~3u;
    const uint64_t row0 = uint64_t(wg) * R;
    float *qs = fs_dyn, *qq = fs_dyn + dim_pad;
    const float *Q = a.Q + uint64_t(qi) * dim;
    for (uint32_t j = tid; j < dim; j += 256) {
        const float v = Q[j];
        qs[j] = v;
        if (COS) qq[j] = v * v;
    }
    const uint32_t nchunk = (dim + G::CW - 1) / G::CW;

    f4 stage[2][FS_PIECES];
    auto load_chunk = [&](uint32_t c, f4 (&st)[FS_PIECES]) {
#pragma unroll
        for (int i = 0; i < FS_PIECES; i++) {
            const uint32_t f = i * 256 + tid, r = f / G::PPR, p = f % G::PPR;
            uint64_t row = row0 + r;
            if (row >= a.n) row = a.n - 1;
            const uint32_t col = c * G::CW + p * 4;
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (col < dim) v = *reinterpret_cast<const f4 *>(a.X + row * dim + col);  // (dim % 4 == 0: a piece is inside the row or outside)
            st[i] = v;
        }
    };
    auto store_chunk = [&](uint32_t c, int buf, const f4 (&st)[FS_PIECES]) {
#pragma unroll
        for (int i = 0; i < FS_PIECES; i++) {
            const uint32_t f = i * 256 + tid, r = f / G::PPR, p = f % G::PPR;
            const uint32_t col = c * G::CW + p * 4;
            f4 pr = {0.f, 0.f, 0.f, 0.f};
            if (col < dim) {
                const f4 q4 = *reinterpret_cast<const f4 *>(qs + col);
                if (COS) {
                    pr = st[i] * q4;
                } else {
                    const f4 df = st[i] - q4;  // (x-q)^2 == (q-x)^2 exactly: the argument order of flat_index.rs:52 is immaterial
                    pr = df * df;
                }
            }
            *reinterpret_cast<f4 *>(&tile[buf][r * G::LDT + p * 4]) = pr;
        }
    };

    load_chunk(0, stage[0]);
    if (nchunk > 1) load_chunk(1, stage[1]);
    __syncthreads();  // qs / qq complete
    store_chunk(0, 0, stage[0]);
    __syncthreads();
    float acc = 0.0f, qacc = 0.0f;
    for (uint32_t c = 0; c < nchunk; c++) {
        const int buf = c & 1;
        // registers: chunk c+1 sits in stage[(c+1)&1]; refill the slot chunk c left with chunk c+2
        if (c + 2 < nchunk) {
            if (buf == 0)
                load_chunk(c + 2, stage[0]);
            else
                load_chunk(c + 2, stage[1]);
        }
        uint32_t cols = dim - c * G::CW;
        if (cols > (uint32_t)G::CW) cols = G::CW;
        // The fold is a chain of dependent adds (4.8 ns each); an LDS round trip in front of every four of them would triple it.
        // 32 products are read ahead of the adds that consume them (two register sets); columns past `cols` hold +0.0 in the
        // tile (store_chunk) and adding +0.0 is exact (the running sum is never -0.0), so the trip count is rounded up to 32.
        if (wave == 0 && lane < R) {
            const f4 *tr = reinterpret_cast<const f4 *>(&tile[buf][lane * G::LDT]);
            const uint32_t nv = (cols + 31) / 32 * 8;  // 16-B pieces, in blocks of 8 (CW % 32 == 0: inside the row)
            f4 cur[8], nxt[8];
#pragma unroll
            for (int u = 0; u < 8; u++) cur[u] = tr[u];
            for (uint32_t j0 = 0; j0 < nv; j0 += 8) {
                const uint32_t jn = j0 + 8 < nv ? j0 + 8 : j0;  // (the last block re-reads itself: no branch around the loads)
#pragma unroll
                for (int u = 0; u < 8; u++) nxt[u] = tr[jn + u];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    acc = acc + cur[u].x;
                    acc = acc + cur[u].y;
                    acc = acc + cur[u].z;
                    acc = acc + cur[u].w;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) cur[u] = nxt[u];
            }
        }
        if (COS && tid == 192) {  // |q|^2 = strict fold of q_j * q_j (distance/mod.rs:72-74 over q, q), this chunk's share: as long as wave 0's fold
            const f4 *qc = reinterpret_cast<const f4 *>(qq + c * G::CW);
            const uint32_t nv = cols / 4;
            for (uint32_t j0 = 0; j0 < nv; j0 += 8) {
                f4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = j0 + u < nv ? qc[j0 + u] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    qacc = qacc + v[u].x;
                    qacc = qacc + v[u].y;
                    qacc = qacc + v[u].z;
                    qacc = qacc + v[u].w;
                }
            }
            if (c + 1 == nchunk) s_qsq = qacc;
        }
        if (c + 1 < nchunk) {
            if (buf == 0)
                store_chunk(c + 1, 1, stage[1]);
            else
                store_chunk(c + 1, 0, stage[0]);
        }
        __syncthreads();
    }
    // pair keys of this workgroup's rows, ascending across wave 0; the first min(ksel, R) go to the query's list
    const uint32_t kw = a.ksel < (uint32_t)R ? a.ksel : (uint32_t)R;
    uint64_t *part = a.part + uint64_t(qi) * nwg * kw;
    if (wave == 0) {
        uint64_t key = PAIR_NONE;
        const uint64_t row = row0 + lane;
        if (lane < R && row < a.n) {
            float d = acc;
            if (COS) {
                const float den = fmaxf(sqrtf(s_qsq) * sqrtf(a.xsq[row]), 1e-10f);  // distance/mod.rs:60-69
                const float r = acc / den;
                d = 1.0f - r;
            }
            key = pair_key(d, (uint32_t)row);
        }
        key = sort64(key, lane);
        // rank-major: the j-th smallest keys of all workgroups sit together, the heads first -- after the heads the merge's
        // running k-th best is close to final and the batches of later ranks are skipped wholesale
        // The lists travel between XCDs, whose L2s are not coherent with each other.  No fences: a release fence writes this
        // XCD's L2 back and an acquire fence invalidates it -- in every workgroup, under the row stream of its neighbours
        // (measured: 8 queries over 16 000 rows 206 us with __threadfence() on both sides, 122 us without any).  The keys go
        // out as agent-scope atomic stores (written through to the coherence point), the arrival is counted once they are
        // acknowledged (vmcnt), and the last workgroup reads them back with agent-scope atomic loads.
        if (lane < kw) __hip_atomic_store(&part[uint64_t(lane) * nwg + wg], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) s_last = atomicAdd(&a.counter[qi], 1u) == nwg - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;  // block-uniform
    const uint64_t best = block_top64<true>(part, nwg * kw, sbest, a.ksel);
    if (wave != 0) return;
    const uint64_t cnt = a.n < a.ksel ? a.n : a.ksel;
    if (lane < a.kstride) {
        const bool ok = lane < cnt && best != PAIR_NONE;
        a.out_idx[uint64_t(qi) * a.kstride + lane] = ok ? uint64_t(uint32_t(best)) + a.id_offset : 0ull;
        a.out_dist[uint64_t(qi) * a.kstride + lane] = ok ? f32_from_orderable(uint32_t(best >> 32)) : 0.0f;
    }
    if (lane == 0) {
        a.out_count[qi] = cnt;
        a.counter[qi] = 0;  // ready for the next launch on this workspace (stream order)
    }
}

template <int R>
void launch_r(const FlatSmallArgs &a, uint32_t nq, hipStream_t s) {
    const uint32_t nwg = (uint32_t)((a.n + R - 1) / R), dim_pad = (a.dim + 3) & ~3u;
    const bool cos = a.metric == MET_COSINE;
    const size_t lds = size_t(dim_pad) * sizeof(float) * (cos ? 2 : 1);
    // 68 KB of tiles + the query image(s): above the 64 KB a launch gets without asking
    if (cos) {
        func_max_lds(reinterpret_cast<const void *>(&k_flat_small<R, true>), 64 * 1024);
        hipLaunchKernelGGL((k_flat_small<R, true>), dim3(nwg, nq), dim3(256), lds, s, a);
    } else {
        func_max_lds(reinterpret_cast<const void *>(&k_flat_small<R, false>), 64 * 1024);
        hipLaunchKernelGGL((k_flat_small<R, false>), dim3(nwg, nq), dim3(256), lds, s, a);
    }
}

}  // namespace

uint32_t flat_small_rows_per_wg(uint64_t n, int num_cu) {
    // as many workgroups as the table has rows to give them, up to ~2 per CU
    if (n <= uint64_t(num_cu) * 2 * 16) return 16;
    if (n <= uint64_t(num_cu) * 2 * 32) return 32;
    return 64;
}
bool flat_small_supported(uint64_t n, uint32_t dim, uint64_t nq, uint64_t k) {
    return n >= 1 && n <= (1u << 20) && (dim & 3) == 0 && dim >= 4 && dim <= 8192 && nq >= 1 && nq <= 64 && k >= 1 && k <= 64;
}
size_t flat_small_part_keys(uint64_t n, uint64_t nq, uint32_t ksel, int num_cu) {
    const uint32_t R = flat_small_rows_per_wg(n, num_cu);
    return size_t(nq) * ((n + R - 1) / R) * std::min<uint32_t>(ksel, R);
}
void launch_flat_small(const FlatSmallArgs &a, uint32_t nq, int num_cu, hipStream_t s) {
    if (nq == 0) return;
    switch (flat_small_rows_per_wg(a.n, num_cu)) {
        case 16: launch_r<16>(a, nq, s); break;
        case 32: launch_r<32>(a, nq, s); break;
        default: launch_r<64>(a, nq, s); break;
    }
    VDB_HIP(hipGetLastError());
}

}  // namespace vdb
