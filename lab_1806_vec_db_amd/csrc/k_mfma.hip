// k_mfma.hip -- batched Flat brute force on the gfx950 matrix cores.
//
// FlatIndex::knn (flat_index.rs:48-57) is called once per query in the reference.  Batched over
// B = 32 queries the distance computation is a dense contraction S = X * Q^T, streamed once over
// the corpus per 32 queries: 3.84 GB of HBM reads per pass for Gist1M instead of 32 x 3.84 GB.
// This kernel produces APPROXIMATE ranking keys  key(r,b) = |x_r|^2 - 2 * S(r,b)  (the expanded
// form of distance/mod.rs:54-57 without the per-query constant); the keys only build a shortlist.
// The exact, reference-order distances of the shortlist are recomputed by k_rerank (k_exact.hip)
// and the shortlist is certified (k_certify); the results that leave the library are bit-identical
// to the reference's strict f32 fold.
//
// Mapping (v_mfma_f32_16x16x4_f32, exact f32 fma chain, 32 cycles per instruction per SIMD):
//   A[i][k] : lane l supplies X[row0 + (l&15)][c],  k-slot = l>>4
//   B[k][j] : lane l supplies Q[qbase + (l&15)][c], same k-slot
//   D[i][j] : lane l holds rows 4*(l>>4)+{0..3}, query (l&15)
// A lane loads one float4 = columns 16*s + 4*(l>>4) + {0..3} of its row per step s and spends its
// four elements on four MFMAs, so the four lanes of a row read 64 contiguous bytes per step and a
// 128-B line is consumed in two steps.  The k order inside the contraction is permuted relative to
// memory order; A and B use the same permutation, and the sum is an approximation anyway.
//
// Q (32 x dim f32 = 120 KB for dim 960) lives in LDS as a fragment-ordered image so that every
// B fragment is one lane-linear, conflict-free ds_read_b128.  X never touches LDS: it is read once,
// by exactly one wave, straight into VGPRs with a PD-step deep register ring (GEMV regime).
#include "common.hpp"
#include "kernels.hpp"

namespace vdb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

size_t mfma_qfrag_floats(uint32_t dim) { return size_t(dim) * MFMA_B; }
bool mfma_supported(uint32_t dim) { return dim >= 32 && (dim % 32) == 0 && size_t(dim) * MFMA_B * 4 <= 150 * 1024; }

__global__ void k_mfma_pack_queries(const float *__restrict__ Q, uint32_t nq, uint32_t dim,
                                    float4 *__restrict__ qfrag) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t S = dim / 16;
    if (i >= S * 128) return;
    uint32_t l = i & 63, h = (i >> 6) & 1, s = i >> 7;
    uint32_t q = h * 16 + (l & 15);
    uint32_t c = s * 16 + 4 * (l >> 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < nq) v = *reinterpret_cast<const float4 *>(Q + size_t(q) * dim + c);
    qfrag[i] = v;
}

void launch_mfma_pack_queries(const float *Q, uint32_t nq, uint32_t dim, float *qfrag, hipStream_t s) {
    uint32_t total = (dim / 16) * 128;
    hipLaunchKernelGGL(k_mfma_pack_queries, dim3((total + 255) / 256), dim3(256), 0, s, Q, nq, dim,
                       reinterpret_cast<float4 *>(qfrag));
}

template <int RT, int PD, int NT>
__global__ __launch_bounds__(NT, 1) void k_flat_mfma(const float *__restrict__ X, uint64_t n, uint32_t dim,
                                                     const float4 *__restrict__ qfrag,
                                                     const float *__restrict__ xsq, float *__restrict__ out,
                                                     uint64_t ld, uint32_t n_items) {
    extern __shared__ __attribute__((aligned(16))) float4 qs[];  // [S][2][64]
    const uint32_t S = dim / 16;
    for (uint32_t i = threadIdx.x; i < S * 128; i += NT) qs[i] = qfrag[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr uint32_t NW = NT / 64;
    const uint32_t r = lane & 15, g = lane >> 4;
    const uint32_t stride = gridDim.x * NW;
    const uint32_t first = blockIdx.x * NW + wave;
    if (first >= n_items) return;

    // The X stream is one continuous sequence of (item, step) pairs per wave; a cursor runs PD steps
    // ahead of the MFMAs and never stops at an item boundary, so every load is unconditional (the
    // compiler can then count vmcnt instead of draining it) and HBM latency is covered across items.
    // Past the wave's last item the cursor re-reads that item (L2 hits, results unused).
    uint32_t c_item = first, c_s = 0;
    const float4 *cp[RT];
    auto set_ptrs = [&](uint32_t item) {
        if (item >= n_items) item = n_items - 1;
#pragma unroll
        for (int t = 0; t < RT; t++) {
            uint64_t row = uint64_t(item) * (16 * RT) + t * 16 + r;
            if (row >= n) row = n - 1;  // tail: read a valid row, the result is never stored
            cp[t] = reinterpret_cast<const float4 *>(X + row * dim) + g;
        }
    };
    set_ptrs(c_item);
    // Register ring of R = PD+1 slots: step i consumes slot i%R while the load for step i+PD lands in
    // slot (i-1)%R, the slot whose MFMAs were issued one step earlier.  With the loop unrolled by R
    // every slot keeps its registers across the back edge (no rotation copies, no vmcnt(0) drain).
    constexpr int R = PD + 1;
    float4 ring[R][RT];
    auto fetch = [&](float4(&dst)[RT]) {
#pragma unroll
        for (int t = 0; t < RT; t++) dst[t] = cp[t][c_s * 4];
        c_s++;
        if (c_s == S) {
            c_s = 0;
            c_item += stride;
            set_ptrs(c_item);
        }
    };
#pragma unroll
    for (int p = 0; p < PD; p++) fetch(ring[p]);

    float4 bfr[2][2];  // B fragments, double-buffered one step ahead (R is even: parity is static)
    bfr[0][0] = qs[lane];
    bfr[0][1] = qs[64 + lane];
    for (uint32_t item = first; item < n_items; item += stride) {
        const uint64_t row0 = uint64_t(item) * (16 * RT);
        f32x4 acc[RT][2];
#pragma unroll
        for (int t = 0; t < RT; t++) {
            acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (uint32_t s0 = 0; s0 < S; s0 += R) {
#pragma unroll
            for (int p = 0; p < R; p++) {
                fetch(ring[(p + R - 1) % R]);
                const uint32_t sn = (s0 + p + 1 == S) ? 0 : s0 + p + 1;
                bfr[(p + 1) & 1][0] = qs[(sn * 2 + 0) * 64 + lane];
                bfr[(p + 1) & 1][1] = qs[(sn * 2 + 1) * 64 + lane];
                const float4 b0 = bfr[p & 1][0], b1 = bfr[p & 1][1];
#pragma unroll
                for (int t = 0; t < RT; t++) {
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[p][t].x, b0.x, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[p][t].x, b1.x, acc[t][1], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < RT; t++) {
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[p][t].y, b0.y, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[p][t].y, b1.y, acc[t][1], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < RT; t++) {
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[p][t].z, b0.z, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[p][t].z, b1.z, acc[t][1], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < RT; t++) {
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[p][t].w, b0.w, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[p][t].w, b1.w, acc[t][1], 0, 0, 0);
                }
            }
        }
        // epilogue: lane holds rows rb..rb+3 of each row tile for queries r and 16+r
#pragma unroll
        for (int t = 0; t < RT; t++) {
            const uint64_t rb = row0 + t * 16 + 4 * g;
            if (rb + 3 < n) {
                const float4 xs = *reinterpret_cast<const float4 *>(xsq + rb);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    float4 key;
                    key.x = xs.x - 2.0f * acc[t][h][0];
                    key.y = xs.y - 2.0f * acc[t][h][1];
                    key.z = xs.z - 2.0f * acc[t][h][2];
                    key.w = xs.w - 2.0f * acc[t][h][3];
                    *reinterpret_cast<float4 *>(out + uint64_t(h * 16 + r) * ld + rb) = key;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (rb + e < n) {
                        float xs = xsq[rb + e];
                        out[uint64_t(r) * ld + rb + e] = xs - 2.0f * acc[t][0][e];
                        out[uint64_t(16 + r) * ld + rb + e] = xs - 2.0f * acc[t][1][e];
                    }
                }
            }
        }
    }
}

template <int RT, int PD, int NT>
static void flat_mfma_launch(const float *X, uint64_t n, uint32_t dim, const float *qfrag, const float *xsq,
                             float *out, uint64_t ld, int num_cu, hipStream_t s) {
    size_t lds = size_t(dim) * MFMA_B * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        VDB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_flat_mfma<RT, PD, NT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    constexpr uint32_t NW = NT / 64;
    uint64_t items = (n + 16 * RT - 1) / (16 * RT);
    VDB_REQUIRE(items < (1ull << 31), "flat_mfma: too many rows for one shard");
    uint32_t grid = (uint32_t)num_cu;
    uint64_t need = (items + NW - 1) / NW;
    if (need < grid) grid = (uint32_t)need;
    if (grid == 0) return;
    hipLaunchKernelGGL((k_flat_mfma<RT, PD, NT>), dim3(grid), dim3(NT), lds, s, X, n, dim,
                       reinterpret_cast<const float4 *>(qfrag), xsq, out, ld, (uint32_t)items);
}

void launch_flat_mfma(const float *X, uint64_t n, uint32_t dim, const float *qfrag, const float *xsq, float *out,
                      uint64_t ld, int num_cu, hipStream_t s) {
    VDB_REQUIRE(mfma_supported(dim), "flat_mfma: dim must be a multiple of 32 with 32*dim*4 <= 150 KiB");
    VDB_REQUIRE((ld & 3) == 0, "flat_mfma: ld must be a multiple of 4");
    if (n == 0) return;
    uint32_t S = dim / 16;
    // ring size R = PD+1 must be even and divide S = dim/16
    if (S % 6 == 0)
        flat_mfma_launch<2, 5, 512>(X, n, dim, qfrag, xsq, out, ld, num_cu, s);
    else if (S % 4 == 0)
        flat_mfma_launch<2, 3, 512>(X, n, dim, qfrag, xsq, out, ld, num_cu, s);
    else
        flat_mfma_launch<2, 1, 512>(X, n, dim, qfrag, xsq, out, ld, num_cu, s);
}

}  // namespace vdb
