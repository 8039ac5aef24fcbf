// k_half.hip -- operands of k_flat_gemm<.., GEMM_F16> (k_gemm.hip): the scaled fp16 mirror of the rows, the scaled
// fp16 query images, and the MEASURED rounding error of both.
//
// Rows are stored as fp16(x * sx), queries as fp16(q * sq[q]); sx and sq[q] are powers of two chosen from the largest
// row norm / the query norm so that no element can overflow (|x_i| <= |x| < 2^c  =>  |x_i * 2^(13-c)| < 2^13), so the
// products are exact in f32 and S = (sum of products) / (sx * sq) differs from x.q only by
//     | x~.q~ - x.q |  <=  |dx| |q| + |x| |dq| + |dx| |dq|,      dx = x - x~/sx,  dq = q - q~/sq      (Cauchy-Schwarz)
// plus the f32 accumulation error that the split-bf16 kernels have as well.  |dx| is not bounded by the format's
// worst case (2^-11 |x|, and an absolute term for fp16 subnormals) but measured: k_row_split_err keeps the maximum
// of |dx_r|^2 and of |dx_r|^2 / |x_r|^2 over all rows ever tiled, k_query_prep_h returns |dq|^2 per query.  Measured
// values are ~2.4x smaller than the worst case and cover subnormals and flushed elements by construction.  These
// numbers go into the certification bound of k_flat_finish (k_exact.hip); the shortlist is re-ranked with the
// reference's strict-order f32 arithmetic as always, so the coarser keys cost certifications, never results.
#include "common.hpp"
#include "kernels.hpp"

namespace vdb {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint4 pack8h(const float4 &a, const float4 &b, float scale) {
    f16x8 h = {(_Float16)(a.x * scale), (_Float16)(a.y * scale), (_Float16)(a.z * scale), (_Float16)(a.w * scale),
               (_Float16)(b.x * scale), (_Float16)(b.y * scale), (_Float16)(b.z * scale), (_Float16)(b.w * scale)};
    return __builtin_bit_cast(uint4, h);  // v_cvt_f16_f32: round to nearest even
}

__device__ __forceinline__ void load8_pad(const float *row, uint32_t dim, uint32_t col, float4 &a, float4 &b) {
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

// rows [16*tile0, 16*tile1) -> T[(tile*KB32 + kb32)*64 + lane] = 8 fp16: row 16*tile + (lane & 15), columns
// 32*kb32 + 8*(lane >> 4) + j -- the A operand of v_mfma_f32_16x16x32_f16; a 64-column k-block of k_flat_gemm is two
// consecutive 1-KB fragments, exactly like the [hi|lo] pair of the split-bf16 mirror.  Rows >= n: zero.
__global__ __launch_bounds__(256) void k_tile_rows_h(const float *__restrict__ X, uint64_t n, uint32_t dim,
                                                     uint64_t tile0, uint64_t tile1, float sx, uint4 *__restrict__ T) {
    const uint32_t KB = ((dim + 63) & ~63u) / 32;
    uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;  // (tile, kb32, lane)
    uint64_t total = (tile1 - tile0) * KB * 64;
    if (i >= total) return;
    uint32_t l = uint32_t(i & 63);
    uint64_t tk = i >> 6;
    uint32_t kb = uint32_t(tk % KB);
    uint64_t tile = tile0 + tk / KB;
    uint64_t row = tile * 16 + (l & 15);
    uint32_t col = kb * 32 + 8 * (l >> 4);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (row < n) load8_pad(X + row * dim, dim, col, a, b);
    T[(tile * KB + kb) * 64 + l] = pack8h(a, b, sx);
}
void launch_tile_rows_h(const float *X, uint64_t n, uint32_t dim, uint64_t tile0, uint64_t tile1, float sx, void *T,
                        hipStream_t s) {
    if (tile1 <= tile0) return;
    uint64_t total = (tile1 - tile0) * (mfma_dim_pad(dim) / 32) * 64;
    hipLaunchKernelGGL(k_tile_rows_h, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, n, dim, tile0, tile1, sx,
                       reinterpret_cast<uint4 *>(T));
}

// the same rounding, row-major: H[i] = fp16(X[i] * sx) (the HNSW walk's pre-pass gathers whole rows; count % 8 == 0)
__global__ __launch_bounds__(256) void k_rows_to_half(const float4 *__restrict__ X, uint64_t n8, float sx, uint4 *__restrict__ H) {
    const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n8) H[i] = pack8h(X[2 * i], X[2 * i + 1], sx);
}
void launch_rows_to_half(const float *X, uint64_t count, float sx, uint16_t *H, hipStream_t s) {
    if (count == 0) return;
    const uint64_t n8 = count / 8;
    hipLaunchKernelGGL(k_rows_to_half, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float4 *>(X), n8, sx,
                       reinterpret_cast<uint4 *>(H));
}

// ---- 8-bit image (half_rows.hpp, "8-bit tier"): one wave per row -- scale = max|x_i| / 127 (1 for an all-zero row),
// q_i = rint(x_i / scale) clamped to [-127, 127], err = |x - scale q| as measured (f32 sums, inflated by 1.001).  A row with a
// non-finite element gets err = +inf: its bound is infinite and the caller keeps it for the exact stage.  dim % 4 == 0.
__global__ __launch_bounds__(256) void k_rows_to_q8(const float *__restrict__ X, uint64_t n, uint32_t dim, int8_t *__restrict__ Q8,
                                                    float *__restrict__ scale, float *__restrict__ err) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t row = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float4 *x4 = reinterpret_cast<const float4 *>(X + row * dim);
    float mx = 0.0f;
    bool bad = false;
    for (uint32_t j = lane; j < dim / 4; j += 64) {
        const float4 v = x4[j];
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        bad |= !(v.x - v.x == 0.0f) || !(v.y - v.y == 0.0f) || !(v.z - v.z == 0.0f) || !(v.w - v.w == 0.0f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    bad = __ballot(bad) != 0;
    const float sc = (mx > 0.0f && !bad) ? mx / 127.0f : 1.0f;
    const float inv = 1.0f / sc;
    float e2 = 0.0f;
    char4 *q4 = reinterpret_cast<char4 *>(Q8 + row * dim);
    for (uint32_t j = lane; j < dim / 4; j += 64) {
        const float4 v = bad ? make_float4(0.f, 0.f, 0.f, 0.f) : x4[j];
        float q[4] = {rintf(v.x * inv), rintf(v.y * inv), rintf(v.z * inv), rintf(v.w * inv)};
        const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            q[e] = fminf(fmaxf(q[e], -127.0f), 127.0f);
            const float d = xv[e] - q[e] * sc;
            e2 += d * d;
        }
        q4[j] = make_char4((signed char)q[0], (signed char)q[1], (signed char)q[2], (signed char)q[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o);
    if (lane == 0) {
        scale[row] = sc;
        err[row] = bad ? INFINITY : sqrtf(e2) * 1.001f;
    }
}
void launch_rows_to_q8(const float *X, uint64_t n, uint32_t dim, int8_t *Q8, float *scale, float *err, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_rows_to_q8, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, X, n, dim, Q8, scale, err);
}

// |v - fp16(v * scale) / scale|^2 summed over a vector by one wave (all differences are exact in f32: v~ is v rounded
// to 11 bits, or a subnormal / zero whose distance to v is representable)
__device__ __forceinline__ float wave_round_err2(const float *v, uint32_t dim, float scale, float inv_scale, uint32_t lane) {
    float acc = 0.0f;
    if ((dim & 3) == 0) {  // 16-B loads, four in flight (the order of the sum is free)
        const float4 *v4 = reinterpret_cast<const float4 *>(v);
#pragma unroll 4
        for (uint32_t j = lane; j < dim / 4; j += 64) {
            const float4 x = v4[j];
            const float d0 = x.x - float((_Float16)(x.x * scale)) * inv_scale, d1 = x.y - float((_Float16)(x.y * scale)) * inv_scale;
            const float d2 = x.z - float((_Float16)(x.z * scale)) * inv_scale, d3 = x.w - float((_Float16)(x.w * scale)) * inv_scale;
            acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
    } else {
        for (uint32_t j = lane; j < dim; j += 64) {
            float x = v[j];
            float d = x - float((_Float16)(x * scale)) * inv_scale;
            acc += d * d;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    return acc;
}

// out2[0] = max over rows of |dx_r|^2, out2[1] = max of |dx_r|^2 / |x_r|^2 (float bits, atomicMax as unsigned: both are
// non-negative).  Rows with a non-finite norm are skipped: their keys are NaN / inf in every kernel and never pass a
// threshold.  One wave per row.
__global__ __launch_bounds__(256) void k_row_split_err(const float *__restrict__ X, const float *__restrict__ xsq,
                                                       uint64_t row0, uint64_t row1, uint32_t dim, float sx, float inv_sx,
                                                       uint32_t *__restrict__ out2) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t row = row0 + uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= row1) return;
    const float e2 = wave_round_err2(X + row * dim, dim, sx, inv_sx, lane);
    const float xs = xsq[row];
    if (lane == 0 && e2 > 0.0f && e2 < INFINITY && xs > 0.0f && xs < INFINITY) {
        // a million same-address atomics serialise in L2 (22 ms for 1M rows): only rows above the running maximum (a
        // plain, possibly stale read: the atomic below is what counts) issue one
        const volatile uint32_t *seen = out2;
        if (__float_as_uint(e2) > seen[0]) atomicMax(&out2[0], __float_as_uint(e2));
        const float rel = e2 / xs;
        if (rel < INFINITY && __float_as_uint(rel) > seen[1]) atomicMax(&out2[1], __float_as_uint(rel));
    }
}
void launch_row_split_err(const float *X, const float *xsq, uint64_t row0, uint64_t row1, uint32_t dim, float sx,
                          uint32_t *out2, hipStream_t s) {
    if (row1 <= row0) return;
    hipLaunchKernelGGL(k_row_split_err, dim3((unsigned)((row1 - row0 + 3) / 4)), dim3(256), 0, s, X, xsq, row0, row1, dim,
                       sx, 1.0f / sx, out2);
}

// per query: |q|^2 in the reference's order (strict f32 fold, distance/mod.rs:72-74 -- what k_row_sqnorm computes for the
// other paths: one launch less), scale sq = 2^(13 - c) with |q| < 2^c, multiplier qmul = 1 / (sx * sq) that undoes both
// scales, and the measured rounding error qerr = |q - q~/sq| (an upper bound: the f32 sum is inflated by 2^-10).  Queries
// whose norm is not a normal number of moderate size get qerr = +inf: they cannot be certified by the fp16 pass and are
// redone.  One wave per query: the query is parked in LDS with coalesced loads, lane 0 folds it in order.  Queries in
// [nq, nq_pad) are padding (zero images, tau = -inf).  hits[q] = 0 readies the filter pass's hit counters.
__global__ __launch_bounds__(256) void k_query_prep_h(const float *__restrict__ Q, uint32_t nq, uint32_t nq_pad, uint32_t dim,
                                                      float inv_sx, float *__restrict__ qsq, float *__restrict__ qscale,
                                                      float *__restrict__ qmul, float *__restrict__ qerr,
                                                      uint32_t *__restrict__ hits, uint4 *__restrict__ qfrag /* null: no image */) {
    extern __shared__ float qp_smem[];  // [4 waves][dim]
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x < 128) hits[nq_pad + threadIdx.x] = 0;  // the filter's set rendezvous words (k_gemm8.hip), behind the counters
    const uint32_t q = blockIdx.x * 4 + wave;
    if (q >= nq_pad) return;
    float sc = 1.0f, err = 0.0f;
    if (q < nq) {
        const float *qv = Q + size_t(q) * dim;
        float *sq = qp_smem + size_t(wave) * dim;
        for (uint32_t j = lane; j < dim; j += 64) sq[j] = qv[j];
        float qs = 0.0f;
        if (lane == 0) {  // same-wave LDS traffic is ordered: no barrier.  The adds are a strict chain, the reads are not:
            uint32_t j = 0;  // eight 16-B reads in flight (dim * 4 B per wave keeps the rows 16-B aligned when dim % 4 == 0)
            if ((dim & 3) == 0) {
                const float4 *s4 = reinterpret_cast<const float4 *>(sq);
                const uint32_t nv = dim / 4;
                uint32_t i = 0;
                for (; i + 8 <= nv; i += 8) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = s4[i + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) {  // products first, then the chain of adds (a multiply in front of every add: 15.6 instead of 11.9
                                                  // cycles per element, tools/fold_chain_probe.cpp); the same separately rounded values
                        const float p0 = v[u].x * v[u].x, p1 = v[u].y * v[u].y, p2 = v[u].z * v[u].z, p3 = v[u].w * v[u].w;
                        qs = qs + p0;
                        qs = qs + p1;
                        qs = qs + p2;
                        qs = qs + p3;
                    }
                }
                j = i * 4;
            }
            for (; j < dim; j++) qs = qs + sq[j] * sq[j];
            qsq[q] = qs;
        }
        qs = __shfl(qs, 0);
        if (qs == 0.0f) {
            // zero query: the image is exactly zero
        } else if (qs >= 0x1p-80f && qs <= 0x1p80f) {
            int e;
            (void)frexpf(qs, &e);            // qs = m * 2^e, m in [0.5, 1)  =>  |q| < 2^ceil(e/2)
            const int c = (e + 1) >> 1;       // arithmetic shift: ceil(e / 2)
            sc = ldexpf(1.0f, 13 - c);
            const float e2 = wave_round_err2(qv, dim, sc, ldexpf(1.0f, c - 13), lane);
            err = sqrtf(e2) * 1.001f;
        } else {
            err = INFINITY;
        }
    }
    if (lane == 0) {
        qscale[q] = sc;
        qmul[q] = inv_sx / sc;  // powers of two: exact
        qerr[q] = err;
        hits[q] = 0;
    }
    if (qfrag != nullptr) {
        // the query's B-operand image (the layout of k_pack_queries_h with NH = 8: one launch less per step): piece (kb32, kg) =
        // 8 columns 32 kb + 8 kg, 16 B at [group][kb64][half][sub][16 kg + (q & 15)]; padding queries and columns are zero
        constexpr uint32_t NH = 8;
        const uint32_t KB = ((dim + 63) & ~63u) / 32;
        const float *sq = qp_smem + size_t(wave) * dim;  // (the wave's own writes: ordered)
        uint4 *dst = qfrag + uint64_t(q / (16 * NH)) * KB * NH * 64;
        const uint32_t h = (q % (16 * NH)) / 16, r = q & 15;
        for (uint32_t p = lane; p < KB * 4; p += 64) {
            const uint32_t kb = p >> 2, kg = p & 3, c = kb * 32 + 8 * kg;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (q < nq && c + i < dim) ? sq[c + i] : 0.0f;
            dst[(((kb >> 1) * NH + h) * 2 + (kb & 1)) * 64 + 16 * kg + r] =
                pack8h(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), sc);
        }
    }
}
void launch_query_prep_h(const float *Q, uint32_t nq, uint32_t nq_pad, uint32_t dim, float sx, float *qsq, float *qscale,
                         float *qmul, float *qerr, uint32_t *hits, void *qfrag, hipStream_t s) {
    if (nq_pad == 0) return;
    hipLaunchKernelGGL(k_query_prep_h, dim3((nq_pad + 3) / 4), dim3(256), size_t(4) * dim * sizeof(float), s, Q, nq, nq_pad, dim,
                       1.0f / sx, qsq, qscale, qmul, qerr, hits, reinterpret_cast<uint4 *>(qfrag));
}

// Q [nq][dim] -> per group of 16*NH queries a B-operand image [kb64][half][sub 0|1][lane]: 8 fp16 of query
// 16*half + (lane & 15), columns 64*kb64 + 32*sub + 8*(lane >> 4) + j, scaled by qscale[query]  (queries >= nq: zero)
__global__ void k_pack_queries_h(const float *__restrict__ Q, uint32_t nq, uint32_t dim, uint32_t NH,
                                 const float *__restrict__ qscale, uint4 *__restrict__ qfrag) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // (kb32, half, lane)
    uint32_t KB = ((dim + 63) & ~63u) / 32;
    if (i >= KB * NH * 64) return;
    qfrag += uint64_t(blockIdx.y) * KB * NH * 64;
    uint32_t l = i & 63, h = (i >> 6) % NH, kb = (i >> 6) / NH;
    uint32_t q = blockIdx.y * 16 * NH + h * 16 + (l & 15);
    uint32_t c = kb * 32 + 8 * (l >> 4);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    float sc = 1.0f;
    if (q < nq) {
        load8_pad(Q + size_t(q) * dim, dim, c, a, b);
        sc = qscale[q];
    }
    qfrag[(((kb >> 1) * NH + h) * 2 + (kb & 1)) * 64 + l] = pack8h(a, b, sc);
}
void launch_pack_queries_h(const float *Q, uint32_t nq, uint32_t nq_cover, uint32_t dim, uint32_t NH, const float *qscale,
                           void *qfrag, hipStream_t s) {
    const uint32_t bq = 16 * NH;
    uint32_t total = (mfma_dim_pad(dim) / 32) * NH * 64;
    uint32_t nbatch = (std::max(nq, nq_cover) + bq - 1) / bq;
    if (nbatch == 0) return;
    hipLaunchKernelGGL(k_pack_queries_h, dim3((total + 255) / 256, nbatch), dim3(256), 0, s, Q, nq, dim, NH, qscale,
                       reinterpret_cast<uint4 *>(qfrag));
}

}  // namespace vdb
