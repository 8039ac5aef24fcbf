// half_rows.hpp -- certified half-precision scoring of gathered rows (device code shared by the HNSW walk, hnsw.hip, and the
// IVF probe-list scan, ivf.hip).
//
// Both paths score rows that are mostly NOT going to be kept: 85 % of the neighbours an HNSW expansion scores fail
// check_candidate, and an IVF scan keeps k of the thousands of rows in its probed clusters.  For those rows the exact value is
// never used -- only the fact that it is above a threshold.  The index keeps a row-major fp16 image of the rows
// (Index::d_rows_h: fp16(x * sx), the scale and rounding of the Flat fp16 mirror, whose MEASURED rounding error
// |dx_r| <= min(dx_abs, dx_rel |x_r|) therefore bounds it as well, k_row_split_err).  half_dots32 returns
// S~ = sum fp16(x_i sx) q_i / sx for up to 32 rows at once at half the bytes of the f32 rows; half_approx turns it into the
// approximate distance a and a bound E with |a - e| <= E for the value e the reference's arithmetic gives:
//
//   L2Sqr, cached form (HNSW)  a = fl(s - 2 S~), e = fl(s - 2 acc) with the SAME s = fl(|x|^2 + |q|^2):  |a - e| <= 2 B + u (|a| + |e|)
//   Cosine                     a = fl(1 - fl(S~/den)), e likewise with the SAME den:       |a - e| <= B/den + u (|S~|/den + ...) + u (|a| + |e|)
//     B >= |S~ - acc| = (gamma_m + gamma_d) |x||q| + |dx||q| (1 + gamma_m): the accumulation error of both sums (m <= d/8 + 3
//     and d terms) and the rounding of the image
//   L2Sqr, direct form (Flat, IVF)  e = strict fold of (x_i - q_i)^2, a = fl(fl(|x|^2 + |q|^2) - 2 S~): both within their own
//     rounding of the real distance D -- |e - D| <= gamma_(d+2) (|x| + |q|)^2, |a - D| <= gamma_d (|x|^2 + |q|^2) + 2 (gamma_m
//     |x||q| + |dx||q|) + 3u (|x| + |q|)^2 -- the sum is below 2.5 (d + 8) u (|x| + |q|)^2 + 2 |dx||q|, the bound of the Flat
//     fp16 tier
//
// Norms are the cached strict folds, inflated by 0.1 % for their own rounding; 1 % is added on the whole.  A caller drops a
// row only on a STRICT inequality a - E > threshold; NaN / infinite values fail it and take the exact path.
#pragma once
#include <utility>

#include "common.hpp"
#include "kernels.hpp"

namespace vdb {

typedef float v4f __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// lanes with fresh == true (all < 32) get S~ of row nb against the query in LDS (f32); dim % 64 == 0; at least one lane fresh.
// Any order, fused multiply-adds, all 64 lanes: lane 8g+j takes chunk j of its group's row (line-major: one L1 access per
// 128-B line), HALF_ROWS_DEPTH lines in flight in registers, partial sums reduced over j at the end -- no LDS transpose, no
// strict chain.  Straight-line code (clamped line indices, duplicate rows for idle lane groups): see hnsw_exact_dists_regs.
constexpr int HALF_ROWS_DEPTH = 5;  // lines (64 columns of 32 rows: 4 KB) in flight per wave
__device__ __forceinline__ float half_dots32(const uint16_t *__restrict__ rows_h, uint32_t dim, float inv_sx, const float *qlds, uint32_t nb,
                                             bool fresh, uint32_t lane) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    constexpr int D = HALF_ROWS_DEPTH;
    const uint32_t nlines = dim / 64, last = nlines - 1;
    const uint64_t fm = __ballot(fresh);
    const uint32_t nfresh = (uint32_t)__builtin_popcountll(fm);
    const uint32_t rank = (uint32_t)__builtin_popcountll(fm & ((1ull << lane) - 1));
    const uint32_t cnb = (uint32_t)__builtin_amdgcn_ds_permute(int((fresh ? rank : nfresh + (lane - rank)) * 4), int(nb));
    const uint32_t gg = lane >> 3, jj = lane & 7;
    const v4u *rp[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t src = 8 * k + gg;
        const uint32_t nbk = __shfl(cnb, src < nfresh ? src : 0u);
        rp[k] = reinterpret_cast<const v4u *>(rows_h + uint64_t(nbk) * dim) + jj;
    }
    v4u buf[D][4];
    static_for<D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const uint32_t Li = (uint32_t)i < last ? (uint32_t)i : last;
#pragma unroll
        for (int k = 0; k < 4; k++) buf[i][k] = rp[k][Li * 8];
        __builtin_amdgcn_sched_barrier(0);
    });
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const v4f *q4 = reinterpret_cast<const v4f *>(qlds) + jj * 2;  // the 8 query columns of this lane's chunk
    for (uint32_t L0 = 0; L0 < nlines; L0 += D) {
        static_for<D>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const uint32_t L = L0 + i;
            const uint32_t Lc = L < last ? L : last, Ln = L + D < last ? L + D : last;
            v4u cur[4];
#pragma unroll
            for (int k = 0; k < 4; k++) cur[k] = buf[i][k];
#pragma unroll
            for (int k = 0; k < 4; k++) buf[i][k] = rp[k][Ln * 8];
            __builtin_amdgcn_sched_barrier(0);
            const v4f qa = q4[Lc * 16], qb = q4[Lc * 16 + 1];
            const float qv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float a = acc[k];
                const uint32_t w[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const h2 h = __builtin_bit_cast(h2, w[e]);
                    a = __builtin_fmaf((float)h.x, qv[2 * e], a);
                    a = __builtin_fmaf((float)h.y, qv[2 * e + 1], a);
                }
                acc[k] = L < nlines ? a : acc[k];
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        acc[k] += __shfl_xor(acc[k], 1);
        acc[k] += __shfl_xor(acc[k], 2);
        acc[k] += __shfl_xor(acc[k], 4);
    }
    // compacted row r = 8k + g: its sum sits in acc[k] of the lanes of group g
    const uint32_t src = 8 * (lane & 7), kr = (lane >> 3) & 3;
    const float s0 = __shfl(acc[0], src), s1 = __shfl(acc[1], src), s2 = __shfl(acc[2], src), s3 = __shfl(acc[3], src);
    const float sr = kr == 0 ? s0 : (kr == 1 ? s1 : (kr == 2 ? s2 : s3));
    return __shfl(sr, rank) * inv_sx;  // back to the lane the neighbour came from; the scale is a power of two
}


// half_diffs32: the same fetch, but sum_i (h_i - q'_i)^2 with q' = q * sx in LDS (the scale is a power of two: q' is exact), returned times
// inv_sx^2 -- |x~ - q|^2 for the image row x~ = h / sx, every term non-negative, so the computed value is within (d + 4) u RELATIVE of the
// real one whatever the order.  This is the form that survives cancellation: the dot form's error is ~ |dx||q|, this one's |dx||x - q|
// (k_redo.hip: k_flat_refine_half).  UNIT: the image row is scaled by inv_sx / |x_r| (cached norm) on the fly and the LDS holds the unit query:
// |x^~ - q^|^2, the difference form of the Cosine keys' terms.
template <bool UNIT = false>
__device__ __forceinline__ float half_diffs32(const uint16_t *__restrict__ rows_h, uint32_t dim, float inv_sx, const float *qlds, uint32_t nb,
                                              bool fresh, uint32_t lane, const float *__restrict__ xsq = nullptr) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    constexpr int D = HALF_ROWS_DEPTH;
    const uint32_t nlines = dim / 64, last = nlines - 1;
    const uint64_t fm = __ballot(fresh);
    const uint32_t nfresh = (uint32_t)__builtin_popcountll(fm);
    const uint32_t rank = (uint32_t)__builtin_popcountll(fm & ((1ull << lane) - 1));
    const uint32_t cnb = (uint32_t)__builtin_amdgcn_ds_permute(int((fresh ? rank : nfresh + (lane - rank)) * 4), int(nb));
    const uint32_t gg = lane >> 3, jj = lane & 7;
    const v4u *rp[4];
    float rs[4] = {1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t src = 8 * k + gg;
        const uint32_t nbk = __shfl(cnb, src < nfresh ? src : 0u);
        rp[k] = reinterpret_cast<const v4u *>(rows_h + uint64_t(nbk) * dim) + jj;
        if (UNIT) rs[k] = inv_sx / sqrtf(xsq[nbk]);  // image row -> (nearly) unit row: h * rs
    }
    v4u buf[D][4];
    static_for<D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const uint32_t Li = (uint32_t)i < last ? (uint32_t)i : last;
#pragma unroll
        for (int k = 0; k < 4; k++) buf[i][k] = rp[k][Li * 8];
        __builtin_amdgcn_sched_barrier(0);
    });
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const v4f *q4 = reinterpret_cast<const v4f *>(qlds) + jj * 2;  // the 8 query columns of this lane's chunk
    for (uint32_t L0 = 0; L0 < nlines; L0 += D) {
        static_for<D>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const uint32_t L = L0 + i;
            const uint32_t Lc = L < last ? L : last, Ln = L + D < last ? L + D : last;
            v4u cur[4];
#pragma unroll
            for (int k = 0; k < 4; k++) cur[k] = buf[i][k];
#pragma unroll
            for (int k = 0; k < 4; k++) buf[i][k] = rp[k][Ln * 8];
            __builtin_amdgcn_sched_barrier(0);
            const v4f qa = q4[Lc * 16], qb = q4[Lc * 16 + 1];
            const float qv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float a = acc[k];
                const uint32_t w[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const h2 h = __builtin_bit_cast(h2, w[e]);
                    const float t0 = UNIT ? __builtin_fmaf((float)h.x, rs[k], -qv[2 * e]) : (float)h.x - qv[2 * e];
                    const float t1 = UNIT ? __builtin_fmaf((float)h.y, rs[k], -qv[2 * e + 1]) : (float)h.y - qv[2 * e + 1];
                    a = __builtin_fmaf(t0, t0, a);
                    a = __builtin_fmaf(t1, t1, a);
                }
                acc[k] = L < nlines ? a : acc[k];
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        acc[k] += __shfl_xor(acc[k], 1);
        acc[k] += __shfl_xor(acc[k], 2);
        acc[k] += __shfl_xor(acc[k], 4);
    }
    // compacted row r = 8k + g: its sum sits in acc[k] of the lanes of group g
    const uint32_t src = 8 * (lane & 7), kr = (lane >> 3) & 3;
    const float s0 = __shfl(acc[0], src), s1 = __shfl(acc[1], src), s2 = __shfl(acc[2], src), s3 = __shfl(acc[3], src);
    const float sr = kr == 0 ? s0 : (kr == 1 ? s1 : (kr == 2 ? s2 : s3));
    return __shfl(sr, rank) * (UNIT ? 1.0f : inv_sx * inv_sx);  // back to the lane the neighbour came from; the scale is a power of two
}


// approximate distance `a` of a row with cached |x|^2 = xs against a query with |q|^2 = qsq from an approximate dot product S
// whose distance to the real x.q is at most `op` (the operand term: rounding of the image[s]), and the bound E (see the
// header comment); metric: MET_L2_CACHED, MET_L2_DIRECT or MET_COSINE
__device__ __forceinline__ void approx_from_dot(int metric, uint32_t dim, float S, float xs, float qsq, float op, float &a, float &E) {
    constexpr float u = 0x1p-24f;
    const float nx = sqrtf(xs) * 1.001f, nq = sqrtf(qsq) * 1.001f;  // (cached norms: strict folds, relative error gamma_d << 1e-3)
    if (metric == MET_COSINE) {
        const float B = 1.002f * (2.0f * float(dim + 2) * u * nx * nq + op);
        const float den = fmaxf(sqrtf(xs) * sqrtf(qsq), 1e-10f);
        const float t = S / den;
        a = 1.0f - t;
        E = 1.01f * (B / den + 2.0f * u * (fabsf(t) + fabsf(a) + B / den + 1.0f));
    } else if (metric == MET_L2_CACHED) {
        const float B = 1.002f * (2.0f * float(dim + 2) * u * nx * nq + op);
        const float s2 = xs + qsq;
        a = s2 - 2.0f * S;
        E = 1.01f * (2.0f * B + 2.0f * u * (fabsf(a) + 2.0f * B));
    } else {
        const float s2 = xs + qsq;
        a = s2 - 2.0f * S;
        const float nn = nx + nq;
        E = 1.01f * (2.5f * float(dim + 8) * u * nn * nn + 2.0f * op);
    }
}
// fp16 image, f32 query: op = |dx||q| with the measured |dx| <= min(dx_abs, dx_rel |x|)
__device__ __forceinline__ void half_approx(int metric, uint32_t dim, float S, float xs, float qsq, float dx_abs, float dx_rel, float &a,
                                            float &E) {
    const float nx = sqrtf(xs) * 1.001f, nq = sqrtf(qsq) * 1.001f;
    approx_from_dot(metric, dim, S, xs, qsq, fminf(dx_abs, dx_rel * nx) * nq, a, E);
}
// true: the reference's distance of the row is certainly above `worst`
__device__ __forceinline__ bool half_rules_out(int metric, uint32_t dim, float S, float xs, float qsq, float dx_abs, float dx_rel,
                                               float worst) {
    float a, E;
    half_approx(metric, dim, S, xs, qsq, dx_abs, dx_rel, a, E);
    return a - E > worst && E < INFINITY;  // (NaN compares false: such rows take the exact path)
}


// ---- 8-bit tier ----------------------------------------------------------------------------------------------------------
// The IVF scan keeps a dozen of the thousands of rows it is offered, so most offers can be settled from an even smaller
// image: rows_q8[r][i] = rint(x_i / s_r) in [-127, 127] with one scale s_r = max|x_i| / 127 per row and the MEASURED
// |dx_r| = |x_r - s_r q8_r| beside it (k_rows_to_q8); the query is quantised the same way per call.  The dot product of
// the two images is an exact integer (|sum| <= dim * 127^2 < 2^24 for dim <= 1040, exact as f32), so
// |s_r s_q isum - x.q| <= |dx_r||q| + |x_r||dq| + |dx_r||dq| (Cauchy-Schwarz) + the two roundings of the scaling.
// q8_dots32: the exact integer sums of up to 32 rows (one per lane 0..31 with fresh == true), whole 128-B lines of a row per group
// of 8 lanes, HALF_ROWS_DEPTH lines in flight.
__device__ __forceinline__ int32_t q8_dots32(const int8_t *__restrict__ rows_q8, uint32_t dim, const int8_t *qlds8, uint32_t nb, bool fresh,
                                             uint32_t lane) {
    // 16-B pieces, 8 per 128-B line of a row; a row of dim bytes has dim/16 pieces, the last line may be half empty (dim % 128 ==
    // 64): lanes past the row's end re-read its last piece against query bytes that are zero (the caller pads the query image in
    // LDS with zeros up to a multiple of 128)
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    constexpr int D = HALF_ROWS_DEPTH;
    const uint32_t npieces = dim / 16, nlines = (dim + 127) / 128, last = nlines - 1;
    const uint64_t fm = __ballot(fresh);
    const uint32_t nfresh = (uint32_t)__builtin_popcountll(fm);
    const uint32_t rank = (uint32_t)__builtin_popcountll(fm & ((1ull << lane) - 1));
    const uint32_t cnb = (uint32_t)__builtin_amdgcn_ds_permute(int((fresh ? rank : nfresh + (lane - rank)) * 4), int(nb));
    const uint32_t gg = lane >> 3, jj = lane & 7;
    const v4u *rp[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t src = 8 * k + gg;
        const uint32_t nbk = __shfl(cnb, src < nfresh ? src : 0u);
        rp[k] = reinterpret_cast<const v4u *>(rows_q8 + uint64_t(nbk) * dim);
    }
    auto piece = [&](uint32_t L) -> uint32_t {  // this lane's piece of line L, clamped into the row
        const uint32_t p = L * 8 + jj;
        return p < npieces ? p : npieces - 1;
    };
    v4u buf[D][4];
    static_for<D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const uint32_t pi = piece((uint32_t)i < last ? (uint32_t)i : last);
#pragma unroll
        for (int k = 0; k < 4; k++) buf[i][k] = rp[k][pi];
        __builtin_amdgcn_sched_barrier(0);
    });
    int32_t acc[4] = {0, 0, 0, 0};
    const v4u *q4 = reinterpret_cast<const v4u *>(qlds8) + jj;  // the 16 query columns of this lane's piece (zero past dim)
    for (uint32_t L0 = 0; L0 < nlines; L0 += D) {
        static_for<D>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const uint32_t L = L0 + i;
            const uint32_t Lc = L < last ? L : last, Ln = L + D < last ? L + D : last;
            v4u cur[4];
#pragma unroll
            for (int k = 0; k < 4; k++) cur[k] = buf[i][k];
            const uint32_t pn = piece(Ln);
#pragma unroll
            for (int k = 0; k < 4; k++) buf[i][k] = rp[k][pn];
            __builtin_amdgcn_sched_barrier(0);
            const v4u qq = q4[Lc * 8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int32_t a = acc[k];
                a = __builtin_amdgcn_sdot4((int)cur[k].x, (int)qq.x, a, false);
                a = __builtin_amdgcn_sdot4((int)cur[k].y, (int)qq.y, a, false);
                a = __builtin_amdgcn_sdot4((int)cur[k].z, (int)qq.z, a, false);
                a = __builtin_amdgcn_sdot4((int)cur[k].w, (int)qq.w, a, false);
                acc[k] = L < nlines ? a : acc[k];
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        acc[k] += __shfl_xor(acc[k], 1);
        acc[k] += __shfl_xor(acc[k], 2);
        acc[k] += __shfl_xor(acc[k], 4);
    }
    const uint32_t src = 8 * (lane & 7), kr = (lane >> 3) & 3;
    const int32_t s0 = __shfl(acc[0], src), s1 = __shfl(acc[1], src), s2 = __shfl(acc[2], src), s3 = __shfl(acc[3], src);
    const int32_t sr = kr == 0 ? s0 : (kr == 1 ? s1 : (kr == 2 ? s2 : s3));
    return __shfl(sr, rank);
}
// S and its operand term from the integer sum: row scale / error sx, dxr, query scale / error sq, dq
__device__ __forceinline__ void q8_approx(int metric, uint32_t dim, int32_t isum, float xs, float qsq, float sx, float dxr, float sq, float dq,
                                          float &a, float &E) {
    constexpr float u = 0x1p-24f;
    const float S = (sx * sq) * float(isum);
    const float nx = sqrtf(xs) * 1.001f, nq = sqrtf(qsq) * 1.001f;
    const float op = 1.001f * (dxr * nq + nx * dq + dxr * dq) + 4.0f * u * fabsf(S);
    approx_from_dot(metric, dim, S, xs, qsq, op, a, E);
}

}  // namespace vdb
