// k_i8.hip -- operands of k_flat_gemm8 (k_gemm8.hip): the centred 8-bit mirror of the rows, its per-row constants, the
// centred 8-bit query images and their per-query constants.  Everything here exists to make
//     key(r, q) + O_q  <=  D(r, q) = |x_r - q|^2                       (L2Sqr; FlatIndex::knn, flat_index.rs:48-57)
// hold for EVERY row and query, with key = C_r + M_r (s_q I(r, q)) evaluated by the filter kernel in f32.
//
// Derivation.  mu is one f32 vector per index (a sample mean of the rows: any vector is valid, a good one makes the centred
// rows small).  With x_c = x - mu, q_c = q - mu as real vectors, D = |x_c|^2 + |q_c|^2 - 2 <x_c, q_c>.  Rows and queries
// are rounded to x~ = s_r x^ and q~ = s_q q^ with int8 vectors x^, q^ (one scale each), dx = x_c - x~, dq = q_c - q~:
//     <x_c, q_c> = <x~, q~> + <dx, q_c> + <x~, dq>  <=  s_r s_q I + |dx||q_c| + |x~||dq|            (Cauchy-Schwarz)
// with I = <x^, q^> an exact integer.  The two products of a row quantity and a query quantity are split by
// 2ab <= l a^2 + b^2 / l (any l > 0):
//     D >= [ |x_c|^2 - |dx|^2 / l1 - |x~|^2 / l2 ]  -  2 s_r s_q I  +  [ |q_c|^2 (1 - l1) - l2 |dq|^2 ]
//        =              C_r                          +  M_r (s_q I)  +               O_q
// l1 = rho, l2 = 1 / rho with rho = the typical |dx| / |x_c| of the index (measured on a row sample when the mirror is
// built): exact for a query that looks like a row, and within (t + 1/t) / 2 of the unsplit bound for a query whose norm
// ratio is off by t.  Every measured quantity is rounded in the safe direction (norms that enter positively are deflated
// by 2 (d + 8) u, residuals are inflated by 1.001 and by the rounding of x - mu itself), so C_r and O_q as stored are
// below their real-number values; the two roundings of the key's own evaluation are part of the certification's margin
// (flat_certify_flag, k_exact.hip).
//
// Cosine (the reference's default metric, pyo3/mod.rs:73; distance/mod.rs:60-69): with a^ = x / |x| and b^ = q / |q| as real
// vectors, 1 - <x, q> / (|x||q|) = |a^ - b^|^2 / 2 -- the cosine distance IS half the L2Sqr distance of the unit vectors, so the
// same construction on unit rows and unit queries (COS = true below: the vector is scaled by 1 / |v| before it is centred,
// the norm and the scaling in f64, so that what is rounded to f32 is a^ - mu up to 1e-13) gives
//     key(r, q) + O_q  <=  2 (1 - cos(x_r, q))
// and the exact stage certifies against (key + O_q) / 2 minus the rounding of the reference's own f32 evaluation of the
// cosine ((2 d + 8) u, flat_certify_lb).  What the real-number cosine does not describe is kept out of the bound: an
// all-zero row is coded as the zero "unit" vector (bound 1/2 <= its reference distance 1 - 0 / 1e-10 = 1); a row whose
// cached |x|^2 (the reference's strict f32 fold) overflowed, underflowed to 0 or lies below 1e-30 gets C_r = -FLT_MAX --
// it passes every threshold and is always evaluated exactly --; a query with such a norm (or a zero one) gets O_q = NaN and
// goes to the next tier; the max(|x||q|, 1e-10) clamp (mod.rs:68) is checked per query by the certification.
#include "common.hpp"
#include "kernels.hpp"

namespace vdb {

constexpr float I8_U = 5.9604645e-8f;  // 2^-24

// ---- column means of a strided row sample: part[chunk][dim] partial sums, then mu[j] = sum / rows --------------------
// xsq != null (Cosine): the mean of the UNIT rows (f32 scaling is enough here: any mu is valid)
__global__ __launch_bounds__(256) void k_col_sum(const float *__restrict__ X, uint64_t n, uint32_t dim, uint64_t stride,
                                                 uint64_t n_s, float *__restrict__ part, const float *__restrict__ xsq) {
    const uint32_t col = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const uint64_t per = (n_s + gridDim.y - 1) / gridDim.y;
    const uint64_t a = uint64_t(blockIdx.y) * per, b = a + per < n_s ? a + per : n_s;
    float acc = 0.0f;
    if (col < dim)
        for (uint64_t i = a + sub; i < b; i += 4) {
            float v = X[(i * stride) * dim + col];
            if (xsq) {
                const float ns = xsq[i * stride];
                v = (ns >= 1e-30f && ns <= 1e30f) ? v / sqrtf(ns) : 0.0f;
            }
            acc += (v - v == 0.0f) ? v : 0.0f;  // non-finite elements do not poison the mean
        }
    __shared__ float sh[256];
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (sub == 0 && col < dim) part[uint64_t(blockIdx.y) * dim + col] = (sh[threadIdx.x] + sh[threadIdx.x + 64]) + (sh[threadIdx.x + 128] + sh[threadIdx.x + 192]);
}
__global__ void k_col_mean(const float *__restrict__ part, uint32_t chunks, uint32_t dim, float inv_rows, float *__restrict__ mu) {
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= dim) return;
    float acc = 0.0f;
    for (uint32_t c = 0; c < chunks; c++) acc += part[uint64_t(c) * dim + col];
    mu[col] = acc * inv_rows;
}
// mu = mean of up to 16 384 rows spread over the table; part: I8_MEAN_CHUNKS * dim floats of scratch
void launch_i8_col_mean(const float *X, uint64_t n, uint32_t dim, float *part, float *mu, hipStream_t s, const float *xsq_cos) {
    const uint64_t n_s = n < 16384 ? n : 16384, stride = n / n_s;
    hipLaunchKernelGGL(k_col_sum, dim3((dim + 63) / 64, I8_MEAN_CHUNKS), dim3(256), 0, s, X, n, dim, stride, n_s, part, xsq_cos);
    hipLaunchKernelGGL(k_col_mean, dim3((dim + 63) / 64), dim3(64), 0, s, part, I8_MEAN_CHUNKS, dim, 1.0f / float(n_s), mu);
}

// ---- one 16-lane group per vector: centre, scale, round, measure -----------------------------------------------------
struct I8Row {
    float s;      // scale (1 for an all-zero / degenerate vector)
    float xs;     // |v - mu|^2 as computed (f32, any order)
    float e2;     // |(v - mu) - s v^|^2 as computed
    float i2;     // |v^|^2 (exact integer below 2^24)
    bool bad;     // a non-finite element (of v - mu)
};
// the centred element: v - mu (L2Sqr), or v / |v| - mu with the scaling in f64 (Cosine; inv = 1 / |v|, 0 for a zero vector)
template <bool COS>
__device__ __forceinline__ float i8_centre(float a, float m, double inv) {
    if constexpr (COS)
        return float(double(a) * inv - double(m));
    else
        return a - m;
}
// Cosine: sum of squares of a vector in f64 over the 16 lanes sub = 0..15 of a group (finite for every finite f32 vector)
__device__ __forceinline__ double i8_sumsq64(const float *__restrict__ v, uint32_t dim, uint32_t sub) {
    double acc = 0.0;
    const float4 *v4 = reinterpret_cast<const float4 *>(v);
    for (uint32_t j = sub; j < dim / 4; j += 16) {
        const float4 a = v4[j];
        acc += double(a.x) * double(a.x) + double(a.y) * double(a.y);
        acc += double(a.z) * double(a.z) + double(a.w) * double(a.w);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    return acc;
}
// pass 1 over a vector by the 16 lanes sub = 0..15 of a group: max |c|, |c|^2, non-finite flag of the centred vector c; dim % 4 == 0
template <bool COS>
__device__ __forceinline__ void i8_pass1(const float *__restrict__ v, const float *__restrict__ mu, uint32_t dim, uint32_t sub, double inv,
                                         float &mx, float &xs, bool &bad) {
    mx = 0.0f;
    xs = 0.0f;
    bad = false;
    const float4 *v4 = reinterpret_cast<const float4 *>(v), *m4 = reinterpret_cast<const float4 *>(mu);
    for (uint32_t j = sub; j < dim / 4; j += 16) {
        const float4 a = v4[j], m = m4[j];
        const float c[4] = {i8_centre<COS>(a.x, m.x, inv), i8_centre<COS>(a.y, m.y, inv), i8_centre<COS>(a.z, m.z, inv), i8_centre<COS>(a.w, m.w, inv)};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            mx = fmaxf(mx, fabsf(c[e]));
            xs += c[e] * c[e];
            bad |= !(c[e] - c[e] == 0.0f);
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o));
        xs += __shfl_xor(xs, o);
        bad |= __shfl_xor((int)bad, o) != 0;
    }
}
// the 16 int8 of columns [c0, c0 + 16) (zero past dim) with their contributions to e2 / i2
template <bool COS>
__device__ __forceinline__ uint4 i8_piece(const float *__restrict__ v, const float *__restrict__ mu, uint32_t dim, uint32_t c0, float s,
                                          float inv, double inv_n, bool zero, float &e2, float &i2) {
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if (zero || c0 >= dim) return make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (c0 + 4 * k >= dim) break;  // dim % 4 == 0
        const float4 a = *reinterpret_cast<const float4 *>(v + c0 + 4 * k), m = *reinterpret_cast<const float4 *>(mu + c0 + 4 * k);
        const float c[4] = {i8_centre<COS>(a.x, m.x, inv_n), i8_centre<COS>(a.y, m.y, inv_n), i8_centre<COS>(a.z, m.z, inv_n),
                            i8_centre<COS>(a.w, m.w, inv_n)};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float q = rintf(c[e] * inv);
            q = fminf(fmaxf(q, -127.0f), 127.0f);
            const float d = c[e] - q * s;
            e2 += d * d;
            i2 += q * q;
            w[k] |= (uint32_t(int(q)) & 0xffu) << (8 * e);
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ float i8_scale(float mx, bool bad) { return (mx >= 1e-30f && mx <= 1e30f && !bad) ? mx / 127.0f : 1.0f; }

// rows [16*tile0, 16*tile1) -> T[(tile*KB + kb)*64 + 16*g + (row & 15)] = 16 int8: columns 64*kb + 16*g + j of the centred,
// scaled row -- the A operand of v_mfma_i32_16x16x64_i8 (the k order inside an instruction is whatever the hardware makes
// of it: queries are packed with the same map, and a dot product does not care) -- and rowc[row] = {C_r, M_r}.
// rows >= n: zero codes, {+inf, 0}; rows with a non-finite element: zero codes, {NaN, 0} (their keys never pass a threshold,
// like everywhere else in the Flat pipeline).  stats != null (sample pass, T == null): stats[2 i] = |dx|^2, [2 i + 1] = |x_c|^2
// of sampled row i = blockIdx * 16 + group, rows taken stride apart.  One workgroup per tile, 16 lanes per row; dim % 4 == 0.
// COS: xsq = the cached strict-fold |x|^2 of the rows (what the reference's cosine divides by): rows it does not describe
// as a real-number norm (overflow, underflow to 0 of a non-zero row, below 1e-30) get {-FLT_MAX, 0} -- always evaluated exactly.
template <bool COS>
__global__ __launch_bounds__(256) void k_tile_rows_i8(const float *__restrict__ X, uint64_t n, uint32_t dim, uint64_t tile0,
                                                      const float *__restrict__ mu, float il1, float il2, uint4 *__restrict__ T,
                                                      float2 *__restrict__ rowc, float *__restrict__ stats, uint64_t stride,
                                                      const float *__restrict__ xsq) {
    const uint32_t KB = ((dim + 63) & ~63u) / 64;
    const uint32_t grp = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const uint64_t tile = tile0 + blockIdx.x;
    const uint64_t row = stats ? (uint64_t(blockIdx.x) * 16 + grp) * stride : tile * 16 + grp;
    const bool live = row < n;
    const float *v = X + (live ? row : 0) * dim;
    double inv_n = 0.0;
    bool odd = false;  // COS: a live, finite row whose reference distance is not the real-number cosine
    if constexpr (COS) {
        const double s64 = i8_sumsq64(v, dim, sub);
        const float ns = xsq[live ? row : 0];
        inv_n = (s64 > 0.0 && s64 < 1e300) ? 1.0 / sqrt(s64) : 0.0;  // zero row: the zero "unit" vector
        odd = !(s64 == 0.0 && ns == 0.0) && !(ns >= 1e-30f && ns <= 1e30f);
    }
    float mx, xs;
    bool bad;
    i8_pass1<COS>(v, mu, dim, sub, inv_n, mx, xs, bad);
    const float s = i8_scale(mx, bad), inv = 1.0f / s;
    const bool zero = !live || bad || odd || !(mx >= 1e-30f && mx <= 1e30f);
    float e2 = 0.0f, i2 = 0.0f;
    for (uint32_t pp = sub; pp < KB * 4; pp += 16) {
        const uint4 w = i8_piece<COS>(v, mu, dim, pp * 16, s, inv, inv_n, zero, e2, i2);
        if (T) T[(tile * KB + (pp >> 2)) * 64 + 16 * (pp & 3) + grp] = w;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        e2 += __shfl_xor(e2, o);
        i2 += __shfl_xor(i2, o);
    }
    if (zero && live && !bad) e2 = xs;  // degenerate magnitudes: coded as zeros, the residual is the whole centred row
    if (sub != 0) return;
    if (stats) {
        stats[2 * (uint64_t(blockIdx.x) * 16 + grp)] = (live && !bad && !odd) ? e2 : 0.0f;
        stats[2 * (uint64_t(blockIdx.x) * 16 + grp) + 1] = (live && !bad && !odd) ? xs : 0.0f;
        return;
    }
    float2 out;
    if (!live) {
        out = make_float2(INFINITY, 0.0f);
    } else if (bad) {
        out = make_float2(__uint_as_float(0x7fc00000u), 0.0f);
    } else if (odd) {
        out = make_float2(-3.4028234664e38f, 0.0f);
    } else {
        const float du = 2.0f * float(dim + 8) * I8_U;
        const float xs_lo = xs * (1.0f - du);
        // |dx| incl. the rounding of x - mu and of s q (COS: + the 1e-13-relative f64 normalisation, far inside the 1e-12)
        const float dxn = sqrtf(e2) * 1.001f + 4.0f * I8_U * sqrtf(xs) + (COS ? 1e-12f : 0.0f);
        const float xt2 = (s * s) * i2 * 1.001f;                          // |x~|^2
        float c = xs_lo - dxn * dxn * il1 - xt2 * il2;
        c -= 8.0f * I8_U * (xs + dxn * dxn * il1 + xt2 * il2);            // the roundings of this very expression
        out = make_float2(c, zero ? 0.0f : -2.0f * s);
    }
    rowc[row] = out;
}
void launch_tile_rows_i8(const float *X, uint64_t n, uint32_t dim, uint64_t tile0, uint64_t tile1, const float *mu, float l1, float l2,
                         void *T, float *rowc, hipStream_t s, const float *xsq_cos) {
    if (tile1 <= tile0) return;
    if (xsq_cos)
        hipLaunchKernelGGL(k_tile_rows_i8<true>, dim3((unsigned)(tile1 - tile0)), dim3(256), 0, s, X, n, dim, tile0, mu, 1.0f / l1, 1.0f / l2,
                           reinterpret_cast<uint4 *>(T), reinterpret_cast<float2 *>(rowc), (float *)nullptr, uint64_t(1), xsq_cos);
    else
        hipLaunchKernelGGL(k_tile_rows_i8<false>, dim3((unsigned)(tile1 - tile0)), dim3(256), 0, s, X, n, dim, tile0, mu, 1.0f / l1, 1.0f / l2,
                           reinterpret_cast<uint4 *>(T), reinterpret_cast<float2 *>(rowc), (float *)nullptr, uint64_t(1), (const float *)nullptr);
}
// |dx|^2 and |x_c|^2 of n_s rows taken `stride` apart: stats[2 i], stats[2 i + 1] (n_s rounded up to 16: the tail is zero)
void launch_i8_row_stats(const float *X, uint64_t n, uint32_t dim, const float *mu, uint64_t n_s, uint64_t stride, float *stats,
                         hipStream_t s, const float *xsq_cos) {
    if (n_s == 0) return;
    if (xsq_cos)
        hipLaunchKernelGGL(k_tile_rows_i8<true>, dim3((unsigned)((n_s + 15) / 16)), dim3(256), 0, s, X, n, dim, uint64_t(0), mu, 1.0f, 1.0f,
                           (uint4 *)nullptr, (float2 *)nullptr, stats, stride, xsq_cos);
    else
        hipLaunchKernelGGL(k_tile_rows_i8<false>, dim3((unsigned)((n_s + 15) / 16)), dim3(256), 0, s, X, n, dim, uint64_t(0), mu, 1.0f, 1.0f,
                           (uint4 *)nullptr, (float2 *)nullptr, stats, stride, (const float *)nullptr);
}

// ---- queries: |q|^2 in the reference's order (what the exact stage and the certification use, as k_query_prep_h), the
// centred 8-bit image in the B-operand layout [group][kb64][half][16 g + (q & 15)], its scale, and the offset O_q.  A query
// with a non-finite element gets scale 0, a zero image and O_q = NaN: it cannot be certified here and goes to the next tier.
// Queries in [nq, nq_pad) are padding (zero image, scale 0; tau = -inf is set by the threshold select).  hits[q] = 0 readies
// the filter pass's counters.  One wave per query: 4 groups of 16 lanes share the pieces; dim % 4 == 0.
// COS: the image is that of q / |q| (norm and scaling in f64); a query whose strict-fold |q|^2 is 0, overflowed or outside
// [1e-30, 1e30] has no real-number cosine the bound could describe and goes to the next tier like a non-finite one.
// Round 4: the strict fold of |q|^2 -- 960 dependent adds, ~4.4 us that nothing else of a query's preparation waits for -- runs on a
// wave of its own (waves 4 - 7 of the workgroup fold the queries whose images waves 0 - 3 build), so a workgroup takes
// max(image, fold) instead of their sum.
template <bool COS>
__global__ __launch_bounds__(512) void k_query_prep_i8(const float *__restrict__ Q, uint32_t nq, uint32_t nq_pad, uint32_t dim,
                                                       const float *__restrict__ mu, float l1, float l2, float *__restrict__ qsq,
                                                       float *__restrict__ qscale, float *__restrict__ qoff,
                                                       uint32_t *__restrict__ hits, uint4 *__restrict__ qfrag) {
    extern __shared__ float qp8_smem[];  // [8 waves][dim]
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x < 128) hits[nq_pad + threadIdx.x] = 0;  // the filter's set rendezvous words (k_gemm8.hip), behind the counters
    const uint32_t q = blockIdx.x * 4 + (wave & 3);
    if (q >= nq_pad) return;
    constexpr uint32_t NH = 8;
    const uint32_t KB = ((dim + 63) & ~63u) / 64;
    uint4 *dst = qfrag + uint64_t(q / (16 * NH)) * KB * NH * 64;
    const uint32_t h = (q % (16 * NH)) / 16, r = q & 15;
    if (q >= nq) {
        if (wave >= 4) return;
        for (uint32_t pp = lane; pp < KB * 4; pp += 64) dst[((pp >> 2) * NH + h) * 64 + 16 * (pp & 3) + r] = make_uint4(0u, 0u, 0u, 0u);
        if (lane == 0) {
            qscale[q] = 0.0f;
            qoff[q] = 0.0f;
            hits[q] = 0;
        }
        return;
    }
    const float *qv = Q + size_t(q) * dim;
    float *sq = qp8_smem + size_t(wave) * dim;
    for (uint32_t j = lane; j < dim; j += 64) sq[j] = qv[j];
    if (wave >= 4) {  // the fold wave of this query
        if (lane != 0) return;
        float qs = 0.0f;
        // strict fold of |q|^2 (distance/mod.rs:72-74), reads eight 16-B pieces ahead of the chain
        const float4 *s4 = reinterpret_cast<const float4 *>(sq);
        const uint32_t nv = dim / 4;
        uint32_t i = 0;
        for (; i + 8 <= nv; i += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = s4[i + u];
#pragma unroll
            for (int u = 0; u < 8; u++) {  // products first, then the chain of adds (a multiply in front of every add: 15.6 instead of 11.9
                                          // cycles per element, tools/fold_chain_probe.cpp); the same separately rounded values
                const float p0 = t[u].x * t[u].x, p1 = t[u].y * t[u].y, p2 = t[u].z * t[u].z, p3 = t[u].w * t[u].w;
                qs = qs + p0;
                qs = qs + p1;
                qs = qs + p2;
                qs = qs + p3;
            }
        }
        for (uint32_t j = i * 4; j < dim; j++) qs = qs + sq[j] * sq[j];
        qsq[q] = qs;
        return;
    }
    // centred image: every 16-lane group computes the same maxima (cheap), then the four groups share the pieces
    float mx, xs;
    bool bad;
    double inv_n = 0.0;
    bool odd = false;
    if constexpr (COS) {
        // (the strict f32 fold of |q|^2 -- another wave's -- differs from this f64 sum by 1e-4 relative at most: a sum inside [1e-29, 1e29]
        // means a fold inside the [1e-30, 1e30] the certification asks for, which checks the fold itself)
        const double s64 = i8_sumsq64(sq, dim, lane & 15);
        odd = !(s64 >= 1e-29 && s64 <= 1e29);
        inv_n = odd ? 0.0 : 1.0 / sqrt(s64);
    }
    i8_pass1<COS>(sq, mu, dim, lane & 15, inv_n, mx, xs, bad);
    bad |= odd;
    const float s = i8_scale(mx, bad), inv = 1.0f / s;
    const bool zero = bad || !(mx >= 1e-30f && mx <= 1e30f);
    float e2 = 0.0f, i2 = 0.0f;
    for (uint32_t pp = lane; pp < KB * 4; pp += 64)
        dst[((pp >> 2) * NH + h) * 64 + 16 * (pp & 3) + r] = i8_piece<COS>(sq, mu, dim, pp * 16, s, inv, inv_n, zero, e2, i2);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o);
    if (zero && !bad) e2 = xs;
    if (lane == 0) {
        float off;
        if (bad) {
            off = __uint_as_float(0x7fc00000u);
        } else {
            const float du = 2.0f * float(dim + 8) * I8_U;
            const float dqn = sqrtf(e2) * 1.001f + 4.0f * I8_U * sqrtf(xs) + (COS ? 1e-12f : 0.0f);
            // |q_c|^2 (1 - l1) - l2 |dq|^2, the norm deflated (inflated if 1 - l1 < 0 ever made its term negative)
            const float w1 = 1.0f - l1;
            off = (w1 >= 0.0f ? xs * (1.0f - du) : xs * (1.0f + du)) * w1 - l2 * dqn * dqn;
            off -= 8.0f * I8_U * (xs * fabsf(w1) + l2 * dqn * dqn);
        }
        qscale[q] = (zero || bad) ? 0.0f : s;
        qoff[q] = off;
        hits[q] = 0;
    }
}
void launch_query_prep_i8(const float *Q, uint32_t nq, uint32_t nq_pad, uint32_t dim, const float *mu, float l1, float l2, float *qsq,
                          float *qscale, float *qoff, uint32_t *hits, void *qfrag, hipStream_t s, int cosine) {
    if (nq_pad == 0) return;
    if (size_t(8) * dim * sizeof(float) > size_t(48) * 1024) {  // (dims above 1536: more dynamic LDS than a kernel gets without asking)
        func_max_lds(reinterpret_cast<const void *>(&k_query_prep_i8<true>), int(64 * 1024));
        func_max_lds(reinterpret_cast<const void *>(&k_query_prep_i8<false>), int(64 * 1024));
    }
    if (cosine)
        hipLaunchKernelGGL(k_query_prep_i8<true>, dim3((nq_pad + 3) / 4), dim3(512), size_t(8) * dim * sizeof(float), s, Q, nq, nq_pad, dim, mu,
                           l1, l2, qsq, qscale, qoff, hits, reinterpret_cast<uint4 *>(qfrag));
    else
        hipLaunchKernelGGL(k_query_prep_i8<false>, dim3((nq_pad + 3) / 4), dim3(512), size_t(8) * dim * sizeof(float), s, Q, nq, nq_pad, dim, mu,
                           l1, l2, qsq, qscale, qoff, hits, reinterpret_cast<uint4 *>(qfrag));
}

}  // namespace vdb
