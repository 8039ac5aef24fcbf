// index.hip -- Index object: HBM-resident VecSet + the Flat search pipeline.
#include "index.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace vdb {

Index::Index(int dev, uint64_t d, int ds, bool u8) : device(dev), dim(d), dist(ds), elem_u8(u8) {
#ifndef VDB_HOST_SANITIZER_BUILD  // (the thread-sanitizer binary of tests/cpp/tsan_host.cpp has no device: host-side state only)
    use_device();
    hipDeviceProp_t prop;
    VDB_HIP(hipGetDeviceProperties(&prop, dev));
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    std::string arch = prop.gcnArchName;
    if (arch.rfind("gfx950", 0) != 0)
        throw Error(4, "libvdbhip is built for gfx950 (MI355X) only; device reports " + arch);
#endif
}

std::unique_ptr<Workspace> Index::acquire_ws() {
    use_device();
    {
        std::lock_guard<std::mutex> g(ws_mu);
        if (!ws_free.empty()) {
            auto ws = std::move(ws_free.back());
            ws_free.pop_back();
            return ws;
        }
    }
    return std::make_unique<Workspace>();
}
void Index::release_ws(std::unique_ptr<Workspace> ws) {
    if (!ws) return;
    std::lock_guard<std::mutex> g(ws_mu);
    ws_free.push_back(std::move(ws));
}

const float *Index::rows_f32_view(DevBuf &scratch, uint64_t r0, uint64_t r1, hipStream_t s) const {
    if (!elem_u8) return d_rows.as<float>();
    scratch.reserve(std::max<uint64_t>(r1 - r0, 1) * dim * sizeof(float));
    launch_widen_u8(d_rows.as<uint8_t>() + r0 * dim, (r1 - r0) * dim, scratch.as<float>(), s);
    return reinterpret_cast<const float *>(reinterpret_cast<uintptr_t>(scratch.p) - r0 * dim * sizeof(float));
}
const float *Index::host_rows() const {
    VDB_REQUIRE(!elem_u8, "this operation needs f32 rows: a VecSet<u8> index serves Flat search only");
    std::lock_guard<std::mutex> g(host_mu);
    if (!host_valid) {
        use_device();
        h_rows.resize(size_t(n) * dim);
        if (n) VDB_HIP(hipMemcpy(h_rows.data(), d_rows.p, size_t(n) * dim * sizeof(float), hipMemcpyDeviceToHost));
        host_valid = true;
    }
    return h_rows.data();
}

// VecSet::push (vec_set.rs:113-118) for `count` rows + dist_cache (hnsw_index.rs:251-254)
void Index::add_rows(const void *rows, uint64_t count, bool on_device) {
    if (count == 0) return;
    use_device();
    VDB_REQUIRE(n + count < (1ull << 32), "a shard holds at most 2^32-1 rows (ids are u32 on the device)");
    WsLease ws(*this);
    hipStream_t s = ws->stream;
    size_t row_bytes = size_t(dim) * elem_size();
    d_rows.grow((n + count) * row_bytes + 16, n * row_bytes, s);  // +16: the u8 kernels may read a 16-B word at the last row's end
    d_sq.grow((n + count + 128) * sizeof(float), n * sizeof(float), s);  // +128: kernels may read a few entries past n
    char *dst = d_rows.as<char>() + n * row_bytes;
    VDB_HIP(hipMemcpyAsync(dst, rows, count * row_bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    bool mirror = mfma_supported((uint32_t)dim) && tiled_built;
    uint64_t tiles_new = ((n + count + 15) / 16 + 11) / 12 * 12;  // whole 64-row items (k_flat_mfma) and whole 2/3-tile units (k_flat_gemm)
    uint64_t tiles_old = n / 16;                                // the partially filled tile is rewritten
    if (mirror) {
        uint64_t tile_bytes = 16 * size_t(mfma_dim_pad((uint32_t)dim)) * sizeof(float);
        try {
            d_tiled.grow(tiles_new * tile_bytes, tiles_old * tile_bytes, s);
        } catch (const AllocError &) {  // the rows went in; the mirror is dropped and rebuilt (or not) by the search that wants it
            d_tiled.release();
            tiled_built = false;
            mirror = false;
            mirror_alloc_failures += 1;
        }
    }
    // row norms of the new rows and the fragment-ordered mirror of every 16-row tile that received rows (a u8 index feeds
    // these f32 build kernels widened chunks; the rows themselves stay at one byte per element)
    for_tile_chunks(*ws, tiles_old, tiles_new, n + count, [&](const float *v, uint64_t ta, uint64_t tb, uint64_t ra, uint64_t rb) {
        const uint64_t a = std::max(ra, n);
        if (rb > a) launch_row_sqnorm(v + a * dim, rb - a, (uint32_t)dim, d_sq.as<float>() + a, s);
        if (mirror) launch_tile_rows(v, n + count, (uint32_t)dim, ta, tb, d_tiled.as<float>(), s);
    });
    std::vector<float> sq(count);
    VDB_HIP(hipMemcpyAsync(sq.data(), d_sq.as<float>() + n, count * sizeof(float), hipMemcpyDeviceToHost, s));
    VDB_SYNC(s);
    for (float v : sq) {
        if (v > xsq_max && std::isfinite(v)) xsq_max = v;
        if (v > 0.0f && v < xsq_min_pos) xsq_min_pos = v;
    }
    h_sq.insert(h_sq.end(), sq.begin(), sq.end());
    if (i8_defers_half() && half_n == n) {
        // the 8-bit pass is this index's first tier: the fp16 mirror waits for its first use (ensure_half)
    } else if (half_n == n) {
        try {
            half_refresh(*ws, n, n + count);  // needs the new xsq_max
            half_n = n + count;
        } catch (const AllocError &) {  // (as above: ensure_half decides at the next search)
            d_tiled_h.release();
            half_valid = false;
            half_n = 0;
            mirror_alloc_failures += 1;
        }
    }  // (else: already behind -- ensure_half catches up)
    {
        std::lock_guard<std::mutex> g(host_mu);
        if (on_device || elem_u8) {
            host_valid = false;
        } else if (host_valid) {
            const float *fr = static_cast<const float *>(rows);
            h_rows.insert(h_rows.end(), fr, fr + count * dim);
        }
    }
    n += count;
}

// VecSet::swap_remove (vec_set.rs:131-137)
void Index::swap_remove(uint64_t i) {
    VDB_REQUIRE(i < n, "swap_remove: index out of bounds");
    use_device();
    WsLease ws(*this);
    hipStream_t s = ws->stream;
    uint64_t last = n - 1;
    const size_t row_bytes = size_t(dim) * elem_size();
    if (i < last) {
        VDB_HIP(hipMemcpyAsync(d_rows.as<char>() + i * row_bytes, d_rows.as<char>() + last * row_bytes, row_bytes, hipMemcpyDeviceToDevice, s));
        VDB_HIP(hipMemcpyAsync(d_sq.as<float>() + i, d_sq.as<float>() + last, sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (mfma_supported((uint32_t)dim)) {  // rewrite the tiles of the moved row and of the removed last row
        for (uint64_t t : {i / 16, last / 16})
            for_tile_chunks(*ws, t, t + 1, last, [&](const float *v, uint64_t ta, uint64_t tb, uint64_t, uint64_t) {
                if (tiled_built) launch_tile_rows(v, last, (uint32_t)dim, ta, tb, d_tiled.as<float>(), s);
                // same scale; the moved row's rounding error is already part of half_dx_*
                if (half_valid && half_n == n) launch_tile_rows_h(v, last, (uint32_t)dim, ta, tb, half_sx(), d_tiled_h.p, s);
            });
    }
    if (half_n == n) {
        half_n = last;
    } else {  // the mirror was behind the table: rebuilt in full by its next use (the moved row's error was never measured)
        half_valid = false;
        half_n = 0;
    }
    if (i8_valid) {  // (f32 rows only) the moved row's and the removed row's tiles, codes and constants
        if (i8_n == n) {
            for (uint64_t t : {i / 16, last / 16})
                launch_tile_rows_i8(d_rows.as<float>(), last, (uint32_t)dim, t, t + 1, d_mu_i8.as<float>(), i8_l1, i8_l2, d_tiled_i8.p,
                                    d_rowc_i8.as<float>(), s, dist == 1 ? d_sq.as<float>() : nullptr);  // (d_sq[i] already holds the moved row's)
            i8_n = last;
        } else {
            i8_valid = false;  // rows were added since the last search: rebuilt by the next one
            i8_n = 0;
        }
    }
    VDB_SYNC(s);
    {
        std::lock_guard<std::mutex> g(host_mu);
        if (host_valid && !elem_u8) {
            if (i < last) std::memcpy(h_rows.data() + i * dim, h_rows.data() + last * dim, dim * sizeof(float));
            h_rows.resize(last * dim);
        }
    }
    if (i < last) h_sq[i] = h_sq[last];
    h_sq.resize(last);
    n = last;
    rows_h_n = 0;  // (rebuilt by the next walk that uses it)
    rows_q8_n = 0;
    // xsq_max stays an upper bound (certification only needs a bound)
}

// ---- split-bf16 mirror, on first need ----------------------------------------------------------------
// A mirror that cannot be allocated is not an error of the search that wanted it: the tier is left to the next one (8-bit -> fp16 ->
// split-bf16 -> exact scan over the rows themselves) and the allocation is not tried again until the table changes.
bool Index::ensure_tiled(Workspace &ws) {
    std::lock_guard<std::mutex> g(tiled_mu);
    if (tiled_built || n == 0) return true;
    if (tiled_failed_n == n) return false;
    const uint64_t tiles = ((n + 15) / 16 + 11) / 12 * 12;  // whole 64-row items (k_flat_mfma) and whole 2/3-tile units (k_flat_gemm)
    const uint64_t tile_bytes = 16 * size_t(mfma_dim_pad((uint32_t)dim)) * sizeof(float);
    try {
        d_tiled.reserve(tiles * tile_bytes);
    } catch (const AllocError &) {
        tiled_failed_n = n;
        mirror_alloc_failures += 1;
        return false;
    }
    for_tile_chunks(ws, 0, tiles, n, [&](const float *v, uint64_t ta, uint64_t tb, uint64_t, uint64_t) {
        launch_tile_rows(v, n, (uint32_t)dim, ta, tb, d_tiled.as<float>(), ws.stream);
    });
    VDB_SYNC(ws.stream);
    tiled_built = true;
    return true;
}

uint64_t Index::hbm_bytes_per_row() const {
    uint64_t b = dim * elem_size() + sizeof(float);  // VecSet row + dist_cache entry
    if (mfma_supported((uint32_t)dim)) {
        if (tiled_built) b += uint64_t(mfma_dim_pad((uint32_t)dim)) * sizeof(float);
        if (half_valid) b += uint64_t(mfma_dim_pad((uint32_t)dim)) * sizeof(uint16_t);
    }
    if (i8_valid) b += uint64_t(mfma_dim_pad((uint32_t)dim)) + 2 * sizeof(float);
    if (rows_h_n) b += dim * sizeof(uint16_t);
    if (rows_q8_n) b += dim + 2 * sizeof(float);
    if (pq.present) b += pq.enc_dim * (pq.codes_t_valid ? 2 : 1);
    if (hnsw.present) b += hnsw.max_m0 * sizeof(uint32_t) + sizeof(uint32_t);
    return b;
}

// ---- scaled fp16 mirror (k_half.hip) ---------------------------------------------------------------
// Called with d_rows, d_sq and xsq_max valid for n_new rows; rows [n_old, n_new) are new.  The scale follows the
// largest row norm: when that grows past the current scale's range the whole mirror is rewritten (one extra bit of
// headroom, so this happens at most once per doubling of the largest norm).
void Index::half_refresh(Workspace &ws, uint64_t n_old, uint64_t n_new) {
    if (!gemm_f16_supported((uint32_t)dim)) return;
    if (!(xsq_max >= 0x1p-80f && xsq_max <= 0x1p80f)) {  // all-zero or extreme data: the split-bf16 / exact paths serve it
        half_valid = false;
        return;
    }
    hipStream_t s = ws.stream;
    int e = 0;
    (void)std::frexp(xsq_max, &e);                           // xsq_max = m * 2^e, m in [0.5, 1)  =>  |x| < 2^ceil(e/2)
    const int need = e >= 0 ? (e + 1) / 2 : -((-e) / 2);
    const bool rebuild = !half_valid || need > half_exp;
    const uint64_t tiles_new = ((n_new + 15) / 16 + 11) / 12 * 12;  // whole units of k_flat_gemm (TW = 2 or 3)
    const uint64_t tile_bytes = 16 * size_t(mfma_dim_pad((uint32_t)dim)) * sizeof(uint16_t);
    uint64_t t0 = n_old / 16, r0 = n_old;
    if (rebuild) {
        half_exp = need + 1;
        half_dx_abs = half_dx_rel = 0.0f;
        t0 = 0;
        r0 = 0;
    }
    d_tiled_h.grow(tiles_new * tile_bytes, t0 * tile_bytes, s);
    d_half_err.reserve(2 * sizeof(uint32_t));
    VDB_HIP(hipMemsetAsync(d_half_err.p, 0, 2 * sizeof(uint32_t), s));
    for_tile_chunks(ws, t0, tiles_new, n_new, [&](const float *v, uint64_t ta, uint64_t tb, uint64_t ra, uint64_t rb) {
        launch_tile_rows_h(v, n_new, (uint32_t)dim, ta, tb, half_sx(), d_tiled_h.p, s);
        const uint64_t a = std::max(ra, r0);
        if (rb > a) launch_row_split_err(v, d_sq.as<float>(), a, rb, (uint32_t)dim, half_sx(), d_half_err.as<uint32_t>(), s);
    });
    uint32_t *h = static_cast<uint32_t *>(ws.pinned(2 * sizeof(uint32_t)));
    VDB_HIP(hipMemcpyAsync(h, d_half_err.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    VDB_SYNC(s);
    float e2[2];
    std::memcpy(e2, h, sizeof(e2));
    half_dx_abs = std::max(half_dx_abs, std::sqrt(e2[0]) * 1.001f);  // the kernel's f32 sums: relative error << 1e-3
    half_dx_rel = std::max(half_dx_rel, std::sqrt(e2[1]) * 1.001f);
    half_valid = true;
}

// ---- centred 8-bit mirror (k_i8.hip) ---------------------------------------------------------------
bool Index::i8_applicable(uint32_t ksel) const {
    if (elem_u8 || flat_i8_mode == 1 || (dim & 3) != 0 || !gemm8_supported((uint32_t)dim)) return false;
    if (!flat_tail_lb_supported((uint32_t)dim, flat_i8_kprime, ksel) || n <= 64) return false;
    if (!(xsq_max <= 0x1p80f)) return false;  // extreme data: the other tiers' own guards decide
    const uint64_t iq = i8_queries.load(), ir = i8_redo.load();
    return flat_i8_mode == 2 || iq < 1024 || ir * 8 <= iq;
}
bool Index::ensure_i8(Workspace &ws) {
    std::lock_guard<std::mutex> g(i8_mu);
    if (i8_valid && i8_n == n) return true;
    if (i8_failed_n == n) return false;
    try {
        ensure_i8_locked(ws);
    } catch (const AllocError &) {
        d_tiled_i8.release();
        d_rowc_i8.release();
        i8_valid = false;
        i8_n = 0;
        i8_failed_n = n;
        mirror_alloc_failures += 1;
        return false;
    }
    return true;
}
void Index::ensure_i8_locked(Workspace &ws) {
    hipStream_t s = ws.stream;
    const uint32_t d = (uint32_t)dim;
    const uint64_t tiles = ((n + 15) / 16 + 11) / 12 * 12;  // whole units of k_flat_gemm8
    const uint64_t tile_bytes = 16 * size_t(mfma_dim_pad(d));
    const bool rebuild = !i8_valid || i8_n > n || n >= 2 * i8_mu_rows;
    const float *xsq_cos = dist == 1 ? d_sq.as<float>() : nullptr;  // Cosine: the mirror of the UNIT rows (k_i8.hip)
    uint64_t t0 = rebuild ? 0 : i8_n / 16;
    d_tiled_i8.grow(tiles * tile_bytes, t0 * tile_bytes, s);
    d_rowc_i8.grow(tiles * 16 * 2 * sizeof(float), t0 * 16 * 2 * sizeof(float), s);
    if (rebuild) {
        // mu = mean of a row sample; rho = typical |dx| / |x_c| of that sample -> l1 = rho, l2 = 1 / rho (k_i8.hip: any
        // positive pair is valid, this one is tight for queries that look like rows)
        d_mu_i8.reserve(size_t(d) * sizeof(float));
        ws.dense.reserve(size_t(I8_MEAN_CHUNKS) * d * sizeof(float) + 2 * 16384 * sizeof(float));
        launch_i8_col_mean(d_rows.as<float>(), n, d, ws.dense.as<float>(), d_mu_i8.as<float>(), s, xsq_cos);
        const uint64_t n_s = std::min<uint64_t>(n, 16384), stride = n / n_s, n_s16 = (n_s + 15) / 16 * 16;
        float *d_stats = ws.dense.as<float>() + size_t(I8_MEAN_CHUNKS) * d;
        launch_i8_row_stats(d_rows.as<float>(), n, d, d_mu_i8.as<float>(), n_s, stride, d_stats, s, xsq_cos);
        std::vector<float> st(2 * n_s16), mu(d);
        VDB_HIP(hipMemcpyAsync(st.data(), d_stats, st.size() * sizeof(float), hipMemcpyDeviceToHost, s));
        VDB_HIP(hipMemcpyAsync(mu.data(), d_mu_i8.p, d * sizeof(float), hipMemcpyDeviceToHost, s));
        VDB_SYNC(s);
        double e2 = 0, xs = 0, m2 = 0;
        for (uint64_t i = 0; i < n_s; i++)
            if (std::isfinite(st[2 * i]) && std::isfinite(st[2 * i + 1])) {
                e2 += st[2 * i];
                xs += st[2 * i + 1];
            }
        for (float v : mu) m2 += double(v) * v;
        float rho = xs > 0 ? (float)std::sqrt(e2 / xs) : 0.01f;
        if (!(rho >= 1e-4f)) rho = 1e-4f;  // exactly representable rows: keep the split finite
        if (rho > 0.5f) rho = 0.5f;
        i8_l1 = rho;
        i8_l2 = 1.0f / rho;
        i8_mu_norm = (float)std::sqrt(m2) * 1.001f;
        i8_mu_rows = n;
    }
    launch_tile_rows_i8(d_rows.as<float>(), n, d, t0, tiles, d_mu_i8.as<float>(), i8_l1, i8_l2, d_tiled_i8.p, d_rowc_i8.as<float>(), s, xsq_cos);
    VDB_SYNC(s);
    i8_n = n;
    i8_valid = true;
}

bool Index::i8_defers_half() const {
    return !elem_u8 && flat_i8_mode != 1 && (dim & 3) == 0 && gemm8_supported((uint32_t)dim);
}
bool Index::ensure_half(Workspace &ws) {
    std::lock_guard<std::mutex> g(half_mu);
    if (half_n != n) {
        if (half_failed_n == n) return false;
        try {
            half_refresh(ws, half_n > n ? 0 : half_n, n);
            half_n = n;
        } catch (const AllocError &) {  // (DevBuf::grow allocates before it frees: the old mirror, if any, is intact but stale)
            d_tiled_h.release();
            half_valid = false;
            half_n = 0;
            half_failed_n = n;
            mirror_alloc_failures += 1;
            return false;
        }
    }
    return half_valid;
}

void Index::prepare_flat(bool all_tiers) {
    if (n == 0) return;
    use_device();
    WsLease ws(*this);
    if (!mfma_supported((uint32_t)dim) || elem_u8) return;  // the exact scan needs nothing beyond the rows
    bool first = false;
    if (i8_defers_half() && n > 64) first = ensure_i8(*ws);
    if (!first || all_tiers) first = ensure_half(*ws) || first;
    if (!first || all_tiers) (void)ensure_tiled(*ws);
}

bool Index::ensure_rows_h(Workspace &ws) {
    if (elem_u8 || dim % 64 != 0 || dim > 4096 || n == 0 || !ensure_half(ws)) return false;
    std::lock_guard<std::mutex> g(rows_h_mu);
    if (rows_h_n == n && rows_h_exp == half_exp) return true;
    if (rows_h_failed_n == n) return false;
    uint64_t r0 = rows_h_n;
    if (rows_h_exp != half_exp || rows_h_n > n) r0 = 0;
    try {
        d_rows_h.grow(n * dim * sizeof(uint16_t) + 16, r0 * dim * sizeof(uint16_t), ws.stream);
    } catch (const AllocError &) {  // the image is an accelerator (pre-passes, key refinement): its callers go on without it
        d_rows_h.release();
        rows_h_n = 0;
        rows_h_failed_n = n;
        mirror_alloc_failures += 1;
        return false;
    }
    launch_rows_to_half(d_rows.as<float>() + r0 * dim, (n - r0) * dim, half_sx(), d_rows_h.as<uint16_t>() + r0 * dim, ws.stream);
    VDB_SYNC(ws.stream);
    rows_h_n = n;
    rows_h_exp = half_exp;
    return true;
}

bool Index::ensure_rows_q8(Workspace &ws) {
    if (elem_u8 || dim % 64 != 0 || dim > 1024 || n == 0) return false;  // (dim * 127^2 < 2^24: the integer sums are exact as f32)
    std::lock_guard<std::mutex> g(rows_q8_mu);
    if (rows_q8_n == n) return true;
    const uint64_t r0 = rows_q8_n > n ? 0 : rows_q8_n;
    d_rows_q8.grow(n * dim + 16, r0 * dim, ws.stream);
    d_q8_scale.grow((n + 64) * sizeof(float), r0 * sizeof(float), ws.stream);
    d_q8_err.grow((n + 64) * sizeof(float), r0 * sizeof(float), ws.stream);
    launch_rows_to_q8(d_rows.as<float>() + r0 * dim, n - r0, (uint32_t)dim, d_rows_q8.as<int8_t>() + r0 * dim, d_q8_scale.as<float>() + r0,
                      d_q8_err.as<float>() + r0, ws.stream);
    VDB_SYNC(ws.stream);
    rows_q8_n = n;
    return true;
}

// ---- timing hooks ------------------------------------------------------------------------------
void Index::prof_begin(Workspace &ws, const char *name, double bytes) {
    if (!prof_on) return;
    if (ws.ev_used == ws.ev_pool.size()) {
        hipEvent_t a, b;
        VDB_HIP(hipEventCreate(&a));
        VDB_HIP(hipEventCreate(&b));
        ws.ev_pool.emplace_back(a, b);
    }
    ws.pending.push_back({name, ws.ev_used, bytes});
    VDB_HIP(hipEventRecord(ws.ev_pool[ws.ev_used].first, ws.stream));
}
void Index::prof_end(Workspace &ws) {
    if (!prof_on) return;
    VDB_HIP(hipEventRecord(ws.ev_pool[ws.ev_used].second, ws.stream));
    ws.ev_used++;
}
void Index::prof_collect(Workspace &ws) {
    if (ws.pending.empty()) return;
    std::lock_guard<std::mutex> g(prof_mu);
    for (auto &p : ws.pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ws.ev_pool[p.ev].first, ws.ev_pool[p.ev].second) == hipSuccess) {
            auto &e = prof[p.name];
            e.ms += ms;
            e.launches += 1;
            e.bytes += p.bytes;
        }
    }
    ws.pending.clear();
    ws.ev_used = 0;
}

// the two search-time readers of the row-major rows, by element type
void Index::scan_rows(uint64_t nrows, uint32_t d, const float *Q, uint32_t nq, int metric, const float *xsq, const float *qsq,
                      float *out, uint64_t ld, bool use_lds, hipStream_t s) const {
    if (elem_u8)
        launch_scan_exact_u8(d_rows.as<uint8_t>(), nrows, d, Q, nq, metric, xsq, qsq, out, ld, s);
    else
        launch_scan_exact(d_rows.as<float>(), nrows, d, Q, nq, metric, xsq, qsq, out, ld, use_lds, s);
}
void Index::rerank_rows(uint32_t d, const float *Q, uint32_t nq, int metric, const float *xsq, const float *qsq, const uint64_t *cand,
                        uint64_t *out, uint32_t ncand, uint32_t ldc, hipStream_t s) const {
    if (elem_u8)
        launch_rerank_u8(d_rows.as<uint8_t>(), d, Q, nq, metric, xsq, qsq, cand, out, ncand, ldc, s);
    else
        launch_rerank(d_rows.as<float>(), d, Q, nq, metric, xsq, qsq, cand, out, ncand, ldc, s);
}

// ---- Flat: exact scan path -----------------------------------------------------------------------
// FlatIndex::knn (flat_index.rs:48-57) for nq queries: strict-order distances for every row, then the
// k smallest pairs by (distance, index).
void Index::flat_exact_device(Workspace &ws, const float *d_q, const float *d_qsq, uint64_t nq, uint32_t ksel,
                              uint64_t k, uint64_t *d_idx, float *d_dist, uint64_t *d_cnt) {
    hipStream_t s = ws.stream;
    const int metric = dist == 0 ? MET_L2_DIRECT : MET_COSINE;
    const uint64_t ld = (n + 63) & ~63ull;
    const uint32_t nl = topk_num_lists(n);
    const uint32_t cap = topk_capacity(ksel);
    constexpr uint32_t BQ = 8;
    ws.dense.reserve(size_t(BQ) * ld * sizeof(float));
    ws.lists.reserve(size_t(BQ) * nl * cap * sizeof(uint64_t));
    ws.keys_c.reserve(size_t(BQ) * cap * sizeof(uint64_t));
    // Small corpus, many queries (tables of a few thousand rows, centroid sets): one thread per (query, row) pair
    // folds in reference order and a wave per query selects -- three launches for the whole batch.  The scan kernel
    // below assigns a thread per ROW and walks 8 queries per launch: 4 workgroups for 1000 rows (measured 119 us per
    // 8 queries, i.e. launch- and latency-bound).
    if (n <= 8192 && nq >= 32) {
        constexpr uint64_t QCH = 4096;  // queries per round: QCH x ld pair keys twice
        const uint64_t qch = std::min<uint64_t>(nq, QCH);
        ws.keys_a.reserve(qch * ld * sizeof(uint64_t));
        ws.keys_b.reserve(qch * ld * sizeof(uint64_t));
        ws.keys_c.reserve(qch * cap * sizeof(uint64_t));
        for (uint64_t q0 = 0; q0 < nq; q0 += qch) {
            const uint32_t nb = (uint32_t)std::min<uint64_t>(qch, nq - q0);
            launch_iota_keys(ws.keys_a.as<uint64_t>(), nb, (uint32_t)n, (uint32_t)ld, s);
            prof_begin(ws, "flat_exact", double(n) * dim * sizeof(float));
            rerank_rows((uint32_t)dim, d_q + q0 * dim, nb, metric, d_sq.as<float>(),
                          d_qsq ? d_qsq + q0 : nullptr, ws.keys_a.as<uint64_t>(), ws.keys_b.as<uint64_t>(), (uint32_t)n,
                          (uint32_t)ld, s);
            prof_end(ws);
            launch_topk_merge(ws.keys_b.as<uint64_t>(), 1, (uint32_t)ld, nb, ksel, ws.keys_c.as<uint64_t>(), s);
            launch_finalize(ws.keys_c.as<uint64_t>(), cap, nb, ksel, (uint32_t)k, id_offset, d_idx + q0 * k, d_dist + q0 * k,
                            d_cnt + q0, s);
        }
        return;
    }
    const bool use_lds = n >= 4096;
    for (uint64_t q0 = 0; q0 < nq; q0 += BQ) {
        uint32_t nb = (uint32_t)std::min<uint64_t>(BQ, nq - q0);
        prof_begin(ws, "flat_exact", double(n) * dim * sizeof(float));
        scan_rows(n, (uint32_t)dim, d_q + q0 * dim, nb, metric, d_sq.as<float>(),
                          d_qsq ? d_qsq + q0 : nullptr, ws.dense.as<float>(), ld, use_lds, s);
        prof_end(ws);
        launch_topk_dense(ws.dense.as<float>(), ld, n, nb, ksel, ws.lists.as<uint64_t>(), s);
        launch_topk_merge(ws.lists.as<uint64_t>(), nl, cap, nb, ksel, ws.keys_c.as<uint64_t>(), s);
        launch_finalize(ws.keys_c.as<uint64_t>(), cap, nb, ksel, (uint32_t)k, id_offset, d_idx + q0 * k, d_dist + q0 * k,
                        d_cnt + q0, s);
    }
}

// ---- Flat, k > 1024: exact distances of every row, full sort of the (distance, index) pairs -------------------
void Index::flat_sorted_device(Workspace &ws, const float *d_q, uint64_t nq, uint64_t ksel, uint64_t k, uint64_t *d_idx,
                               float *d_dist, uint64_t *d_cnt) {
    hipStream_t s = ws.stream;
    VDB_REQUIRE(k < (1ull << 31), "flat knn: k too large");
    const int metric = dist == 0 ? MET_L2_DIRECT : MET_COSINE;
    const uint64_t ld = (n + 63) & ~63ull;
    constexpr uint32_t BQ = 8;
    size_t tb = sort_pairs_temp_bytes(n);
    ws.qsq.reserve(nq * sizeof(float));
    launch_row_sqnorm(d_q, nq, (uint32_t)dim, ws.qsq.as<float>(), s);
    ws.dense.reserve(size_t(BQ) * ld * sizeof(float));
    ws.keys_a.reserve(n * sizeof(uint64_t));
    ws.keys_b.reserve(n * sizeof(uint64_t));
    ws.lists.reserve(tb + 256);
    if (k > ksel) {
        VDB_HIP(hipMemsetAsync(d_idx, 0, nq * k * sizeof(uint64_t), s));
        VDB_HIP(hipMemsetAsync(d_dist, 0, nq * k * sizeof(float), s));
    }
    for (uint64_t q0 = 0; q0 < nq; q0 += BQ) {
        uint32_t nb = (uint32_t)std::min<uint64_t>(BQ, nq - q0);
        scan_rows(n, (uint32_t)dim, d_q + q0 * dim, nb, metric, d_sq.as<float>(),
                          ws.qsq.as<float>() + q0, ws.dense.as<float>(), ld, n >= 4096, s);
        for (uint32_t b = 0; b < nb; b++) {
            launch_sort_pairs(ws.dense.as<float>() + b * ld, n, ws.keys_a.as<uint64_t>(), ws.keys_b.as<uint64_t>(),
                              ws.lists.p, tb, s);
            launch_finalize(ws.keys_b.as<uint64_t>(), (uint32_t)n, 1, (uint32_t)ksel, (uint32_t)k, id_offset,
                            d_idx + (q0 + b) * k, d_dist + (q0 + b) * k, d_cnt + q0 + b, s);
        }
    }
}

// ---- Flat: small table, few queries -> one launch (k_small.hip) ------------------------------------------------------------
bool Index::flat_small_applies(uint64_t nq, uint64_t k) const {
    if (flat_small_mode == 1 || elem_u8 || n == 0 || !flat_small_supported(n, (uint32_t)dim, nq, k)) return false;
    if (flat_small_mode == 2) return true;
    if (flat_mode != 0) return false;
    if (n <= flat_small_max_rows) return nq < 32;  // below the MFMA shortlist's domain the alternative is the five-launch scan
    // Beyond it the one-launch kernel competes with the MFMA pipeline, whose cost is a fixed ~100 us of dependent launches plus
    // ~0.3 us per 1000 rows, while this kernel re-reads the rows per query: ~25 us + ~0.9 us per 1000 (row, query) pairs
    // (tools/probe_small_vs_mfma.py: one query wins up to ~125k rows of 960 columns, four queries up to ~23k).  Rows of other
    // widths scale both sides alike.
    return mfma_supported((uint32_t)dim) ? double(n) * (0.9 * double(nq) - 0.3) < 75000.0 : nq < 32 && n <= (1u << 17);
}
void Index::flat_small_device(Workspace &ws, const float *q, uint64_t nq, uint64_t k, uint64_t *o_idx, float *o_dist, uint64_t *o_cnt) {
    hipStream_t s = ws.stream;
    const uint32_t ksel = (uint32_t)std::min<uint64_t>(k, n);
    ws.small_part.reserve(flat_small_part_keys(n, nq, ksel, num_cu) * sizeof(uint64_t));
    if (ws.small_cnt.cap < 64 * sizeof(uint32_t)) {
        ws.small_cnt.reserve(64 * sizeof(uint32_t));
        VDB_HIP(hipMemsetAsync(ws.small_cnt.p, 0, 64 * sizeof(uint32_t), s));  // the kernel leaves them zero
    }
    FlatSmallArgs a{};
    a.X = d_rows.as<float>();
    a.n = n;
    a.dim = (uint32_t)dim;
    a.Q = q;
    a.metric = dist == 0 ? MET_L2_DIRECT : MET_COSINE;
    a.xsq = d_sq.as<float>();
    a.part = ws.small_part.as<uint64_t>();
    a.counter = ws.small_cnt.as<uint32_t>();
    a.ksel = ksel;
    a.kstride = (uint32_t)k;
    a.id_offset = id_offset;
    a.out_idx = o_idx;
    a.out_dist = o_dist;
    a.out_count = o_cnt;
    prof_begin(ws, "flat_small", double(nq) * double(n) * dim * sizeof(float));
    launch_flat_small(a, (uint32_t)nq, num_cu, s);
    prof_end(ws);
}

// ---- Flat: full pipeline ---------------------------------------------------------------------------
void Index::flat_knn_device(Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t *d_idx,
                            float *d_dist, uint64_t *d_cnt, bool allow_half, uint32_t kprime_min, bool allow_i8, const float *d_dk_hint) {
    FlatPending p;
    flat_knn_enqueue(ws, d_q, nq, k, d_idx, d_dist, d_cnt, allow_half, kprime_min, p, allow_i8, d_dk_hint);
    flat_knn_finish(ws, p);
}

// Everything of a Flat call up to (not including) the host's look at the certification flags.  p.active on return: the MFMA
// pipeline is enqueued on ws.stream and flat_knn_finish must follow (same workspace); otherwise the call took one of the
// synchronous-by-nature paths (small table, exact scan, k > 1024) and is enqueued in full -- nothing left but the stream sync.
void Index::flat_knn_enqueue(Workspace &ws, const float *d_q, uint64_t nq, uint64_t k, uint64_t *d_idx, float *d_dist,
                             uint64_t *d_cnt, bool allow_half, uint32_t kprime_min, FlatPending &p, bool allow_i8, const float *d_dk_hint) {
    hipStream_t s = ws.stream;
    p = FlatPending{};
    if (nq == 0) return;
    if (k == 0 || n == 0) {  // ResultSet::new(0) rejects everything; empty VecSet -> empty result
        VDB_HIP(hipMemsetAsync(d_cnt, 0, nq * sizeof(uint64_t), s));
        return;
    }
    if (flat_small_applies(nq, k)) {
        flat_small_device(ws, d_q, nq, k, d_idx, d_dist, d_cnt);
        return;
    }
    const uint64_t ksel64 = std::min<uint64_t>(k, n);
    if (ksel64 > 1024) {  // beyond the register-resident select: exact scan + full radix sort per query
        flat_sorted_device(ws, d_q, nq, ksel64, k, d_idx, d_dist, d_cnt);
        return;
    }
    const uint32_t ksel = (uint32_t)ksel64;
    if (k > ksel) {  // slots beyond min(k, len) are defined (zero) but not counted
        VDB_HIP(hipMemsetAsync(d_idx, 0, nq * k * sizeof(uint64_t), s));
        VDB_HIP(hipMemsetAsync(d_dist, 0, nq * k * sizeof(float), s));
    }
    ws.qsq.reserve(nq * sizeof(float));

    uint32_t kprime = std::max<uint32_t>(32, 2 * ksel);
    // a redo of queries the fp16 pass could not certify keeps that pass's (longer) shortlist: tiny margins -- clusters of
    // near-duplicates -- need rows, not precision, and the split-bf16 tier should never certify less than the tier before it
    if (kprime_min > kprime && kprime_min <= 1024 && n > kprime_min) kprime = kprime_min;
    const int cosine = dist == 1 ? 1 : 0;
    bool mfma = mfma_supported((uint32_t)dim) && kprime <= 1024 && n > kprime &&
                (flat_mode == 2 || (flat_mode == 0 && n >= 16384));
    if (flat_mode == 1) mfma = false;
    if (!mfma) {
        launch_row_sqnorm(d_q, nq, (uint32_t)dim, ws.qsq.as<float>(), s);
        flat_exact_device(ws, d_q, ws.qsq.as<float>(), nq, ksel, k, d_idx, d_dist, d_cnt);
        return;
    }

    // --- MFMA shortlist -> exact re-rank -> certification --------------------------------------
    // phase 1: keys of a strided row sample for every query -> tau[q] = r-th smallest sampled key, an upper bound of
    //          the r-th smallest key over all rows (the sample is a subset of the rows; mfma_sample_plan picks r)
    // phase 2: one corpus pass per 128 queries (k_flat_gemm; calls of at most 64 queries without an fp16 mirror: per
    //          2 x 32 queries, k_flat_mfma) that parks only the keys <= tau[q] (expected ~r * step ~ 1000 hits)
    // phase 3: shortlist = k' smallest parked pairs per query -> exact re-rank -> top-k -> certification
    const uint64_t bq = mfma_batch((uint32_t)dim);  // queries per workgroup batch (32, or 16 for 1024 < dim <= 2048)
    // First pass with fp16 operands (k_half.hip): half the HBM bytes and a third of the matrix work per row, coarser
    // keys -> a longer shortlist and a wider certification margin; what it cannot certify is redone below with the
    // split-bf16 operands.  Switched off (auto mode) once more than 1/8 of the queries had to be redone.  When it is
    // available it also serves small calls (one 128-query group, mostly padding: the pass is HBM-bound and reads half
    // the bytes of the small-batch kernel's -- 0.36 instead of 0.63 ms per pass at 1M x 960).
    const uint32_t kprime_h = std::max<uint32_t>(64, flat_half_kmul * ksel);
    const uint64_t hq = half_queries.load(), hr = half_redo.load();
    // First pass on the centred 8-bit mirror (k_gemm8.hip): half the bytes of the fp16 pass again; its keys are lower
    // bounds of the distances, its exact stage walks the hit list until the k-th distance is below the next bound
    // (k_flat_tail_lb).  What it cannot close in flat_i8_kprime rows goes through this function again (fp16 pass next).
    // Every tier falls through to the next when its mirror cannot be allocated (ensure_*: false).
    const bool i8_first = allow_i8 && allow_half && kprime_min == 0 && flat_gemm_mode != 1 && (d_dk_hint ? i8_valid.load() : i8_applicable(ksel));
    const bool i8 = i8_first && ensure_i8(ws);
    const bool half_wanted = !i8 && allow_half && flat_half_mode != 1 && kprime_h <= 1024 && n > kprime_h &&
                             (flat_half_mode == 2 || hq < 1024 || hr * 8 <= hq);
    const bool half_ok = half_wanted && ensure_half(ws);
    const bool gemm = i8 || flat_gemm_mode == 2 || (flat_gemm_mode == 0 && (nq > 64 || half_ok));
    const bool half = !i8 && half_ok && gemm;
    if (half) kprime = kprime_h;
    constexpr uint32_t CAND_CAP = 8192;
    const bool i8_second = i8 && d_dk_hint != nullptr;
    if (i8) kprime = i8_second ? CAND_CAP : flat_i8_kprime;  // (second attempt: the exact stage may walk the whole candidate list)
    if (!half && !i8) launch_row_sqnorm(d_q, nq, (uint32_t)dim, ws.qsq.as<float>(), s);  // (the fp16 / 8-bit passes: k_query_prep_*)
    if (!half && !i8 && !ensure_tiled(ws)) {  // no mirror at all: the strict-order scan over the rows themselves
        flat_exact_device(ws, d_q, ws.qsq.as<float>(), nq, ksel, k, d_idx, d_dist, d_cnt);
        return;
    }
    const uint32_t capp = topk_capacity(i8 ? 64u : kprime);  // (8-bit pass: only its sample's lists -- rank <= 64 -- use these buffers)
    const uint32_t capk = topk_capacity(ksel);
    const uint64_t gq = gemm_group();
    const uint64_t ngroups = gemm ? (nq + gq - 1) / gq : 0;
    const uint64_t nq_pad = gemm ? ngroups * gq : (nq + bq - 1) / bq * bq;
    const uint64_t nbatch = nq_pad / bq;
    uint32_t s_step = 1, s_rank = kprime;  // threshold sample: every s_step-th item, tau = s_rank-th smallest sampled key
    // (8-bit pass: planned for 64 guaranteed hits, ~1000 expected -- its exact stage takes tau itself as the bound of everything
    // outside the hit list, so a list shorter than flat_i8_kprime is no failure)
    mfma_sample_plan(n, i8 ? 64u : kprime, &s_step, &s_rank, i8 ? flat_i8_hits : 1024u);
    // 8-bit pass with many sampled units: the sample kernel hands the selection ONE value per (query, unit), the unit's smallest key --
    // 48 x fewer values to write and select from (1M rows: 31 MB -> 0.65 MB per 1000 queries); k_gemm8.hip, launch_flat_gemm8_sample
    const bool unit_min = i8 && flat_i8_unit_min != 1 && gemm8_sample_units(n, s_step) >= (flat_i8_unit_min == 2 ? 2ull : 16ull) * s_rank;  // (2: tests force it on short samples)
    const uint64_t n_s = i8 ? (unit_min ? gemm8_sample_units(n, s_step) : gemm8_sample_rows(n, s_step))
                            : (gemm ? gemm_sample_rows(n, s_step) : mfma_sample_rows(n, s_step));
    const uint64_t ld_s = (n_s + 63) & ~63ull;
    const uint32_t nl_s = topk_num_lists(n_s);
    const size_t qf = mfma_qfrag_floats((uint32_t)dim);
    ws.qfrag.reserve(nbatch * qf * sizeof(float));
    ws.dense.reserve(nq_pad * ld_s * sizeof(float));
    ws.lists.reserve(std::max<size_t>(nq_pad * nl_s * capp, nq_pad * size_t(CAND_CAP)) * sizeof(uint64_t));
    ws.keys_a.reserve(nq_pad * capp * sizeof(uint64_t));  // approximate shortlist, sorted
    ws.keys_b.reserve(nq_pad * capp * sizeof(uint64_t));  // exact keys of the shortlist, unsorted
    ws.keys_c.reserve(nq_pad * capk * sizeof(uint64_t));  // exact top-k, sorted
    const size_t sync_words = mfma_sync_words((uint32_t)nbatch, num_cu);
    ws.misc.reserve(nq_pad * (sizeof(float) + sizeof(uint32_t)) + sync_words * sizeof(uint32_t));  // tau | hit counters | rendezvous
    float *d_tau = ws.misc.as<float>();
    uint32_t *d_hits = reinterpret_cast<uint32_t *>(d_tau + nq_pad);
    if (!gemm) launch_mfma_pack_queries(d_q, (uint32_t)nq, (uint32_t)nq_pad, (uint32_t)dim, ws.qfrag.as<float>(), s);
    const float *xt = half ? d_tiled_h.as<float>() : d_tiled.as<float>();
    float *d_qscale = nullptr, *d_qmul = nullptr, *d_qerr = nullptr, *d_qoff = nullptr;
    if (gemm) {
        ws.qfrag_g.reserve(nq_pad * size_t(mfma_dim_pad((uint32_t)dim)) * sizeof(float));
        if (i8) {
            ws.qaux.reserve(3 * nq_pad * sizeof(float));
            d_qscale = ws.qaux.as<float>();
            d_qoff = d_qscale + nq_pad;
            launch_query_prep_i8(d_q, (uint32_t)nq, (uint32_t)nq_pad, (uint32_t)dim, d_mu_i8.as<float>(), i8_l1, i8_l2, ws.qsq.as<float>(),
                                 d_qscale, d_qoff, d_hits, ws.qfrag_g.p, s, cosine);
        } else if (half) {
            ws.qaux.reserve(3 * nq_pad * sizeof(float));
            d_qscale = ws.qaux.as<float>();
            d_qmul = d_qscale + nq_pad;
            d_qerr = d_qmul + nq_pad;
            launch_query_prep_h(d_q, (uint32_t)nq, (uint32_t)nq_pad, (uint32_t)dim, half_sx(), ws.qsq.as<float>(), d_qscale, d_qmul,
                                d_qerr, d_hits, ws.qfrag_g.p, s);  // norms, scales, rounding errors AND the packed fp16 image
        } else {
            launch_mfma_pack_queries_nh(d_q, (uint32_t)nq, (uint32_t)nq_pad, (uint32_t)dim, 8, ws.qfrag_g.as<float>(), s);
        }
    }
    if (i8_second) {
        // thresholds from the k-th exact distances the first walk left behind (k_redo.hip): no sample, no selection
        launch_i8_tau_from_dk(d_dk_hint, (uint32_t)nq, (uint32_t)nq_pad, d_qoff, ws.qsq.as<float>(), xsq_max, i8_mu_norm, (uint32_t)dim, cosine,
                              d_tau, s);
    } else {
        if (i8)
            launch_flat_gemm8_sample(d_tiled_i8.p, n, (uint32_t)dim, ws.qfrag_g.p, d_qscale, (uint32_t)ngroups, d_rowc_i8.as<float>(), s_step,
                                     ws.dense.as<float>(), ld_s, num_cu, s, unit_min ? 1 : 0);
        else if (gemm)  // the sample through the 128-query kernel too: same arithmetic as the filter, 4x fewer re-reads of the sample
            launch_flat_gemm_sample(xt, n, (uint32_t)dim, ws.qfrag_g.as<float>(), d_qmul, (uint32_t)ngroups,
                                    d_sq.as<float>(), cosine, s_step, ws.dense.as<float>(), ld_s, num_cu, s);
        else
            launch_flat_mfma_sample(d_tiled.as<float>(), n, (uint32_t)dim, ws.qfrag.as<float>(), (uint32_t)nbatch,
                                    d_sq.as<float>(), cosine, s_step, ws.dense.as<float>(), ld_s, num_cu, s);
        if (n_s <= select_tau_max_n()) {  // tau only needs the k'-th smallest sampled key, not a sorted sample shortlist
            launch_select_tau(ws.dense.as<float>(), ld_s, (uint32_t)n_s, (uint32_t)nq_pad, (uint32_t)nq, s_rank, d_tau, s);
        } else {
            // (lists of topk_capacity(s_rank) slots -- NOT of k' slots: a thinned sample's rank is below k', and so is the 8-bit pass's)
            const uint32_t cap_s = topk_capacity(s_rank);
            launch_topk_dense(ws.dense.as<float>(), ld_s, n_s, (uint32_t)nq_pad, s_rank, ws.lists.as<uint64_t>(), s);
            launch_topk_merge(ws.lists.as<uint64_t>(), nl_s, cap_s, (uint32_t)nq_pad, s_rank, ws.keys_a.as<uint64_t>(), s);
            launch_extract_tau(ws.keys_a.as<uint64_t>(), cap_s, (uint32_t)nq_pad, s_rank, d_tau, s);
        }
        // padding queries are zero vectors: under Cosine every row ties at key 0 = tau and would flood the hit buffers of
        // the real queries that share their workgroup batch; tau = -inf lets nothing through
        if (nq_pad > nq && n_s > select_tau_max_n())  // (k_select_tau does it itself)
            VDB_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_tau + nq), (int)0xFF800000u, nq_pad - nq, s));
    }
    uint64_t *d_cand = ws.lists.as<uint64_t>();  // the sample lists are dead now
    uint32_t *d_sync = d_hits + nq_pad;
    if (!half && !i8) VDB_HIP(hipMemsetAsync(d_hits, 0, (nq_pad + sync_words) * sizeof(uint32_t), s));  // (k_query_prep_* zero the counters)
    // algorithmic bytes: one corpus pass (N*d*4) serves 32*share queries (bytes/query = N*d*4 / B)
    // (the fp16 pass streams N*d*2 bytes per 128 queries: its own counter, so that GB/s are the bytes really read)
    const uint64_t hbm_passes = gemm ? ngroups : (nbatch + mfma_share() - 1) / mfma_share();
    // Calls in flight on other workspaces (re-entrant readers, vdb_flat_knn_device_begin): their corpus passes take turns.
    // Each pass wants every CU (one persistent workgroup per CU); two of them resident at once only wait for each other's
    // workgroups, and their HIP-event durations would measure that wait.  The small kernels around the passes still overlap.
    {
        std::lock_guard<std::mutex> g(pass_mu);
        if (pass_ev_valid) VDB_HIP(hipStreamWaitEvent(s, pass_ev, 0));
    }
    prof_begin(ws, i8 ? "flat_i8" : (half ? "flat_half" : "flat_mfma"),
               double(hbm_passes) * double(n) * dim * (i8 ? 1 : (half ? sizeof(uint16_t) : sizeof(float))));
    if (i8)
        launch_flat_gemm8_filter(d_tiled_i8.p, n, (uint32_t)dim, ws.qfrag_g.p, d_qscale, (uint32_t)ngroups, d_rowc_i8.as<float>(), d_tau,
                                 d_cand, d_hits, CAND_CAP, flat_gemm_debug, num_cu, s,
                                 i8_second ? CAND_CAP : std::max<uint32_t>(64u, s_step * s_rank));  // (expected hits per query: sizes the hand-over blocks)
    else if (gemm)
        launch_flat_gemm_filter(xt, n, (uint32_t)dim, ws.qfrag_g.as<float>(), d_qmul, (uint32_t)ngroups,
                                d_sq.as<float>(), cosine, d_tau, d_cand, d_hits, CAND_CAP, flat_gemm_debug, num_cu, s);
    else
        launch_flat_mfma_filter(d_tiled.as<float>(), n, (uint32_t)dim, ws.qfrag.as<float>(), (uint32_t)nbatch,
                                d_sq.as<float>(), cosine, d_tau, d_cand, d_hits, CAND_CAP, d_sync, num_cu, s);
    prof_end(ws);
    {
        std::lock_guard<std::mutex> g(pass_mu);
        if (!pass_ev) VDB_HIP(hipEventCreateWithFlags(&pass_ev, hipEventDisableTiming));
        VDB_HIP(hipEventRecord(pass_ev, s));
        pass_ev_valid = true;
    }
    SplitErr se;
    if (half) {
        se.qerr = d_qerr;
        se.dx_abs = half_dx_abs;
        se.dx_rel = half_dx_rel;
    }
    if (i8) {
        se.qoff = d_qoff;
        se.mu_norm = i8_mu_norm;
    }
    if (i8 || (flat_tail_mode != 1 && !elem_u8 && flat_tail64_supported((uint32_t)dim, kprime, ksel))) {
        FlatTailArgs t{};
        t.cand = d_cand;
        t.cap = CAND_CAP;
        t.cnt = d_hits;
        t.kprime = kprime;
        t.ksel = ksel;
        t.kstride = (uint32_t)k;
        t.X = d_rows.as<float>();
        t.dim = (uint32_t)dim;
        t.Q = d_q;
        t.metric = cosine ? MET_COSINE : MET_L2_DIRECT;
        t.xsq = d_sq.as<float>();
        t.qsq = ws.qsq.as<float>();
        t.n_rows = n;
        t.xsq_max = xsq_max;
        t.xsq_min_pos = xsq_min_pos;
        t.cosine = cosine;
        t.se = se;
        t.id_offset = id_offset;
        // long walks (tight clusters): the keys of the hits are tightened from the row-major fp16 image first (k_redo.hip).  Decided -- and the
        // image built, which may use the pinned block -- before this call takes its flags from that block.
        bool refine = false;
        if (i8 && !i8_second && nq >= 64 && flat_i8_refine != 1) {
            bool on = flat_i8_refine == 2;
            if (flat_i8_refine == 0 && i8_refine_on.load() != 0) on = (i8_refine_calls.fetch_add(1) % 32u) != 31u;  // (the 32nd: a probe without)
            refine = on && ensure_rows_h(ws);
        }
        // (flags, then -- measurement builds of a call, flat_i8_stats -- one word per query of exact-stage statistics)
        const size_t st_off = (nq + 15) & ~size_t(15);
        const bool want_stats = i8 && flat_i8_stats;
        t.flags = static_cast<uint8_t *>(ws.pinned(want_stats ? st_off + nq * sizeof(uint32_t) : nq));
        if (want_stats) t.qstat = reinterpret_cast<uint32_t *>(t.flags + st_off);
        p.stats = want_stats;
        p.refined = refine;
        if (refine) {
            launch_flat_refine_half(d_rows_h.as<uint16_t>(), (uint32_t)dim, half_sx(), half_dx_abs, half_dx_rel, cosine, d_q, d_sq.as<float>(),
                                    ws.qsq.as<float>(), d_qoff, d_cand, CAND_CAP, d_hits, (uint32_t)nq, CAND_CAP, s);  // (a list holds up to CAND_CAP hits; the walk selects among all of them)
            i8_refine_queries += nq;
        }
        t.out_idx = d_idx;
        t.out_dist = d_dist;
        t.out_count = d_cnt;
        t.tau = d_tau;
        if (i8 && i8_second && nq <= 96 && flat_i8_full != 1) {
            // a handful of queries on their second attempt: all their candidates at once instead of a chain of 63-row rounds (k_exact.hip)
            ws.keys_b.reserve(nq * size_t(CAND_CAP) * sizeof(uint64_t));
            launch_flat_full_lb(t, (uint32_t)nq, ws.keys_b.as<uint64_t>(), ws.keys_c.as<uint64_t>(), s);
        } else if (i8)
            launch_flat_tail_lb(t, (uint32_t)nq, s);
        else
            launch_flat_tail64(t, (uint32_t)nq, s);
    } else {
        launch_topk_merge_counted(d_cand, CAND_CAP, d_hits, (uint32_t)nq, kprime, ws.keys_a.as<uint64_t>(), s);
        rerank_rows((uint32_t)dim, d_q, (uint32_t)nq, cosine ? MET_COSINE : MET_L2_DIRECT, d_sq.as<float>(),
                      ws.qsq.as<float>(), ws.keys_a.as<uint64_t>(), ws.keys_b.as<uint64_t>(), kprime, capp, s);  // pads its rows
        launch_topk_merge(ws.keys_b.as<uint64_t>(), 1, capp, (uint32_t)nq, ksel, ws.keys_c.as<uint64_t>(), s);
        launch_flat_finish(ws.keys_c.as<uint64_t>(), capk, ws.keys_a.as<uint64_t>(), capp, (uint32_t)nq, ksel, (uint32_t)k, kprime,
                           n, ws.qsq.as<float>(), xsq_max, xsq_min_pos, cosine, (uint32_t)dim, se, d_hits, CAND_CAP, id_offset,
                           static_cast<uint8_t *>(ws.pinned(nq)), d_idx, d_dist, d_cnt, s);
    }
    p.active = true;
    p.half = half;
    p.i8 = i8;
    p.i8_second = i8_second;
    p.kprime = kprime;
    p.ksel = ksel;
    p.nq = nq;
    p.k = k;
    p.d_q = d_q;
    p.d_idx = d_idx;
    p.d_dist = d_dist;
    p.d_cnt = d_cnt;
}

// the host's half of a Flat call: wait for the stream, read the certification flags, redo what was not certified
void Index::flat_knn_finish(Workspace &ws, FlatPending &p) {
    hipStream_t s = ws.stream;
    if (!p.active) return;
    p.active = false;
    const bool half = p.half, i8 = p.i8, i8_second = p.i8_second;
    const uint32_t kprime = p.kprime, ksel = p.ksel;
    const uint64_t nq = p.nq, k = p.k;
    const float *d_q = p.d_q;
    uint64_t *d_idx = p.d_idx;
    float *d_dist = p.d_dist;
    uint64_t *d_cnt = p.d_cnt;
    // the flags go straight to pinned host memory (device-visible): no copy kernel between the last kernel and the sync
    const uint8_t *flags = static_cast<const uint8_t *>(ws.pinned(nq));
    VDB_SYNC(s);
    // uncertified queries: gather them, redo them (8-bit pass: once more with thresholds from the k-th distances the first walk found,
    // then the fp16 pass; fp16 pass: through this function again with the split-bf16 operands; split-bf16 pass: 8 per corpus pass with
    // the exact scan), scatter the results
    std::vector<uint64_t> redo;
    for (uint64_t q = 0; q < nq; q++)
        if (flags[q] & 1u) redo.push_back(q);  // (bits 1..7: rounds the 8-bit pass's exact stage walked)
    if (half) {
        half_queries += nq;
        half_redo += redo.size();
    }
    if (i8 && i8_second) {
        i8_second_queries += nq;
        i8_second_redo += redo.size();
    }
    const bool try_second = i8 && !i8_second && flat_i8_second != 1 && !redo.empty();
    if (i8 && !i8_second) {
        i8_queries += nq;
        if (!try_second) i8_redo += redo.size();  // (with a second attempt: what THAT passes on, counted below)
        if (flat_i8_refine == 0 && nq >= 64) {
            // auto rule of the fp16 refinement: it costs half the f32 bytes of every hit (~0.5 ms per 1000 queries), a round of the walk
            // ~0.1 ms per 1000 queries -- on when the walks average 6 rounds, off again when a probe call without it averages under 4
            // (the rounds come with the flags: bits 1..7)
            uint64_t rs = 0;
            for (uint64_t q = 0; q < nq; q++) rs += std::min<uint32_t>(flags[q] >> 1, 32u);
            const double mean_rounds = double(rs) / double(nq);
            if (!p.refined) {
                const int was = i8_refine_on.load();
                const int now = was ? (mean_rounds >= 4.0 ? 1 : 0) : (mean_rounds >= 6.0 ? 1 : 0);
                if (now != was) {
                    i8_refine_on = now;
                    i8_refine_calls = 0;
                }
            }
        }
        if (flat_i8_stats) {
            const uint32_t *qs = reinterpret_cast<const uint32_t *>(flags + ((nq + 15) & ~size_t(15)));
            uint64_t hs = 0, hm = 0;
            for (uint64_t q = 0; q < nq; q++) {
                const uint32_t r = qs[q] & 0xFFu, h = qs[q] >> 8;
                i8_rounds_hist[r < 8 ? r : 8] += 1;
                hs += h;
                hm = std::max<uint64_t>(hm, h);
            }
            i8_hits_sum += hs;
            i8_stat_queries += nq;
            uint64_t cur = i8_hits_max.load();
            while (hm > cur && !i8_hits_max.compare_exchange_weak(cur, hm)) {
            }
        }
    }
    if (redo.empty()) return;
    if (!half && !i8) fallback_count += redo.size();
    const uint64_t nr = redo.size();
    DevBuf rx, rq, rqs, ri, rd, rc, rdk;  // rare: allocated on demand
    rx.reserve(nr * sizeof(uint64_t));
    rq.reserve(nr * dim * sizeof(float));
    rqs.reserve(nr * sizeof(float));
    ri.reserve(nr * k * sizeof(uint64_t));
    rd.reserve(nr * k * sizeof(float));
    rc.reserve(nr * sizeof(uint64_t));
    VDB_HIP(hipMemcpyAsync(rx.p, redo.data(), nr * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    launch_gather_rows_f32(d_q, rx.as<uint64_t>(), nr, (uint32_t)dim, rq.as<float>(), s);
    launch_gather_rows_f32(ws.qsq.as<float>(), rx.as<uint64_t>(), nr, 1, rqs.as<float>(), s);
    if (try_second) {
        rdk.reserve(nr * sizeof(float));
        launch_gather_dk(d_dist, d_cnt, rx.as<uint64_t>(), nr, (uint32_t)k, ksel, rdk.as<float>(), s);
    }
    VDB_HIP(hipMemsetAsync(ri.p, 0, nr * k * sizeof(uint64_t), s));
    VDB_HIP(hipMemsetAsync(rd.p, 0, nr * k * sizeof(float), s));
    VDB_SYNC(s);  // (`redo` is pageable host memory: the copy above must have read it before it goes out of scope in a nested call's unwinding)
    if (try_second) {
        // the second 8-bit attempt; what it still cannot close goes to the fp16 tier from inside that call
        const uint64_t before = i8_second_redo.load();
        flat_knn_device(ws, rq.as<float>(), nr, k, ri.as<uint64_t>(), rd.as<float>(), rc.as<uint64_t>(), true, 0, true, rdk.as<float>());
        i8_redo += i8_second_redo.load() - before;  // the auto rule counts what left the 8-bit tier for good
    } else if (i8)  // next tier: the fp16 pass (or whatever this index has instead), with its own shortlist rules
        flat_knn_device(ws, rq.as<float>(), nr, k, ri.as<uint64_t>(), rd.as<float>(), rc.as<uint64_t>(), true, 0, false);
    else if (half)
        flat_knn_device(ws, rq.as<float>(), nr, k, ri.as<uint64_t>(), rd.as<float>(), rc.as<uint64_t>(), false, kprime);
    else
        flat_exact_device(ws, rq.as<float>(), rqs.as<float>(), nr, ksel, k, ri.as<uint64_t>(), rd.as<float>(), rc.as<uint64_t>());
    launch_scatter_results(ri.as<uint64_t>(), rd.as<float>(), rc.as<uint64_t>(), rx.as<uint64_t>(), nr, (uint32_t)k, d_idx, d_dist, d_cnt, s);
    VDB_SYNC(s);  // rx..rc are freed on return
}

// ---- the approximate keys of the Flat shortlist pass, for every row ------------------------------------------------
// What the filter kernels compare with tau, produced by the SAME kernel in its dense (sample) mode with unit_step = 1:
// L2Sqr key(r, q) = |x_r|^2 - 2 S~(r, q)  (approximate distance = key + |q|^2);  Cosine key = -S~ / |x_r|  (approximate
// distance = 1 + key / |q|).  tier 0 = fp16 operands (k_flat_gemm<GEMM_F16>), tier 1 = split-bf16 operands.  Also the
// quantities the certification bound is built from: |q|^2 (strict fold), the measured |dq| of the fp16 query image and
// the measured row rounding errors (dx_abs = max |dx_r|, dx_rel = max |dx_r| / |x_r|).  Test / measurement entry point.
void Index::flat_debug_keys(Workspace &ws, const float *d_q, uint64_t nq, int tier, float *h_keys, float *h_qsq, float *h_qerr,
                            float *h_dx) {
    hipStream_t s = ws.stream;
    VDB_REQUIRE(nq >= 1 && nq <= 1024 && n >= 1, "debug keys: 1..1024 queries on a non-empty index");
    VDB_REQUIRE(mfma_supported((uint32_t)dim), "debug keys: the dimension has no MFMA shortlist path");
    const int cosine = dist == 1 ? 1 : 0;
    const uint64_t gq = gemm_group(), ngroups = (nq + gq - 1) / gq, nq_pad = ngroups * gq;
    if (tier == 2) {  // 8-bit operands: keys are lower bounds, D >= key + qoff (h_qerr = qoff; h_dx = l1, l2, |mu|, 0)
        VDB_REQUIRE(!elem_u8 && (dim & 3) == 0 && gemm8_supported((uint32_t)dim), "debug keys: the index has no 8-bit pass");
        VDB_REQUIRE(ensure_i8(ws), "debug keys: the 8-bit mirror could not be allocated");
        const uint64_t n_s8 = gemm8_sample_rows(n, 1), ld8 = (n_s8 + 63) & ~63ull;
        ws.qsq.reserve(nq_pad * sizeof(float));
        ws.qfrag_g.reserve(nq_pad * size_t(mfma_dim_pad((uint32_t)dim)));
        ws.dense.reserve(nq_pad * ld8 * sizeof(float));
        ws.qaux.reserve(2 * nq_pad * sizeof(float));
        ws.misc.reserve((nq_pad + 128) * sizeof(uint32_t));  // (the preparation kernel also zeroes the 128 rendezvous words behind the counters)
        float *d_qs8 = ws.qaux.as<float>(), *d_qoff = d_qs8 + nq_pad;
        launch_query_prep_i8(d_q, (uint32_t)nq, (uint32_t)nq_pad, (uint32_t)dim, d_mu_i8.as<float>(), i8_l1, i8_l2, ws.qsq.as<float>(), d_qs8,
                             d_qoff, ws.misc.as<uint32_t>(), ws.qfrag_g.p, s, cosine);
        launch_flat_gemm8_sample(d_tiled_i8.p, n, (uint32_t)dim, ws.qfrag_g.p, d_qs8, (uint32_t)ngroups, d_rowc_i8.as<float>(), 1,
                                 ws.dense.as<float>(), ld8, num_cu, s);
        VDB_HIP(hipMemcpy2DAsync(h_keys, n * sizeof(float), ws.dense.p, ld8 * sizeof(float), n * sizeof(float), nq, hipMemcpyDeviceToHost, s));
        if (h_qsq) VDB_HIP(hipMemcpyAsync(h_qsq, ws.qsq.p, nq * sizeof(float), hipMemcpyDeviceToHost, s));
        if (h_qerr) VDB_HIP(hipMemcpyAsync(h_qerr, d_qoff, nq * sizeof(float), hipMemcpyDeviceToHost, s));
        VDB_SYNC(s);
        if (h_dx) {
            h_dx[0] = i8_l1;
            h_dx[1] = i8_l2;
            h_dx[2] = i8_mu_norm;
            h_dx[3] = 0.0f;
        }
        return;
    }
    const bool half = tier == 0;
    VDB_REQUIRE(!half || ensure_half(ws), "debug keys: the index holds no fp16 mirror");
    if (!half) VDB_REQUIRE(ensure_tiled(ws), "debug keys: the split-bf16 mirror could not be allocated");
    const uint64_t n_s = gemm_sample_rows(n, 1), ld = (n_s + 63) & ~63ull;
    ws.qsq.reserve(nq_pad * sizeof(float));
    ws.qfrag_g.reserve(nq_pad * size_t(mfma_dim_pad((uint32_t)dim)) * sizeof(float));
    ws.dense.reserve(nq_pad * ld * sizeof(float));
    ws.qaux.reserve(3 * nq_pad * sizeof(float));
    ws.misc.reserve((nq_pad + 128) * sizeof(uint32_t));  // (the preparation kernel also zeroes the 128 rendezvous words behind the counters)
    float *d_qscale = ws.qaux.as<float>(), *d_qmul = d_qscale + nq_pad, *d_qerr = d_qmul + nq_pad;
    if (half) {
        launch_query_prep_h(d_q, (uint32_t)nq, (uint32_t)nq_pad, (uint32_t)dim, half_sx(), ws.qsq.as<float>(), d_qscale, d_qmul, d_qerr,
                            ws.misc.as<uint32_t>(), nullptr, s);  // (the stand-alone packing kernel below: the two must agree)
        launch_pack_queries_h(d_q, (uint32_t)nq, (uint32_t)nq_pad, (uint32_t)dim, 8, d_qscale, ws.qfrag_g.p, s);
    } else {
        launch_row_sqnorm(d_q, nq, (uint32_t)dim, ws.qsq.as<float>(), s);
        launch_mfma_pack_queries_nh(d_q, (uint32_t)nq, (uint32_t)nq_pad, (uint32_t)dim, 8, ws.qfrag_g.as<float>(), s);
    }
    launch_flat_gemm_sample(half ? d_tiled_h.as<float>() : d_tiled.as<float>(), n, (uint32_t)dim, ws.qfrag_g.as<float>(),
                            half ? d_qmul : nullptr, (uint32_t)ngroups, d_sq.as<float>(), cosine, 1, ws.dense.as<float>(), ld, num_cu, s);
    VDB_HIP(hipMemcpy2DAsync(h_keys, n * sizeof(float), ws.dense.p, ld * sizeof(float), n * sizeof(float), nq, hipMemcpyDeviceToHost, s));
    if (h_qsq) VDB_HIP(hipMemcpyAsync(h_qsq, ws.qsq.p, nq * sizeof(float), hipMemcpyDeviceToHost, s));
    if (h_qerr) {
        if (half)
            VDB_HIP(hipMemcpyAsync(h_qerr, d_qerr, nq * sizeof(float), hipMemcpyDeviceToHost, s));
        else
            std::memset(h_qerr, 0, nq * sizeof(float));
    }
    VDB_SYNC(s);
    if (h_dx) {
        h_dx[0] = half ? half_dx_abs : 0.0f;
        h_dx[1] = half ? half_dx_rel : 0.0f;
        h_dx[2] = xsq_max;
        h_dx[3] = xsq_min_pos;
    }
}

}  // namespace vdb
