"""GPU: the error bound E behind the certification of the Flat shortlist is MEASURED, not only
derived.  For every (query, row) pair of adversarial corpora the approximate distance the shortlist kernels rank by
(vdb_flat_shortlist_keys: the production kernel in its dense mode) and the strict-order f32 distance the library returns
are compared with the distance in float64:

        |a(r, q) - d64(r, q)| + |e(r, q) - d64(r, q)| <= E(|x_r|, q)

(a row outside the shortlist has a >= kappa + |q|^2, hence e >= a - |a - d64| - |e - d64| >= kappa + |q|^2 - E)

with E evaluated exactly as flat_certify_flag does (k_exact.hip), at the row's own norm -- the certification evaluates it
at a norm bound that is at least as large, and E grows with the norm.  Both tiers (fp16 operands with their measured
rounding errors, split-bf16 operands with the format constants), both metrics.  The largest observed ratio error / E is
printed: it is the margin the constants 2.5 * (d + 8) * 2^-24, 5e-5 / 6e-5 and 1.001 leave.
"""
import numpy as np
import pytest

from conftest import gist_like

pytestmark = pytest.mark.gpu

U = 2.0 ** -24


def _corpora(dim):
    rng = np.random.default_rng(2026)
    n, nq = 3000, 12
    out = {}
    g = gist_like(n, dim=dim, seed=101)
    out["gistlike"] = (g, gist_like(nq, dim=dim, seed=102))
    # cancellation-heavy: every row and query = one large common vector + a small perturbation, so |x|^2 + |q|^2 - 2 x.q
    # cancels ~5 digits and the distance is tiny against (|x| + |q|)^2
    c = rng.standard_normal(dim).astype(np.float32)
    c *= np.float32(50.0 / np.linalg.norm(c))
    out["cancellation"] = ((c + 1e-2 * rng.standard_normal((n, dim))).astype(np.float32),
                           (c + 1e-2 * rng.standard_normal((nq, dim))).astype(np.float32))
    # mixed magnitudes: row norms over four decades (the fp16 mirror's scale follows the largest; small rows lose bits)
    s = (10.0 ** rng.uniform(-2, 2, size=(n, 1))).astype(np.float32)
    out["mixed_magnitudes"] = ((rng.standard_normal((n, dim)) * s).astype(np.float32) / np.float32(np.sqrt(dim)),
                               (rng.standard_normal((nq, dim)) * 3).astype(np.float32) / np.float32(np.sqrt(dim)))
    # |q| >> sqrt(D_k): queries sit on top of rows of norm ~30
    b = rng.standard_normal((n, dim)).astype(np.float32)
    b *= np.float32(30.0) / np.linalg.norm(b, axis=1, keepdims=True).astype(np.float32)
    out["near_duplicates_large_norm"] = (b, (b[:nq] + 1e-3 * rng.standard_normal((nq, dim))).astype(np.float32))
    return out


def _bound(kind, tier, dim, rx, qn, qe, st):
    """E of flat_certify_flag at row norm rx (float64 arithmetic on the same expression)"""
    if kind == 1:
        if tier == 0:
            qr = qe / qn
            split = (st["dx_rel"] + qr + st["dx_rel"] * qr) * 1.001
        else:
            split = 6e-5
        return 2.5 * (dim + 8) * U + split + 0.0 * rx
    if tier == 0:
        dxa = np.minimum(st["dx_abs"], st["dx_rel"] * rx)
        split = 2.0 * (dxa * qn + rx * qe + dxa * qe) * 1.001
    else:
        split = 5e-5 * rx * qn
    return 2.5 * (dim + 8) * U * (rx + qn) ** 2 + split


@pytest.mark.parametrize("dist,kind", [("l2sqr", 0), ("cosine", 1)])
@pytest.mark.parametrize("dim", [960, 192])
def test_error_bound_holds_with_margin(dist, kind, dim):
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O

    worst = {}
    for name, (base, qs) in _corpora(dim).items():
        ix = vdb.GpuIndex(dim, dist)
        ix.set_param("flat_i8", 1)  # (the 8-bit pass's keys are lower bounds, not estimates: tests/test_flat_i8_gpu.py; off, the fp16 mirror is built at add time)
        ix.batch_add(base)
        x64, q64 = base.astype(np.float64), qs.astype(np.float64)
        rx = np.linalg.norm(x64, axis=1)
        qn = np.linalg.norm(q64, axis=1)
        if kind == 0:
            d64 = ((x64[None, :, :] - q64[:, None, :]) ** 2).sum(axis=2)
        else:
            d64 = 1.0 - (q64 @ x64.T) / (qn[:, None] * rx[None, :])
        # the library's exact values: strict-order fold of the reference, all rows (k = len -> the full-sort path)
        e = np.zeros_like(d64)
        for q in range(len(qs)):
            oi, od = O.flat_knn(base, qs[q], len(base), kind)
            e[q, oi.astype(np.int64)] = od
        gi, gd, _ = ix.flat_knn(qs[:2], len(base))
        for q in range(2):  # (and they ARE the library's values)
            assert np.array_equal(e[q, gi[q].astype(np.int64)], gd[q])
        for tier in (0, 1):
            if tier == 0 and not ix.get_stat("flat_half_valid"):
                continue
            keys, qsq, qerr, st = ix.flat_shortlist_keys(qs, tier)
            if kind == 0:
                a = (keys + qsq[:, None]).astype(np.float64)               # kappa + |q|^2 as the certification forms it
            else:
                a = (np.float32(1.0) + keys / np.sqrt(qsq)[:, None]).astype(np.float64)
            E = np.stack([_bound(kind, tier, dim, rx, qn[q], float(qerr[q]), st) for q in range(len(qs))])
            ra = np.abs(a - d64) / E
            re = np.abs(e - d64) / E
            assert np.isfinite(ra).all() and np.isfinite(re).all(), (name, tier)
            worst[(name, tier)] = (float(ra.max()), float(re.max()))
            assert (ra + re).max() <= 1.0, (name, tier, "key error + fold error exceed the certification bound", (ra + re).max())
        ix.close()
    print(f"\ncertification bound, {dist} dim {dim}: max |err| / E per corpus and tier (approximate key, strict fold)")
    for (name, tier), (ma, me) in sorted(worst.items()):
        print(f"  {name:28s} {'fp16 ' if tier == 0 else 'bf16x3'}  key {ma:.3f}  fold {me:.3f}")
    # the constants must leave real margin on ordinary data, not just hold
    assert max(v[0] for (nm, t), v in worst.items() if nm == "gistlike") < 0.7
