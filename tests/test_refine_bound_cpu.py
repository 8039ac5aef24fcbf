"""The difference-form lower bound behind k_flat_refine_half (lab_1806_vec_db_amd/csrc/k_redo.hip), restated in numpy and checked against float64:
for the image row x~ = fp16(x * sx) / sx with the measured |dx_r| = |x - x~|,
    sqrt(D) = |x - q| >= |x~ - q| - |dx_r|        (triangle inequality)
so  lb = max(0, sqrt(a (1 - (d + 8) u)) - |dx_r|)^2 <= D  for a = fl32(sum (h - q sx)^2) / sx^2 (every term non-negative: a is within (d + 4) u
relative of the real sum whatever the order), and the same for the unit vectors of the Cosine keys with the slack |dx_r| / |x| + (d + 16) u.
The test also pins WHY the form matters: on a tight cluster the bound is within a few per cent of D where a dot-product bound of the same
image (a - E of half_rows.hpp: E ~ 2 |dx||q| + gamma_d (|x| + |q|)^2) is not.  (The reference's distances: src/distance/mod.rs:31-75.)"""
import numpy as np
import pytest

U = np.float32(2.0 ** -24)


def _image(x, sx):
    h = (x.astype(np.float32) * np.float32(sx)).astype(np.float16)  # round to nearest even, as v_cvt_f16_f32
    xt = h.astype(np.float64) / sx
    return h, xt


def _lb_l2(h, q, sx, dxr, dim):
    t = h.astype(np.float32) - (q.astype(np.float32) * np.float32(sx))[None, :]
    a = np.sum(t * t, axis=1, dtype=np.float32) / np.float32(sx * sx)
    sr = np.sqrt(a * (np.float32(1) - np.float32(dim + 8) * U)) * (np.float32(1) - 4 * U) - dxr.astype(np.float32) * np.float32(1.001)
    sr = np.where(sr > 0, sr * (np.float32(1) - 4 * U), np.float32(0))
    return (sr * sr * (np.float32(1) - 4 * U)).astype(np.float32)


@pytest.mark.parametrize("spread", [0.05, 0.15, 1.0])
@pytest.mark.parametrize("dim", [64, 960])
def test_difference_form_is_a_lower_bound_and_tight(dim, spread):
    rng = np.random.default_rng(dim * 7 + int(spread * 100))
    n = 4000
    centre = np.abs(rng.normal(0.07, 0.045, dim))
    x = np.round(np.clip(np.abs(centre + spread * 0.045 * rng.standard_normal((n, dim))), 0, 0.8), 4).astype(np.float32)
    q = np.round(np.clip(np.abs(centre + spread * 0.045 * rng.standard_normal(dim)), 0, 0.8), 4).astype(np.float32)
    sx = 2.0 ** 13  # Index::half_sx(): a power of two that puts the largest element near the top of the fp16 range
    h, xt = _image(x, sx)
    dxr = np.linalg.norm(x.astype(np.float64) - xt, axis=1)
    D = np.sum((x.astype(np.float64) - q.astype(np.float64)) ** 2, axis=1)
    lb = _lb_l2(h, q, sx, dxr, dim).astype(np.float64)
    assert np.all(lb <= D), float(np.max(lb - D))
    # the dot-product form of the same image: a - E with E >= 2 |dx||q| alone
    slack_dot = 2.0 * dxr * np.linalg.norm(q.astype(np.float64))
    rel_diff = np.median((D - lb) / D)
    rel_dot = np.median(slack_dot / D)
    if spread <= 0.15 and dim == 960:
        assert rel_diff < 0.05 and rel_dot > 3 * rel_diff, (rel_diff, rel_dot)


@pytest.mark.parametrize("spread", [0.15, 1.0])
def test_difference_form_of_the_unit_vectors(spread):
    dim, n = 960, 3000
    rng = np.random.default_rng(int(spread * 1000))
    centre = np.abs(rng.normal(0.07, 0.045, dim))
    x = np.round(np.clip(np.abs(centre + spread * 0.045 * rng.standard_normal((n, dim))), 0, 0.8), 4).astype(np.float32)
    q = np.round(np.clip(np.abs(centre + spread * 0.045 * rng.standard_normal(dim)), 0, 0.8), 4).astype(np.float32)
    sx = 2.0 ** 13
    h, xt = _image(x, sx)
    dxr = np.linalg.norm(x.astype(np.float64) - xt, axis=1).astype(np.float32)
    # cached squared norms: strict f32 folds (the oracle's order); a plain f32 sum is within the same gamma_d
    xs = np.array([np.float32(0)] * n)
    for i in range(dim):
        xs = (xs + x[:, i] * x[:, i]).astype(np.float32)
    qs = np.float32(0)
    for i in range(dim):
        qs = np.float32(qs + q[i] * q[i])
    qhat = (q * (np.float32(1) / np.sqrt(qs))).astype(np.float32)
    rs = (np.float32(1.0 / sx) / np.sqrt(xs)).astype(np.float32)
    t = (h.astype(np.float32) * rs[:, None] - qhat[None, :]).astype(np.float32)  # (the kernel fuses this into one rounding)
    a = np.sum(t * t, axis=1, dtype=np.float32)
    slack = dxr * np.float32(1.001) / (np.sqrt(xs) * np.float32(0.999)) * np.float32(1.002) + np.float32(dim + 16) * U
    sr = np.sqrt(a * (np.float32(1) - np.float32(dim + 8) * U)) * (np.float32(1) - 4 * U) - slack
    sr = np.where(sr > 0, sr * (np.float32(1) - 4 * U), np.float32(0))
    lb = (sr * sr * (np.float32(1) - 4 * U)).astype(np.float64)
    x64, q64 = x.astype(np.float64), q.astype(np.float64)
    unit = np.sum((x64 / np.linalg.norm(x64, axis=1)[:, None] - q64 / np.linalg.norm(q64)) ** 2, axis=1)  # = 2 (1 - cos)
    assert np.all(lb <= unit), float(np.max(lb - unit))
    if spread <= 0.15:
        assert np.median((unit - lb) / unit) < 0.05
