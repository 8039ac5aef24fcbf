"""GPU: a Flat call in two halves (vdb_flat_knn_device_begin / _end): several batches in flight on one index give the synchronous
call's answers bit for bit -- MFMA pipeline (fp16 pass, redo tier, exact fallback through near-duplicate clusters), the
one-launch small-table path and the exact scan (which complete inside begin), out-of-order ends, and the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(torch, a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


@pytest.mark.parametrize("n,dim,style", [(40000, 128, "normal"), (30000, 192, "clusters"), (3000, 64, "normal")])
def test_begin_end_equals_synchronous_call(n, dim, style):
    torch = pytest.importorskip("torch")
    import lab_1806_vec_db_amd as vdb
    from oracle import oracle as O

    rng = np.random.default_rng(n + dim)
    if style == "clusters":  # near-duplicates: tiny margins -> redo tier and exact fallbacks inside _end
        c = rng.standard_normal((n // 50 + 1, dim)).astype(np.float32)
        base = (np.repeat(c, 50, axis=0)[:n] + 1e-4 * rng.standard_normal((n, dim))).astype(np.float32)
        mk = lambda m: (c[rng.integers(0, len(c), m)] + 1e-4 * rng.standard_normal((m, dim))).astype(np.float32)  # noqa: E731
    else:
        base = rng.standard_normal((n, dim)).astype(np.float32)
        mk = lambda m: rng.standard_normal((m, dim)).astype(np.float32)  # noqa: E731
    ix = vdb.GpuIndex(dim, "l2sqr")
    ix.batch_add(base)
    k = 10
    batches = [mk(m) for m in (100, 7, 129, 64)]
    dq = [_dev(torch, b) for b in batches]
    outs = [(torch.zeros((len(b), k), dtype=torch.int64, device="cuda"), torch.zeros((len(b), k), dtype=torch.float32, device="cuda"),
             torch.zeros((len(b),), dtype=torch.int64, device="cuda")) for b in batches]
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    hs = [ix.flat_knn_device_begin(q.data_ptr(), len(b), k, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), stream=st)
          for q, b, o in zip(dq, batches, outs)]  # four calls in flight
    for j in (2, 0, 3, 1):  # ended in another order than begun
        ix.flat_knn_device_end(hs[j])
    for b, o in zip(batches, outs):
        si, sd, sc = ix.flat_knn(b, k)  # the synchronous host call
        assert np.array_equal(o[0].cpu().numpy().astype(np.uint64), si) and np.array_equal(o[1].cpu().numpy(), sd)
        assert o[2].cpu().numpy().tolist() == sc.tolist()
    oi, od, oc = O.flat_knn_batch(base, batches[0][:16], k, 0, nthreads=8)
    assert np.array_equal(outs[0][0][:16].cpu().numpy().astype(np.uint64), oi) and np.array_equal(outs[0][1][:16].cpu().numpy(), od)
    if style == "clusters":
        # the redo paths ran inside _end: the second 8-bit attempt (k_redo.hip) and / or the tiers behind it
        assert ix.get_stat("flat_i8_second_queries") + ix.get_stat("flat_i8_redo") + ix.get_stat("flat_half_redo") + ix.flat_fallback_count() > 0
    with pytest.raises(vdb.VdbError):
        ix.flat_knn_device_end(None)
    ix.close()
