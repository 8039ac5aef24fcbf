"""GPU: the two measurement hooks bench.py quotes next to its roofline numbers return sane figures for an MI355X
(vdb_stream_probe: attainable HBM read rate; vdb_mfma_probe: sustained rate of the Flat filter's MFMA instruction and the
shader clock held meanwhile) and reject bad arguments through the ABI's error path."""
import pytest

pytestmark = pytest.mark.gpu


def test_stream_and_mfma_probes():
    import lab_1806_vec_db_amd as vdb
    from lab_1806_vec_db_amd.index import mfma_probe, stream_probe
    gbps = stream_probe(0, 512 << 20, 2)
    assert 1000.0 < gbps < 8000.0, gbps  # below the 8 TB/s HBM3E peak, far above anything PCIe or a single CU could give
    tfl, ghz = mfma_probe(0, 2, 20000)
    assert 100.0 < tfl < 2600.0, tfl     # dense fp16: nominal 2.5 PFLOP/s, ~1.3 sustained
    assert 0.3 < ghz < 3.0, ghz
    with pytest.raises(vdb.VdbError):
        stream_probe(0, 1 << 20, 1)      # smaller than the Infinity Cache: refused
    with pytest.raises(vdb.VdbError):
        mfma_probe(0, 0, 10)


def test_latency_and_row_fragment_probes():
    """vdb_latency_probe (the dependent-load latency the graph walks' floor is quoted on) and vdb_stream_probe_rows (MFMA
    fragment loads from a row-major image: the A/B behind keeping two fp16 copies of the rows)."""
    import lab_1806_vec_db_amd as vdb
    from lab_1806_vec_db_amd.index import fold_probe, latency_probe, stream_probe_rows
    ns = latency_probe(0, 512 << 20, 4000)
    assert 100.0 < ns < 5000.0, ns       # an HBM round trip: a few hundred nanoseconds
    na = fold_probe(0, 1 << 20)
    assert 0.5 < na < 20.0, na           # a dependent v_add_f32: a handful of cycles
    gbps = stream_probe_rows(0, 512 << 20, 2, 1920)
    assert 200.0 < gbps < 8000.0, gbps
    with pytest.raises(vdb.VdbError):
        latency_probe(0, 1 << 10, 100)
    with pytest.raises(vdb.VdbError):
        stream_probe_rows(0, 512 << 20, 2, 1000)  # rows must be whole 128-B lines
