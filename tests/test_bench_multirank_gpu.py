"""GPU: rehearsal of bench.py's N>1 path on a one-GPU box (VDB_DIST_BACKEND=gloo: ranks share the GPU, the
all-gather goes through host memory).  The merged results of 2 and 3 row shards must equal the 1-GPU results."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(world, dump, port, extra=()):
    env = dict(os.environ, VDB_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    args = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--rows", "60000", "--nq", "96", "--cpu-queries",
            "0", "--dump", dump] + list(extra)
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_sharded_bench_equals_single(tmp_path):
    r1 = _run(1, str(tmp_path / "w1.npz"), 29611)
    assert r1["n_gpus"] == 1 and r1["roofline"]["frac"] > 0
    a = np.load(tmp_path / "w1.npz")
    for world, port in ((2, 29612), (3, 29613)):
        r = _run(world, str(tmp_path / f"w{world}.npz"), port)
        assert r["n_gpus"] == world and r["scaling"] == "strong"
        b = np.load(tmp_path / f"w{world}.npz")
        assert np.array_equal(a["idx"], b["idx"])
        assert np.array_equal(a["dist"], b["dist"])
        assert np.array_equal(a["cnt"], b["cnt"])


def test_one_rank_rccl_exchange(tmp_path):
    """The per-step exchange of bench.py's N > 1 path -- torch.distributed `nccl` (= RCCL) all_gather_into_tensor on the
    send block + vdb_merge_topk_gathered on the receive buffer -- executed for real with ONE rank (VDB_FORCE_EXCHANGE=1):
    the only way to run those calls on a one-GPU box.  Results must equal the plain single-GPU run."""
    r0 = _run(1, str(tmp_path / "plain.npz"), 29631)
    env_backup = os.environ.get("VDB_FORCE_EXCHANGE")
    os.environ["VDB_FORCE_EXCHANGE"] = "1"
    os.environ.pop("VDB_DIST_BACKEND", None)
    try:
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29632", VDB_FORCE_EXCHANGE="1")
        env.pop("VDB_DIST_BACKEND", None)
        args = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--rows", "60000", "--nq", "96", "--cpu-queries", "0", "--dump",
                str(tmp_path / "x.npz")]
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
    finally:
        if env_backup is None:
            os.environ.pop("VDB_FORCE_EXCHANGE", None)
        else:
            os.environ["VDB_FORCE_EXCHANGE"] = env_backup
    a, b = np.load(tmp_path / "plain.npz"), np.load(tmp_path / "x.npz")
    assert r0["n_gpus"] == 1
    assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["dist"], b["dist"]) and np.array_equal(a["cnt"], b["cnt"])


@pytest.mark.parametrize("workload,rows,dim", [("pq_flat", "70000", "96"), ("hnsw", "3000", "48")])
def test_sharded_other_workloads_equal_single(tmp_path, workload, rows, dim):
    """pq_flat: row shards + ADC-order merge + re-sort; hnsw: replicas with split queries."""
    extra = ["--workload", workload, "--rows", rows, "--dim", dim]
    r1 = _run(1, str(tmp_path / "w1.npz"), 29621, extra)
    assert r1["n_gpus"] == 1 and r1["roofline"]["launches"] > 0 and r1["roofline"]["bytes_per_launch"] > 0
    a = np.load(tmp_path / "w1.npz")
    r = _run(2, str(tmp_path / "w2.npz"), 29622, extra)
    assert r["n_gpus"] == 2
    b = np.load(tmp_path / "w2.npz")
    assert np.array_equal(a["idx"], b["idx"])
    assert np.array_equal(a["dist"], b["dist"])
    assert np.array_equal(a["cnt"], b["cnt"])


def test_pipelined_steps_with_the_rccl_exchange_and_self_launch(tmp_path):
    """Three steps in flight (vdb_flat_knn_device_begin / _end, what bench.py keeps on row shards) together with the real
    RCCL all-gather + gathered merge of a 1-rank group (VDB_FORCE_EXCHANGE=1): the results of the last step equal the plain
    synchronous run's.  And `bench.py --gpus 2` WITHOUT a launcher starts its own two ranks (gloo rehearsal on the one GPU)."""
    r0 = _run(1, str(tmp_path / "plain.npz"), 29641, ["--pipeline", "1"])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29642", VDB_FORCE_EXCHANGE="1")
    env.pop("VDB_DIST_BACKEND", None)
    args = ["--gpus", "1", "--steps", "5", "--warmup", "2", "--rows", "60000", "--nq", "96", "--cpu-queries", "0", "--pipeline", "3",
            "--dump", str(tmp_path / "p3.npz")]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["steps_in_flight"] == 3 and r0["config"]["steps_in_flight"] == 1
    a, b = np.load(tmp_path / "plain.npz"), np.load(tmp_path / "p3.npz")
    assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["dist"], b["dist"]) and np.array_equal(a["cnt"], b["cnt"])
    # self-launch: no torch.distributed.run around it, WORLD_SIZE unset
    env2 = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env2["VDB_DIST_BACKEND"] = "gloo"
    args2 = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "60000", "--nq", "96", "--cpu-queries", "0", "--dump", str(tmp_path / "s2.npz")]
    out2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args2, env=env2, capture_output=True, text=True, timeout=900)
    assert out2.returncode == 0, out2.stderr[-2000:]
    line2 = json.loads([l for l in out2.stdout.splitlines() if l.startswith("{")][-1])
    assert line2["n_gpus"] == 2 and line2["config"]["steps_in_flight"] == 3
    c = np.load(tmp_path / "s2.npz")
    assert np.array_equal(a["idx"], c["idx"]) and np.array_equal(a["dist"], c["dist"]) and np.array_equal(a["cnt"], c["cnt"])
    # a launcher that disagrees with --gpus is an error, not a silent one-rank run
    env3 = dict(env2, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out3 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env3, capture_output=True, text=True, timeout=300)
    assert out3.returncode != 0 and "WORLD_SIZE" in (out3.stderr + out3.stdout)


def test_bench_reads_raw_f32_files(tmp_path):
    """--base-file / --query-file: raw row-major f32 without header, the layout src/bin/convert_fvecs.rs:29-31 writes -- here the
    reference's own gist_1000.bin / gist_test.bin; rows come from the file size, `data` names the files, parity against the oracle."""
    g = os.path.join(ROOT, "tests", "golden")
    args = ["--base-file", os.path.join(g, "gist_1000.bin"), "--query-file", os.path.join(g, "gist_test.bin"), "--steps", "2", "--warmup", "1",
            "--legs", "none", "--cpu-queries", "48", "--nq", "200", "--dump", str(tmp_path / "f.npz")]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["rows"] == 1000 and line["data"].startswith("file: gist_1000.bin x gist_test.bin")
    assert line["parity"] == {"queries_checked": 48, "indices_identical": True, "distances_bit_exact": True}
    d = np.load(tmp_path / "f.npz")
    assert d["idx"][0].tolist()[:3] == [918, 467, 725]  # query 0 of gist_test against gist_1000
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--base-file", os.path.join(g, "gist_1000.bin")], capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "go together" in (bad.stderr + bad.stdout)
