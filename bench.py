#!/usr/bin/env python3
"""bench.py -- headline benchmark: Flat brute-force k-NN, Gist1M-shaped corpus, queries/s at recall@10.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1] / configs[4]): corpus N=1,000,000 x 960 f32, L2Sqr, k=10, one step = one
batch of nq=1000 queries (the size of data/gist_test.bin) through FlatIndex::knn semantics.  Gist1M is not
available offline, so rows are synthetic gist-shaped (per-dimension mean/std of data/gist_1000.bin,
|N(mu,sigma)| clipped to [0,0.8], 4 decimals), generated on the GPU from a fixed seed; queries likewise.
With N ranks the SAME 1M corpus is row-sharded (strong scaling): per-shard local top-k, one RCCL
all-gather of [nq,k], exact merge by (distance, index) on every rank.

Timed region: queries and corpus already resident in HBM; K steps between barrier+synchronize pairs; the
maximum over ranks is reported.  `roofline` is computed from HIP-event timings of the dominant kernel
(flat_mfma) taken inside the library on its own stream during the timed steps.  `cpu_baseline` is the CPU
oracle (a C restatement of the reference's Rust path, kind "port") timed on rank 0 at N=1 on a bounded
query sample of the same corpus, and doubles as a parity check of the GPU results.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def gist_like_gpu(torch, n, dim, seed, device, chunk=131072):
    stats = np.load(os.path.join(ROOT, "tests", "golden", "gist_dim_stats.npy"))
    mu = torch.from_numpy(np.resize(stats[0], dim).astype(np.float32)).to(device)
    sd = torch.from_numpy(np.resize(stats[1], dim).astype(np.float32)).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n, dim), dtype=torch.float32, device=device)
    for r0 in range(0, n, chunk):
        r1 = min(n, r0 + chunk)
        x = torch.randn((r1 - r0, dim), generator=g, device=device, dtype=torch.float32)
        x.mul_(sd).add_(mu).abs_().clamp_(0.0, 0.8)
        x.mul_(10000.0).round_().div_(10000.0)
        out[r0:r1] = x
    return out


def gist_lowrank_gpu(torch, n, dim, seed, device, latent=32, chunk=131072):
    """Same per-dimension mean / std / clipping / 4-decimal grid as gist_like_gpu, but 81 % of every coordinate's variance
    comes from a `latent`-dimensional Gaussian factor shared through one fixed mixing matrix: neighbours are meaningfully
    closer than random rows (as in real GIST descriptors), so graph / PQ recall is informative.  `--data lowrank`."""
    stats = np.load(os.path.join(ROOT, "tests", "golden", "gist_dim_stats.npy"))
    mu = torch.from_numpy(np.resize(stats[0], dim).astype(np.float32)).to(device)
    sd = torch.from_numpy(np.resize(stats[1], dim).astype(np.float32)).to(device)
    gw = torch.Generator(device=device)
    gw.manual_seed(977)  # the mixing matrix is part of the distribution: same for base and queries
    w = torch.randn((latent, dim), generator=gw, device=device, dtype=torch.float32) / float(np.sqrt(latent))
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n, dim), dtype=torch.float32, device=device)
    for r0 in range(0, n, chunk):
        r1 = min(n, r0 + chunk)
        z = torch.randn((r1 - r0, latent), generator=g, device=device, dtype=torch.float32)
        e = torch.randn((r1 - r0, dim), generator=g, device=device, dtype=torch.float32)
        x = (z @ w).mul_(0.9).add_(e.mul_(0.4359))
        x.mul_(sd).add_(mu).abs_().clamp_(0.0, 0.8)
        x.mul_(10000.0).round_().div_(10000.0)
        out[r0:r1] = x
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=0, help="corpus rows (default: 1,000,000 = Gist1M)")
    ap.add_argument("--dim", type=int, default=960)
    ap.add_argument("--nq", type=int, default=1000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--cpu-queries", type=int, default=512,
                    help="queries timed on the CPU oracle and parity-checked (0 = skip); 512 = ~12 s on 16 cores")
    ap.add_argument("--workload", choices=["flat", "pq_flat", "hnsw", "hnsw_pq", "ivf"], default="flat",
                    help="flat = the headline (BASELINE metric); pq_flat / hnsw = the other SURVEY 8d configs")
    ap.add_argument("--ef", type=int, default=0, help="pq_flat: ADC shortlist (default 100); hnsw: search ef (default 128); ivf: n_probes (default 4)")
    ap.add_argument("--dist", choices=["l2sqr", "cosine"], default="l2sqr",
                    help="l2sqr = the BASELINE metric; cosine = the reference's default table distance (pyo3/mod.rs:73)")
    ap.add_argument("--data", choices=["gistlike", "lowrank"], default="gistlike",
                    help="gistlike = per-dimension Gaussians (SURVEY 8d generator); lowrank = same marginals with a 32-d "
                         "latent factor, for informative ANN recall")
    ap.add_argument("--mode", type=int, default=0, help="flat mode: 0 auto, 1 exact scan, 2 MFMA forced")
    ap.add_argument("--half", type=int, default=0, help="flat: fp16 first pass of large query batches: 0 auto, 1 off, 2 forced")
    ap.add_argument("--half-kmul", type=int, default=0, help="flat: shortlist of the fp16 pass = max(64, kmul*k) (0: library default)")
    ap.add_argument("--dump", type=str, default="", help="rank 0 saves the last step's results to this .npz (tests)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import lab_1806_vec_db_amd as vdb
    from lab_1806_vec_db_amd.shard import (ShardExchange, allgather_concat, allgather_merge, allgather_merge_pq,
                                           replica_query_slice, shard_bounds)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # VDB_DIST_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: ranks share the visible
    # GPUs and the all-gather goes through host memory.  The driver's runs use the default (nccl = RCCL).
    backend = os.environ.get("VDB_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wl = args.workload
    if args.rows <= 0:
        args.rows = 1_000_000  # hnsw: the host builder (hnsw_index.rs:391-457 batches) takes ~3-4 min for this graph
    ef = args.ef or {"pq_flat": 100, "hnsw": 128, "hnsw_pq": 128, "ivf": 4}.get(wl, 0)
    n, dim, nq, k = args.rows, args.dim, args.nq, args.k
    # identical corpus on every rank (same seed), each keeps its row block
    gen = gist_like_gpu if args.data == "gistlike" else gist_lowrank_gpu
    base = gen(torch, n, dim, 1806, device)
    queries = gen(torch, nq, dim, 1807, device)
    r0, r1 = shard_bounds(n, world, rank) if wl in ("flat", "pq_flat") else (0, n)  # HNSW / IVF: full replica per GPU
    shard = base[r0:r1].contiguous()
    torch.cuda.synchronize()

    ix = vdb.GpuIndex(dim, args.dist, device=local_rank)
    ix.add_device(shard.data_ptr(), r1 - r0)
    ix.set_id_offset(r0)
    ix.set_flat_mode(args.mode)
    if args.half:
        ix.set_param("flat_half", args.half)
    if args.half_kmul:
        ix.set_param("flat_half_kmul", args.half_kmul)
    host_base = None
    if rank == 0 and world == 1 and args.cpu_queries > 0:
        host_base = base.cpu().numpy()
    if wl in ("pq_flat", "hnsw_pq"):
        # config/bench_pq_hnsw.toml:16-23: n_bits 4, m = dim/3, k_means_size 10000, max_iter 20, tol 1e-6.  Every rank
        # trains on the same first 10000 rows (same seed) -> identical centroids; codes are encoded per shard on the GPU.
        m = dim // 3
        tr = vdb.GpuIndex(dim, args.dist, device=local_rank)
        tr.add_device(base.data_ptr(), min(n, 10000))
        tr.pq_build(n_bits=4, m=m, train_n=0, max_iter=20, tol=1e-6, seed=42)
        cent = tr.pq_export()["centroids"]
        del tr
        ix.pq_attach(4, m, cent, None)
    if wl == "ivf":
        # IVFIndex::from_vec_set (ivf_index.rs:66-118): sqrt(N) clusters, k-means on 10000 sampled rows, 10 iterations
        t_b = time.perf_counter()
        ix.ivf_build(int(round(n ** 0.5)), train_n=10000, max_iter=10, tol=1e-6, seed=42)
        build_s = time.perf_counter() - t_b
    if wl in ("hnsw", "hnsw_pq"):
        t_b = time.perf_counter()
        ix.hnsw_build(M=16, ef_construction=200, seed=42, batch=64, nthreads=min(len(os.sched_getaffinity(0)), 16))
        build_s = time.perf_counter() - t_b
    del base, shard
    torch.cuda.empty_cache()

    # the local results live in the send block of the per-step exchange (typed views, no packing)
    ex = ShardExchange(nq, k, device, world if backend == "nccl" else 1)
    o_idx, o_dist, o_cnt = ex.idx, ex.dist, ex.cnt

    host_xchg = backend != "nccl" and world > 1
    efk = max(ef, k)
    if wl == "pq_flat" and world > 1:
        s_adc = torch.zeros((nq, efk), dtype=torch.int64, device=device)
        s_ex = torch.zeros((nq, efk), dtype=torch.int64, device=device)
    q0, q1 = replica_query_slice(nq, world, rank)

    def step():
        if wl == "flat":
            ix.flat_knn_device(queries.data_ptr(), nq, k, o_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr())
            if host_xchg:
                return allgather_merge(o_idx.cpu(), o_dist.cpu(), o_cnt.cpu(), k)
            return ex.exchange_merge(ix)
        if wl == "pq_flat":
            if world == 1:
                ix.knn_pq_device(queries.data_ptr(), nq, k, ef, o_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr())
                return o_idx, o_dist, o_cnt
            ix.knn_pq_shard_device(queries.data_ptr(), nq, k, ef, s_adc.data_ptr(), s_ex.data_ptr())
            if host_xchg:
                return allgather_merge_pq(s_adc.cpu(), s_ex.cpu(), k)
            return allgather_merge_pq(s_adc, s_ex, k, gpu_index=ix)
        # hnsw / ivf: replicas, each rank answers its block of the queries, blocks are concatenated
        if q1 > q0 and wl == "ivf":
            ix.ivf_knn_device(queries[q0:q1].data_ptr(), q1 - q0, k, ef, o_idx.data_ptr(), o_dist.data_ptr(),
                              o_cnt.data_ptr())
        elif q1 > q0:
            ix.hnsw_knn_device(queries[q0:q1].data_ptr(), q1 - q0, k, ef, o_idx.data_ptr(), o_dist.data_ptr(),
                               o_cnt.data_ptr(), use_pq=(wl == "hnsw_pq"))
        li, ld, lc = o_idx[: q1 - q0], o_dist[: q1 - q0], o_cnt[: q1 - q0]
        if host_xchg:
            return allgather_concat(li.cpu(), ld.cpu(), lc.cpu(), nq)
        return allgather_concat(li, ld, lc, nq)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    ix.prof_enable(True)
    ix.prof_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    ix.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if wl == "flat":
        # dominant kernel: the fp16 first pass (k_flat_gemm<GEMM_F16>, calls with more than 64 queries) when it ran, else
        # the split-bf16 pass, else the exact scan
        kernel = next((kn for kn in ("flat_half", "flat_mfma") if ix.prof_get(kn)["launches"]), "flat_exact")
    else:
        kernel = {"pq_flat": "pq_adc", "hnsw": "hnsw", "hnsw_pq": "hnsw", "ivf": "ivf_rerank"}[wl]
    p = ix.prof_get(kernel)
    roofline = None
    if p["launches"] and wl != "flat":
        # pq_adc: code bytes of one scan = rows x ceil(m*n_bits/8), one scan serves 4 queries (LUTs side by side in
        # LDS); hnsw: n_dist x (dim*4 + 4) + n_expanded x max_m0*4 counted by the kernel
        avg_ms = p["ms"] / p["launches"]
        bytes_per_launch = p["bytes"] / p["launches"]
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": kernel,
                    "avg_launch_ms": round(avg_ms, 4), "launches": p["launches"], "bytes_per_launch": bytes_per_launch}
    elif p["launches"]:
        avg_ms = p["ms"] / p["launches"]
        # bytes the kernel has to stream: passes x shard_rows x dim x 4 for the split-bf16 / exact kernels;
        # the fp16 first pass reads a 2-byte mirror, so ITS operand bytes are half of that -- the roofline fraction is
        # quoted on the bytes really needed, the f32-equivalent rate is reported next to it
        elem = 2 if kernel == "flat_half" else 4
        bytes_per_launch = p["bytes"] / p["launches"]
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        # queries one corpus pass serves: 128 (k_flat_gemm) when a call carries more than 64 queries, else 2 x 32
        # (k_flat_mfma, XCD-shared passes) -- the rule of Index::flat_knn_device
        qpp = 128 if (nq > 64 or kernel == "flat_half") else 64
        traffic = None  # HBM bytes per launch from the committed PMC passes (same kernel, same shard size only)
        try:
            pmc_name = "pmc_flat_half.json" if kernel == "flat_half" else ("pmc_flat_gemm.json" if qpp == 128 else "pmc_flat_mfma.json")
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
            passes = round(bytes_per_launch / pmc["algorithmic_bytes_per_pass"])  # HBM passes in one launch
            if kernel in ("flat_mfma", "flat_half") and abs(pmc["algorithmic_bytes_per_pass"] * passes - bytes_per_launch) < 1:
                traffic = pmc["hbm_bytes_per_pass"] * passes
        except (OSError, KeyError, ValueError):
            pass
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": kernel,
                    "avg_launch_ms": round(avg_ms, 4), "launches": p["launches"],
                    "bytes_per_launch": bytes_per_launch,
                    "units_per_launch": f"{round(bytes_per_launch / ((r1 - r0) * dim * elem))} corpus passes x {r1 - r0} rows "
                                        f"x {dim} x {elem} B; one pass serves {qpp} queries"}
        if kernel == "flat_half":
            roofline["operand"] = "scaled fp16 mirror of the rows (2 B/element); exact f32 re-rank + certification downstream"
            roofline["f32_equivalent_GBps"] = round(2 * achieved, 1)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    if args.dump:
        np.savez(args.dump, idx=res[0].cpu().numpy(), dist=res[1].cpu().numpy(), cnt=res[2].cpu().numpy())
    qps = nq * args.steps / elapsed
    dname = "L2Sqr" if args.dist == "l2sqr" else "Cosine"
    names = {"flat": ("Flat brute force", "flat_knn_gist1m"), "pq_flat": (f"PQ-Flat 4-bit m={dim // 3}, ADC ef={ef}", "pq_flat_knn_gist1m"),
             "hnsw": (f"HNSW M=16 efc=200, ef={ef}", f"hnsw_knn_gistlike_{n}"),
             "hnsw_pq": (f"HNSW M=16 efc=200 + PQ 4-bit m={dim // 3}, ef={ef}", f"hnsw_pq_knn_gistlike_{n}"),
             "ivf": (f"IVF {int(round(n ** 0.5))} clusters, n_probes={ef}", f"ivf_knn_gistlike_{n}")}[wl]
    par = {"flat": f"row-shard x{world}", "pq_flat": f"row-shard x{world}", "hnsw": f"replica x{world}, queries split",
           "hnsw_pq": f"replica x{world}, queries split", "ivf": f"replica x{world}, queries split"}[wl]
    out = {
        "metric": f"queries/sec at recall@10, Gist1M d=960 ({names[0]}, {dname}, k=10)",
        "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic" if args.data == "gistlike" else "synthetic (low-rank gist-like)",
        "config": {"workload": names[1], "rows": n, "dim": dim, "queries_per_step": nq, "k": k, "dist": dname,
                   "parallelism": par if world > 1 else "single GPU"},
        "roofline": roofline, "recall_at_10": None,
    }
    if wl == "flat":
        out["config"]["queries_per_corpus_pass"] = 128 if (nq > 64 or (roofline or {}).get("kernel") == "flat_half") else 64
        out["fallback_queries"] = ix.flat_fallback_count()
        out["half_pass"] = {"queries": ix.get_stat("flat_half_queries"), "redone_split_bf16": ix.get_stat("flat_half_redo")}
    else:
        out["config"]["ef"] = ef
    if wl == "ivf":
        out["config"]["host_build_s"] = round(build_s, 1)
    if wl in ("hnsw", "hnsw_pq"):
        nd, ne = ix.hnsw_last_stats()
        out["config"]["host_build_s"] = round(build_s, 1)
        out["hnsw_work_per_query"] = {"n_dist": round(nd / max(q1 - q0, 1), 1), "n_expanded": round(ne / max(q1 - q0, 1), 1)}

    # ---- CPU baseline + parity (rank 0, N=1 only) ------------------------------------------------------
    if host_base is not None:
        from concurrent.futures import ThreadPoolExecutor

        from oracle import oracle as O

        O.build()
        okind = O.L2SQR if args.dist == "l2sqr" else O.COSINE
        ncpu = min(args.cpu_queries, nq)
        threads = min(len(os.sched_getaffinity(0)), 16)  # the GPU box's CPU share for one GPU
        hq = queries[:ncpu].cpu().numpy()
        gi = res[0][:ncpu].cpu().numpy().astype(np.uint64)
        gd = res[1][:ncpu].cpu().numpy()
        if wl == "flat":
            t0 = time.perf_counter()
            ci, cd, cc = O.flat_knn_batch(host_base, hq, k, okind, nthreads=threads)
            cpu_s = time.perf_counter() - t0
            truth = ci
            what = "FlatIndex::knn"
        else:
            # recall is against Flat ground truth (gen_gnd.rs:54-72), taken from the GPU Flat path of the same index
            t_idx = torch.zeros((nq, k), dtype=torch.int64, device=device)
            t_dist = torch.zeros((nq, k), dtype=torch.float32, device=device)
            t_cnt = torch.zeros((nq,), dtype=torch.int64, device=device)
            ix.flat_knn_device(queries.data_ptr(), nq, k, t_idx.data_ptr(), t_dist.data_ptr(), t_cnt.data_ptr())
            truth = t_idx[:ncpu].cpu().numpy().astype(np.uint64)
            if wl == "pq_flat":
                pq = ix.pq_export()
                opq = O.PQ.from_centroids(dim, pq["m"], pq["n_bits"], okind, pq["centroids"])
                opq.set_codes(pq["codes"])  # GPU-encoded codes (bit-equal to the oracle's encoder, tests/test_pq_gpu.py)
                t0 = time.perf_counter()
                with ThreadPoolExecutor(threads) as pool:  # ctypes releases the GIL: one query per thread
                    r = list(pool.map(lambda q: O.flat_knn_pq(host_base, opq, hq[q], k, ef, okind), range(ncpu)))
                cpu_s = time.perf_counter() - t0
                what = "FlatIndex::knn_pq"
            elif wl == "ivf":
                ex_ivf = ix.ivf_export()  # centroids and clusters are inputs of the oracle (assignment parity: tests/test_ivf_gpu.py)
                oiv = O.IVF(host_base, ex_ivf["centroids"], okind, assign=ex_ivf["assign"])
                t0 = time.perf_counter()
                with ThreadPoolExecutor(threads) as pool:
                    r = list(pool.map(lambda q: oiv.knn(hq[q], k, ef), range(ncpu)))
                cpu_s = time.perf_counter() - t0
                what = "IVFIndex::knn_with_ef on the same centroids and clusters"
            elif wl == "hnsw_pq":
                pq = ix.pq_export()
                opq = O.PQ.from_centroids(dim, pq["m"], pq["n_bits"], okind, pq["centroids"])
                opq.set_codes(pq["codes"])
                oh = O.HNSW.from_graph(host_base, okind, 16, 200, ix.hnsw_export())
                t0 = time.perf_counter()
                with ThreadPoolExecutor(threads) as pool:
                    r = list(pool.map(lambda q: oh.knn_pq(opq, hq[q], k, ef), range(ncpu)))
                cpu_s = time.perf_counter() - t0
                what = "HNSWIndex::knn_pq on the same graph and PQ table"
            else:
                oh = O.HNSW.from_graph(host_base, okind, 16, 200, ix.hnsw_export())
                t0 = time.perf_counter()
                with ThreadPoolExecutor(threads) as pool:
                    r = list(pool.map(lambda q: oh.knn(hq[q], k, ef), range(ncpu)))
                cpu_s = time.perf_counter() - t0
                what = "HNSWIndex::knn_with_ef on the same graph"
            ci = np.stack([np.pad(x[0], (0, k - len(x[0]))) for x in r]).astype(np.uint64)
            cd = np.stack([np.pad(x[1], (0, k - len(x[1]))) for x in r]).astype(np.float32)
        idx_equal = bool(np.array_equal(gi, ci))
        dist_equal = bool(np.array_equal(gd, cd))
        out["recall_at_10"] = float(np.mean([O.recall(truth[q], gi[q]) for q in range(ncpu)]))
        out["parity"] = {"queries_checked": ncpu, "indices_identical": idx_equal, "distances_bit_exact": dist_equal}
        out["cpu_baseline"] = {"value": round(ncpu / cpu_s, 2), "unit": "queries/s", "cores": threads,
                               "kind": "port",
                               "sample": f"{what}: {ncpu} of the {nq} queries against the full {n}x{dim} corpus, "
                                         f"one query per thread (mirrors rayon par_iter, examples/bench.rs:414-416)"}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
