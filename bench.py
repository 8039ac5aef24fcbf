#!/usr/bin/env python3
"""bench.py -- headline benchmark: Flat brute-force k-NN, Gist1M-shaped corpus, queries/s at recall@10.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints ONE JSON line on stdout: the COMPACT record
(< 6 KB: the contract's fields, roofline incl. the N*d*4 leg, cpu_baseline, parity, one brief object per leg); the full record
with every leg's roofline / notes goes to gpurun_out/bench_full.json (--full-out).

Headline workload (BASELINE.json configs[1] / configs[4]): corpus N=1,000,000 x 960 f32, L2Sqr, k=10, one step = one
batch of nq=1000 queries (the size of data/gist_test.bin) through FlatIndex::knn semantics.  Gist1M is not
available offline, so rows are synthetic gist-shaped (per-dimension mean/std of data/gist_1000.bin,
|N(mu,sigma)| clipped to [0,0.8], 4 decimals), generated on the GPU from a fixed seed; queries likewise.
With N ranks the SAME 1M corpus is row-sharded (strong scaling): per-shard local top-k, one RCCL
all-gather of [nq,k], exact merge by (distance, index) on every rank.

Timed region: queries and corpus already resident in HBM; K steps between barrier+synchronize pairs; the
maximum over ranks is reported.  `roofline` is computed from HIP-event timings of the dominant kernel
taken inside the library on its own stream during the timed steps.  `cpu_baseline` is the CPU
oracle (a C restatement of the reference's Rust path, kind "port") timed on rank 0 at N=1 on a bounded
query sample of the same corpus, and doubles as a parity check of the GPU results.

At N=1 (default flags) the same run also reports, under "legs", every other item SURVEY.md 8(d) asks for:
  flat_f32_operands  the same 1000-query step with the fp16 first pass switched off: the split-bf16 kernel streams the
                     4-B/element rows mirror, so ITS roofline is on N*d*4 bytes per corpus pass -- the figure the
                     ">= 70 % of HBM roofline" target is defined on
  flat_cosine        the same 1000-query step under Cosine, the reference's default table distance (8-bit pass on unit rows)
  flat_B32, flat_B1  calls of 32 queries (SURVEY 8d's headline batch) and of 1 query
  config1_gist_1000  BASELINE config 1: the reference's own data/gist_1000.bin x data/gist_test.bin, Flat, L2Sqr, k=10,
                     GPU beside the CPU oracle serial and on all cores (protocol of examples/bench.rs:403-433)
  pq_flat            PQ-Flat ADC (4-bit, m=320, ef=100) on a 1M low-rank gist-like corpus
  ivf                IVF (sqrt(N) = 1000 clusters, 4 probes) on the same corpus: the probe-list scan as a certified 8-bit -> fp16 -> exact cascade
  hnsw               HNSW (M=16, efc=200, ef=128) on the first --hnsw-rows rows of that corpus (default: all 1M)
  hnsw_pq            the same graph walked with ADC distances + cached-form re-sort (HNSWIndex::knn_pq)
each with its own roofline / cpu_baseline / parity, plus `attainable_peak_GBps` from a streaming-read probe in this run.
`--legs none` prints the headline only (what N>1 runs always do).
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


def gist_clustered_gpu(torch, n, dim, seed, device, clusters=1024, spread=0.15, chunk=131072):
    """`clusters` Gaussian clusters with the gist-like marginals: the centres are gist-like rows (seed 4711: part of the distribution,
    the same for base and queries), a row is its centre + spread x sigma_j noise -- squared distances ~ 2 spread^2 sum sigma_j^2 inside a
    cluster (0.09 at 0.15) against ~3.9 between clusters.  Neighbours are then separated by margins far below the resolution of an
    8-bit (or fp16) key: the case the Flat filter's auto-off rule exists for.  `--data clustered`, `--cluster-spread`."""
    stats = np.load(os.path.join(ROOT, "tests", "golden", "gist_dim_stats.npy"))
    sd = torch.from_numpy(np.resize(stats[1], dim).astype(np.float32)).to(device)
    centres = gist_like_gpu(torch, clusters, dim, 4711, device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n, dim), dtype=torch.float32, device=device)
    for r0 in range(0, n, chunk):
        r1 = min(n, r0 + chunk)
        c = torch.randint(0, clusters, (r1 - r0,), generator=g, device=device)
        x = torch.randn((r1 - r0, dim), generator=g, device=device, dtype=torch.float32)
        x.mul_(sd * spread).add_(centres[c]).abs_().clamp_(0.0, 0.8)
        x.mul_(10000.0).round_().div_(10000.0)
        out[r0:r1] = x
    return out


def gist_neardup_gpu(torch, n, dim, seed, device, frac=0.01, group=20):
    """gist-like rows of which `frac` are near-duplicates: groups of `group` copies of one source row each, every copy off by one or two
    steps of the 4-decimal grid in a few coordinates -- more near-identical neighbours than k for the queries that land on them.  For the
    queries (seed != 1806) the same fraction sits on source rows of the base.  `--data neardup`."""
    out = gist_like_gpu(torch, n, dim, seed, device)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 99)
    ndup = int(n * frac)
    if ndup == 0:
        return out
    if seed == 1806:  # base: rows [n - ndup, n) are the copies, sources are rows 0, 1, ... (deterministic: queries can aim at them)
        nsrc = max(1, ndup // group)
        src = torch.arange(ndup, device=device) % nsrc
        noise = (torch.rand((ndup, dim), generator=g, device=device) < 0.01).float() * 1e-4
        out[n - ndup:] = (out[src] + noise).clamp_(0.0, 0.8)
    else:  # queries: the first max(1, frac * nq) of them sit next to source rows 0, 1, ... of the base
        base_head = gist_like_gpu(torch, max(ndup, 1), dim, 1806, device)[:ndup]
        noise = (torch.rand((ndup, dim), generator=g, device=device) < 0.01).float() * 1e-4
        out[:ndup] = (base_head + noise).clamp_(0.0, 0.8)
    return out


def load_rows_file(torch, path, dim, r0, r1, device, chunk=131072, cycle=False):
    """rows [r0, r1) of a raw row-major f32 file without header (src/bin/convert_fvecs.rs:29-31 writes exactly this), streamed
    through a memory map; cycle=True repeats the file's rows when it holds fewer than r1"""
    mm = np.memmap(path, dtype=np.float32, mode="r")
    total = mm.shape[0] // dim
    if total * dim != mm.shape[0] or total == 0:
        raise SystemExit(f"bench.py: {path}: size is not a positive multiple of dim*4 = {dim * 4} bytes")
    if r1 > total and not cycle:
        raise SystemExit(f"bench.py: {path} holds {total} rows of dim {dim}, {r1} requested")
    out = torch.empty((r1 - r0, dim), dtype=torch.float32, device=device)
    for a in range(r0, r1, chunk):
        b = min(r1, a + chunk)
        ids = np.arange(a, b) % total
        blk = mm.reshape(total, dim)[ids] if (cycle and b > total) else mm[a * dim:b * dim].reshape(b - a, dim)
        out[a - r0:b - r0] = torch.from_numpy(np.ascontiguousarray(blk)).to(device)
    return out


def hbm_roofline(kernel, p, extra=None):
    """roofline object of one library kernel from its HIP-event record (vdb_prof_get): bytes and ms are sums over the
    launches of the timed region; `traffic` (PMC HBM bytes) cannot be collected inside a bench run and stays null --
    the rocprofv3 --pmc summaries of the same command are under profiles/."""
    if not p["launches"]:
        return None
    avg_ms = p["ms"] / p["launches"]
    bpl = p["bytes"] / p["launches"]
    achieved = bpl / (avg_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": kernel,
         "avg_launch_ms": round(avg_ms, 4), "launches": p["launches"], "bytes_per_launch": bpl}
    if extra:
        r.update(extra)
    return r


def pmc_traffic(kernel, rows, dim, nq):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this very command (profiles/
    r04_pmc_traffic.json (then r03_pmc_traffic.json): counters cannot be collected inside a timed run) -- only for the workload they were collected on"""
    e = src = None
    for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json"):  # the latest passes that hold this kernel
        try:
            e = json.load(open(os.path.join(ROOT, "profiles", name))).get(kernel)
        except (OSError, ValueError):
            e = None
        if e and e.get("workload") == {"rows": rows, "dim": dim, "queries_per_step": nq}:
            src = name
            break
    if not src:
        return None
    return {"traffic": e["hbm_side_bytes"], "traffic_source": f"profiles/{src} (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate "
                                                              "passes of `bench.py --legs none`, per launch; not collected in this run)"}


def flat_kernel_of(ix):
    """dominant kernel of the Flat steps just timed: the 8-bit first pass (k_flat_gemm8) when it ran, else the fp16 first pass
    (k_flat_gemm<GEMM_F16>), else the split-bf16 pass (k_flat_gemm<GEMM_BF16X3> / k_flat_mfma), else the exact scan"""
    return next((kn for kn in ("flat_i8", "flat_half", "flat_mfma") if ix.prof_get(kn)["launches"]), "flat_exact")


def flat_roofline(ix, rows, dim, nq):
    """Flat roofline per SURVEY 8(d): algorithmic bytes of one corpus pass = rows*dim*4 (the f32 VecSet the reference
    scans).  The split-bf16 and exact kernels read exactly that; the fp16 first pass reads a 2-B/element mirror, so for
    it `achieved`/`frac` are quoted on the operand bytes the kernel really streams (named in `frac_of`), and the 8(d)
    figure -- the f32-equivalent rate, which exceeds the HBM peak because half the bytes are never read -- is given next
    to it as `achieved_8d` / `frac_8d`."""
    kernel = flat_kernel_of(ix)
    p = ix.prof_get(kernel)
    if not p["launches"]:
        return None
    elem = {"flat_i8": 1, "flat_half": 2}.get(kernel, 4)
    bpl = p["bytes"] / p["launches"]
    passes = round(bpl / (rows * dim * elem))
    qpp = 128 if (nq > 64 or kernel in ("flat_half", "flat_i8")) else 64  # the rule of Index::flat_knn_device
    if kernel == "flat_exact":
        qpp = 8
    avg_s = p["ms"] / p["launches"] * 1e-3
    alg = passes * rows * dim * 4
    extra = {"units_per_launch": f"{passes} corpus passes x {rows} rows x {dim} x {elem} B; one pass serves up to {qpp} queries",
             "algorithmic_bytes": alg,
             "algorithmic_bytes_def": "SURVEY 8(d): corpus passes x N x d x 4 B (f32 rows)",
             "achieved_8d": round(alg / avg_s / 1e9, 1), "frac_8d": round(alg / avg_s / 1e9 / HBM_PEAK_GBS, 4)}
    if kernel == "flat_i8":
        extra["frac_of"] = ("operand bytes: the centred int8 mirror of the rows (1 B/element) the first pass streams; its keys are lower bounds of the "
                            "distances, the exact f32 stage walks the hit list until the k-th distance is below the next bound")
        extra["operand_bytes"] = bpl
        tf = nq * rows * dim * 2 / avg_s / 1e12  # the call's queries (not the padded slots of its last pass) x rows x dim x 2
        extra["matrix_pipe"] = {"achieved_TOPs": round(tf, 1), "nominal_peak_TOPs": 5000.0, "frac_of_nominal": round(tf / 5000.0, 4),
                                "instruction": "v_mfma_i32_16x16x64_i8"}
        coop = ix.get_stat("flat_gemm8_coop_sets")
        if coop > 1:
            # cooperative sets: `coop` workgroups of an XCD walk the same rows for different query groups, so a row is read from HBM once
            # per set and from the XCD's L2 by the other members.  The operand bytes the CUs consume then exceed what HBM delivers (and
            # can exceed its peak): HBM is no longer what bounds the kernel, the matrix pipe is -- the roofline is quoted on it, the
            # byte rates stay beside it
            r = hbm_roofline(kernel, p, extra)
            r["hbm_operand_rate"] = {"achieved": r["achieved"], "peak": r["peak"], "unit": "GB/s", "frac": r["frac"],
                                     "note": f"operand bytes consumed per second; sets of {coop} workgroups share one row stream through their XCD's L2, "
                                             "HBM-side traffic per launch is `traffic` (PMC), about 1 / set size of the operand bytes"}
            r.update({"bound": "mfma", "achieved": round(tf, 1), "peak": 5000.0, "unit": "TOP/s", "frac": round(tf / 5000.0, 4),
                      "peak_def": "dense int8 MFMA peak of the chip (MI355X_MICROARCH.md: 2 x the ~2.5 PF BF16 rate); what back-to-back "
                                  "v_mfma issue sustains on a box is about half of it (vdb_mfma_probe)",
                      "cooperative_set": coop})
            return r
    elif kernel == "flat_half":
        extra["frac_of"] = "operand bytes: the scaled fp16 mirror of the rows (2 B/element) the first pass streams; exact f32 re-rank + certification downstream"
        extra["operand_bytes"] = bpl
        # the filter is a dense contraction too: 2 flops per (row, column, query slot) with 128 slots per pass
        tf = nq * rows * dim * 2 / avg_s / 1e12
        extra["matrix_pipe"] = {"achieved_TFLOPs": round(tf, 1), "nominal_peak_TFLOPs": 2500.0, "frac_of_nominal": round(tf / 2500.0, 4),
                                "instruction": "v_mfma_f32_16x16x32_f16"}
    else:
        extra["frac_of"] = "SURVEY 8(d) algorithmic bytes (the kernel streams 4 B/element)"
        if kernel == "flat_mfma":  # the split-bf16 kernel: three bf16 MFMAs per operand pair (hi*hi, hi*lo, lo*hi)
            tf = nq * rows * dim * 2 * 3 / avg_s / 1e12
            extra["matrix_pipe"] = {"achieved_TFLOPs": round(tf, 1), "nominal_peak_TFLOPs": 2500.0, "frac_of_nominal": round(tf / 2500.0, 4),
                                    "instruction": "v_mfma_f32_16x16x16_bf16 x 3 (split operands)"}
    r = hbm_roofline(kernel, p, extra)
    coop = ix.get_stat("flat_gemm_coop_sets") if kernel in ("flat_half", "flat_mfma") else 0
    mp = extra.get("matrix_pipe")
    if r and coop > 1 and mp:  # cooperative sets (see the flat_i8 branch): the matrix pipe bounds the kernel, the byte rates stay beside it
        r["hbm_operand_rate"] = {"achieved": r["achieved"], "peak": r["peak"], "unit": "GB/s", "frac": r["frac"],
                                 "note": f"operand bytes consumed per second; sets of {coop} workgroups share one row stream through their XCD's L2, "
                                         "so about 1 / set size of them comes from HBM"}
        r.update({"bound": "mfma", "achieved": mp["achieved_TFLOPs"], "peak": mp["nominal_peak_TFLOPs"], "unit": "TFLOP/s",
                  "frac": mp["frac_of_nominal"], "cooperative_set": coop})
    return r


COMPACT_LIMIT = 6144  # the driver keeps the last 8 KB of stdout: the FINAL line must fit with room to spare


def _leg_brief(leg):
    """{qps, ms_per_step, frac, parity_ok} of one leg of the full record (plus the two or three numbers a leg is judged on)"""
    if not isinstance(leg, dict):
        return None
    r = leg.get("roofline") or {}
    b = {"qps": leg.get("value"), "ms_per_step": leg.get("ms_per_step"), "frac": r.get("frac"), "bound": r.get("bound")}
    par = leg.get("parity")
    if par is not None:
        b["parity_ok"] = bool(par.get("indices_identical") and par.get("distances_bit_exact"))
    elif "results_equal_headline" in leg:
        b["parity_ok"] = bool(leg["results_equal_headline"])
    if leg.get("recall_at_10") is not None:
        b["recall_at_10"] = round(leg["recall_at_10"], 4)
    if (leg.get("cpu_baseline") or {}).get("value") is not None:
        b["cpu_qps"] = leg["cpu_baseline"]["value"]
    for nm in ("one_call_of_1000", "one_call_of_1"):
        if nm in leg:
            b[nm + "_ms"] = leg[nm].get("ms_per_step")
    lf = r.get("latency_floor")
    if lf:
        b["latency_floor_frac"] = lf.get("frac")
    return {k_: v for k_, v in b.items() if v is not None}


def compact_line(out, full_path=None):
    """The FINAL stdout line of a run: the contract's fields of the full record `out`, no prose, < COMPACT_LIMIT bytes.
    The full record (every leg's roofline / cpu_baseline / notes) goes to `full_path`."""
    r = out.get("roofline") or {}
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches", "units_per_launch",
            "bytes_per_launch", "algorithmic_bytes", "cooperative_set", "attainable_peak_GBps", "frac_of_attainable")
    cr = {k_: r[k_] for k_ in keep if r.get(k_) is not None or k_ == "traffic"} if r else None
    if cr is not None:
        if r.get("traffic") and r.get("avg_launch_ms"):
            cr["hbm_from_pmc_GBps"] = round(r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9, 1)
        h = r.get("hbm_operand_rate")
        if h:
            cr["operand_GBps"] = h.get("achieved")
            cr["attainable_peak_GBps"] = h.get("attainable_peak_GBps")
        mp = r.get("matrix_pipe") or {}
        for k_ in ("instruction", "sustained_peak_TOPs", "sustained_peak_TFLOPs", "frac_of_sustained", "clock_GHz_at_sustained_peak"):
            if mp.get(k_) is not None:
                cr[k_] = mp[k_]
        for nm in ("f32_operand_leg", "fp16_pass_leg", "one_step_in_flight"):
            leg = r.get(nm)
            if leg:
                cr[nm] = {k_: leg[k_] for k_ in ("qps", "ms_per_step", "kernel", "avg_launch_ms", "achieved", "frac", "unit", "results_equal_headline")
                          if leg.get(k_) is not None}
    sm = out.get("step_ms")
    cb = out.get("cpu_baseline")
    c = {"metric": out["metric"], "value": out["value"], "unit": out["unit"], "n_gpus": out["n_gpus"], "steps": out["steps"],
         "warmup": out["warmup"], "ms_per_step": out["ms_per_step"],
         "step_ms": {k_: sm[k_] for k_ in ("min", "median", "max")} if sm else None,
         "higher_is_better": out["higher_is_better"], "scaling": out["scaling"], "vs_baseline": out["vs_baseline"],
         "dtype": out["dtype"], "data": out["data"], "config": out["config"], "recall_at_10": out.get("recall_at_10"),
         "parity": out.get("parity"), "roofline": cr,
         "cpu_baseline": {k_: cb[k_] for k_ in ("value", "unit", "cores", "kind", "sample")} if cb else None}
    for k_ in ("i8_pass", "half_pass", "fallback_queries", "filter_work", "pq8_pass"):
        if out.get(k_) is not None:
            c[k_] = out[k_]
    legs = out.get("legs")
    if legs:
        cl = {}
        for nm, leg in legs.items():
            if nm == "ivf":  # out of scope; stays in the full record only
                continue
            if nm == "config1_gist_1000":
                cl[nm] = {"gpu_one_call_qps": leg["gpu_one_call"]["value"], "gpu_per_query_calls_qps": leg["gpu_per_query_calls"]["value"],
                          "gpu_per_query_calls_threads_qps": leg["gpu_per_query_calls_threads"]["value"],
                          "cpu_qps": (leg.get("cpu_baseline") or {}).get("value"),
                          "cpu_serial_qps": ((leg.get("cpu_baseline") or {}).get("serial") or {}).get("value"),
                          "parity_ok": bool((leg.get("parity") or {}).get("indices_identical") and (leg.get("parity") or {}).get("distances_bit_exact"))}
                continue
            cl[nm] = _leg_brief(leg)
            if isinstance(leg.get("n_bits_8"), dict):
                cl[nm + "_8bit"] = _leg_brief(leg["n_bits_8"])
        c["legs"] = cl
    if full_path:
        c["full_record"] = full_path
    line = json.dumps(c, separators=(",", ":"))
    if len(line) >= COMPACT_LIMIT:  # never let prose push the line past what the driver keeps: drop the optional parts, largest first
        for k_ in ("legs", "filter_work", "half_pass", "i8_pass"):
            c.pop(k_, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) < COMPACT_LIMIT:
                break
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
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
    ap.add_argument("--data", choices=["gistlike", "lowrank", "clustered", "neardup"], default="gistlike",
                    help="gistlike = per-dimension Gaussians (SURVEY 8d generator); lowrank = same marginals with a 32-d "
                         "latent factor, for informative ANN recall; clustered = 1024 Gaussian clusters (--cluster-spread x sigma inside); "
                         "neardup = gistlike with 1 %% near-duplicate rows in groups of 20")
    ap.add_argument("--cluster-spread", type=float, default=0.15, help="--data clustered: noise inside a cluster in units of the per-dimension sigma")
    ap.add_argument("--pq-bits", type=int, choices=[4, 8], default=4, help="pq_flat / hnsw_pq: bits per code (8: 256 centroids per group, pq_table.rs:142-145)")
    ap.add_argument("--mode", type=int, default=0, help="flat mode: 0 auto, 1 exact scan, 2 MFMA forced")
    ap.add_argument("--half", type=int, default=0, help="flat: fp16 first pass of large query batches: 0 auto, 1 off, 2 forced")
    ap.add_argument("--i8", type=int, default=0, help="flat (L2Sqr): 8-bit first pass: 0 auto, 1 off, 2 forced")
    ap.add_argument("--half-kmul", type=int, default=0, help="flat: shortlist of the fp16 pass = max(64, kmul*k) (0: library default)")
    ap.add_argument("--param", action="append", default=[], metavar="NAME=VALUE",
                    help="developer tuning switch passed to vdb_set_param before the first step (A/B runs; results never depend on them)")
    ap.add_argument("--full-out", type=str, default="", help="file of the full record (default gpurun_out/bench_full.json); stdout gets the compact line")
    ap.add_argument("--dump", type=str, default="", help="rank 0 saves the last step's results to this .npz (tests)")
    ap.add_argument("--legs", choices=["auto", "all", "none"], default="auto",
                    help="the SURVEY 8(d) report items beside the headline (N=1, flat): auto = all when the headline runs at its "
                         "default size, none = headline only")
    ap.add_argument("--hnsw-rows", type=int, default=1_000_000,
                    help="rows of the hnsw leg (BASELINE config 3 names Gist1M); the build -- batches of 1024 points as "
                         "add_parallel forms them (hnsw_index.rs:391-457), their candidate phase on the GPU -- takes ~55 s for 1M rows")
    ap.add_argument("--hnsw-queries", type=int, default=8192,
                    help="queries per step of the hnsw / hnsw_pq legs: a walk is one wavefront and the chip keeps 2048 of them "
                         "resident, so the rate is flat from ~4096 queries per call on (a 1000-query call is reported beside it)")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="flat: query batches (steps) in flight: with 2, step i+1 is enqueued (vdb_flat_knn_device_begin) before the host "
                         "looks at step i's certification flags (_end), so the corpus passes of consecutive steps run back to back and "
                         "the exact stage of one step overlaps the query preparation of the next; 1 = every step is one synchronous call; "
                         "0 = auto: 1 on one GPU (the pipelining is worth 0 - 2 % there and would blur the HIP-event duration of the corpus pass "
                         "the roofline is quoted on), 3 on row shards, where the per-step fixed cost is the larger share of a step "
                         "(one of 8 shards: 0.44 -> 0.375 ms per step)")
    ap.add_argument("--base-file", type=str, default="",
                    help="raw row-major f32 corpus, no header (the output of src/bin/convert_fvecs.rs:29-31, e.g. a real gist_base "
                         "converted from .fvecs): used instead of the synthetic rows when given; rows = file size / (dim * 4) unless --rows")
    ap.add_argument("--query-file", type=str, default="", help="raw row-major f32 queries (same layout); the first --nq rows are used")
    ap.add_argument("--hnsw-batch", type=int, default=1024,
                    help="points per builder batch (the reference uses 4 x rayon threads); >= 256 puts the candidate phase on the GPU")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves -- as a CHILD process, before
    # anything in this process touches the GPU (no exec of a GPU-initialised process), relay rank 0's JSON line (the child
    # shares stdout) and exit with the child's code.  Under a launcher (WORLD_SIZE set) the flag must agree with it.
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} ranks")

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
    # VDB_FORCE_EXCHANGE=1: a single rank still forms the RCCL process group and runs the per-step all-gather + gathered
    # merge (the N > 1 code path with one rank): lets a one-GPU box execute the nccl calls of the exchange
    force_x = world == 1 and backend == "nccl" and os.environ.get("VDB_FORCE_EXCHANGE") == "1"
    if world > 1 or force_x:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wl = args.workload
    if bool(args.base_file) != bool(args.query_file):
        raise SystemExit("bench.py: --base-file and --query-file go together")
    from_file = bool(args.base_file)
    if from_file and args.rows <= 0:
        args.rows = os.path.getsize(args.base_file) // (args.dim * 4)
    default_size = args.rows in (0, 1_000_000) and args.dim == 960 and args.nq == 1000
    if args.rows <= 0:
        args.rows = 1_000_000
    data_name = (f"file: {os.path.basename(args.base_file)} x {os.path.basename(args.query_file)} (raw row-major f32, convert_fvecs.rs layout)"
                 if from_file else {"gistlike": "synthetic", "lowrank": "synthetic (low-rank gist-like)",
                                    "clustered": f"synthetic (1024 gist-like clusters, spread {args.cluster_spread} sigma)",
                                    "neardup": "synthetic (gist-like, 1 % near-duplicate rows in groups of 20)"}[args.data])
    legs_on = world == 1 and wl == "flat" and (args.legs == "all" or (args.legs == "auto" and default_size))
    ef = args.ef or {"pq_flat": 100, "hnsw": 128, "hnsw_pq": 128, "ivf": 4}.get(wl, 0)
    n, dim, nq, k = args.rows, args.dim, args.nq, args.k
    threads = min(len(os.sched_getaffinity(0)), 16)  # the GPU box's CPU share for one GPU

    attainable = None
    if rank == 0 and world == 1:
        from lab_1806_vec_db_amd.index import stream_probe

        attainable = round(stream_probe(local_rank, 3_840_000_000, 5), 1)  # before the corpus exists: 3.84 GB of its own

    # identical corpus on every rank (same seed), each keeps its row block
    gen = {"gistlike": gist_like_gpu, "lowrank": gist_lowrank_gpu, "neardup": gist_neardup_gpu,
           "clustered": lambda t, n_, d_, sd_, dev_: gist_clustered_gpu(t, n_, d_, sd_, dev_, spread=args.cluster_spread)}[args.data]
    if from_file:
        base = load_rows_file(torch, args.base_file, dim, 0, n, device)
        queries = load_rows_file(torch, args.query_file, dim, 0, nq, device)
    else:
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
    if args.i8:
        ix.set_param("flat_i8", args.i8)
    for pv in args.param:
        pn, _, pval = pv.partition("=")
        ix.set_param(pn, int(pval))
    host_base = None
    if rank == 0 and world == 1 and args.cpu_queries > 0:
        host_base = base.cpu().numpy()
    build_s = 0.0
    if wl in ("pq_flat", "hnsw_pq"):
        # config/bench_pq_hnsw.toml:16-23: n_bits 4, m = dim/3, k_means_size 10000, max_iter 20, tol 1e-6.  Every rank
        # trains on the same first 10000 rows (same seed) -> identical centroids; codes are encoded per shard on the GPU.
        m = dim // 3
        tr = vdb.GpuIndex(dim, args.dist, device=local_rank)
        tr.add_device(base.data_ptr(), min(n, 10000 if args.pq_bits == 4 else 20000))
        tr.pq_build(n_bits=args.pq_bits, m=m, train_n=0, max_iter=20 if args.pq_bits == 4 else 5, tol=1e-6, seed=42)
        cent = tr.pq_export()["centroids"]
        del tr
        ix.pq_attach(args.pq_bits, m, cent, None)
    if wl == "ivf":
        # IVFIndex::from_vec_set (ivf_index.rs:66-118): sqrt(N) clusters, k-means on 10000 sampled rows, 10 iterations
        t_b = time.perf_counter()
        ix.ivf_build(int(round(n ** 0.5)), train_n=10000, max_iter=10, tol=1e-6, seed=42)
        build_s = time.perf_counter() - t_b
    if wl in ("hnsw", "hnsw_pq"):
        t_b = time.perf_counter()
        ix.hnsw_build(M=16, ef_construction=200, seed=42, batch=args.hnsw_batch, nthreads=threads)
        build_s = time.perf_counter() - t_b
    del base, shard
    torch.cuda.empty_cache()

    # the local results live in the send block of the per-step exchange (typed views, no packing)
    depth = (max(1, min(args.pipeline, 4)) if args.pipeline > 0 else (1 if world == 1 else 3)) if wl == "flat" else 1
    ex = ShardExchange(nq, k, device, world if backend == "nccl" else 1, force=force_x, min_depth=depth + 1 if depth > 1 else 1)
    o_idx, o_dist, o_cnt = ex.idx, ex.dist, ex.cnt

    host_xchg = backend != "nccl" and world > 1
    efk = max(ef, k)
    if wl == "pq_flat" and world > 1:
        s_adc = torch.zeros((nq, efk), dtype=torch.int64, device=device)
        s_ex = torch.zeros((nq, efk), dtype=torch.int64, device=device)
    q0, q1 = replica_query_slice(nq, world, rank)

    inflight = []  # flat, depth > 1: (pending handle, buffer slot, views) of the steps begun and not yet ended
    last = [None]

    def flat_finish_one():
        h, slot, (b_idx, b_dist, b_cnt) = inflight.pop(0)
        ix.flat_knn_device_end(h)  # the step's local answer is complete (certified or redone) when this returns
        if host_xchg:
            last[0] = allgather_merge(b_idx.cpu(), b_dist.cpu(), b_cnt.cpu(), k)
        else:
            last[0] = ex.exchange_merge(ix, slot=slot)  # enqueued on torch's stream: runs under the steps already begun
        return last[0]

    def drain():
        while inflight:
            flat_finish_one()
        return last[0]

    def step():
        if wl == "flat":
            # the step's send block (the sets rotate: the all-gather + merge of step i run on torch's stream under the search of
            # step i+1, which the library issues on its own stream; everything is complete at the fence that ends the timed region)
            b_idx, b_dist, b_cnt = ex.begin_step()
            if depth > 1:
                h = ix.flat_knn_device_begin(queries.data_ptr(), nq, k, b_idx.data_ptr(), b_dist.data_ptr(), b_cnt.data_ptr(),
                                             stream=torch.cuda.current_stream().cuda_stream)
                inflight.append((h, ex.slot, (b_idx, b_dist, b_cnt)))
                if len(inflight) >= depth:
                    flat_finish_one()
                return last[0]
            ix.flat_knn_device(queries.data_ptr(), nq, k, b_idx.data_ptr(), b_dist.data_ptr(), b_cnt.data_ptr())
            if host_xchg:
                return allgather_merge(b_idx.cpu(), b_dist.cpu(), b_cnt.cpu(), k)
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

    def timed(index, fn, steps, warmup, drain=None):
        """W untimed + EXACTLY `steps` timed calls of fn between fences; the library's HIP-event records cover the timed calls.
        drain (pipelined steps): completes the steps still in flight -- before the opening fence, so that the timed region starts
        with an empty pipeline, and before the closing one, so that all `steps` steps are answered inside it"""
        r = None
        for _ in range(warmup):
            r = fn()
        if drain is not None:
            r = drain() or r
        index.prof_enable(True)
        index.prof_reset()
        fence()
        t0 = time.perf_counter()
        marks = [t0]
        for _ in range(steps):
            r = fn()
            marks.append(time.perf_counter())
        if drain is not None:
            r = drain() or r
        fence()
        el = time.perf_counter() - t0
        index.prof_enable(False)
        marks[-1] = t0 + el  # the closing fence belongs to the last step
        step_times.clear()
        step_times.extend((b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:]))
        return el, r

    step_times = []  # ms per step of the LAST timed() region (host clock between consecutive step returns)

    def step_stats():
        """host clock between consecutive step returns inside the one timed region (a Flat / PQ / HNSW step at N=1 ends in the library's own
        stream synchronisation; with N>1 the exchange of step i overlaps step i+1, so single steps are enqueue-to-enqueue and only the
        total is fenced)"""
        st = sorted(step_times)
        if not st:
            return None
        return {"min": round(st[0], 4), "median": round(st[len(st) // 2], 4), "max": round(st[-1], 4)}

    elapsed, res = timed(ix, step, args.steps, args.warmup, drain if depth > 1 else None)
    head_steps = step_stats()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if wl == "flat":
        roofline = flat_roofline(ix, r1 - r0, dim, nq)
        if roofline and world == 1:
            roofline.update(pmc_traffic(roofline["kernel"], r1 - r0, dim, nq) or {})
        if roofline and depth > 1:
            roofline["overlap_note"] = (f"{depth} steps in flight: the HIP-event duration of a corpus pass includes the CUs it waited for while the "
                                        "neighbouring steps' exact stage / query preparation ran beside it (passes themselves take turns)")
            if world == 1:
                # the same kernel with nothing beside it: a short region of synchronous calls right after the timed one
                el1, _ = timed(ix, lambda: ix.flat_knn_device(queries.data_ptr(), nq, k, o_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr()),
                               min(args.steps, 10), 2)
                r1s = flat_roofline(ix, r1 - r0, dim, nq) or {}
                roofline["one_step_in_flight"] = {"qps": round(nq * min(args.steps, 10) / el1, 1), "ms_per_step": round(el1 / min(args.steps, 10) * 1e3, 3),
                                                  "avg_launch_ms": r1s.get("avg_launch_ms"), "achieved": r1s.get("achieved"), "frac": r1s.get("frac"),
                                                  "unit": "GB/s"}
    else:
        # pq_adc: code bytes of one scan = rows x ceil(m*n_bits/8), one scan serves the queries whose LUTs sit side by side
        # in LDS; hnsw: n_dist x (dim*4 + 4) + n_expanded x max_m0*4 counted by the kernel
        kernel = {"pq_flat": "pq_adc", "hnsw": "hnsw", "hnsw_pq": "hnsw", "ivf": "ivf_rerank"}[wl]
        if wl == "ivf":  # the scan's certified pre-pass (8-bit tier, fp16 tier) holds the dominant kernel when it runs
            kernel = next((kn for kn in ("ivf_q8", "ivf_half") if ix.prof_get(kn)["launches"]), kernel)
        roofline = hbm_roofline(kernel, ix.prof_get(kernel))
        if wl == "ivf" and roofline:
            offers, kept, kept8 = ix.get_stat("ivf_last_offers"), ix.get_stat("ivf_last_kept"), ix.get_stat("ivf_last_kept_q8")
            fetched = ix.get_stat("ivf_last_rows_fetched_q8")
            if kernel == "ivf_q8" and fetched:
                roofline["rows_fetched_per_step"] = fetched
                roofline["visits_per_fetched_row"] = round(offers / fetched, 2)
            roofline["units_per_launch"] = {"ivf_q8": ("rows of the visited clusters x (dim + 12) bytes of the 8-bit image, each read ONCE per step (cluster-major "
                                                       "k_ivf_q8_bounds_cm scores a row against every query that visits its cluster from registers) + 8 B of "
                                                       "bounds per offer" if fetched else "offers x (dim + 12) bytes of the 8-bit image (k_ivf_q8_bounds)") +
                                                      "; the fp16 tier reads dim*2 bytes for the offers it passes on, the exact stage dim*4 for what is left",
                                            "ivf_half": "offers x (dim*2 + 4) bytes of the fp16 image (k_ivf_half_bounds); the exact stage then fetches "
                                                        "dim*4 bytes for the offers the pre-pass kept",
                                            "ivf_rerank": "offers x (dim*4 + 4) bytes (k_rerank_t)"}[kernel]
            roofline["offers_per_query"] = round(offers / max(q1 - q0, 1), 1)
            if kernel == "ivf_q8":
                roofline["kept_by_8bit_tier_per_query"] = round(kept8 / max(q1 - q0, 1), 1)
            roofline["kept_for_exact_stage_per_query"] = round(kept / max(q1 - q0, 1), 1)
            later = {"ivf_q8": ("ivf_half", "ivf_rerank"), "ivf_half": ("ivf_rerank",)}.get(kernel, ())
            if later:
                roofline["later_stages"] = [{"kernel": kn, "avg_launch_ms": round(ix.prof_get(kn)["ms"] / max(ix.prof_get(kn)["launches"], 1), 4),
                                             "bytes_per_launch": ix.prof_get(kn)["bytes"] / max(ix.prof_get(kn)["launches"], 1)} for kn in later]
            alg = offers * (dim * 4 + 4)  # SURVEY 8(d)-style figure: every offered row as f32
            tot_ms = sum(ix.prof_get(kn)["ms"] for kn in ("ivf_q8", "ivf_half", "ivf_rerank")) / max(ix.prof_get("ivf_rerank")["launches"], 1)
            roofline["f32_equivalent_GBps"] = round(alg / (tot_ms * 1e-3) / 1e9, 1) if tot_ms > 0 else None
    if roofline and attainable and roofline.get("bound") != "hbm":  # (cooperative sets: the byte rates sit in hbm_operand_rate)
        h = roofline.get("hbm_operand_rate")
        if h:
            h["attainable_peak_GBps"] = attainable
            h["frac_of_attainable"] = round(h["achieved"] / attainable, 4)
    elif roofline and attainable:
        roofline["attainable_peak_GBps"] = attainable
        roofline["frac_of_attainable"] = round(roofline["achieved"] / attainable, 4)
        roofline["attainable_peak_how"] = ("vdb_stream_probe in this run: best of contiguous-chunk and grid-stride streaming reads of a 3.84-GB buffer, "
                                           "default and non-temporal loads (the non-temporal forms reach ~7.0 TB/s on this chip, the default ones ~6.2)")

    filter_work = None
    if wl == "flat" and world == 1 and roofline and roofline.get("kernel") == "flat_i8":
        # what the 8-bit pass's exact stage had to do on THIS data: one more step with per-query statistics (after the timed region)
        if depth > 1:
            drain()
        ix.set_param("flat_i8_stats", 1)
        r0_ = ix.get_stat("flat_i8_redo")
        ix.flat_knn_device(queries.data_ptr(), nq, k, o_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr())
        sq_ = max(ix.get_stat("flat_i8_stat_queries"), 1)
        filter_work = {"queries": sq_, "hits_per_query_mean": round(ix.get_stat("flat_i8_hits_sum") / sq_, 1), "hits_per_query_max": ix.get_stat("flat_i8_hits_max"),
                       "queries_by_rounds_of_63_rows": {str(r_): ix.get_stat(f"flat_i8_rounds_{r_}") for r_ in range(9) if ix.get_stat(f"flat_i8_rounds_{r_}")},
                       "passed_on": ix.get_stat("flat_i8_redo") - r0_}
        ix.set_param("flat_i8_stats", 0)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    if args.dump:
        np.savez(args.dump, idx=res[0].cpu().numpy(), dist=res[1].cpu().numpy(), cnt=res[2].cpu().numpy())
    qps = nq * args.steps / elapsed
    dname = "L2Sqr" if args.dist == "l2sqr" else "Cosine"
    names = {"flat": ("Flat brute force", "flat_knn_gist1m"), "pq_flat": (f"PQ-Flat {args.pq_bits}-bit m={dim // 3}, ADC ef={ef}", "pq_flat_knn_gist1m" + ("_8bit" if args.pq_bits == 8 else "")),
             "hnsw": (f"HNSW M=16 efc=200, ef={ef}", f"hnsw_knn_gistlike_{n}"),
             "hnsw_pq": (f"HNSW M=16 efc=200 + PQ 4-bit m={dim // 3}, ef={ef}", f"hnsw_pq_knn_gistlike_{n}"),
             "ivf": (f"IVF {int(round(n ** 0.5))} clusters, n_probes={ef}", f"ivf_knn_gistlike_{n}")}[wl]
    par = {"flat": f"row-shard x{world}", "pq_flat": f"row-shard x{world}", "hnsw": f"replica x{world}, queries split",
           "hnsw_pq": f"replica x{world}, queries split", "ivf": f"replica x{world}, queries split"}[wl]
    out = {
        "metric": f"queries/sec at recall@10, Gist1M d=960 ({names[0]}, {dname}, k=10)",
        "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "step_ms": head_steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None,
        # the arithmetic type of everything that leaves the library (strict-order f32 folds); the Flat filter in front of it ranks
        # its shortlist from a scaled fp16 mirror and every answer is certified against the f32 rows
        "dtype": {"flat_half": "f32 (certified fp16 filter pass)", "flat_i8": "f32 (int8 lower-bound filter pass)"}.get((roofline or {}).get("kernel"), "f32"),
        "data": data_name,
        "config": {"workload": names[1], "rows": n, "dim": dim, "queries_per_step": nq, "k": k, "dist": dname,
                   "parallelism": par if world > 1 else "single GPU", "steps_in_flight": depth},
        "roofline": roofline, "recall_at_10": None,
    }
    if args.param:
        out["config"]["tuning_switches"] = list(args.param)
    if wl == "flat":
        out["config"]["queries_per_corpus_pass"] = 128 if (nq > 64 or (roofline or {}).get("kernel") in ("flat_half", "flat_i8")) else 64
        out["fallback_queries"] = ix.flat_fallback_count()
        out["i8_pass"] = {"queries": ix.get_stat("flat_i8_queries"), "second_attempts": ix.get_stat("flat_i8_second_queries"), "passed_on_to_fp16": ix.get_stat("flat_i8_redo")}
        out["half_pass"] = {"queries": ix.get_stat("flat_half_queries"), "redone_split_bf16": ix.get_stat("flat_half_redo")}
        out["hbm_bytes_per_row"] = ix.get_stat("hbm_bytes_per_row")
        if filter_work:
            out["filter_work"] = filter_work
    else:
        out["config"]["ef"] = ef
    if wl == "pq_flat" and args.pq_bits == 8:
        # what the quantised pass of the 8-bit codes handed on (whole run incl. warm-up and the parity call)
        out["pq8_pass"] = {"queries": ix.get_stat("pq_adc16_queries"), "candidates": ix.get_stat("pq_q8_hits_sum"),
                           "largest_list": ix.get_stat("pq_q8_hits_max"), "lists_overflowed_to_f32_scan": ix.get_stat("pq_q8_overflow"),
                           "lists_short_to_f32_scan": ix.get_stat("pq_q8_short")}
    if wl == "ivf":
        out["config"]["host_build_s"] = round(build_s, 1)
    if wl in ("hnsw", "hnsw_pq"):
        nd, ne = ix.hnsw_last_stats()
        out["config"]["host_build_s"] = round(build_s, 1)
        out["hnsw_work_per_query"] = {"n_dist": round(nd / max(q1 - q0, 1), 1), "n_expanded": round(ne / max(q1 - q0, 1), 1)}

    # ---- CPU baseline + parity (rank 0, N=1 only) ------------------------------------------------------
    O = None
    if host_base is not None:
        from oracle import oracle as O

        O.build()
        okind = O.L2SQR if args.dist == "l2sqr" else O.COSINE
        gt = None
        if wl != "flat":
            # recall is against Flat ground truth (gen_gnd.rs:54-72), taken from the GPU Flat path of the same index
            t_idx = torch.zeros((nq, k), dtype=torch.int64, device=device)
            t_dist = torch.zeros((nq, k), dtype=torch.float32, device=device)
            t_cnt = torch.zeros((nq,), dtype=torch.int64, device=device)
            ix.flat_knn_device(queries.data_ptr(), nq, k, t_idx.data_ptr(), t_dist.data_ptr(), t_cnt.data_ptr())
            gt = t_idx.cpu().numpy().astype(np.uint64)
        out.update(cpu_and_parity(O, wl, ix, host_base, queries, res, gt, min(args.cpu_queries, nq), k, ef, okind, threads, n, dim, nq))

    if legs_on:
        legs = {}
        outs = (o_idx, o_dist, o_cnt)
        bytes_per_row = {"flat": out.get("hbm_bytes_per_row")}
        legs["flat_f32_operands"] = f32 = leg_flat_f32(ix, timed, step_stats, queries, nq, k, outs, args, n, dim, res, attainable)
        # SURVEY 8(d)'s own figure -- N*d*4 bytes per corpus pass -- where the driver's record keeps it: the same 1000-query step
        # with the fp16 pass off, i.e. the split-bf16 kernel streaming 4 B/element (the ">= 70 % of HBM roofline" target)
        fr = f32["roofline"] or {}
        out["roofline"]["f32_operand_leg"] = {"qps": f32["value"], "ms_per_step": f32["ms_per_step"], "kernel": fr.get("kernel"),
                                              "avg_launch_ms": fr.get("avg_launch_ms"), "bytes_per_launch": fr.get("bytes_per_launch"),
                                              "achieved": fr.get("achieved"), "frac": fr.get("frac"), "unit": "GB/s",
                                              "bytes_def": "SURVEY 8(d): corpus passes x N x d x 4 B, which this kernel really streams (cooperative sets off for this leg)",
                                              "results_equal_headline": f32["results_equal_headline"]}
        bytes_per_row["flat_with_redo_tier"] = f32["hbm_bytes_per_row"]
        if (out["roofline"] or {}).get("kernel") == "flat_i8":  # the previous rounds' headline path beside the new one
            legs["flat_fp16_pass"] = h16 = leg_flat_f32(ix, timed, step_stats, queries, nq, k, outs, args, n, dim, res, attainable, fp16=True)
            hr = h16["roofline"] or {}
            out["roofline"]["fp16_pass_leg"] = {"qps": h16["value"], "ms_per_step": h16["ms_per_step"], "kernel": hr.get("kernel"),
                                                "avg_launch_ms": hr.get("avg_launch_ms"), "bytes_per_launch": hr.get("bytes_per_launch"),
                                                "achieved": (hr.get("hbm_operand_rate") or hr).get("achieved"), "frac": (hr.get("hbm_operand_rate") or hr).get("frac"), "unit": "GB/s",
                                                "cooperative_set": hr.get("cooperative_set"), "matrix_pipe_frac": hr.get("frac") if hr.get("bound") == "mfma" else None,
                                                "bytes_def": "corpus passes x N x d x 2 B (the scaled fp16 mirror) CONSUMED; with cooperative sets about 1 / set size of them comes from HBM",
                                                "results_equal_headline": h16["results_equal_headline"]}
        for b in (32, 1):
            legs[f"flat_B{b}"] = leg_flat_small(ix, timed, queries, b, k, outs, args, n, dim, res, attainable)
        ix.close()
        del ix
        torch.cuda.empty_cache()
        if args.dist == "l2sqr" and not from_file:
            legs["flat_cosine"] = leg_flat_other_metric(vdb, O, torch, device, local_rank, timed, step_stats, queries, outs, args, n, dim, nq, k, host_base,
                                                        threads, attainable, gen)
        legs["config1_gist_1000"] = leg_config1(vdb, O, torch, device, local_rank, threads, k)
        del host_base
        legs.update(legs_ann(vdb, O, torch, device, local_rank, timed, step_stats, args, threads, attainable, from_file))
        for nm in ("pq_flat", "ivf", "hnsw", "hnsw_pq"):
            bytes_per_row[nm] = legs[nm].get("hbm_bytes_per_row")
        out["hbm_bytes_per_row"] = dict(bytes_per_row, note="resident HBM bytes per 960-d row by index kind: f32 rows + norms + the mirrors / images / codes / "
                                        "links that kind keeps (flat = f32 rows + fragment-ordered fp16 mirror; the redo tier's split-bf16 mirror, the "
                                        "row-major fp16 image of the walks / IVF scan and IVF's 8-bit image are built on first use)")
        out["legs"] = legs
    mp = (out.get("roofline") or {}).get("matrix_pipe")
    if mp is not None and world == 1 and mp.get("instruction", "").endswith(("f16", "i8")):
        # the matrix pipe's rate under sustained load on this box (the chip is power-limited well below the nominal peaks);
        # measured AFTER every timed region: half a second of nothing but MFMAs is not what a timed step should start behind
        from lab_1806_vec_db_amd.index import mfma_probe

        is8 = mp["instruction"].endswith("i8")
        best = max((tuple(round(v, 2) for v in mfma_probe(local_rank, w, 200_000, i8=is8)) + (w,) for w in (1, 2, 4)), key=lambda t: t[0])
        unit = "TOPs" if is8 else "TFLOPs"
        mp.update({f"sustained_peak_{unit}": best[0], "clock_GHz_at_sustained_peak": best[1], "waves_per_simd_at_sustained_peak": best[2],
                   "frac_of_sustained": round(mp[f"achieved_{unit}"] / best[0], 4),
                   "note": "sustained peak = vdb_mfma_probe(_i8) at the end of this run: the same instruction back to back on every SIMD, "
                           "best of 1 / 2 / 4 waves per SIMD; the chip lowers its clock under that load"})
    # the full record goes to a side file (and stays out of stdout); the FINAL stdout line is the compact one the driver parses
    full_path = args.full_out or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(full_path)), exist_ok=True)
        with open(full_path, "w") as fh:
            json.dump(out, fh)
            fh.write("\n")
        shown = os.path.relpath(full_path, ROOT)
    except OSError:
        shown = None
    print(compact_line(out, shown), flush=True)
    if world > 1 or force_x:
        dist.destroy_process_group()


def cpu_and_parity(O, wl, ix, host_base, queries, res, gt, ncpu, k, ef, okind, threads, n, dim, nq):
    """the CPU oracle on `ncpu` of the queries, one query per thread (mirrors rayon par_iter, examples/bench.rs:414-416),
    timed; its answers are the parity reference of the GPU results `res`; recall@10 against `gt` (Flat ground truth; None:
    the oracle's Flat answers themselves)"""
    from concurrent.futures import ThreadPoolExecutor

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
        truth = gt[:ncpu]
        if wl == "pq_flat":
            pq = ix.pq_export()
            opq = O.PQ.from_centroids(dim, pq["m"], pq["n_bits"], okind, pq["centroids"])
            opq.set_codes(pq["codes"])  # GPU-encoded codes (bit-equal to the oracle's encoder, tests/test_pq_gpu.py)
            fn = lambda q: O.flat_knn_pq(host_base, opq, hq[q], k, ef, okind)  # noqa: E731
            what = "FlatIndex::knn_pq"
        elif wl == "ivf":
            ex_ivf = ix.ivf_export()  # centroids and clusters are inputs of the oracle (assignment parity: tests/test_ivf_gpu.py)
            oiv = O.IVF(host_base, ex_ivf["centroids"], okind, assign=ex_ivf["assign"])
            fn = lambda q: oiv.knn(hq[q], k, ef)  # noqa: E731
            what = "IVFIndex::knn_with_ef on the same centroids and clusters"
        elif wl == "hnsw_pq":
            pq = ix.pq_export()
            opq = O.PQ.from_centroids(dim, pq["m"], pq["n_bits"], okind, pq["centroids"])
            opq.set_codes(pq["codes"])
            oh = O.HNSW.from_graph(host_base, okind, 16, 200, ix.hnsw_export())
            fn = lambda q: oh.knn_pq(opq, hq[q], k, ef)  # noqa: E731
            what = "HNSWIndex::knn_pq on the same graph and PQ table"
        else:
            oh = O.HNSW.from_graph(host_base, okind, 16, 200, ix.hnsw_export())
            fn = lambda q: oh.knn(hq[q], k, ef)  # noqa: E731
            what = "HNSWIndex::knn_with_ef on the same graph"
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as pool:  # ctypes releases the GIL: one query per thread
            r = list(pool.map(fn, range(ncpu)))
        cpu_s = time.perf_counter() - t0
        ci = np.stack([np.pad(x[0], (0, k - len(x[0]))) for x in r]).astype(np.uint64)
        cd = np.stack([np.pad(x[1], (0, k - len(x[1]))) for x in r]).astype(np.float32)
    return {
        "recall_at_10": float(np.mean([O.recall(truth[q], gi[q]) for q in range(ncpu)])),
        "parity": {"queries_checked": ncpu, "indices_identical": bool(np.array_equal(gi, ci)),
                   "distances_bit_exact": bool(np.array_equal(gd, cd))},
        "cpu_baseline": {"value": round(ncpu / cpu_s, 2), "unit": "queries/s", "cores": threads, "kind": "port",
                         "sample": f"{what}: {ncpu} of the {nq} queries against the full {n}x{dim} corpus, "
                                   f"one query per thread (mirrors rayon par_iter, examples/bench.rs:414-416)"}}


def with_attainable(roofline, attainable):
    if roofline and attainable and roofline.get("bound") != "hbm":
        h = roofline.get("hbm_operand_rate")
        if h:
            h["attainable_peak_GBps"] = attainable
            h["frac_of_attainable"] = round(h["achieved"] / attainable, 4)
        return roofline
    if roofline and attainable:
        roofline["attainable_peak_GBps"] = attainable
        roofline["frac_of_attainable"] = round(roofline["achieved"] / attainable, 4)
    return roofline


def leg_flat_f32(ix, timed, step_stats, queries, nq, k, outs, args, n, dim, ref, attainable, fp16=False):
    """the headline step with the 8-bit and fp16 first passes off: every corpus pass streams the 4-B/element split-bf16 mirror,
    i.e. SURVEY 8(d)'s N*d*4 algorithmic bytes; results must equal the headline's bit for bit.  fp16=True: only the 8-bit
    pass off (the headline path of rounds 2 - 3: 2 B/element)"""
    ref_idx, ref_dist = ref[0].clone(), ref[1].clone()
    ix.set_param("flat_half", 2 if fp16 else 1)
    ix.set_param("flat_i8", 1)
    # The N*d*4 leg is the measurement of SURVEY 8(d)'s HBM figure: the filter's cooperative sets (a row from HBM once per 8 groups, the
    # rest from the XCDs' L2) are switched off for it, so that every pass really streams its 4 B/element from HBM.  The fp16 leg runs
    # the way a caller gets it (sets on; its roofline is then quoted on the matrix pipe, the byte rates in hbm_operand_rate).
    ix.set_param("flat_gemm_coop", 0 if fp16 else 1)
    fn = lambda: ix.flat_knn_device(queries.data_ptr(), nq, k, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr())  # noqa: E731
    el, _ = timed(ix, fn, args.steps, max(1, args.warmup))
    r = with_attainable(flat_roofline(ix, n, dim, nq), attainable)
    same = bool((outs[0] == ref_idx).all().item()) and bool((outs[1] == ref_dist).all().item())
    ix.set_param("flat_half", args.half)
    ix.set_param("flat_i8", args.i8)
    ix.set_param("flat_gemm_coop", 0)
    return {"value": round(nq * args.steps / el, 1), "unit": "queries/s", "steps": args.steps,
            "ms_per_step": round(el / args.steps * 1e3, 3), "step_ms": step_stats(), "queries_per_step": nq, "queries_per_corpus_pass": 128,
            "roofline": r, "results_equal_headline": same, "fallback_queries_total": ix.flat_fallback_count(),
            "hbm_bytes_per_row": ix.get_stat("hbm_bytes_per_row")}


def leg_flat_other_metric(vdb, O, torch, device, local_rank, timed, step_stats, queries, outs, args, n, dim, nq, k, host_base, threads, attainable, gen):
    """the headline step under Cosine -- the reference's DEFAULT table distance (pyo3/mod.rs:73, distance/mod.rs:60-69): the same corpus
    and queries, the 8-bit pass on unit rows; parity against the oracle's Cosine answers on a bounded sample (the CPU needs ~70 ms per query
    and core for a Cosine scan of 1M x 960)"""
    base = gen(torch, n, dim, 1806, device)
    torch.cuda.synchronize()  # (the library copies the rows on its own stream)
    cx = vdb.GpuIndex(dim, "cosine", device=local_rank)
    cx.add_device(base.data_ptr(), n)
    del base
    torch.cuda.empty_cache()
    fn = lambda: cx.flat_knn_device(queries.data_ptr(), nq, k, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr())  # noqa: E731
    el, _ = timed(cx, fn, args.steps, max(1, args.warmup))
    r = with_attainable(flat_roofline(cx, n, dim, nq), attainable)
    leg = {"value": round(nq * args.steps / el, 1), "unit": "queries/s", "steps": args.steps, "ms_per_step": round(el / args.steps * 1e3, 3),
           "step_ms": step_stats(), "queries_per_step": nq, "dist": "Cosine", "roofline": r,
           "i8_pass": {"queries": cx.get_stat("flat_i8_queries"), "passed_on_to_fp16": cx.get_stat("flat_i8_redo")},
           "hbm_bytes_per_row": cx.get_stat("hbm_bytes_per_row")}
    if O is not None and host_base is not None:
        ncpu = min(96, nq, max(args.cpu_queries, 0))
        if ncpu:
            t0 = time.perf_counter()
            ci, cd, cc = O.flat_knn_batch(host_base, queries[:ncpu].cpu().numpy(), k, O.COSINE, nthreads=threads)
            cpu_s = time.perf_counter() - t0
            gi, gd = outs[0][:ncpu].cpu().numpy().astype(np.uint64), outs[1][:ncpu].cpu().numpy()
            leg["parity"] = {"queries_checked": ncpu, "indices_identical": bool(np.array_equal(gi, ci)), "distances_bit_exact": bool(np.array_equal(gd, cd))}
            leg["recall_at_10"] = float(np.mean([O.recall(ci[q], gi[q]) for q in range(ncpu)]))
            leg["cpu_baseline"] = {"value": round(ncpu / cpu_s, 2), "unit": "queries/s", "cores": threads, "kind": "port",
                                   "sample": f"FlatIndex::knn (Cosine): {ncpu} of the {nq} queries against the full {n}x{dim} corpus, one query per thread"}
    cx.close()
    return leg


def leg_flat_small(ix, timed, queries, b, k, outs, args, n, dim, ref, attainable):
    """calls of b queries (B=32 is the batch the HBM-bound roofline QPS = B*BW/(N*d*4) is quoted on; B=1 the
    reference's own per-call semantics); the first b rows of the headline's results are the check"""
    ref_idx, ref_dist = ref[0][:b].clone(), ref[1][:b].clone()
    steps = max(args.steps, 20)
    fn = lambda: ix.flat_knn_device(queries.data_ptr(), b, k, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr())  # noqa: E731
    el, _ = timed(ix, fn, steps, 2)
    r = with_attainable(flat_roofline(ix, n, dim, b), attainable)
    same = bool((outs[0][:b] == ref_idx).all().item()) and bool((outs[1][:b] == ref_dist).all().item())
    return {"value": round(b * steps / el, 1), "unit": "queries/s", "steps": steps, "ms_per_step": round(el / steps * 1e3, 4),
            "queries_per_step": b, "roofline": r, "results_equal_headline": same,
            "roofline_qps_8d": round(b * HBM_PEAK_GBS * 1e9 / (n * dim * 4), 1)}


def leg_config1(vdb, O, torch, device, local_rank, threads, k):
    """BASELINE config 1 (config/gist_1000.toml): the reference's own 1000 x 960 base and 1000 queries, Flat, L2Sqr, k=10.
    Protocol of examples/bench.rs:403-433: wall time of the whole query loop / nq -- the CPU oracle serial and with one
    query per thread on all cores; the GPU (i) the same loop, one vdb_flat_knn call per query with host pointers, and (ii)
    the 1000 queries in ONE call.  Every GPU answer is compared with the oracle's."""
    g = os.path.join(ROOT, "tests", "golden")
    base = np.fromfile(os.path.join(g, "gist_1000.bin"), dtype=np.float32).reshape(1000, 960)
    test = np.fromfile(os.path.join(g, "gist_test.bin"), dtype=np.float32).reshape(1000, 960)
    nq = test.shape[0]
    ix = vdb.GpuIndex(960, "l2sqr", device=local_rank)
    ix.batch_add(base)
    ix.flat_knn(test[:8], k)  # warm-up (workspace allocation)
    t0 = time.perf_counter()
    per = [ix.flat_knn(test[q], k) for q in range(nq)]
    t_loop = time.perf_counter() - t0
    # the same one-query calls from `threads` host threads at once -- what the CPU figure beside it is (one query per thread,
    # bench.rs:414-416): read-side calls are re-entrant, every thread gets its own workspace and stream, the kernels overlap.
    # Straight through the C ABI with per-thread result arrays (the Python wrapper's allocations would serialise on the GIL).
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    from lab_1806_vec_db_amd import _lib as L

    lib = L.load()
    mt_i = np.zeros((nq, k), dtype=np.uint64)
    mt_d = np.zeros((nq, k), dtype=np.float32)
    mt_c = np.zeros(nq, dtype=np.uint64)

    def worker(t):
        for q in range(t, nq, threads):
            rc = lib.vdb_flat_knn(ix._h, test[q].ctypes.data_as(L.f32p), 1, 960, k, mt_i[q].ctypes.data_as(L.u64p),
                                  mt_d[q].ctypes.data_as(L.f32p), mt_c[q:].ctypes.data_as(L.u64p))
            if rc != 0:
                raise RuntimeError("vdb_flat_knn failed")

    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(worker, range(threads)))  # warm-up: one workspace per thread
        t0 = time.perf_counter()
        list(pool.map(worker, range(threads)))
        t_mt = time.perf_counter() - t0
    ix.flat_knn(test, k)
    t0 = time.perf_counter()
    gi, gd, gc = ix.flat_knn(test, k)
    t_batch = time.perf_counter() - t0
    leg = {"data": "tests/golden/gist_1000.bin x gist_test.bin (the reference's data/ files, sha256 in SURVEY 8c)", "rows": 1000,
           "dim": 960, "queries": nq, "k": k, "dist": "L2Sqr",
           "gpu_per_query_calls": {"ms_per_query": round(t_loop / nq * 1e3, 4), "value": round(nq / t_loop, 1), "unit": "queries/s",
                                   "note": "host pointers, one call per query (PCIe-inclusive), as bench.rs loops"},
           "gpu_per_query_calls_threads": {"threads": threads, "value": round(nq / t_mt, 1), "unit": "queries/s",
                                           "note": "one vdb_flat_knn call per query (host pointers) from this many host threads at once"},
           "gpu_one_call": {"ms_per_query": round(t_batch / nq * 1e3, 5), "value": round(nq / t_batch, 1), "unit": "queries/s",
                            "note": "host pointers, 1000 queries in one call (PCIe-inclusive)"}}
    if O is not None:
        t0 = time.perf_counter()
        ci, cd, cc = O.flat_knn_batch(base, test, k, O.L2SQR, nthreads=1)
        t_ser = time.perf_counter() - t0
        t0 = time.perf_counter()
        O.flat_knn_batch(base, test, k, O.L2SQR, nthreads=threads)
        t_par = time.perf_counter() - t0
        loop_i = np.stack([p[0] for p in per]).astype(np.uint64)
        loop_d = np.stack([p[1] for p in per])
        leg["cpu_baseline"] = {"value": round(nq / t_par, 1), "unit": "queries/s", "cores": threads, "kind": "port",
                               "sample": "FlatIndex::knn, all 1000 queries, one query per thread",
                               "serial": {"value": round(nq / t_ser, 1), "ms_per_query": round(t_ser / nq * 1e3, 4), "cores": 1}}
        leg["parity"] = {"queries_checked": nq,
                         "indices_identical": bool(np.array_equal(gi.astype(np.uint64), ci) and np.array_equal(loop_i, ci) and np.array_equal(mt_i, ci)),
                         "distances_bit_exact": bool(np.array_equal(gd, cd) and np.array_equal(loop_d, cd) and np.array_equal(mt_d, cd))}
        leg["recall_at_10"] = float(np.mean([O.recall(ci[q], gi[q].astype(np.uint64)) for q in range(nq)]))
    ix.close()
    return leg


def legs_ann(vdb, O, torch, device, local_rank, timed, step_stats, args, threads, attainable, from_file=False):
    """BASELINE configs 3 and 4 on ONE low-rank gist-like corpus (recall on per-dimension-Gaussian rows is uninformative:
    0.2 for HNSW, 0.5 for PQ): PQ-Flat ADC on 1M rows, HNSW on the first --hnsw-rows of them.  With
    --base-file / --query-file the file's rows are used instead (real Gist1M when present)."""
    n, dim, nq, k = 1_000_000, 960, 1000, 10
    legs = {}
    if from_file:
        base = load_rows_file(torch, args.base_file, dim, 0, n, device)
        queries = load_rows_file(torch, args.query_file, dim, 0, nq, device)
        data_name = f"file: {os.path.basename(args.base_file)} x {os.path.basename(args.query_file)}"
    else:
        base = gist_lowrank_gpu(torch, n, dim, 1806, device)
        queries = gist_lowrank_gpu(torch, nq, dim, 1807, device)
        data_name = "synthetic (low-rank gist-like)"
    o_idx = torch.zeros((nq, k), dtype=torch.int64, device=device)
    o_dist = torch.zeros((nq, k), dtype=torch.float32, device=device)
    o_cnt = torch.zeros((nq,), dtype=torch.int64, device=device)
    t_idx = torch.zeros((nq, k), dtype=torch.int64, device=device)
    torch.cuda.synchronize()  # (the library reads the rows on its own streams)
    ncpu = min(args.cpu_queries, nq)
    okind = O.L2SQR if O is not None else 0

    def run(ix, wl, rows, ef, fn, extra_cfg, nq=nq, queries=queries, outs=(o_idx, o_dist, o_cnt), t_idx=t_idx):
        o_idx, o_dist, o_cnt = outs
        el, _ = timed(ix, fn, args.steps, max(1, args.warmup))
        kernel = {"pq_flat": "pq_adc", "hnsw": "hnsw", "hnsw_pq": "hnsw", "ivf": "ivf_rerank"}[wl]
        if wl == "ivf":  # the scan's certified cascade holds the dominant kernel when it runs
            kernel = next((kn for kn in ("ivf_q8", "ivf_half") if ix.prof_get(kn)["launches"]), kernel)
        leg = {"value": round(nq * args.steps / el, 1), "unit": "queries/s", "steps": args.steps,
               "ms_per_step": round(el / args.steps * 1e3, 3), "step_ms": step_stats(), "data": data_name,
               "hbm_bytes_per_row": ix.get_stat("hbm_bytes_per_row"),
               "config": dict({"rows": rows, "dim": dim, "queries_per_step": nq, "k": k, "dist": "L2Sqr", "ef": ef}, **extra_cfg),
               "roofline": with_attainable(hbm_roofline(kernel, ix.prof_get(kernel)), attainable)}
        if wl in ("hnsw", "hnsw_pq"):
            nd, ne = ix.hnsw_last_stats()
            leg["hnsw_work_per_query"] = {"n_dist": round(nd / nq, 1), "n_expanded": round(ne / nq, 1)}
            if wl == "hnsw":
                # random whole-row gathers do not reach the streaming peak: MI355X_MICROARCH.md measures 5.5 - 5.8 TB/s for them
                r = leg["roofline"]
                r["gather_peak_GBps"] = 5500.0
                r["frac_of_gather_peak"] = round(r["achieved"] / 5500.0, 4)
                # n_dist f32 rows (+ cached norm) + n_expanded link rows.  The walk's certified half-precision
                # pre-pass reads 2 B/element for every row and the f32 row only when it cannot rule the row out, so the
                # bytes it asks for (`achieved`) are fewer than 8(d)'s; both rates are given
                dropped = ix.get_stat("hnsw_half_dropped")
                alg = nd * (dim * 4 + 4) + ne * 32 * 4
                r["algorithmic_bytes_8d"] = alg
                r["achieved_8d"] = round(alg / (r["avg_launch_ms"] * 1e-3) / 1e9, 1)
                r["frac_8d"] = round(r["achieved_8d"] / HBM_PEAK_GBS, 4)
                r["frac_of"] = "bytes requested: fp16 image rows for all scored neighbours + f32 rows for those not ruled out"
                leg["hnsw_work_per_query"]["ruled_out_by_half_precision_pre_pass"] = round(dropped / nq, 1)
            leg["roofline"]["units_per_launch"] = ("n_half x dim*2 + (n_dist - ruled_out) x dim*4 + n_dist x 4 + n_expanded x max_m0*4 bytes, counted by the kernel" if wl == "hnsw" else
                                                   "n_dist x 160-B code rows + n_expanded x max_m0*4 bytes (ADC walk: latency-bound by construction), counted by the kernel")
        elif wl == "ivf":
            r = leg["roofline"]
            offers, kept, kept8, fetched = (ix.get_stat(n_) for n_ in ("ivf_last_offers", "ivf_last_kept", "ivf_last_kept_q8", "ivf_last_rows_fetched_q8"))
            leg["ivf_work_per_query"] = {"offers": round(offers / nq, 1), "kept_by_8bit_tier": round(kept8 / nq, 1), "exact_stage": round(kept / nq, 1)}
            if fetched:
                r["rows_fetched_per_step"] = fetched
                r["visits_per_fetched_row"] = round(offers / fetched, 2)
            r["units_per_launch"] = ("rows of the visited clusters x (dim + 12) B of the 8-bit image, each read once per step by the cluster-major tier "
                                     "(k_ivf_q8_bounds_cm) + 8 B of bounds per offer; then dim*2 B per offer the tier keeps (fp16 tier) and dim*4 B per "
                                     "offer that reaches the exact stage" if fetched else "bytes of the dominant tier, counted by the library")
            alg = offers * (dim * 4 + 4)  # SURVEY 8(d)-style figure: every offered row as f32
            tot_ms = sum(ix.prof_get(kn)["ms"] for kn in ("ivf_q8", "ivf_half", "ivf_rerank")) / max(ix.prof_get("ivf_rerank")["launches"], 1)
            r["f32_equivalent_GBps"] = round(alg / (tot_ms * 1e-3) / 1e9, 1) if tot_ms > 0 else None
        else:
            r = leg["roofline"]
            r["units_per_launch"] = ("rows x ceil(m*n_bits/8) code bytes per scan; one scan (k_pq_adc16) serves the 8 queries whose "
                                     "16-bit tables share LDS; the 160-MB code mirror is re-read from the Infinity Cache, not HBM")
            # the scan's own bound is the LDS gather rate: one 16-B table entry per (row, group, 8 queries)
            lds_bytes = r["bytes_per_launch"] * 32.0  # m * 16 B per row = 32 x the row's m/2 code bytes (4-bit codes)
            lds_tbps = lds_bytes / (r["avg_launch_ms"] * 1e-3) / 1e12
            r["lds_gather"] = {"achieved_TBps": round(lds_tbps, 1), "peak_TBps": 150.0, "frac": round(lds_tbps / 150.0, 4),
                               "note": "ds_read_b128 aggregate of 256 CUs (MI355X_MICROARCH.md, LDS section); the kernel's binding resource"}
        if O is not None and ncpu > 0:
            ix.flat_knn_device(queries.data_ptr(), nq, k, t_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr())  # Flat ground truth
            gt = t_idx.cpu().numpy().astype(np.uint64)
            fn()
            hb = base[:rows].cpu().numpy()
            leg.update(cpu_and_parity(O, wl, ix, hb, queries, (o_idx, o_dist, o_cnt), gt, min(ncpu, nq), k, ef, okind, threads, rows, dim, nq))
            del hb
        return leg

    def small_call(ix, fn, nq_call=1000):
        """the same search as ONE call of nq_call queries (1000 = the size of the other legs' steps; 1 = the reference's own
        per-call shape, db.search): bounded by the longest walk / the fixed launches of a call, not a rate"""
        steps = args.steps if nq_call >= 100 else max(args.steps, 50)
        el, _ = timed(ix, fn, steps, 2)
        return {"queries_per_step": nq_call, "value": round(nq_call * steps / el, 1), "unit": "queries/s",
                "ms_per_step": round(el / steps * 1e3, 4), "step_ms": step_stats()}

    lat_ns = add_ns = None
    if attainable is not None:
        from lab_1806_vec_db_amd.index import fold_probe, latency_probe

        lat_ns = round(latency_probe(local_rank, 1 << 30, 20000), 1)
        add_ns = round(fold_probe(local_rank, 1 << 22), 3)

    def latency_floor(leg, one_call, round_trips, what, chain_adds):
        """A graph walk is a CHAIN: per expansion the popped node's link row, the visited words of its neighbours and the first
        lines of their rows are three dependent HBM accesses (hnsw_index.rs:258-291 cannot start one before the previous
        returned), and a neighbour that may enter the result set is scored by the reference's strict left fold -- chain_adds
        dependent f32 adds (dim for the exact walk, hnsw_index.rs:351-358; m table entries for the ADC walk, pq_table.rs:254-292)
        which no lane count shortens.  A call of <= 2048 queries is one round of resident walks, so its time is the longest
        chain, not bytes: floor = expansions per query x (round trips x dependent-load latency + chain_adds x dependent-add
        latency), both latencies measured in this run."""
        if lat_ns is None:
            return
        ne = leg["hnsw_work_per_query"]["n_expanded"]
        floor_ms = ne * (round_trips * lat_ns + chain_adds * add_ns) * 1e-6
        leg["roofline"]["latency_floor"] = {
            "dependent_load_ns": lat_ns, "round_trips_per_expansion": round_trips, "what": what, "expansions_per_query": ne,
            "dependent_add_ns": add_ns, "fold_chain_adds_per_expansion": chain_adds,
            "floor_ms_per_call": round(floor_ms, 4), "one_call_of_1000_ms": one_call["ms_per_step"],
            "frac": round(floor_ms / one_call["ms_per_step"], 4),
            "note": "vdb_latency_probe (pointer chase over a 1-GiB buffer) and vdb_fold_probe in this run; the mean walk, so calls bounded by their longest "
                    "walk sit further above it; the HBM-bytes fraction beside it is the yardstick of the large calls only"}

    # -- PQ-Flat: config/bench_pq_hnsw.toml:16-23 (n_bits 4, m = dim/3, k_means_size 10000, max_iter 20, tol 1e-6), ef = 100
    ix = vdb.GpuIndex(dim, "l2sqr", device=local_rank)
    ix.add_device(base.data_ptr(), n)
    tr = vdb.GpuIndex(dim, "l2sqr", device=local_rank)
    tr.add_device(base.data_ptr(), 10000)
    t_b = time.perf_counter()
    tr.pq_build(n_bits=4, m=dim // 3, train_n=0, max_iter=20, tol=1e-6, seed=42)
    cent = tr.pq_export()["centroids"]
    tr.close()
    ix.pq_attach(4, dim // 3, cent, None)
    pq_build_s = time.perf_counter() - t_b
    legs["pq_flat"] = run(ix, "pq_flat", n, 100,
                          lambda: ix.knn_pq_device(queries.data_ptr(), nq, k, 100, o_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr()),
                          {"workload": "pq_flat_knn_gist1m", "n_bits": 4, "m": dim // 3, "train_and_encode_s": round(pq_build_s, 1)})
    legs["pq_flat"]["one_call_of_1"] = small_call(
        ix, lambda: ix.knn_pq_device(queries.data_ptr(), 1, k, 100, o_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr()), 1)
    # the same table shape with 8-bit codes (n_bits = 8, pq_table.rs:142-145: 256 centroids per group, 320 B per code row): one query
    # per pass on a one-byte table in LDS (k_pq_adc8), so the scan is bound by the code bytes -- 320 MB per query
    tr8 = vdb.GpuIndex(dim, "l2sqr", device=local_rank)
    tr8.add_device(base.data_ptr(), 20000)
    tr8.pq_build(n_bits=8, m=dim // 3, train_n=0, max_iter=5, tol=1e-6, seed=42)
    cent8 = tr8.pq_export()["centroids"]
    tr8.close()
    ix.pq_attach(8, dim // 3, cent8, None)
    legs["pq_flat"]["n_bits_8"] = run(ix, "pq_flat", n, 100,
                                      lambda: ix.knn_pq_device(queries.data_ptr(), nq, k, 100, o_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr()),
                                      {"workload": "pq_flat_knn_gist1m_8bit", "n_bits": 8, "m": dim // 3})
    r8 = legs["pq_flat"]["n_bits_8"]["roofline"]
    if r8:
        r8.pop("lds_gather", None)
        r8["units_per_launch"] = ("queries x rows x m code bytes: one query per pass of k_pq_adc8 over the 320-MB code mirror (partly served by the "
                                  "Infinity Cache from one pass to the next, so the rate can exceed what HBM alone delivers)")
    ix.close()
    del ix
    torch.cuda.empty_cache()

    # -- IVF (IVFIndex::from_vec_set, ivf_index.rs:66-118: sqrt(N) clusters, k-means on 10000 sampled rows, 10
    #    iterations; knn with the default 4 probes)
    ix = vdb.GpuIndex(dim, "l2sqr", device=local_rank)
    ix.add_device(base.data_ptr(), n)
    t_b = time.perf_counter()
    ix.ivf_build(int(round(n ** 0.5)), train_n=10000, max_iter=10, tol=1e-6, seed=42)
    ivf_build_s = time.perf_counter() - t_b
    legs["ivf"] = run(ix, "ivf", n, 4,
                      lambda: ix.ivf_knn_device(queries.data_ptr(), nq, k, 4, o_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr()),
                      {"workload": "ivf_knn_gistlike_1000000", "clusters": int(round(n ** 0.5)), "n_probes": 4, "build_s": round(ivf_build_s, 1)})
    # the reference's default is 4 probes (ivf_index.rs:108); recall there is low on any data, so the same index at 16 and 64 probes too
    # (rate + recall@10 against the GPU Flat ground truth; parity against the oracle is the 4-probe line's)
    ix.flat_knn_device(queries.data_ptr(), nq, k, t_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr())
    gt = t_idx.cpu().numpy().astype(np.uint64)
    legs["ivf"]["more_probes"] = []
    for npb in (16, 64):
        fn_p = lambda: ix.ivf_knn_device(queries.data_ptr(), nq, k, npb, o_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr())  # noqa: E731
        el, _ = timed(ix, fn_p, args.steps, 2)
        gi = o_idx.cpu().numpy().astype(np.uint64)
        rec = float(np.mean([len(set(gt[q].tolist()) & set(gi[q].tolist())) / k for q in range(nq)]))
        legs["ivf"]["more_probes"].append({"n_probes": npb, "value": round(nq * args.steps / el, 1), "unit": "queries/s",
                                           "ms_per_step": round(el / args.steps * 1e3, 3), "recall_at_10": round(rec, 4)})
    ix.close()
    del ix
    torch.cuda.empty_cache()

    # -- HNSW: config/bench_hnsw.toml:12-14 (M 16, ef_construction 200), search ef = 128
    hr = max(1000, min(args.hnsw_rows, n))
    ix = vdb.GpuIndex(dim, "l2sqr", device=local_rank)
    ix.add_device(base.data_ptr(), hr)
    t_b = time.perf_counter()
    ix.hnsw_build(M=16, ef_construction=200, seed=42, batch=args.hnsw_batch, nthreads=threads)
    hb_s = time.perf_counter() - t_b
    hq = max(1000, args.hnsw_queries)
    # (a query file shorter than the step is repeated cyclically: the rate of a step does not depend on the queries being distinct)
    queries_h = load_rows_file(torch, args.query_file, dim, 0, hq, device, cycle=True) if from_file else gist_lowrank_gpu(torch, hq, dim, 1807, device)
    h_idx = torch.zeros((hq, k), dtype=torch.int64, device=device)
    h_dist = torch.zeros((hq, k), dtype=torch.float32, device=device)
    h_cnt = torch.zeros((hq,), dtype=torch.int64, device=device)
    ht_idx = torch.zeros((hq, k), dtype=torch.int64, device=device)
    hkw = {"nq": hq, "queries": queries_h, "outs": (h_idx, h_dist, h_cnt), "t_idx": ht_idx}
    legs["hnsw"] = run(ix, "hnsw", hr, 128,
                       lambda: ix.hnsw_knn_device(queries_h.data_ptr(), hq, k, 128, h_idx.data_ptr(), h_dist.data_ptr(), h_cnt.data_ptr()),
                       {"workload": f"hnsw_knn_gistlike_{hr}", "M": 16, "ef_construction": 200, "build_s": round(hb_s, 1),
                        "build_batch": args.hnsw_batch,
                        "build_note": "HNSWIndex::add_parallel batches; candidate phase of a batch on the GPU (k_hnsw_search, k = ef = "
                                      "ef_construction, over the device mirror of the graph), linking on host threads; the graph equals the all-host builder's"},
                       **hkw)
    legs["hnsw"]["one_call_of_1000"] = small_call(
        ix, lambda: ix.hnsw_knn_device(queries_h.data_ptr(), 1000, k, 128, h_idx.data_ptr(), h_dist.data_ptr(), h_cnt.data_ptr()))
    legs["hnsw"]["one_call_of_1"] = small_call(
        ix, lambda: ix.hnsw_knn_device(queries_h.data_ptr(), 1, k, 128, h_idx.data_ptr(), h_dist.data_ptr(), h_cnt.data_ptr()), 1)
    latency_floor(legs["hnsw"], legs["hnsw"]["one_call_of_1000"], 3, "link row -> visited words -> first row lines", dim)
    # -- HNSW + PQ (hnsw_index.rs:672-697; config/bench_pq_hnsw.toml: the reference's fastest published point): the same graph,
    #    the PQ leg's centroids, codes encoded on the GPU; ADC walk + cached-form re-sort
    ix.pq_attach(4, dim // 3, cent, None)
    legs["hnsw_pq"] = run(ix, "hnsw_pq", hr, 128,
                          lambda: ix.hnsw_knn_device(queries_h.data_ptr(), hq, k, 128, h_idx.data_ptr(), h_dist.data_ptr(), h_cnt.data_ptr(),
                                                     use_pq=True),
                          {"workload": f"hnsw_pq_knn_gistlike_{hr}", "M": 16, "ef_construction": 200, "n_bits": 4, "m": dim // 3}, **hkw)
    legs["hnsw_pq"]["one_call_of_1000"] = small_call(
        ix, lambda: ix.hnsw_knn_device(queries_h.data_ptr(), 1000, k, 128, h_idx.data_ptr(), h_dist.data_ptr(), h_cnt.data_ptr(), use_pq=True))
    legs["hnsw_pq"]["one_call_of_1"] = small_call(
        ix, lambda: ix.hnsw_knn_device(queries_h.data_ptr(), 1, k, 128, h_idx.data_ptr(), h_dist.data_ptr(), h_cnt.data_ptr(), use_pq=True), 1)
    latency_floor(legs["hnsw_pq"], legs["hnsw_pq"]["one_call_of_1000"], 3, "link row -> visited words -> code rows", dim // 3)
    ix.close()
    return legs


if __name__ == "__main__":
    main()
