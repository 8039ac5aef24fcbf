"""Replays the RNG stream of tools/fuzz_flat.py on the CPU (no GPU calls), checks every drawn configuration against the lines of a
soak log and writes the generator state in front of a chosen configuration -- how configuration #87 of
gpurun_out/fuzz_fs.log (round 2: the run that ended in a GPU memory-access fault) was recovered.
usage: python tools/replay_fuzz_flat.py LOG SEED STOP_AT OUT.json"""
import json, re, sys
import numpy as np

log, seed, stop_at, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
lines = {}
for ln in open(log):
    m = re.match(r"#(\d+) dim (\d+) n (\d+) nq (\d+) k (\d+) (\w+) style (\d+):", ln)
    if m:
        lines[int(m.group(1))] = (int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), m.group(6), int(m.group(7)))
rng = np.random.default_rng(seed)
it = 0
while True:
    it += 1
    if it == stop_at:
        json.dump({"seed": seed, "configuration": it, "state": rng.bit_generator.state}, open(out, "w"))
    dim = int(rng.choice([64, 96, 100, 128, 192, 256, 320, 384, 512, 768, 960, 1000, 1024, 1536]))
    n = int(rng.integers(17000, 60000))
    nq = int(rng.choice([1, 3, 17, 64, 65, 100, 128, 129, 200, 257]))
    k = int(rng.choice([1, 2, 5, 10, 16, 17, 33, 64, 70]))
    dist = str(rng.choice(["l2sqr", "cosine"]))
    style = int(rng.integers(0, 4))
    if style == 0:
        rng.standard_normal((n, dim)); rng.standard_normal((nq, dim))
    elif style == 1:
        rng.normal(0.07, 0.045, (n, dim)); rng.normal(0.07, 0.045, (nq, dim))
    elif style == 2:
        c = rng.standard_normal((n // 50 + 1, dim)); rng.standard_normal((n, dim)); rng.integers(0, len(c), nq); rng.standard_normal((nq, dim))
    else:
        rng.standard_normal((n, dim)); rng.normal(0, 1.0, (n, 1)); rng.standard_normal((nq, dim))
    two = not (rng.random() < 0.5)
    cut = int(rng.integers(1, n - 1)) if two else 0
    mode = int(rng.choice([0, 2])); half = int(rng.choice([0, 0, 1, 2])); tail = int(rng.choice([0, 0, 1]))
    cfg = (dim, n, nq, k, dist, style)
    if it in lines:
        assert lines[it] == cfg, (it, lines[it], cfg)
    print(f"#{it} dim {dim} n {n} nq {nq} k {k} {dist} style {style} two_part {two} cut {cut} flat_mode {mode} flat_half {half} flat_tail {tail}"
          + ("" if it in lines else "   <- not in the log"), flush=True)
    if it >= stop_at:
        break
