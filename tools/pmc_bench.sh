#!/bin/bash
# usage: tools/pmc_bench.sh <tag> [bench.py args...] -- HBM traffic counters of bench.py's kernels, one --pmc pass per set
# (counters only, no tracing mix), then a per-kernel average per launch.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/pmc_$tag
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$tag/p$i -- python bench.py --cpu-queries 0 "$@" > gpurun_out/pmc_$tag/p$i.log 2>&1 || echo "pass $i failed: $set"
done
python - <<'PY' $tag
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'gpurun_out/pmc_{tag}/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(f'gpurun_out/pmc_{tag}/summary.txt', 'w') as out:
    for k, cs in agg.items():
        if 'vdb' not in k: continue
        line = f"{k}: " + ", ".join(f"{c}={sum(v)/len(v):.6g} (n={len(v)})" for c, v in sorted(cs.items()))
        print(line); out.write(line + "\n")
PY
