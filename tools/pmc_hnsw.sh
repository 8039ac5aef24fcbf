#!/bin/bash
# usage: tools/pmc_hnsw.sh <rows> [queries per step] -- TA / L1 / LDS counters of the HNSW walk kernel (bench.py --workload hnsw), one --pmc pass per set
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rows=$1
mkdir -p gpurun_out/pmc_hnsw
rocprofv3 -L > gpurun_out/pmc_hnsw/counters.txt 2>&1
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" "FETCH_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_hnsw/p$i -- python bench.py --workload hnsw --data lowrank --rows $rows --legs none --cpu-queries 0 --nq ${2:-8192} --steps 5 --warmup 2 > gpurun_out/pmc_hnsw/p$i.log 2>&1 || echo "pass $i failed: $set"
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_hnsw/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/pmc_hnsw/summary.txt', 'w') as out:
    for k, cs in agg.items():
        if 'hnsw' not in k: continue
        line = f"{k}: " + ", ".join(f"{c}={sum(v)/len(v):.6g} (n={len(v)})" for c, v in sorted(cs.items()))
        print(line); out.write(line + "\n")
PY
rm -rf gpurun_out/pmc_hnsw/p*/
