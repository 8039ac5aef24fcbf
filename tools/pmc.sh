#!/bin/bash
# usage: tools/pmc.sh <tag> <python script + args...>   -- runs several separate --pmc passes (counters only, no tracing mix)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/pmc_$tag
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$tag/p$i -- python "$@" > gpurun_out/pmc_$tag/p$i.log 2>&1 || echo "pass $i failed: $set"
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
        line = f"{k}: " + ", ".join(f"{c}={sum(v)/len(v):.4g} (n={len(v)})" for c, v in sorted(cs.items()))
        print(line); out.write(line + "\n")
PY
