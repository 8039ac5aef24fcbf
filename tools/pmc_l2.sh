#!/bin/bash
# usage: tools/pmc_l2.sh <tag> <python args...>  -- L2 hit/miss + fetch size only (2 passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/pmc_$tag
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$tag/p$i -- python "$@" > gpurun_out/pmc_$tag/p$i.log 2>&1 || echo "pass $i failed"
done
python - <<'PY' $tag
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'gpurun_out/pmc_{tag}/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    if 'k_flat_mfma<5, 1>' not in k: continue
    # launches alternate per the probe order; print every launch
    n = len(next(iter(cs.values())))
    for j in range(n):
        print(k, j, {c: v[j] for c, v in cs.items() if j < len(v)})
PY
