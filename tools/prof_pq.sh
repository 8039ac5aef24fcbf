# kernel profile of one workload's bench step: bash tools/prof_pq.sh <workload> [bench args]  -> gpurun_out/kstats_<workload>.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
wl=${1:-pq_flat}; shift
rocprofv3 --kernel-trace -d /tmp/prof_$wl -o p -- python3 bench.py --workload $wl --legs none --cpu-queries 0 --steps 10 "$@" > gpurun_out/prof_$wl.log 2>&1 || { tail -5 gpurun_out/prof_$wl.log; exit 2; }
grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_$wl.log | head -1
python3 tools/kstats.py /tmp/prof_$wl > gpurun_out/kstats_$wl.csv
