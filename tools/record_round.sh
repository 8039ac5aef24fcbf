# the round's record on one box: GPU suite, ONE default bench run, its rocprofv3 kernel-stats twin, shard lines, small-call lines
# usage (GPU box): bash tools/record_round.sh <tag>     -> gpurun_out/<tag>_*
tag=${1:-rec}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests -x -q -m gpu > gpurun_out/${tag}_suite.log 2>&1 || { tail -20 gpurun_out/${tag}_suite.log; exit 1; }
tail -1 gpurun_out/${tag}_suite.log
python3 bench.py --steps 20 --warmup 5 --full-out gpurun_out/${tag}_bench_all_full.json > gpurun_out/${tag}_bench_all.json 2> gpurun_out/${tag}_bench_all.err || { tail -5 gpurun_out/${tag}_bench_all.err; exit 2; }
echo bench done
rocprofv3 --kernel-trace --stats -d /tmp/prof_all -o p -- python3 bench.py --steps 20 --warmup 5 --full-out gpurun_out/${tag}_bench_all_under_rocprof_full.json > gpurun_out/${tag}_bench_all_under_rocprof.json 2> gpurun_out/${tag}_rocprof.err || { tail -5 gpurun_out/${tag}_rocprof.err; exit 3; }
python3 tools/kstats.py /tmp/prof_all "" > gpurun_out/${tag}_bench_all_kernel_stats.csv
echo rocprof done
# the headline alone (no legs): the dominant kernel's average launch duration is readable straight from this summary
rocprofv3 --kernel-trace --stats -d /tmp/prof_head -o p -- python3 bench.py --legs none --steps 20 --warmup 5 --cpu-queries 0 --full-out gpurun_out/${tag}_bench_headline_under_rocprof_full.json > gpurun_out/${tag}_bench_headline_under_rocprof.json 2>> gpurun_out/${tag}_err.log || exit 8
python3 tools/kstats.py /tmp/prof_head "" > gpurun_out/${tag}_bench_headline_kernel_stats.csv
echo headline rocprof done
for r in 125000 250000 500000; do for p in 1 3; do
  python3 bench.py --rows $r --legs none --pipeline $p --steps 30 > gpurun_out/${tag}_bench_flat_rows${r}_pipeline$p.json 2>> gpurun_out/${tag}_err.log || exit 4
done; done
echo shards done
rocprofv3 --kernel-trace --stats -d /tmp/prof_125 -o p -- python3 bench.py --rows 125000 --legs none --pipeline 1 --steps 30 --cpu-queries 0 > /dev/null 2>> gpurun_out/${tag}_err.log || exit 5
python3 tools/kstats.py /tmp/prof_125 > gpurun_out/${tag}_bench_flat_rows125000_kernel_stats.csv
python3 bench.py --nq 32 --legs none --steps 50 > gpurun_out/${tag}_bench_flat_nq32.json 2>> gpurun_out/${tag}_err.log || exit 6
python3 bench.py --nq 1 --legs none --steps 50 > gpurun_out/${tag}_bench_flat_nq1.json 2>> gpurun_out/${tag}_err.log || exit 7
echo done
