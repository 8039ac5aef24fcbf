# A/B of the 8-bit pass's exact stage (k_flat_tail_lb) under rocprofv3: run on the GPU box as `bash tools/ab_tail.sh [nw...]`
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # tag, bench args...
  tag=$1; shift
  rocprofv3 --kernel-trace -d /tmp/prof_$tag -o p -- python3 bench.py --legs none --pipeline 1 --cpu-queries 0 "$@" > gpurun_out/ab_$tag.log 2>&1 || { tail -5 gpurun_out/ab_$tag.log; exit 2; }
  echo "$tag: step $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_$tag.log | head -1 | cut -d' ' -f2) ms; tail kernel avg $(python3 tools/kstats.py /tmp/prof_$tag tail_lb | tail -1 | awk -F, '{printf "%.1f us (min %.1f)", $(NF-2)/1e3, $(NF-1)/1e3}')"
}
for nw in ${@:-4}; do
  run s125_nw$nw --rows 125000 --steps 30 --param flat_tail_lb_nw=$nw
  run m1_nw$nw --steps 20 --param flat_tail_lb_nw=$nw
  run m1_nq32_nw$nw --steps 50 --nq 32 --param flat_tail_lb_nw=$nw
  run m1_nq1_nw$nw --steps 50 --nq 1 --param flat_tail_lb_nw=$nw
done
