#!/bin/bash
# usage (GPU box): bash tools/ab_step.sh <rows> <tag>=<bench args, comma separated> ...
#   one `bench.py --legs none --pipeline 1 --cpu-queries 0 --rows <rows>` per variant under rocprofv3 --kernel-trace: the step time of the
#   compact line and the average duration of every library kernel, side by side in gpurun_out/ab_step_<rows>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rows=$1; shift
out=gpurun_out/ab_step_$rows.txt
: > $out
for v in "$@"; do
  tag=${v%%=*}; args=${v#*=}; args=${args//,/ }
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace -d /tmp/prof_$tag -o p -- python3 bench.py --legs none --pipeline 1 --cpu-queries 0 --rows $rows --steps 30 --warmup 5 $args > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || { tail -5 gpurun_out/ab_$tag.err; exit 2; }
  echo "== $tag ($args)" >> $out
  python3 -c "import json,sys; r=json.loads(open('gpurun_out/ab_$tag.json').read().strip().splitlines()[-1]); print('step_ms', r['ms_per_step'], 'qps', r['value'], 'filter_ms', r['roofline']['avg_launch_ms'], 'passed_on', r.get('i8_pass'))" >> $out
  python3 tools/kstats.py /tmp/prof_$tag | grep -v "k_probe\|k_tile_rows\|k_col_\|k_row_sqnorm\|^#\|\"Name\"" | awk -F'",' '{split($1,a,"("); n=a[1]; gsub(/"/,"",n); split($2,b,","); printf "   %-60s calls %4d avg_us %8.1f\n", substr(n,1,60), b[1], b[3]/1000}' >> $out
done
cat $out
