#!/bin/bash
# usage (GPU box): bash tools/ab_build.sh <file.hip> <EXTRA define> [bench args...] -- same-box A/B of a compile-time switch: the headline
# (bench.py --legs none --cpu-queries 0) with the library as shipped, with <file>.o rebuilt under EXTRA, and as shipped again
cd $GRAFT_REPO_ROOT
f=$1; extra=$2; shift 2
run() { python3 bench.py --legs none --cpu-queries 0 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', 'step_ms', r['ms_per_step'], 'filter_ms', r['roofline']['avg_launch_ms'], 'qps', r['value'])"; }
tag=shipped; run "$@"
cp lab_1806_vec_db_amd/libvdbhip.so /tmp/libvdbhip.shipped.so; cp lab_1806_vec_db_amd/csrc/${f%.hip}.o /tmp/ab_shipped.o
touch lab_1806_vec_db_amd/csrc/$f; make -C lab_1806_vec_db_amd/csrc -s EXTRA="$extra" > /tmp/ab_make.log 2>&1 || { tail -5 /tmp/ab_make.log; exit 2; }
tag="variant($extra)"; run "$@"; run "$@"
cp /tmp/ab_shipped.o lab_1806_vec_db_amd/csrc/${f%.hip}.o; cp /tmp/libvdbhip.shipped.so lab_1806_vec_db_amd/libvdbhip.so
tag=shipped; run "$@"
