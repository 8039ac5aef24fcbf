#!/bin/bash
# usage: tools/prof_stats.sh <tag> [bench.py args...] -- rocprofv3 --kernel-trace --stats of one bench.py command;
# the per-kernel summary lands in gpurun_out/prof_<tag>/kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag/raw -- python bench.py "$@" > gpurun_out/prof_$tag/bench.json 2> gpurun_out/prof_$tag/bench.err || echo "rocprofv3 failed"
f=$(find gpurun_out/prof_$tag/raw -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/prof_$tag/kernel_stats.csv
rm -rf gpurun_out/prof_$tag/raw
head -25 gpurun_out/prof_$tag/kernel_stats.csv
