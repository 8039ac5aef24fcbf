#!/bin/bash
# usage (GPU box): bash tools/ab_hnsw.sh "<EXTRA defines>" [rows] -- same-box A/B of a compile-time switch of hnsw.hip on small HNSW calls
# (tools/probe_hnsw_small.py, exact walk): as shipped, rebuilt under EXTRA, as shipped again -> gpurun_out/ab_hnsw.txt
cd $GRAFT_REPO_ROOT
export HNSW_PROBE_QUICK=1
run() { echo "== $1" >> gpurun_out/ab_hnsw.txt; python3 tools/probe_hnsw_small.py ${ROWS} 2>/dev/null | grep -v build >> gpurun_out/ab_hnsw.txt; }
ROWS=${2:-1000000}
: > gpurun_out/ab_hnsw.txt
run shipped
cp lab_1806_vec_db_amd/libvdbhip.so /tmp/libvdbhip.shipped.so; cp lab_1806_vec_db_amd/csrc/hnsw.o /tmp/hnsw.shipped.o
touch lab_1806_vec_db_amd/csrc/hnsw.hip; make -C lab_1806_vec_db_amd/csrc -s EXTRA="$1" > /tmp/ab_make.log 2>&1 || { tail -5 /tmp/ab_make.log; exit 2; }
run "variant($1)"
cp /tmp/hnsw.shipped.o lab_1806_vec_db_amd/csrc/hnsw.o; cp /tmp/libvdbhip.shipped.so lab_1806_vec_db_amd/libvdbhip.so
run shipped
cat gpurun_out/ab_hnsw.txt
