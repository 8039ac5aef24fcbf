#!/bin/bash
# usage (GPU box): bash tools/hnsw_ablate.sh [rows] -- where does the exact walk's fold spend its time?  hnsw.o rebuilt with -DHNSW_STAMP -DHNSW_STAMP2
# and -DHNSW_ABL=0 / 1 / 2 / 4 / 7 (parts of hnsw_exact_dists_regs's inner loop left out: the distances are WRONG, only the stamps of the
# fold are read), one-query and 1000-query calls -> gpurun_out/hnsw_ablate.txt; restores the shipped library
cd $GRAFT_REPO_ROOT
cp lab_1806_vec_db_amd/libvdbhip.so /tmp/libvdbhip.shipped.so; cp lab_1806_vec_db_amd/csrc/hnsw.o /tmp/hnsw.shipped.o
: > gpurun_out/hnsw_ablate.txt
for abl in 0 1 2 4 7; do
  touch lab_1806_vec_db_amd/csrc/hnsw.hip; make -C lab_1806_vec_db_amd/csrc -s EXTRA="-DHNSW_STAMP -DHNSW_STAMP2 -DHNSW_ABL=$abl" > /tmp/st_make.log 2>&1 || { tail -5 /tmp/st_make.log; exit 2; }
  echo "== HNSW_ABL=$abl" >> gpurun_out/hnsw_ablate.txt
  HNSW_STAMPS_EXACT_ONLY=1 python3 tools/probe_hnsw_stamps.py ${1:-300000} 2>&1 | grep "hnsw_exact_dists_regs\|== \|per expansion)" >> gpurun_out/hnsw_ablate.txt
done
cp /tmp/hnsw.shipped.o lab_1806_vec_db_amd/csrc/hnsw.o; cp /tmp/libvdbhip.shipped.so lab_1806_vec_db_amd/libvdbhip.so
cat gpurun_out/hnsw_ablate.txt
