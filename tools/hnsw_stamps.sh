#!/bin/bash
# usage (GPU box): bash tools/hnsw_stamps.sh [rows] -- rebuilds hnsw.o with -DHNSW_STAMP, runs tools/probe_hnsw_stamps.py (stdout + the
# kernel's stamp lines interleaved into gpurun_out/hnsw_stamps.txt), restores the shipped library
cd $GRAFT_REPO_ROOT
cp lab_1806_vec_db_amd/libvdbhip.so /tmp/libvdbhip.shipped.so; cp lab_1806_vec_db_amd/csrc/hnsw.o /tmp/hnsw.shipped.o
touch lab_1806_vec_db_amd/csrc/hnsw.hip; make -C lab_1806_vec_db_amd/csrc -s EXTRA="-DHNSW_STAMP $HNSW_EXTRA" > /tmp/st_make.log 2>&1 || { tail -5 /tmp/st_make.log; exit 2; }
python3 tools/probe_hnsw_stamps.py ${1:-1000000} > gpurun_out/hnsw_stamps.txt 2>&1
cp /tmp/hnsw.shipped.o lab_1806_vec_db_amd/csrc/hnsw.o; cp /tmp/libvdbhip.shipped.so lab_1806_vec_db_amd/libvdbhip.so
grep -v "amdgpu.ids\|distance evaluations" gpurun_out/hnsw_stamps.txt
