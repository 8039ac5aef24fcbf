#!/bin/bash
# usage (GPU box): bash tools/ab_hnsw_r03.sh -- same-box A/B of the HNSW walk at bench size (8192 queries per step, 1M rows): the library as
# shipped, then with round 3's hnsw.hip (a copy you place at tools/_hnsw_r03.hip.txt = git show <round-3 head>:lab_1806_vec_db_amd/csrc/hnsw.hip) -> gpurun_out/ab_hnsw_r03.txt
cd $GRAFT_REPO_ROOT
run() { echo "== $1" >> gpurun_out/ab_hnsw_r03.txt; python3 bench.py --workload hnsw --nq 8192 --data lowrank --steps 10 --warmup 3 --cpu-queries 32 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['parity'], r.get('recall_at_10'))" >> gpurun_out/ab_hnsw_r03.txt; }
: > gpurun_out/ab_hnsw_r03.txt
run shipped
cp lab_1806_vec_db_amd/csrc/hnsw.hip /tmp/hnsw.shipped.hip; cp lab_1806_vec_db_amd/libvdbhip.so /tmp/libvdbhip.shipped.so; cp lab_1806_vec_db_amd/csrc/hnsw.o /tmp/hnsw.shipped.o
cp tools/_hnsw_r03.hip.txt lab_1806_vec_db_amd/csrc/hnsw.hip; make -C lab_1806_vec_db_amd/csrc -s > /tmp/ab_make.log 2>&1 || { tail -5 /tmp/ab_make.log; exit 2; }
run "round 3 hnsw.hip"
cp /tmp/hnsw.shipped.hip lab_1806_vec_db_amd/csrc/hnsw.hip; cp /tmp/hnsw.shipped.o lab_1806_vec_db_amd/csrc/hnsw.o; cp /tmp/libvdbhip.shipped.so lab_1806_vec_db_amd/libvdbhip.so
cat gpurun_out/ab_hnsw_r03.txt
