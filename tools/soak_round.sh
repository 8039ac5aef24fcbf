# soaks on the final library: bash tools/soak_round.sh <tag>  -> gpurun_out/<tag>_fuzz_*.txt (each fuzzer prints one line per configuration)
tag=${1:-soak}
cd $GRAFT_REPO_ROOT
python3 tools/fuzz_flat.py 150 9031 > gpurun_out/${tag}_fuzz_flat_a.txt 2>&1 || { tail -5 gpurun_out/${tag}_fuzz_flat_a.txt; exit 1; }
echo "flat a: $(grep -c ' ok ' gpurun_out/${tag}_fuzz_flat_a.txt) ok, $(grep -c MISMATCH gpurun_out/${tag}_fuzz_flat_a.txt) mismatches"
python3 tools/fuzz_flat.py 100 9032 -1 300 > gpurun_out/${tag}_fuzz_flat_b.txt 2>&1 || { tail -5 gpurun_out/${tag}_fuzz_flat_b.txt; exit 2; }
echo "flat b: $(grep -c ' ok ' gpurun_out/${tag}_fuzz_flat_b.txt) ok, $(grep -c MISMATCH gpurun_out/${tag}_fuzz_flat_b.txt) mismatches"
python3 tools/fuzz_pq.py 150 9033 > gpurun_out/${tag}_fuzz_pq.txt 2>&1 || { tail -5 gpurun_out/${tag}_fuzz_pq.txt; exit 3; }
echo "pq: $(tail -1 gpurun_out/${tag}_fuzz_pq.txt)"
python3 tools/fuzz_hnsw.py 120 9034 > gpurun_out/${tag}_fuzz_hnsw.txt 2>&1 || { tail -5 gpurun_out/${tag}_fuzz_hnsw.txt; exit 4; }
echo "hnsw: $(tail -1 gpurun_out/${tag}_fuzz_hnsw.txt)"
python3 tools/fuzz_ivf.py 80 9035 > gpurun_out/${tag}_fuzz_ivf.txt 2>&1 || { tail -5 gpurun_out/${tag}_fuzz_ivf.txt; exit 5; }
echo "ivf: $(tail -1 gpurun_out/${tag}_fuzz_ivf.txt)"
python3 tools/fuzz_mutate.py 80 9036 > gpurun_out/${tag}_fuzz_mutate.txt 2>&1 || { tail -5 gpurun_out/${tag}_fuzz_mutate.txt; exit 6; }
echo "mutate: $(tail -1 gpurun_out/${tag}_fuzz_mutate.txt)"
