# phase stamps of k_flat_tail_lb (needs .probe/libvdbhip_stamps.so = a build with -DVDB_TAIL_STAMPS, _ablate.so = + -DVDB_TAIL_ABLATE=1)
cd $GRAFT_REPO_ROOT
for lib in ${LIBS:-stamps ablate}; do
  for nw in ${@:-4 40 41 8}; do
    for a in "--rows 125000" "" "--nq 1"; do
      echo "== $lib nw=$nw $a: $(VDBHIP_LIB=$PWD/.probe/libvdbhip_$lib.so VDB_TAIL_STAMPS=12 python3 bench.py --legs none --pipeline 1 --cpu-queries 0 --steps 20 --param flat_tail_lb_nw=$nw $a 2>&1 | grep TAIL_STAMPS | cut -c12- | sed 's/queries by rounds//')"
    done
  done
done
