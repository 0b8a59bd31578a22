O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c21}; mkdir -p $O
for m in 0 1 2 3 4 8 15; do
  echo "== FABIND_EDGE_BWD4_EXP=$m (1 dT/dP2 copy-outs, 2 S1/dP1 stores, 4 row scan, 8 contractions skipped)" | tee -a $O/fe4_sens.txt
  FABIND_EDGE_BWD4_EXP=$m python tools/probes/edge_bwd4_time.py 8 new 2>&1 | tail -1 | tee -a $O/fe4_sens.txt
done
