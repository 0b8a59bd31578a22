O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c20}; mkdir -p $O
echo "== default build" | tee -a $O/fe4.txt
python tools/probes/edge_bwd4_time.py 8 all 2>&1 | tail -4 | tee -a $O/fe4.txt
for i in 1 2 3 4 5; do
  echo "== variant $(grep "^$i:" fabind_amd/_ab/variants.txt)" | tee -a $O/fe4.txt
  FABIND_LIB=$GRAFT_REPO_ROOT/fabind_amd/_ab/libfabind_fe4_$i.so python tools/probes/edge_bwd4_time.py 8 new 2>&1 | tail -2 | tee -a $O/fe4.txt
done
echo "== default build, pocket-sized (100 / 40)" | tee -a $O/fe4.txt
N_PROT=100 python tools/probes/edge_bwd4_time.py 20 all 2>&1 | tail -4 | tee -a $O/fe4.txt
