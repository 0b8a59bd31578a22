O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c24}; mkdir -p $O
for a in "bf16x3" "bf16" "bf16 grad" "bf16x3 grad"; do timeout 600 python tools/probes/poison_alloc.py $a 2>&1 | tail -6 | tee -a $O/poison.txt; done
