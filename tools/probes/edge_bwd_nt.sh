# Fused edge backward: plain vs non-temporal operand-tile stores (knob bit 32) -- time per launch (stand-alone, interleaved), HBM
# traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes), headline A/B.  Output under gpurun_out/$1.
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3nt}; mkdir -p $O
cd $GRAFT_REPO_ROOT
for r in 1 2; do for m in 0 32; do
  FABIND_EDGE_BWD3_EXP=$m python tools/probes/edge_bwd_time.py 10 2>/dev/null | grep backward | sed "s/^/NT round $r mask $m: /"
done; done | tee $O/edge_bwd_nt_time.txt
cd /tmp && export TMPDIR=/tmp
for m in 0 32; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$m -o f -- python3 $GRAFT_REPO_ROOT/tools/probes/edge_bwd_one.py 5 $m > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w$m -o w -- python3 $GRAFT_REPO_ROOT/tools/probes/edge_bwd_one.py 5 $m > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/t$m -o t -- python3 $GRAFT_REPO_ROOT/tools/probes/edge_bwd_one.py 5 $m > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/probes/pmc_table.py "fused edge backward (store wave), operand-store mask $m, per launch, E = 1,539,196, H = 512; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md)" gcl_edge_fused_bwd3_kernel $(find $O/f$m $O/w$m $O/t$m -name "*counter_collection.csv") >> $O/edge_bwd_nt_pmc.txt 2>&1
  rm -rf $O/f$m $O/w$m $O/t$m
  cd /tmp
done
cd $GRAFT_REPO_ROOT
cat $O/edge_bwd_nt_pmc.txt
for r in 1 2 3; do for m in 0 32; do
  FABIND_EDGE_BWD3_EXP=$m python bench.py --no-cpu-baseline --no-extras --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('HEADLINE round $r mask $m', round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['avg_us'],1))"
done; done | tee $O/edge_bwd_nt_headline.txt
