O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c11}; mkdir -p $O
show() { python -c "
import json,sys
d=json.load(sys.stdin)
print('$1', 'headline', round(d['value'],1), {k: (round(v['value'],1), round(v['ms_per_step'],1)) for k,v in d.items() if isinstance(v,dict) and 'value' in v and k not in ('roofline','cpu_baseline')})"; }
FABIND_BENCH_PREFETCH=0 FABIND_BENCH_ONLY=pocket,model_fwdbwd python bench.py --no-cpu-baseline --steps 5 2>/dev/null | show noprefetch_pocket_model | tee -a $O/ab.txt
FABIND_BENCH_ST_BACKWARD=0 FABIND_BENCH_ONLY=pocket,model_fwdbwd python bench.py --no-cpu-baseline --steps 5 2>/dev/null | show mt_pocket_model | tee -a $O/ab.txt
FABIND_TN_DEFER=0 FABIND_BENCH_ONLY=pocket,model_fwdbwd python bench.py --no-cpu-baseline --steps 5 2>/dev/null | show nodefer_pocket_model | tee -a $O/ab.txt
FABIND_BENCH_ONLY=pocket,model_fwdbwd,model_gate,config3_whole_graph python bench.py --no-cpu-baseline --steps 5 2>/dev/null | show pocket_model_more | tee -a $O/ab.txt
