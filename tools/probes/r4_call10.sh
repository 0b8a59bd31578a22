O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c10}; mkdir -p $O
show() { python -c "
import json,sys
d=json.load(sys.stdin)
print('$1', 'headline', round(d['value'],1), {k: (round(v['value'],1), round(v['ms_per_step'],1)) for k,v in d.items() if isinstance(v,dict) and 'value' in v and k not in ('roofline','cpu_baseline')})"; }
FABIND_BENCH_ONLY=model_fwdbwd python bench.py --no-cpu-baseline --steps 5 2>/dev/null | show only_model | tee -a $O/ab.txt
FABIND_BENCH_ONLY=pocket,model_fwdbwd python bench.py --no-cpu-baseline --steps 5 2>/dev/null | show pocket_model | tee -a $O/ab.txt
FABIND_BENCH_ONLY=fp32,model_fwdbwd python bench.py --no-cpu-baseline --steps 5 2>/dev/null | show fp32_model | tee -a $O/ab.txt
FABIND_BENCH_ONLY=gate_mode,model_fwdbwd python bench.py --no-cpu-baseline --steps 5 2>/dev/null | show gate_model | tee -a $O/ab.txt
FABIND_BENCH_ONLY=n_iter8,fwd,model_fwdbwd python bench.py --no-cpu-baseline --steps 5 2>/dev/null | show iter8_fwd_model | tee -a $O/ab.txt
python bench.py --mode model --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | show standalone | tee -a $O/ab.txt
