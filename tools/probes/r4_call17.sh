# full GPU suite with the size-dispatched fused cross-attention training path, then the sub-objects that touch it
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c17}; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; tail -5 $O/tests_gpu.log
grep "headline shape bf16 gradients" -A3 $O/tests_gpu.log | cut -c1-200
FABIND_BENCH_ONLY=pocket,n_iter8,model_fwdbwd,model_fwdbwd_train_n_iter8,train_mode python bench.py --no-cpu-baseline > $O/bench_subs.json 2> $O/bench_subs.err
python - <<PY
import json
d = json.load(open("$O/bench_subs.json"))
print("headline", round(d["value"], 1), d["roofline"]["frac"])
for k, v in d.get("extras", d).items():
    if isinstance(v, dict) and "value" in v: print(k, round(v["value"], 1))
PY
