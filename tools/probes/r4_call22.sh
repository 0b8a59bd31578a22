# full GPU suite + the default bench line with the saved-forward edge path and the fused attention training path
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c22}; mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; tail -5 $O/tests_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("headline", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2))
print("roofline", {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_us")}, d["roofline"].get("other_roofline"))
print("cpu_baseline", {k: v for k, v in d["cpu_baseline"].items() if k != "protocol_8d"})
for k, v in d.items():
    if isinstance(v, dict) and "value" in v: print(k, round(v["value"], 1))
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
