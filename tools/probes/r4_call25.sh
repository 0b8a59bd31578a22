O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c25}; mkdir -p $O
for i in 1 2; do
FABIND_BENCH_ONLY=pocket,model_fwdbwd,fwd python bench.py --no-cpu-baseline > $O/bench_subs_$i.json 2> $O/bench_subs.err
python - <<PY
import json
d = json.load(open("$O/bench_subs_$i.json"))
print("headline", round(d["value"], 1), d["roofline"]["bound"], round(d["roofline"]["frac"], 3), d["roofline"]["kernel"][:40])
for k, v in d.items():
    if isinstance(v, dict) and "value" in v: print("  ", k, round(v["value"], 1), v.get("roofline", {}).get("kernel", "")[:50], v.get("roofline", {}).get("bound"), round(v.get("roofline", {}).get("frac", 0), 3))
PY
done
