# last check of the committed tree: full GPU suite + smoke
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c32}; mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline', round(d['value'],1), d['roofline']['bound'], round(d['roofline']['frac'],3))"
