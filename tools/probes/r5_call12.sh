#!/bin/bash
# round 5, call 12: full GPU suite + smoke on the final tree, the evidence run (tools/probes/r5_final.sh), the driver's default bench line
O=$GRAFT_REPO_ROOT/gpurun_out/r5fin2; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -4 $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/probes/r5_final.sh r5fin2 > $O/final.log 2>&1; tail -12 $O/final.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
