#!/bin/bash
# round 5: bf16x3 with the exact backward through the unfused edge pipeline: gradient test at the headline shape + the step
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c29; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_headline.py -x -q -s -k "gradients_match" > $O/tests.log 2>&1; grep -n "headline shape\|whole-gradient\|passed\|failed\|Error" $O/tests.log | head -30
FABIND_X3_WGRAD=x3 python bench.py --precision bf16x3 --no-cpu-baseline --no-extras --steps 4 --warmup 2 2>$O/err.log | python -c "import json,sys; d=json.load(sys.stdin); print('gate_mode exact', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/speed.txt
tail -3 $O/err.log
