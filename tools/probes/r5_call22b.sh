#!/bin/bash
# round 5: the GPU suite twice more (flakiness check), exactly as the driver runs it
O=$GRAFT_REPO_ROOT/gpurun_out/r5c28; mkdir -p $O
for i in 1 2; do timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/tests_gpu_$i.log 2>&1; tail -3 $O/tests_gpu_$i.log; done
