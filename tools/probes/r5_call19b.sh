#!/bin/bash
# round 5: torch-op census of the FABind+ training step and of the full IaBNet step (FABIND_BENCH_ATEN)
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/r5c25; mkdir -p $O
FABIND_BENCH_ATEN=$O/aten_plus_train.txt python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 2 --warmup 2 > /dev/null 2> $O/err1.log
FABIND_BENCH_ATEN=$O/aten_model.txt python bench.py --mode model --no-cpu-baseline --no-extras --steps 2 --warmup 2 > /dev/null 2> $O/err2.log
FABIND_BENCH_ATEN=$O/aten_headline.txt python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 2 > /dev/null 2> $O/err3.log
head -45 $O/aten_plus_train.txt | cut -c1-250
