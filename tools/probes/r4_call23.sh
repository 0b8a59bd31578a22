O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c23}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > $O/tests_gemm.log 2>&1; tail -4 $O/tests_gemm.log
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_headline.py -x -q -k "test_bench_batch_properties and x3" -s 2>&1 | grep "run-to-run\|passed\|failed" | cut -c1-200; done | tee $O/x3_repeat.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline big-tile GEMM rounds', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
FABIND_GEMM_BIG=0 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline 256x128 tiles only    ', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
done
FABIND_BENCH_DUMP_PROFILE=$O/launch_groups.txt python bench.py --no-cpu-baseline --no-extras > /dev/null 2>&1; grep "fabind_gemm <" $O/launch_groups.txt | head -12 | cut -c1-150
