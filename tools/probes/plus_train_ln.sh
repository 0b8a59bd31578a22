python -m pytest tests/test_gpu_plus.py tests/test_gpu_plus_dense_api.py tests/test_gpu_kernels.py -q -m gpu -k "plus or layernorm or ln" 2>&1 | tail -2
for i in 1 2; do python bench.py --no-cpu-baseline --mode plus_train 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('plus_train new', d['value'], d['ms_per_step'])"; done
mkdir -p gpurun_out/pt; R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pt -o pt -- python3 $R/bench.py --no-cpu-baseline --mode plus_train --steps 2 --warmup 1 > /dev/null 2>&1
cd $R; python tools/prof_summary.py gpurun_out/pt/pt_results.db | grep "layernorm\|total GPU" | cut -c1-150; rm -rf gpurun_out/pt
