# one-node relu MLP + row-dot dropout node: kernel tests, FABind+ parity tests, plus_train bench (same box: new, then the knobs off)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3k}; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "relu_mlp or rowdot or dropout or gemm" 2>&1 | tail -6 > $O/tests_kernels.txt
python -m pytest tests/test_gpu_plus.py tests/test_gpu_production.py -q -m gpu -k "plus" 2>&1 | tail -8 > $O/tests_plus.txt
for i in 1 2; do
python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])" >> $O/ab.txt
FABIND_PLUS_MLP2_NODE=0 FABIND_PLUS_ROWDOT_DROP=0 python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
cat $O/tests_kernels.txt $O/tests_plus.txt $O/ab.txt
