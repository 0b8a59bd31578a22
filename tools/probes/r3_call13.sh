# pair-embedding gradient sink + 64-padded narrow contractions: FABind+ tests, plus_train A/B (sink on / off), whole kernel test file
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3m}; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -3 > $O/tests_kernels.txt
python -m pytest tests/test_gpu_plus.py tests/test_gpu_production.py tests/test_gpu_plus_dense_api.py -q -m gpu -k "plus" 2>&1 | tail -8 > $O/tests_plus.txt
for i in 1 2; do
python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sink', d['value'], d['ms_per_step'])" >> $O/ab.txt
FABIND_PLUS_Z_SINK=0 python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 4 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nosink', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
cat $O/tests_kernels.txt $O/tests_plus.txt $O/ab.txt
