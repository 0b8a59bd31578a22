# FABind+ parameter pack as one launch (kept plan): FABind+ tests incl. gradients vs goldens / oracle, same-box A/B of plus_train
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3x}; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_plus.py tests/test_gpu_production.py tests/test_gpu_plus_dense_api.py tests/test_gpu_param_pack.py -q -m gpu -x 2>&1 | tail -3 > $O/tests.txt
one() { python bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for i in 1 2 3; do
echo "pack   $(one --mode plus_train --steps 4 --warmup 2)" >> $O/ab.txt
echo "eager  $(FABIND_PLUS_PARAM_PACK=0 one --mode plus_train --steps 4 --warmup 2)" >> $O/ab.txt
done
cat $O/tests.txt $O/ab.txt
