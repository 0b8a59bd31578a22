# feeder-stream prefetch of the next batch's layout + input graph: tests that touch the stack path, headline / fwd / x3 A/B on one box
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3p}; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_stack.py tests/test_gpu_bench.py tests/test_gpu_headline.py -q -m gpu -x 2>&1 | tail -4 > $O/tests.txt
one() { python bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for i in 1 2 3; do
echo "prefetch  $(one)" >> $O/ab.txt
echo "serial    $(FABIND_BENCH_PREFETCH=0 one)" >> $O/ab.txt
done
echo "fwd prefetch $(one --mode fwd)" >> $O/ab.txt
echo "fwd serial   $(FABIND_BENCH_PREFETCH=0 one --mode fwd)" >> $O/ab.txt
echo "x3 prefetch  $(one --precision bf16x3)" >> $O/ab.txt
echo "x3 serial    $(FABIND_BENCH_PREFETCH=0 one --precision bf16x3)" >> $O/ab.txt
echo "pocket prefetch $(one --n-prot 100 --steps 20 --warmup 5)" >> $O/ab.txt
echo "pocket serial   $(FABIND_BENCH_PREFETCH=0 one --n-prot 100 --steps 20 --warmup 5)" >> $O/ab.txt
cat $O/tests.txt $O/ab.txt
