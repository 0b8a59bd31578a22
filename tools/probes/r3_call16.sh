# kept parameter-pack requests: tests, same-box A/B at the headline and the pocket shape; kernel windows around the pack's launches
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3q}; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_stack.py tests/test_gpu_model.py tests/test_gpu_kernels.py -q -m gpu -x -k "grad or pack or retained or loss" 2>&1 | tail -4 > $O/tests.txt
one() { python bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for i in 1 2 3; do
echo "plan     $(one)" >> $O/ab.txt
echo "rebuild  $(FABIND_PACK_PLAN=0 one)" >> $O/ab.txt
done
for i in 1 2; do
echo "pocket plan    $(one --n-prot 100 --steps 20 --warmup 5)" >> $O/ab.txt
echo "pocket rebuild $(FABIND_PACK_PLAN=0 one --n-prot 100 --steps 20 --warmup 5)" >> $O/ab.txt
done
echo "model plan    $(one --mode model_fwdbwd 2>/dev/null)" >> $O/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof -o fb -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $O/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_window.py $O/prof/fb_results.db multi_copy 14 10 4 > $O/windows.txt 2>&1
python tools/prof_gaps.py $O/prof/fb_results.db 10 edges_kernelILb1 2 > $O/gaps.txt 2>&1
rm -rf $O/prof
cat $O/tests.txt $O/ab.txt; head -12 $O/gaps.txt; cat $O/windows.txt
