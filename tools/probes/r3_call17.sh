# raw stream accessor: same-box A/B is not possible by knob -- compare against the r3r numbers of the same command; cost of the bench's per-launch events
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3t}; mkdir -p $O
cd $GRAFT_REPO_ROOT
one() { python bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for i in 1 2 3; do
echo "events    $(one)" >> $O/ab.txt
echo "no events $(FABIND_BENCH_NO_PROFILE=1 one)" >> $O/ab.txt
done
for i in 1 2; do
echo "pocket events    $(one --n-prot 100 --steps 20 --warmup 5)" >> $O/ab.txt
echo "pocket no events $(FABIND_BENCH_NO_PROFILE=1 one --n-prot 100 --steps 20 --warmup 5)" >> $O/ab.txt
done
python tools/probes/stack_hostprof.py 100 10 2>&1 | head -12 >> $O/ab.txt
cat $O/ab.txt
