# Round 3's final tree against round 2's final tree (_ab_prev = commit 19517ec with its own build) on ONE box, interleaved, under ROUND 2's
# bench protocol (one resident batch re-served: FABIND_BENCH_REUSE_BATCH=1 on the current tree; that tree's bench.py does it by itself),
# then the current tree under its own protocol.
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3z}; mkdir -p $O
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"; }
for i in 1 2 3; do
echo "r3 tree, r2 protocol   $(FABIND_BENCH_REUSE_BATCH=1 one)" >> $O/vs_r2.txt
echo "r2 tree (19517ec)      $(cd _ab_prev && one)" >> $O/vs_r2.txt
echo "r3 tree, r3 protocol   $(one)" >> $O/vs_r2.txt
done
cat $O/vs_r2.txt
