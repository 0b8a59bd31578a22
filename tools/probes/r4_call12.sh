O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c12}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/profpt -o pt -- python3 $GRAFT_REPO_ROOT/bench.py --mode plus_train --no-cpu-baseline --steps 3 --warmup 2 > $O/bench_pt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/profpt/pt_results.db > $O/plus_train_kernel_stats.txt 2>&1
rm -rf $O/profpt
FABIND_BENCH_DUMP_PROFILE=$O/plus_train_launch_groups.txt python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/bench_plus_train.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_plus_train.json')); print('plus_train', round(d['value'],1), round(d['ms_per_step'],1))"
head -45 $O/plus_train_kernel_stats.txt | cut -c1-150
