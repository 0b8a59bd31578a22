# round 5, call 5: in-place inter-edge attention in no-grad passes (parity + A/B); where the config-3 (whole graph) step goes; pocket launch count; FABind+ training launch groups
O=$GRAFT_REPO_ROOT/gpurun_out/r5c5; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_stack.py tests/test_gpu_headline.py tests/test_gpu_model.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
tools/ab.sh r5c5/inplace_fwd 2 "FABIND_INTER_ATTN_INPLACE=1" "FABIND_INTER_ATTN_INPLACE=0" --mode fwd --steps 10 --warmup 3
tools/ab.sh r5c5/inplace_n8 1 "FABIND_INTER_ATTN_INPLACE=1" "FABIND_INTER_ATTN_INPLACE=0" --n-iter 8 --steps 4 --warmup 2
python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/plus_train.json 2>/dev/null
FABIND_BENCH_DUMP_PROFILE=$O/plus_train_launch_groups.txt python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 2 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o fb -- python3 $GRAFT_REPO_ROOT/bench.py --mode model --whole-pocket --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $O/bench_c3.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $O/prof_c3/fb_results.db > $O/config3_kernel_stats.txt 2>&1; rm -rf $O/prof_c3
head -30 $O/config3_kernel_stats.txt | cut -c1-170; tail -1 $O/bench_c3.log | cut -c1-300
rocprofv3 --kernel-trace --stats -d $O/prof_p -o fb -- python3 $GRAFT_REPO_ROOT/bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_p.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $O/prof_p/fb_results.db > $O/pocket_kernel_stats.txt 2>&1; rm -rf $O/prof_p
head -3 $O/pocket_kernel_stats.txt | cut -c1-170
