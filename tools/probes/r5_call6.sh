# round 5, call 6: FABind+ pair update with the fused training forward (parity + A/B); in-place inter-edge attention in no-grad passes (parity + A/B);
# where the config-3 (whole graph) step goes; pocket launch count; FABind+ training launch groups
O=$GRAFT_REPO_ROOT/gpurun_out/r5c6; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_plus.py tests/test_gpu_plus_dense_api.py -x -q -s > $O/tests_plus.log 2>&1; grep -E "pair update|passed|failed|Error" $O/tests_plus.log | tail -6
timeout 1800 python -m pytest tests/test_gpu_stack.py tests/test_gpu_headline.py tests/test_gpu_model.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 1800 python -m pytest tests/test_gpu_production.py -x -q -k "plus" > $O/tests_prod.log 2>&1; tail -3 $O/tests_prod.log
tools/ab.sh r5c6/plus_fuse 2 "FABIND_PLUS_FUSE_PAIR_TRAIN=1" "FABIND_PLUS_FUSE_PAIR_TRAIN=0" --mode plus_train --steps 3 --warmup 2
tools/ab.sh r5c6/inplace_fwd 2 "FABIND_INTER_ATTN_INPLACE=1" "FABIND_INTER_ATTN_INPLACE=0" --mode fwd --steps 10 --warmup 3
tools/ab.sh r5c6/inplace_n8 1 "FABIND_INTER_ATTN_INPLACE=1" "FABIND_INTER_ATTN_INPLACE=0" --n-iter 8 --steps 4 --warmup 2
FABIND_BENCH_DUMP_PROFILE=$O/plus_train_launch_groups.txt python bench.py --mode plus_train --no-cpu-baseline --no-extras --steps 3 --warmup 2 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o fb -- python3 $GRAFT_REPO_ROOT/bench.py --mode model --whole-pocket --no-cpu-baseline --no-extras --steps 4 --warmup 2 > $O/bench_c3.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $O/prof_c3/fb_results.db > $O/config3_kernel_stats.txt 2>&1; rm -rf $O/prof_c3
head -24 $O/config3_kernel_stats.txt | cut -c1-170; tail -1 $O/bench_c3.log | cut -c1-200
rocprofv3 --kernel-trace --stats -d $O/prof_p -o fb -- python3 $GRAFT_REPO_ROOT/bench.py --n-prot 100 --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_p.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $O/prof_p/fb_results.db > $O/pocket_kernel_stats.txt 2>&1; rm -rf $O/prof_p
head -3 $O/pocket_kernel_stats.txt | cut -c1-170
