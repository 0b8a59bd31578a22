O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c13}; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -k "mlp2 or epilogue or gemm_bf16_lds or fused_edge_pipeline" > $O/tests_a.log 2>&1; tail -3 $O/tests_a.log
python tools/probes/gemm_node_epi.py > $O/gemm_node_epi.txt 2>&1; tail -12 $O/gemm_node_epi.txt
python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a $O/ab.txt
