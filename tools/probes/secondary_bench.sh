run() { tag="$1"; shift; python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 > gpurun_out/sec_$tag.json; python -c "import json; d=json.load(open('gpurun_out/sec_$tag.json')); print('SEC $tag', round(d['value'],1), round(d['ms_per_step'],1))"; }
run fwd_a --mode fwd
run fwd_b --mode fwd
run fwd_niter8 --mode fwd --n-iter 8
run model_niter1 --mode model --n-iter 1
run model_niter8 --mode model --n-iter 8
run pocket_fwdbwd --mode fwdbwd --n-prot 100
run pocket_fwd8 --mode fwd --n-prot 100 --n-iter 8
