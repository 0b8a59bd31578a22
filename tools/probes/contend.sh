#!/bin/bash
# Round 6: device-sharing stress of the weight-gradient contraction + LAS step (tests/contention_child.py), the round-6 library against the
# round-5 state (tools/probes/contend/libfabind_hip_r5.so: gemm.hip / attn.hip of commit c74caec, everything else current) in ONE call.
#   tools/probes/contend.sh [children] [passes]
O=${GRAFT_REPO_ROOT:-.}/gpurun_out/contend; mkdir -p $O
N=${1:-4}; P=${2:-50}
run() {  # label, env
  D=$(mktemp -d)
  for i in $(seq 1 $N); do (env $2 python tests/contention_child.py c$i $D $N $P 2>/dev/null | grep "^{" > $O/$1_c$i.json) & done
  wait; rm -rf $D
  echo "== $1 ($N processes x $P passes; counts = passes that differ from the child's first)"; cat $O/$1_c*.json
}
N1=$N; N=1; run solo_r6 "X=1"; run solo_r5 "FABIND_LIB=$PWD/tools/probes/contend/libfabind_hip_r5.so"; N=$N1
run r5 "FABIND_LIB=$PWD/tools/probes/contend/libfabind_hip_r5.so"
run r6 "X=1"
