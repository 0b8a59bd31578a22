#!/bin/bash
# round 5: same-box A/B of the FABind+ training step with this tree's norm.hip vs the previous commit's (FABIND_LIB)
timeout 1200 bash tools/ab.sh r5c22/lnfold_u 3 "FABIND_LIB=" "FABIND_LIB=fabind_amd/_ab/libfabind_normprev.so" --mode plus_train --steps 3 --warmup 2
