# A/B builds of single csrc files with compile-time knobs as separate libraries under fabind_amd/_ab/ (selected with FABIND_LIB).
#   tools/probes/build_variants.sh <name> "<file.hip>:<opts>" ["<file2.hip>:<opts>" ...]   ->  fabind_amd/_ab/libfabind_<name>.so
# (the default objects must be current: python -m fabind_amd.build first)
cd "$(dirname "$0")/../../fabind_amd/csrc"
NAME=$1; shift
mkdir -p ../_ab
EXCL=""; NEW=""
for spec in "$@"; do
  f=${spec%%:*}; opt=${spec#*:}
  o=/tmp/ab_${NAME}_${f%.hip}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -I../../include -I. $opt -c $f -o $o 2>/dev/null || { echo "compile failed: $spec"; exit 1; }
  EXCL="$EXCL ${f%.hip}.o"; NEW="$NEW $o"
done
OBJS=$(ls *.o | grep -v -x -F "$(echo $EXCL | tr ' ' '\n')")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../_ab/libfabind_$NAME.so $OBJS $NEW && echo "built fabind_amd/_ab/libfabind_$NAME.so: $*"
