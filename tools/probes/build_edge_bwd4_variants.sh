# A/B builds of csrc/fused_edge_bwd4.hip (compile-time knobs) as separate libraries under fabind_amd/_ab/ (selected with FABIND_LIB)
cd "$(dirname "$0")/../../fabind_amd/csrc"
OBJS=$(ls *.o | grep -v fused_edge_bwd4.o)
i=0
for opt in "-DFE4_E3_DB=1" "-DFE4_P5_BATCH=8" "-DFE4_E3_DB=1 -DFE4_P5_BATCH=8" "-DFE4_SCAN64=1" "-DFE4_P5_BATCH=2"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -I../../include -I. $opt -c fused_edge_bwd4.hip -o /tmp/fe4_$i.o 2>/dev/null &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../_ab/libfabind_fe4_$i.so $OBJS /tmp/fe4_$i.o && echo "$i: $opt" 
done > ../_ab/variants.txt
cat ../_ab/variants.txt
