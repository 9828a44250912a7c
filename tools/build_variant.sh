#!/bin/bash
# build a kernel A/B variant of the library: tools/build_variant.sh <tag> <edge|featurizer|node> <extra hipcc flags...>
# Only the named forward source is rebuilt with the extra flags; every other object comes from the main build (run make first).
tag=$1; which=$2; shift 2
C=mpnn_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$C $@"
[ "$which" = featurizer ] && F="$F -ffp-contract=off -fno-slp-vectorize"
mkdir -p build/$tag variants
/opt/rocm/bin/hipcc $F -c $C/mpnn_$which.hip -o build/$tag/v.o || exit 1
OBJS=""
for o in featurizer edge edge_bwd node node_bwd data loss abi featurizer_f16 edge_f16 node_f16; do
  if [ "$o" = "$which" ]; then OBJS="$OBJS build/$tag/v.o"; else OBJS="$OBJS $C/mpnn_$o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -o variants/libmpnn_$tag.so
