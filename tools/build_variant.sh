#!/bin/bash
# build a kernel A/B variant of the library: tools/build_variant.sh <tag> <extra hipcc flags...>
# Only the forward edge kernels are rebuilt with the extra flags; every other object comes from the main build (run make first).
tag=$1; shift
C=mpnn_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$C $@"
mkdir -p build/$tag variants
/opt/rocm/bin/hipcc $F -c $C/mpnn_edge.hip -o build/$tag/e.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build/$tag/e.o $C/mpnn_featurizer.o $C/mpnn_edge_bwd.o $C/mpnn_node.o \
  $C/mpnn_node_bwd.o $C/mpnn_abi.o $C/mpnn_featurizer_f16.o $C/mpnn_edge_f16.o $C/mpnn_node_f16.o -o variants/libmpnn_$tag.so
