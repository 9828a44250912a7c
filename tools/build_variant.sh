#!/bin/bash
# build a kernel A/B variant of the library: tools/build_variant.sh <tag> <extra hipcc flags...>
tag=$1; shift
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Impnn_amd/csrc $@"
mkdir -p build/$tag
/opt/rocm/bin/hipcc $F -ffp-contract=off -c mpnn_amd/csrc/mpnn_featurizer.hip -o build/$tag/f.o &
/opt/rocm/bin/hipcc $F -c mpnn_amd/csrc/mpnn_edge.hip -o build/$tag/e.o &
/opt/rocm/bin/hipcc $F -c mpnn_amd/csrc/mpnn_node.hip -o build/$tag/n.o &
/opt/rocm/bin/hipcc $F -c mpnn_amd/csrc/mpnn_abi.cpp -o build/$tag/a.o &
wait
mkdir -p variants
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build/$tag/*.o -o variants/libmpnn_$tag.so
