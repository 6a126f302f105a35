#!/bin/bash
# Builds a variant of the library for A/B measurements (selected at run time with TFHE_AMD_LIBRARY):
#   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DTFHE_AMD_WG_PHASE_CLOCKS=1]   ->  build/libtfhe_amd_<name>.so
# Translation units are compiled in parallel.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
OBJ=build/obj_$NAME
mkdir -p $OBJ
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 $*"
pids=()
for f in kernels generic_kernels wg_kernels keygen_kernels replica_kernels capi; do
    /opt/rocm/bin/hipcc $FLAGS -c ${SRCDIR:-tfhe_amd/csrc}/$f.hip -o $OBJ/$f.o 2> $OBJ/$f.err &
    pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { cat $OBJ/*.err | grep -E "error" ; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $OBJ/*.o -o build/libtfhe_amd_$NAME.so
echo "built build/libtfhe_amd_$NAME.so"
