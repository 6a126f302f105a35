#!/bin/bash
# usage: ab.sh "<python command>" variant...   (variant "head" = the in-tree library); two alternating rounds
cd "$GRAFT_REPO_ROOT"
CMD=$1; shift
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/build/libtfhe_amd_$v.so
  [ "$v" = head ] && lib=$PWD/tfhe_amd/libtfhe_amd.so
  echo "== $v"
  TFHE_AMD_LIBRARY=$lib timeout -k 10 300 $CMD 2>&1 | grep -v amdgpu.ids || exit 1
done
done
