#!/bin/bash
# A/B of library variants on the N = 2048 set: same box, alternating runs
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/build/libtfhe_amd_$v.so
  [ "$v" = head ] && lib=$PWD/tfhe_amd/libtfhe_amd.so
  echo "== $v"
  TFHE_AMD_LIBRARY=$lib timeout -k 10 120 python3 tools/n2048_perf.py 2048 2>&1 | grep -v amdgpu.ids | tail -9 || exit 1
done
done
