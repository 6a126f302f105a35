#!/bin/bash
# A/B: exchange pads (head vs oldpad), latency key prefetch depth, stamps of the N = 2048 kernel
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/r3/ab1.log; mkdir -p gpurun_out/r3; : > $L
run() { echo "== $1: $3" >> $L; lib=$PWD/build/libtfhe_amd_$1.so; [ "$1" = head ] && lib=$PWD/tfhe_amd/libtfhe_amd.so
        TFHE_AMD_LIBRARY=$lib timeout -k 10 $2 $3 2>&1 | grep -v amdgpu.ids >> $L || exit 1; }
timeout -k 10 600 python -u -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread >> $L 2>&1 || { tail -20 $L; exit 1; }
for rep in 1 2; do
  for v in head oldpad; do run $v 200 "python3 tools/quick_perf.py 8192" ; done
done
for rep in 1 2; do
  for v in head oldpad; do run $v 200 "python3 tools/n2048_perf.py 2048" ; done
done
for rep in 1 2; do
  for v in head lat2; do run $v 200 "python3 tools/scratch/lat_perf.py" ; done
done
run wgclk 200 "python3 tools/n2048_perf.py 512"
