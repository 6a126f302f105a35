#!/bin/bash
# All profiling runs of a round: kernel-trace statistics of the benchmark and of every workload of
# tools/prof_target.py, then the counter passes (each rocprofv3 run traces OR counts, never both).
#   tools/profile_round.sh <tag>     outputs under gpurun_out/prof_<tag>/ and gpurun_out/pmc_<tag>*/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
stats() { name=$1; shift; timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1 || { echo "stats $name failed"; tail -5 $OUT/$name.log; exit 1; }; echo "stats $name ok"; }
stats bench python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras
for mode in latency1 latency256 n2048 reference-fft keygen; do
    stats $mode python3 tools/prof_target.py $mode
done
for mode in default n2048 latency1 latency256 reference-fft keygen; do
    echo "== counters: $mode"
    bash tools/pmc_passes.sh $TAG $mode || exit 1
done
