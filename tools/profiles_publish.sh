#!/bin/bash
# Copies the results of one tools/profile_round.sh run into profiles/ under a round's name.
#   tools/profiles_publish.sh <run tag> <round name>      e.g.  r02c r02
# Kernel-trace statistics: gpurun_out/prof_<run>/<workload>/…_kernel_stats.csv -> profiles/<round>[_<workload>]_kernel_stats.csv
# Counters: gpurun_out/pmc_<run>*/ -> profiles/<round>_pmc_summary.json (tools/pmc_summarize.py)
set -e
RUN=$1; ROUND=$2
[ -d gpurun_out/prof_$RUN/bench ] && [ -d gpurun_out/pmc_$RUN ] || { echo "no results of run $RUN under gpurun_out/"; exit 1; }
python3 tools/pmc_summarize.py $RUN $ROUND > /dev/null
cp gpurun_out/prof_$RUN/bench/*/*_kernel_stats.csv profiles/${ROUND}_kernel_stats.csv
for d in gpurun_out/prof_$RUN/*/; do
    w=$(basename $d)
    [ $w = bench ] || cp $d*/*_kernel_stats.csv profiles/${ROUND}_${w}_kernel_stats.csv
done
echo "profiles/${ROUND}_* refreshed from run $RUN"
