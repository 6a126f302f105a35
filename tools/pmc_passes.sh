#!/bin/bash
# rocprofv3 counter passes over tools/prof_target.py, one run per pass (no trace domains mixed in).
#   tools/pmc_passes.sh <tag> [mode]      ->  gpurun_out/pmc_<tag>[_<mode>]/<pass>/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
MODE=${2:-default}
OUT=gpurun_out/pmc_$1
[ "$MODE" != default ] && OUT=${OUT}_$MODE
mkdir -p $OUT
run() { name=$1; shift; timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/prof_target.py $MODE > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }; echo "pass $name ok"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
