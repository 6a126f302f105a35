#!/bin/bash
# texture-path counters of one tools/prof_target.py workload:  pmc_tcp.sh <tag> <mode>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/pmc_$1_$2; mkdir -p $OUT
run() { name=$1; shift; timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/prof_target.py $MODE > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }; echo "pass $name ok"; }
MODE=$2
run tcp1 TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
run tcp2 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
