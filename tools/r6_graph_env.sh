#!/bin/bash
# hipGraphLaunch's host time (tools/host_phases.py) against the HIP runtime's graph / kernarg switches, K1's 32-cloud share.
export TMPDIR=/tmp
OUT=gpurun_out/r6env; mkdir -p $OUT
run() { # name, env assignments...
  name=$1; shift
  env "$@" python3 tools/host_phases.py ${BATCH:-32} 2000 > $OUT/$name.txt 2>&1
  echo "== $name ($*)"; grep -E "hipGraphLaunch|turn-around|mean whole" $OUT/$name.txt
}
run default A=1
run packet_capture0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run packet_capture1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run graph_queues1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run graph_queues2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run dev_kernarg0 HIP_FORCE_DEV_KERNARG=0
run dev_kernarg1 HIP_FORCE_DEV_KERNARG=1
run kernarg_copy_opt0 DEBUG_HIP_KERNARG_COPY_OPT=0
run kernarg_copy_opt1 DEBUG_HIP_KERNARG_COPY_OPT=1
run fgs_kernarg0 ROC_USE_FGS_KERNARG=0
run default_again A=1
