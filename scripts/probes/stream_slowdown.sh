#!/bin/bash
# The stream / graph slowdown probe under several runtime environments, one box.
# usage (GPU box): bash scripts/probes/stream_slowdown.sh > gpurun_out/stream_slowdown.txt
cd "${GRAFT_REPO_ROOT:-$PWD}"
P=scripts/probes/stream_slowdown.py
run() { echo "## $*"; env "$@" python3 $P "$ORDER" 2>&1 | grep -v Warning; }
ORDER=ABCDEFG run DUMMY=1
ORDER=ACE run GPU_MAX_HW_QUEUES=1
ORDER=ACE run GPU_MAX_HW_QUEUES=2
ORDER=ACE run GPU_MAX_HW_QUEUES=8
ORDER=APC run DUMMY=1
ORDER=ANC run DUMMY=1
ORDER=AEC run DUMMY=1
ORDER=ACE run HSA_ENABLE_SDMA=0
ORDER=ACE run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
ORDER=AAA run DUMMY=1
