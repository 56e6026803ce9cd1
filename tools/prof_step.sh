#!/bin/bash
# tools/prof_step.sh NAME PAIRS: rocprofv3 kernel trace of five eager steps at PAIRS pairs (tools/step_prof.py)
# -> gpurun_out/NAME_kernel_stats.txt
set -e
NAME=$1; PAIRS=${2:-128}
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$NAME
rocprofv3 --kernel-trace --stats -d /tmp/ks_$NAME -o s -- python3 "$REPO/tools/step_prof.py" $PAIRS > /dev/null 2> "$REPO/gpurun_out/${NAME}_prof.err" || true
DB=$(find /tmp/ks_$NAME -name '*_results.db' | head -1)
python3 "$REPO/tools/rocpd_stats.py" "$DB" 90 > "$REPO/gpurun_out/${NAME}_kernel_stats.txt"
head -60 "$REPO/gpurun_out/${NAME}_kernel_stats.txt"
