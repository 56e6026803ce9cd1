#!/bin/bash
# tools/pmc_view.sh NAME REGEX "CTR1 CTR2 ...": one rocprofv3 --pmc pass over five 1024-pair training steps
# (tools/step_prof.py), limited to the kernels matching REGEX -> gpurun_out/NAME_pm.txt (per kernel and counter) and
# gpurun_out/NAME_view.txt (shares of SQ_WAVE_CYCLES: parked, issue-stalled, VALU; LDS busy / conflicts; MFMA busy).
# Counters only with --kernel-trace (no sys/runtime traces beside --pmc on this pool).
NAME=$1; REGEX=$2; CTRS=$3
REPO=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm_$NAME
rocprofv3 --kernel-trace --pmc $CTRS --kernel-include-regex "$REGEX" -d /tmp/pm_$NAME -o s -- python3 "$REPO/tools/step_prof.py" 1024 > "$REPO/gpurun_out/${NAME}_pm.log" 2>&1 || true
DB=$(find /tmp/pm_$NAME -name '*_results.db' | head -1)
python3 "$REPO/tools/rocpd_stats.py" "$DB" --pmc > "$REPO/gpurun_out/${NAME}_pm.txt"
python3 "$REPO/tools/pmc_view_parse.py" "$REPO/gpurun_out/${NAME}_pm.txt" > "$REPO/gpurun_out/${NAME}_view.txt"
cat "$REPO/gpurun_out/${NAME}_view.txt"
