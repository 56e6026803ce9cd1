#!/bin/bash
# tools/pmc.sh NAME COUNTER [bench args...]: one rocprofv3 --pmc pass (kernel trace + ONE counter; FETCH_SIZE and WRITE_SIZE
# cannot share a pass) over the grafp kernels of bench.py (or of $PROG) -> gpurun_out/NAME_pmc_COUNTER.txt
set -e
NAME=$1; CTR=$2; shift 2
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_${NAME}_$CTR
rocprofv3 --kernel-trace --pmc $CTR --kernel-include-regex grafp -d /tmp/pmc_${NAME}_$CTR -o s -- python3 "$REPO/${PROG:-bench.py}" "$@" > /dev/null 2> "$REPO/gpurun_out/${NAME}_pmc_${CTR}.err" || true
DB=$(find /tmp/pmc_${NAME}_$CTR -name '*_results.db' | head -1)
python3 "$REPO/tools/rocpd_stats.py" "$DB" --pmc > "$REPO/gpurun_out/${NAME}_pmc_${CTR}.txt"
head -20 "$REPO/gpurun_out/${NAME}_pmc_${CTR}.txt"
