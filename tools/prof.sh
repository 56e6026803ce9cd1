#!/bin/bash
# tools/prof.sh NAME [bench args...]: rocprofv3 kernel trace of bench.py on the GPU box -> gpurun_out/NAME_kernel_stats.txt
set -e
NAME=$1; shift
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$NAME
rocprofv3 --kernel-trace --stats -d /tmp/ks_$NAME -o s -- python3 "$REPO/bench.py" "$@" > "$REPO/gpurun_out/${NAME}_bench.json" 2> "$REPO/gpurun_out/${NAME}_prof.err" || true
DB=$(find /tmp/ks_$NAME -name '*_results.db' | head -1)
python3 "$REPO/tools/rocpd_stats.py" "$DB" 70 > "$REPO/gpurun_out/${NAME}_kernel_stats.txt"
head -45 "$REPO/gpurun_out/${NAME}_kernel_stats.txt"
