#!/bin/bash
# tools/knn_prof.sh CLIPS [extra env]: per-kernel times of the certified k-NN path (tools/knn_bench.py under rocprofv3
# --kernel-trace).  With the measurement library, GRAFP_KX_STOP=1|2 ends knn_exact_clip_kernel after its flag scan / after
# its light queries (results are then incomplete: timing only).
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_knn
rocprofv3 --kernel-trace --stats -d /tmp/ks_knn -o s -- python3 "$REPO/tools/knn_bench.py" --clips $1 --dtype bf16 > /dev/null 2>&1
DB=$(find /tmp/ks_knn -name '*_results.db' | head -1)
python3 "$REPO/tools/rocpd_stats.py" "$DB" 30 | grep -i "knn_exact\|knn_topk_raw\|knn_norms\|calls"
