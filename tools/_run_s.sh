REPO=$(pwd)
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "search or merge" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
prof() {  # name, nq
  rm -rf /tmp/ks_$1
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$1 -o s -- python3 $REPO/tools/search_bench.py --reps 30 256 512 1024 $2 > /tmp/ks_$1.log 2>&1
  grep nq= /tmp/ks_$1.log
  DB=$(find /tmp/ks_$1 -name '*_results.db' | head -1)
  echo "== $1"; python3 $REPO/tools/rocpd_stats.py "$DB" 12 | grep -i "search_\|calls" | cut -c1-110
}
prof pipe2_4096 4096
cd $REPO
python tools/search_bench.py --reps 30 1 8 41 128 256 512 1024 2048 4096 2>&1 | grep nq=
