#!/bin/bash
# Kernel-trace statistics and the two HBM-traffic counter passes for the bench workloads, as separate
# rocprofv3 runs.  Run on the GPU box from the repo root; results land under gpurun_out/profile/.
# usage: tools/profile_all.sh <workload>...
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/profile
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  extra=""
  [ "$w" = hdr20 ] || extra="--rotate 1"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${w}_stats" -o s -- python3 "$R/bench.py" --workload "$w" --cpu-seconds 0 > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err" || exit 1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${w}_fetch" -o f -- python3 "$R/bench.py" --workload "$w" --steps 3 --warmup 1 --cpu-seconds 0 $extra > /dev/null 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${w}_write" -o w -- python3 "$R/bench.py" --workload "$w" --steps 3 --warmup 1 --cpu-seconds 0 $extra > /dev/null 2>&1 || exit 1
  echo "profiled $w"
done
