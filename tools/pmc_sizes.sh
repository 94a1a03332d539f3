#!/bin/bash
# Hardware counters of k_verify / k_csum per record size: separate rocprofv3 --pmc passes over tools/size_sweep.py
# (HBM read bytes, HBM written bytes, two sets of SQ counters).  Run on the GPU box from the repo root.
# usage: tools/pmc_sizes.sh <tag> <size>...      -> gpurun_out/pmc_sizes_<tag>/{fetch,write,sq,sq2}/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; shift
OUT=$R/gpurun_out/pmc_sizes_$tag
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export UNET_SWEEP_TOTAL=$((1<<30)) UNET_SWEEP_ITERS=3
pass() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o c -- python3 "$R/tools/size_sweep.py" "${SIZES[@]}" > "$OUT/$name.jsonl" 2> "$OUT/$name.err" || { echo "pass $name failed"; exit 1; }
  echo "pass $name done"
}
SIZES=("$@")
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
