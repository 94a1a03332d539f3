#!/bin/bash
# Runs one bench workload against the default library and every variant build under unet_amd/lib/variants/.
# usage: tools/variants.sh <workload> [extra bench flags]
w=$1; shift
mkdir -p gpurun_out
for lib in unet_amd/lib/libunet_offload.so unet_amd/lib/variants/*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  UNET_OFFLOAD_LIB=$PWD/$lib timeout -k 10 120 python bench.py --workload $w --steps 20 --cpu-seconds 0 "$@" 2>/dev/null \
    | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.1f GB/s  %.3f ms  parity=%s copy=%s" % (r["achieved"], r["kernel_ms"], d.get("parity_check"), r.get("d2d_copy_GBps")))' || exit 1
done
