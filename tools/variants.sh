#!/bin/bash
# Runs one bench workload against the default library and every variant build under unet_amd/lib/variants/.
# usage: tools/variants.sh <workload> [extra bench flags]
#        tools/variants.sh debug      build unet_amd/lib/variants/debug.so with -DUNET_DEBUG (every LDS area store bounds-checked
#                                     and counted, unet_debug_lds_violations); run the soak against it with
#                                     UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/debug.so python -m pytest tests/test_fuzz_short.py -m gpu
if [ "$1" = debug ]; then
  mkdir -p unet_amd/lib/variants
  exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -DUNET_DEBUG unet_amd/csrc/unet_offload.hip -o unet_amd/lib/variants/debug.so
fi
w=$1; shift
mkdir -p gpurun_out
for lib in unet_amd/lib/libunet_offload.so unet_amd/lib/variants/*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  UNET_OFFLOAD_LIB=$PWD/$lib timeout -k 10 120 python bench.py --workload $w --steps 20 --cpu-seconds 0 "$@" 2>/dev/null \
    | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.1f GB/s  %.3f ms  parity=%s copy=%s" % (r["achieved"], r["kernel_ms"], d.get("parity_check"), r.get("d2d_copy_GBps")))' || exit 1
done
