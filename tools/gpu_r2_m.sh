#!/bin/bash
# counters: stream route vs lane-group / per-record routes at 512, 1500 and 9000 bytes
S="512 1500 9000"
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/libunet_offload.so bash tools/pmc_sizes.sh stream $S || exit 1
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/nostream.so bash tools/pmc_sizes.sh nostream $S || exit 2
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/t16.so bash tools/pmc_sizes.sh t16 $S || exit 3
for t in stream nostream t16; do python tools/pmc_sizes.py gpurun_out/pmc_sizes_$t 5 $S > gpurun_out/pmc_sizes_$t.jsonl; done
