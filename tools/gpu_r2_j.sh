#!/bin/bash
mkdir -p gpurun_out
for v in blk64 blk128; do
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/$v.so timeout -k 10 200 python -u -m pytest tests/test_tx_build.py tests/test_short_records.py -x -q --timeout 150 --timeout-method thread > gpurun_out/r2_j_tests_$v.log 2>&1 || exit 1
for w in txbuild tximix udp1500 imix hdr20; do
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/$v.so timeout -k 10 200 python bench.py --workload $w --cpu-seconds 0 --steps 20 > gpurun_out/r2_j_${w}_$v.json 2> gpurun_out/r2_j_${w}_$v.err || exit 6
done
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/$v.so timeout -k 10 300 python tools/size_sweep.py 40 64 100 128 256 576 > gpurun_out/r2_j_sweep_$v.jsonl 2> gpurun_out/r2_j_sweep_$v.err || exit 3
done
