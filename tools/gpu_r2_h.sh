#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 200 python -u -m pytest tests/test_tx_build.py -x -q --timeout 150 --timeout-method thread > gpurun_out/r2_h_tests.log 2>&1 || exit 1
for v in txd4 txd5 txd6; do
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/$v.so timeout -k 10 200 python -u -m pytest tests/test_tx_build.py -x -q --timeout 150 --timeout-method thread > gpurun_out/r2_h_tests_$v.log 2>&1 || exit 1
for w in txbuild tximix; do
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/$v.so timeout -k 10 200 python bench.py --workload $w --cpu-seconds 0 --steps 20 > gpurun_out/r2_h_${w}_$v.json 2> gpurun_out/r2_h_${w}_$v.err || exit 6
done; done
for w in txbuild tximix; do timeout -k 10 200 python bench.py --workload $w --cpu-seconds 0 --steps 20 > gpurun_out/r2_h_${w}_base.json 2> gpurun_out/r2_h_${w}_base.err || exit 6; done
