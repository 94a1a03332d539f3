#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 200 python -u -m pytest tests/test_async_counters.py tests/test_zz_no_alloc.py tests/test_short_records.py tests/test_tx_build.py -x -q --timeout 150 --timeout-method thread > gpurun_out/r2_c_tests.log 2>&1 || exit 1
timeout -k 10 300 tools/copy_probe.bin > gpurun_out/r2_copy_probe2.log 2>&1 || exit 2
S="20 40 64 128"
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/occ8_exp.so timeout -k 10 300 python tools/size_sweep.py 20 32 48 64 > gpurun_out/r2_sweep_occ8.jsonl 2> gpurun_out/r2_sweep_occ8.err || exit 3
for t in 4 8 16 32; do UNET_TX_TILE=$t timeout -k 10 200 python bench.py --workload txbuild --cpu-seconds 0 --steps 20 > gpurun_out/r2_tx_tile$t.json 2> gpurun_out/r2_tx_tile$t.err || exit 4; done
for t in 8 16 32; do UNET_TX_TILE=$t timeout -k 10 200 python bench.py --workload tximix --cpu-seconds 0 --steps 20 > gpurun_out/r2_tximix_tile$t.json 2> gpurun_out/r2_tximix_tile$t.err || exit 5; done
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/txocc5.so timeout -k 10 200 python bench.py --workload txbuild --cpu-seconds 0 --steps 20 > gpurun_out/r2_tx_occ5.json 2> gpurun_out/r2_tx_occ5.err || exit 6
