#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 tools/copy_probe.bin > gpurun_out/r2_copy_probe4.log 2>&1 || exit 2
UNET_BENCH_TX_OSTRIDE=96 timeout -k 10 200 python bench.py --workload txbuild --cpu-seconds 0 --steps 20 > gpurun_out/r2_i_txbuild_o96.json 2> gpurun_out/r2_i_txbuild_o96.err || exit 6
UNET_BENCH_TX_OSTRIDE=96 timeout -k 10 200 python bench.py --workload txbuild --stride16 96 --cpu-seconds 0 --steps 20 > gpurun_out/r2_i_txbuild_io96.json 2> gpurun_out/r2_i_txbuild_io96.err || exit 6
