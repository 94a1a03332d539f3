#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 tools/copy_probe.bin > gpurun_out/r2_copy_probe3.log 2>&1 || exit 2
UNET_TILE=32 timeout -k 10 300 python tools/size_sweep.py 20 40 64 65 80 96 100 112 113 128 160 256 > gpurun_out/r2_sweep_tile32.jsonl 2> gpurun_out/r2_sweep_tile32.err || exit 3
UNET_TILE=16 timeout -k 10 300 python tools/size_sweep.py 64 112 128 160 256 512 > gpurun_out/r2_sweep_tile16.jsonl 2> gpurun_out/r2_sweep_tile16.err || exit 3
