#!/bin/bash
mkdir -p gpurun_out
for v in nt16_3 nt16_7 nt16_1; do
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/$v.so timeout -k 10 300 python tools/size_sweep.py 576 800 1024 > gpurun_out/r2_sweep_$v.jsonl 2> gpurun_out/r2_sweep_$v.err || exit 3
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/$v.so timeout -k 10 200 python bench.py --workload imix --cpu-seconds 0 > gpurun_out/r2_g_imix_$v.json 2> gpurun_out/r2_g_imix_$v.err || exit 6
done
for v in nt8_1 nt8_3; do
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/$v.so timeout -k 10 300 python tools/size_sweep.py 288 400 512 > gpurun_out/r2_sweep_$v.jsonl 2> gpurun_out/r2_sweep_$v.err || exit 3
done
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/nt4_1.so timeout -k 10 300 python tools/size_sweep.py 160 208 256 > gpurun_out/r2_sweep_nt4_1.jsonl 2> gpurun_out/r2_sweep_nt4_1.err || exit 3
timeout -k 10 300 python tools/size_sweep.py 160 208 256 288 400 512 576 800 1024 > gpurun_out/r2_sweep_base.jsonl 2> gpurun_out/r2_sweep_base.err || exit 3
