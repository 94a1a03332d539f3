#!/bin/bash
# span path: parity, then A/B sweep, copy probe, counters per size
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_short_records.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r2_short.log 2>&1 || exit 1
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_gputests_b.log 2>&1 || exit 2
S="20 40 64 65 80 81 96 100 112 113 128 160"
timeout -k 10 300 python tools/size_sweep.py $S > gpurun_out/r2_sweep_span.jsonl 2> gpurun_out/r2_sweep_span.err || exit 3
UNET_OFFLOAD_LIB=$PWD/unet_amd/lib/variants/nospan.so timeout -k 10 300 python tools/size_sweep.py $S > gpurun_out/r2_sweep_nospan.jsonl 2> gpurun_out/r2_sweep_nospan.err || exit 4
timeout -k 10 300 tools/copy_probe.bin > gpurun_out/r2_copy_probe.log 2>&1 || exit 5
for w in imix udp1500 hdr20; do timeout -k 10 200 python bench.py --workload $w --cpu-seconds 0 > gpurun_out/r2_bench_$w.json 2> gpurun_out/r2_bench_$w.err || exit 6; done
bash tools/pmc_sizes.sh span 40 64 65 100 128 > gpurun_out/r2_pmc_span.log 2>&1 || exit 7
