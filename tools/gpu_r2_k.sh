#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_gputests_k.log 2>&1 || exit 1
for w in imix odd udp1500; do timeout -k 10 200 python bench.py --workload $w --cpu-seconds 0 > gpurun_out/r2_k_bench_$w.json 2> gpurun_out/r2_k_bench_$w.err || exit 6; done
