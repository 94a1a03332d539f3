#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_short_records.py tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread -k "not full_size and not flow_trace" > gpurun_out/r2_e_tests.log 2>&1 || exit 1
timeout -k 10 300 python tools/size_sweep.py 20 40 64 65 80 96 100 112 113 128 160 256 576 1500 > gpurun_out/r2_sweep_e.jsonl 2> gpurun_out/r2_sweep_e.err || exit 3
for w in imix udp1500; do timeout -k 10 200 python bench.py --workload $w --cpu-seconds 0 > gpurun_out/r2_e_bench_$w.json 2> gpurun_out/r2_e_bench_$w.err || exit 6; done
