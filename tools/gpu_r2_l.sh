#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_short_records.py tests/test_gpu_parity.py tests/test_stream_route.py tests/test_zz_no_alloc.py -x -q --timeout 150 --timeout-method thread > gpurun_out/r2_l_tests.log 2>&1 || { tail -30 gpurun_out/r2_l_tests.log; exit 1; }
tail -2 gpurun_out/r2_l_tests.log
timeout -k 10 300 python tools/size_sweep.py 20 32 40 48 64 65 80 96 112 128 > gpurun_out/r2_l_sweep.jsonl 2> gpurun_out/r2_l_sweep.err || exit 3
python - <<PY
import json
print(' '.join(f"{d['len']}:{d['verify_GBps']:.0f}/{d['csum_GBps']:.0f}" for d in map(json.loads, open('gpurun_out/r2_l_sweep.jsonl'))))
PY
