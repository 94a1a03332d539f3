#!/bin/bash
# first GPU pass of the round: full GPU tests, the no-allocation test, a size sweep with the kernels as they stand
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_gputests.log 2>&1 || exit 1
timeout -k 10 120 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > gpurun_out/r2_noalloc.log 2>&1 || exit 2
timeout -k 10 400 python tools/size_sweep.py 20 40 64 65 80 81 96 100 128 160 256 400 512 576 577 1024 1040 1499 1500 2048 4096 9000 65535 > gpurun_out/r2_sweep0.jsonl 2> gpurun_out/r2_sweep0.err || exit 3
