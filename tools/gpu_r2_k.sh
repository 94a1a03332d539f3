#!/bin/bash
# stream route: parity first, then the sizes it serves
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > gpurun_out/r2_k_tests.log 2>&1 || { tail -30 gpurun_out/r2_k_tests.log; exit 1; }
tail -3 gpurun_out/r2_k_tests.log
for v in default t16 nostream; do
lib=$PWD/unet_amd/lib/variants/$v.so; [ $v = default ] && lib=$PWD/unet_amd/lib/libunet_offload.so
echo "== $v"
if [ $v != default ]; then UNET_OFFLOAD_LIB=$lib timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_short_records.py -x -q --timeout 150 --timeout-method thread > gpurun_out/r2_k_tests_$v.log 2>&1 || { tail -30 gpurun_out/r2_k_tests_$v.log; exit 1; }; fi
UNET_OFFLOAD_LIB=$lib timeout -k 10 300 python tools/size_sweep.py 65 128 160 256 400 512 577 800 1000 1024 1040 1500 2048 4096 9000 65535 > gpurun_out/r2_k_sweep_$v.jsonl 2> gpurun_out/r2_k_sweep_$v.err || exit 3
python - <<PY
import json
print(' '.join(f"{d['len']}:{d['verify_GBps']:.0f}/{d['csum_GBps']:.0f}" for d in map(json.loads, open('gpurun_out/r2_k_sweep_$v.jsonl'))))
PY
for w in udp1500 imix odd trace generate fields; do
UNET_OFFLOAD_LIB=$lib timeout -k 10 200 python bench.py --workload $w --cpu-seconds 0 --steps 20 > gpurun_out/r2_k_${w}_$v.json 2> gpurun_out/r2_k_${w}_$v.err || exit 6
python - <<PY
import json
d=json.load(open('gpurun_out/r2_k_${w}_$v.json')); print('$w', round(d['roofline']['achieved']), round(d['value'],3))
PY
done
done
