#!/bin/bash
mkdir -p gpurun_out
for v in default d2 d6 d8 nostream; do
lib=$PWD/unet_amd/lib/variants/$v.so; [ $v = default ] && lib=$PWD/unet_amd/lib/libunet_offload.so
for w in imix odd trace; do
UNET_OFFLOAD_LIB=$lib timeout -k 10 200 python bench.py --workload $w --cpu-seconds 0 --steps 20 > gpurun_out/r2_l_${w}_$v.json 2> gpurun_out/r2_l_${w}_$v.err || exit 6
python - <<PY
import json
d=json.load(open('gpurun_out/r2_l_${w}_$v.json')); print('$v $w', round(d['roofline']['achieved']), round(d['value'],3))
PY
done
done
