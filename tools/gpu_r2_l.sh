#!/bin/bash
mkdir -p gpurun_out
for v in default hb128 hb64; do
lib=$PWD/unet_amd/lib/variants/$v.so; [ $v = default ] && lib=$PWD/unet_amd/lib/libunet_offload.so
for f in 16777216 67108864; do
UNET_OFFLOAD_LIB=$lib timeout -k 10 200 python bench.py --workload hdr20 --cpu-seconds 0 --steps 30 --frames $f > gpurun_out/r2_l_hdr20_$v.json 2> gpurun_out/r2_l_hdr20_$v.err || exit 6
python - <<PY
import json
d=json.load(open('gpurun_out/r2_l_hdr20_$v.json')); print('$v $f', round(d['roofline']['achieved']), round(d['value'],3))
PY
done
done
