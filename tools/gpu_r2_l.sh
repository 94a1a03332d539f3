#!/bin/bash
mkdir -p gpurun_out
for f in 2097152 8388608 33554432 67108864; do
for w in imix; do
timeout -k 10 300 python bench.py --workload $w --cpu-seconds 0 --steps 20 --frames $f > gpurun_out/r2_l_${w}_$f.json 2> gpurun_out/r2_l_${w}_$f.err || exit 6
python - <<PY
import json
d=json.load(open('gpurun_out/r2_l_${w}_$f.json')); print('$f $w', round(d['roofline']['achieved']), round(d['value'],3), d['ms_per_step'])
PY
done
done
