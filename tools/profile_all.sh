#!/bin/bash
# Every measurement the design document cites, produced in one go on the GPU box, under
# gpurun_out/profile_<tag>/ (tools/collect_profiles.py <tag> then copies it to profiles/<tag>/):
#   per workload: bench JSON under rocprofv3 --kernel-trace --stats, and the two HBM-traffic counter passes
#   bench JSONs without a profiler for the remaining workloads, the emulated ranks of an 8-GPU job
#   size sweeps of k_verify / k_csum and of k_tx_build, the batch-size sweep, the host-pointer rate
# usage: tools/profile_all.sh <tag> [part]      part: pmc | bench | sweeps | all (default)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
tag=${1:?tag}; part=${2:-all}
OUT=$R/gpurun_out/profile_$tag
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$part" = pmc ] || [ "$part" = all ]; then
for w in udp1500 imix odd hdr20 csum generate txbuild tximix; do
  extra=""
  [ "$w" = hdr20 ] || extra="--rotate 1"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${w}_stats" -o s -- python3 "$R/bench.py" --workload "$w" --cpu-seconds 0 > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err" || exit 1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${w}_fetch" -o f -- python3 "$R/bench.py" --workload "$w" --steps 3 --warmup 1 --cpu-seconds 0 $extra > /dev/null 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${w}_write" -o w -- python3 "$R/bench.py" --workload "$w" --steps 3 --warmup 1 --cpu-seconds 0 $extra > /dev/null 2>&1 || exit 1
  echo "profiled $w"
done
fi
if [ "$part" = bench ] || [ "$part" = all ]; then
cd "$R"
for w in fields eth1514; do
  timeout -k 10 200 python3 bench.py --workload $w --cpu-seconds 0 > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err" || exit 2
  echo "bench $w"
done
timeout -k 10 200 python3 bench.py --workload txbuild --tx-layout packed --cpu-seconds 0 > "$OUT/bench_txbuild_packed.json" 2>> "$OUT/bench.err" || exit 2
timeout -k 10 200 python3 bench.py --force-dist --cpu-seconds 0 > "$OUT/bench_udp1500_nccl_world1.json" 2>> "$OUT/bench.err" || exit 2
( cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tcptrack_stats" -o s -- python3 "$R/bench.py" --workload tcptrack --steps 10 --cpu-seconds 0 > "$OUT/bench_tcptrack.json" 2> "$OUT/bench_tcptrack.err" ) || exit 2
echo "bench tcptrack"
timeout -k 10 300 python3 bench.py --workload udp1500 --cpu-seconds 10 > "$OUT/bench_udp1500_with_cpu.json" 2> "$OUT/bench_udp1500_with_cpu.err" || exit 2
timeout -k 10 300 python3 bench.py --workload udp1500 --emulate-world 8 --emulate-rank 3 --cpu-seconds 0 > "$OUT/bench_udp1500_emulated_rank3of8.json" 2>> "$OUT/bench.err" || exit 2
timeout -k 10 300 python3 bench.py --workload trace --emulate-world 8 --emulate-rank 2 --steps 4 --cpu-seconds 0 > "$OUT/bench_trace_emulated_rank2of8_4chunks.json" 2>> "$OUT/bench.err" || exit 2
timeout -k 10 400 python3 bench.py --workload trace --steps 8 --total-frames $((1<<27)) --cpu-seconds 0 > "$OUT/bench_trace_8chunks.json" 2>> "$OUT/bench.err" || exit 2
echo "bench extra done"
timeout -k 10 300 python3 tools/host_path_bench.py > "$OUT/host_path_pcie_inclusive.json" 2>> "$OUT/bench.err" || exit 2
fi
if [ "$part" = sweeps ] || [ "$part" = all ]; then
cd "$R"
UNET_SWEEP_BYTES=1 timeout -k 10 500 python3 tools/size_sweep.py 20 28 40 46 60 64 65 74 80 81 96 100 112 113 128 160 208 256 400 512 576 577 800 1024 1040 1499 1500 2048 4096 9000 65535 > "$OUT/size_sweep.jsonl" 2> "$OUT/size_sweep.err" || exit 3
UNET_SWEEP_L2=1 UNET_SWEEP_BYTES=1 timeout -k 10 300 python3 tools/size_sweep.py 60 64 74 100 128 590 1514 > "$OUT/size_sweep_l2.jsonl" 2>> "$OUT/size_sweep.err" || exit 3
echo "size sweeps done"
timeout -k 10 500 python3 tools/tx_sweep.py > "$OUT/tx_sweep.jsonl" 2> "$OUT/tx_sweep.err" || exit 3
TX_SWEEP_LINE_ALIGNED=1 timeout -k 10 500 python3 tools/tx_sweep.py 1000 1472 2000 3000 4000 6000 8972 20000 65507 > "$OUT/tx_sweep_lines_fill.jsonl" 2>> "$OUT/tx_sweep.err" || exit 3
echo "tx sweeps done"
timeout -k 10 300 python3 tools/split_latency.py both > "$OUT/split_latency.jsonl" 2> "$OUT/split_latency.err" || exit 3
timeout -k 10 500 python3 tools/batch_sweep.py > "$OUT/batch_sweep.jsonl" 2> "$OUT/batch_sweep.err" || exit 3
echo "batch sweep done"
fi
if [ "$part" = sizes ] || [ "$part" = all ]; then
cd "$R"
bash tools/pmc_sizes.sh $tag 20 40 64 65 100 128 256 > "$OUT/pmc_sizes.log" 2>&1 || exit 4
python3 tools/pmc_sizes.py "$R/gpurun_out/pmc_sizes_$tag" 5 20 40 64 65 100 128 256 > "$OUT/pmc_short_records.jsonl" 2>> "$OUT/pmc_sizes.log" || exit 4
echo "pmc sizes done"
fi
