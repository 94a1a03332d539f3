#!/usr/bin/env python3
"""Copies what `tools/profile_all.sh <tag>` left under gpurun_out/profile_<tag>/ into profiles/<tag>/ and rebuilds
profiles/traffic.json (the per-launch HBM bytes bench.py reports as roofline.traffic, each tied to the digest of
the kernel sources the passes ran on).  Source and destination are both named by the tag, so a later round can
never pick up an older round's files; a file is only replaced by a newer one.
usage: tools/collect_profiles.py <tag>"""
import csv
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
P, D = f"gpurun_out/profile_{tag}", os.path.join("profiles", tag)
os.makedirs(D, exist_ok=True)


def put(src, dst):
    """copy unless the destination is newer than the source"""
    if not os.path.exists(src):
        return False
    if os.path.exists(dst) and os.path.getmtime(dst) > os.path.getmtime(src):
        print(f"kept {dst}: newer than {src}")
        return False
    shutil.copy2(src, dst)
    return True


kern = {"udp1500": "k_verify", "imix": "k_verify", "odd": "k_verify", "hdr20": "k_ipv4_validate_rows", "csum": "k_csum", "generate": "k_generate",
        "txbuild": "k_tx_build_wave", "tximix": "k_tx_build<"}
t = {}
try:
    t = json.load(open("profiles/traffic.json"))
except (OSError, ValueError):
    pass
for w, k in kern.items():
    if not os.path.exists(f"{P}/bench_{w}.json") or not os.path.exists(f"{P}/{w}_fetch/f_counter_collection.csv"):
        continue
    put(f"{P}/{w}_stats/s_kernel_stats.csv", f"{D}/{w}_kernel_stats.csv")
    put(f"{P}/bench_{w}.json", f"{D}/bench_{w}_1gpu.json")
    j = json.load(open(f"{P}/bench_{w}.json"))
    ab, n, alg = j["config"]["arena_bytes_rank0"], j["config"]["rank0_frames"], j["roofline"]["algorithmic_bytes_per_launch"]
    out = subprocess.check_output([sys.executable, "tools/pmc_traffic.py", f"{P}/{w}_fetch/f_counter_collection.csv",
                                   f"{P}/{w}_write/w_counter_collection.csv", k, str(ab)]).decode()
    open(f"{D}/{w}_traffic.json", "w").write(out)
    r = json.loads(out)
    extra = "" if w == "hdr20" else " --rotate 1"
    t[w] = {"frames": n, "kernel": j["roofline"]["kernel"], "traffic_bytes": r["traffic_bytes"], "read_bytes": r["read_bytes"],
            "write_bytes": r["write_bytes"], "kernels_sha256": j["roofline"]["kernels_sha256"],
            "source": f"profiles/{tag}/{w}_traffic.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `python3 bench.py "
                      f"--workload {w} --steps 3 --warmup 1{extra}`; KiB units, FETCH_SIZE doubled per the gfx950 calibration on k_stream_read)"}
    # the bench line was printed before this pass existed: give it the traffic of the same session's counter passes
    # (same kernel digest), and say so
    if j["roofline"].get("traffic") is None:
        j["roofline"]["traffic"] = r["traffic_bytes"]
        j["roofline"]["traffic_source"] = t[w]["source"] + "; joined to this line by tools/collect_profiles.py, same run of tools/profile_all.sh, same kernel digest"
        json.dump(j, open(f"{D}/bench_{w}_1gpu.json", "w"))
    rows = [r for r in csv.DictReader(open(f"{P}/{w}_stats/s_kernel_stats.csv")) if k in r["Name"]] or list(csv.DictReader(open(f"{P}/{w}_stats/s_kernel_stats.csv")))
    ro = j["roofline"]
    print(w, rows[0]["Name"][:36], rows[0]["Calls"], round(float(rows[0]["AverageNs"]) / 1e3, 1), "us | bench ms", round(ro["kernel_ms"], 4), "GB/s",
          round(ro["achieved"]), "frac", round(ro["frac"], 3), "value", round(j["value"], 3), "traffic/alg", round(r["traffic_bytes"] / alg, 4),
          "parity", j.get("parity_check"))
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
put(f"{P}/tcptrack_stats/s_kernel_stats.csv", f"{D}/tcptrack_kernel_stats.csv")
for src, dst in (("bench_fields.json", "bench_fields_1gpu.json"), ("bench_txbuild_packed.json", "bench_txbuild_packed_layout_1gpu.json"),
                 ("bench_udp1500_nccl_world1.json", "bench_udp1500_nccl_world1.json"), ("bench_tcptrack.json", "bench_tcptrack_1gpu.json"),
                 ("size_sweep_l2.jsonl", "size_sweep_verify_csum_l2.jsonl"), ("tx_sweep_lines_fill.jsonl", "size_sweep_tx_build_lines_fill.jsonl"),
                 ("split_latency.jsonl", "split_latency_few_long_records.jsonl"),
                 ("bench_eth1514.json", "bench_eth1514_l2_1gpu.json"), ("bench_udp1500_with_cpu.json", "bench_udp1500_with_cpu_baseline.json"),
                 ("bench_udp1500_emulated_rank3of8.json", "bench_udp1500_emulated_rank3of8.json"),
                 ("bench_trace_emulated_rank2of8_4chunks.json", "bench_trace_emulated_rank2of8_4chunks.json"),
                 ("bench_trace_8chunks.json", "bench_trace_8chunks_1gpu.json"), ("host_path_pcie_inclusive.json", "host_path_pcie_inclusive.json"),
                 ("size_sweep.jsonl", "size_sweep_verify_csum.jsonl"), ("tx_sweep.jsonl", "size_sweep_tx_build.jsonl"),
                 ("batch_sweep.jsonl", "batch_size_sweep.jsonl"), ("pmc_short_records.jsonl", "pmc_short_records.jsonl")):
    if put(f"{P}/{src}", f"{D}/{dst}") and src.startswith("bench_"):
        j = json.load(open(f"{P}/{src}"))
        print(src, round(j["value"], 3), j["unit"], round(j["roofline"]["achieved"]), "GB/s", "frac", round(j["roofline"]["frac"], 3))
