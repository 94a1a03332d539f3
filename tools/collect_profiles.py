#!/usr/bin/env python3
"""Copies what tools/profile_all.sh left under gpurun_out/profile/ into profiles/<round>/ and rebuilds
profiles/traffic.json (the per-launch HBM bytes bench.py reports as roofline.traffic).
usage: tools/collect_profiles.py r1"""
import csv
import json
import os
import shutil
import subprocess
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r1"
P, D = "gpurun_out/profile", os.path.join("profiles", rnd)
os.makedirs(D, exist_ok=True)
kern = {"udp1500": "k_verify", "imix": "k_verify", "hdr20": "k_ipv4_validate_rows", "csum": "k_csum", "generate": "k_generate",
        "txbuild": "k_tx_build"}
t = {}
for w, k in kern.items():
    if not os.path.exists(f"{P}/bench_{w}.json"):
        continue
    shutil.copy(f"{P}/{w}_stats/s_kernel_stats.csv", f"{D}/{w}_kernel_stats.csv")
    shutil.copy(f"{P}/bench_{w}.json", f"{D}/bench_{w}_1gpu.json")
    j = json.load(open(f"{P}/bench_{w}.json"))
    ab, n, alg = j["config"]["arena_bytes_rank0"], j["config"]["rank0_frames"], j["roofline"]["algorithmic_bytes_per_launch"]
    out = subprocess.check_output([sys.executable, "tools/pmc_traffic.py", f"{P}/{w}_fetch/f_counter_collection.csv",
                                   f"{P}/{w}_write/w_counter_collection.csv", k, str(ab)]).decode()
    open(f"{D}/{w}_traffic.json", "w").write(out)
    r = json.loads(out)
    extra = "" if w == "hdr20" else " --rotate 1"
    t[w] = {"frames": n, "kernel": j["roofline"]["kernel"], "traffic_bytes": r["traffic_bytes"], "read_bytes": r["read_bytes"],
            "write_bytes": r["write_bytes"],
            "source": f"profiles/{rnd}/{w}_traffic.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `python3 bench.py "
                      f"--workload {w} --steps 3 --warmup 1{extra}`; KiB units, FETCH_SIZE doubled per the gfx950 calibration on k_stream_read)"}
    rows = list(csv.DictReader(open(f"{P}/{w}_stats/s_kernel_stats.csv")))
    ro = j["roofline"]
    print(w, rows[0]["Name"][:36], rows[0]["Calls"], round(float(rows[0]["AverageNs"]) / 1e3, 1), "us | bench ms", round(ro["kernel_ms"], 4), "GB/s",
          round(ro["achieved"]), "frac", round(ro["frac"], 3), "value", round(j["value"], 3), "traffic/alg", round(r["traffic_bytes"] / alg, 4),
          "parity", j.get("parity_check"))
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
for src, dst in (("bench_trace.json", "bench_trace_2p30_1gpu.json"), ("bench_eth1514.json", "bench_eth1514_l2_1gpu.json"),
                 ("size_sweep.jsonl", "size_sweep_verify_csum.jsonl"), ("tx_sweep.jsonl", "size_sweep_tx_build.jsonl")):
    if os.path.exists(f"{P}/{src}"):
        shutil.copy(f"{P}/{src}", f"{D}/{dst}")
        if src.endswith(".json"):
            j = json.load(open(f"{P}/{src}"))
            print(src, round(j["value"], 3), j["unit"], round(j["roofline"]["achieved"]), "GB/s", "frac", round(j["roofline"]["frac"], 3))
