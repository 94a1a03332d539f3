#!/usr/bin/env python3
"""Prints the by-batch-size latency tables (markdown) from profiles/<tag>/batch_size_sweep.jsonl.
usage: tools/design_tables.py <tag> [--write]    --write replaces the two tables in the design document in place"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", tag, "batch_size_sweep.jsonl"))]


def f(v):
    return "—" if v is None else (f"{v:.1f}" if v < 100 else f"{v:.0f}")


t1 = ["| records × bytes | `unet_verify` | `unet_verify_static` | `unet_verify_async` (2 in flight) | `unet_verify_dev` + sync | oracle, 1 core |",
      "|---|---|---|---|---|---|"]
t2 = ["| `unet_tx_build` (host pointers, Ethernet frames out) | µs per call | oracle, 1 core |", "|---|---|---|"]
for r in rows:
    if "record_bytes" in r and "n" in r:
        t1.append(f"| {r['n']} × {r['record_bytes']} | {f(r.get('unet_verify_us'))} | {f(r.get('unet_verify_static_us'))} | "
                  f"{f(r.get('unet_verify_async_us'))} | {f(r.get('unet_verify_dev_us'))} | {f(r.get('oracle_1core_us'))} |")
    elif "tx_build" in r:
        t2.append(f"| {r['tx_build'].replace(' -> Ethernet frames', '')} | {f(r['unet_tx_build_us'])} | {f(r['oracle_1core_us'])} |")
text = "\n".join(t1) + "\n\n" + "\n".join(t2) + "\n\n"
if "--write" in sys.argv:
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    i = s.index(t1[0])
    k = s.index("A host-pointer call used to cost")
    open(p, "w").write(s[:i] + text + s[k:])
else:
    print(text)
