#!/usr/bin/env python3
"""Per record size: HBM traffic over algorithmic bytes and instructions per record for k_verify / k_csum, from the
counter CSVs tools/pmc_sizes.sh leaves.  Launches are attributed to sizes by dispatch order (size_sweep.py runs, per
size, 2 warm-up + UNET_SWEEP_ITERS timed launches of verify, then the same of csum).
usage: pmc_sizes.py <dir> <launches per op per size> <size>..."""
import csv
import glob
import json
import sys
from collections import defaultdict


def rows(d, name):
    f = glob.glob(f"{d}/{name}/**/*counter_collection.csv", recursive=True)
    if not f:
        return []
    return sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Dispatch_Id"]))


def per_size(d, name, kernel, per, sizes):
    """counter -> list (one entry per size) of the mean over that size's launches"""
    byc = defaultdict(list)
    for r in rows(d, name):
        if kernel in r["Kernel_Name"]:
            byc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for c, v in byc.items():
        if len(v) != per * len(sizes):
            out[c] = None
            continue
        out[c] = [sum(v[i * per:(i + 1) * per]) / per for i in range(len(sizes))]
    return out


def main():
    d, per = sys.argv[1], int(sys.argv[2])
    sizes = [int(x) for x in sys.argv[3:]]
    sweep = [json.loads(l) for l in open(f"{d}/fetch.jsonl") if l.startswith("{")]
    recs = {r["len"]: r["records"] for r in sweep}
    for kernel, extra in (("k_verify", 8), ("k_csum", 6)):
        c = {}
        for name in ("fetch", "write", "sq", "sq2"):
            c.update(per_size(d, name, kernel, per, sizes))
        for i, L in enumerate(sizes):
            n = recs[L]
            alg = n * (L + extra)
            slot = (L + 15) // 16 * 16
            row = {"kernel": kernel, "len": L, "records": n, "algorithmic_bytes": alg,
                   "layout_bytes": n * (slot + extra)}  # what a 16-byte-aligned arena makes any reader fetch
            if c.get("FETCH_SIZE") and c.get("WRITE_SIZE"):
                rd, wr = 2 * c["FETCH_SIZE"][i] * 1024, c["WRITE_SIZE"][i] * 1024
                row.update(read_bytes=rd, write_bytes=wr, traffic_over_algorithmic=(rd + wr) / alg,
                           traffic_over_layout=(rd + wr) / row["layout_bytes"])
            for k in sorted(c):
                if k.startswith("SQ_") and c[k]:
                    row[k + "_per_record"] = c[k][i] / n
            print(json.dumps(row))


if __name__ == "__main__":
    main()
