#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs into HBM bytes per launch.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE counts 128-byte read requests at 64 bytes, i.e. reports exactly half of a wide coalesced
streaming read, so it is doubled; WRITE_SIZE is exact for dword/dwordx4 streaming stores.  The
correction is calibrated in the same run on k_stream_read, whose byte count is known.

usage: pmc_traffic.py <fetch.csv> <write.csv> <kernel substring> [known_stream_bytes]
"""
import csv
import json
import sys


def mean_counter(path, kernel, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_csv, write_csv, kernel = sys.argv[1:4]
    f, nf = mean_counter(fetch_csv, kernel, "FETCH_SIZE")
    w, nw = mean_counter(write_csv, kernel, "WRITE_SIZE")
    out = {"kernel": kernel, "launches_fetch": nf, "launches_write": nw,
           "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
           "read_bytes": 2 * f * 1024, "write_bytes": w * 1024, "traffic_bytes": 2 * f * 1024 + w * 1024}
    if len(sys.argv) > 4:
        sf, _ = mean_counter(fetch_csv, "k_stream_read", "FETCH_SIZE")
        out["calibration"] = {"kernel": "k_stream_read", "known_bytes": int(sys.argv[4]),
                              "FETCH_SIZE_KiB_raw": sf, "corrected_over_known": 2 * sf * 1024 / int(sys.argv[4])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
