#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point unet_verify(): pageable host arena -> pinned
staging -> H2D -> kernel -> D2H of results.  A PCIe-inclusive figure for the record; never the benchmark's `value`."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import unet_amd

SEED = 0x756E6574


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
    ifc = unet_amd.Iface()
    arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_UDP1500, SEED, 0, n, ifc)
    with unet_amd.Offload(0) as ctx:
        ctx.verify(arena, off16, lens, ifc)  # warm: allocations, pinned buffers
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            res = ctx.verify(arena, off16, lens, ifc)
        dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"entry_point": "unet_verify (host pointers)", "records": n, "arena_bytes": int(arena.nbytes),
                      "seconds_per_call": dt, "Gpkt_s": n / dt / 1e9, "GB_s_incl_pcie": arena.nbytes / dt / 1e9,
                      "accepted": int((res["verdict"] == 20).sum())}))


if __name__ == "__main__":
    main()
