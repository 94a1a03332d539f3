#!/usr/bin/env python3
"""Bandwidth of k_verify / k_csum as a function of record size: packed arenas of equal-length random
records (about 4 GiB each), HIP-event timing on the launch stream.  UNET_SWEEP_L2=1 judges the records as
Ethernet frames (MODE_L2) instead of L3 packets."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import unet_amd

sizes = [int(x) for x in sys.argv[1:]] or [40, 64, 128, 144, 256, 576, 1024, 1040, 1500, 2048, 4096, 9000, 65535]
ctx = unet_amd.Offload(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
l2 = bool(int(os.environ.get("UNET_SWEEP_L2", "0")))
ifc = unet_amd.Iface(ext_tcp=True, mode=1 if l2 else 0)
total = int(os.environ.get("UNET_SWEEP_TOTAL", 4 << 30))
iters = int(os.environ.get("UNET_SWEEP_ITERS", 10))
buf = torch.randint(0, 2**31 - 1, (total // 4 + 16,), dtype=torch.int32, device="cuda").view(torch.uint8)
out = []
for L in sizes:
    slots = (L + 15) // 16
    n = min(total // (slots * 16), 1 << 26)
    d_len = torch.full((n,), L, dtype=torch.int32, device="cuda")
    d_out = torch.zeros(n, dtype=torch.int32, device="cuda")
    nbytes = n * slots * 16
    row = {"len": L, "records": n, "mode": "L2" if l2 else "L3"}
    names = ("verify", "csum")
    if os.environ.get("UNET_SWEEP_BYTES") and L <= 256:
        names = ("verify", "csum", "verify_bytes")  # the same records at a byte stride of L: no slots, no tables
    for name in names:
        def run():
            if name == "verify_bytes":
                ctx.verify_bytes_dev(buf.data_ptr(), (n * L + 15) // 16 * 16, L, L, n, ifc, d_out.data_ptr(), st.cuda_stream)
            elif name == "verify":
                ctx.verify_dev(buf.data_ptr(), nbytes, None, slots, d_len.data_ptr(), n, ifc, d_out.data_ptr(), st.cuda_stream)
            else:
                ctx.csum_batch_dev(buf.data_ptr(), nbytes, None, slots, d_len.data_ptr(), None, n, d_out.data_ptr(), st.cuda_stream)
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        # algorithmic bytes: the record, its 4-byte result and -- where the layout needs one -- its 4-byte length
        row[name + "_GBps"] = round(n * (L + (4 if name == "verify_bytes" else 8)) / ms / 1e6, 1)
        row[name + "_Gpkt_s"] = round(n / ms / 1e6, 3)
    out.append(row)
    print(json.dumps(row), flush=True)
