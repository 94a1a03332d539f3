#!/usr/bin/env python3
"""Bandwidth of k_tx_build as a function of payload size (UDP, L3 and L2 output), HIP-event timing."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import unet_amd
from unet_amd._ffi import TX_DESC_DTYPE, tx_frame_len

sizes = [int(x) for x in sys.argv[1:]] or [18, 64, 128, 256, 548, 1000, 1472, 2000, 3000, 4000, 6000, 8972, 20000, 65507]
ctx = unet_amd.Offload(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
total = 2 << 30
buf = torch.randint(0, 2**31 - 1, (total // 4 + 16,), dtype=torch.int32, device="cuda").view(torch.uint8)
for mode in (0, 1):
    ifc = unet_amd.Iface(mode=mode)
    for L in sizes:
        slots = max(1, (L + 15) // 16)
        n = min(total // (slots * 16), 1 << 24)
        flen = int(tx_frame_len(L, mode))
        oslots = (flen + 15) // 16
        if os.environ.get("TX_SWEEP_LINE_ALIGNED"):
            oslots = (oslots + 7) // 8 * 8  # every packet starts on a 128-byte line
        desc = np.zeros(n, dtype=TX_DESC_DTYPE)
        desc["src"], desc["dst"], desc["proto"], desc["l4a"], desc["l4b"] = ifc.unicast, 0xC6336407, 17, 1234, 4321
        desc["ip_id"] = np.arange(n) & 0xFFFF
        d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1)).cuda()
        d_plen = torch.full((n,), L, dtype=torch.int32, device="cuda")
        d_ooff = (torch.arange(n, dtype=torch.int64, device="cuda") * oslots).to(torch.int32)
        d_out = torch.empty(n * oslots * 16 + 16, dtype=torch.uint8, device="cuda")
        d_olen = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")

        flags = 1 if os.environ.get("TX_SWEEP_LINE_ALIGNED") else 0  # UNET_TX_FILL_LINES: the slots are the caller's up to the line's end

        def run():
            ctx.tx_build_dev(buf.data_ptr(), n * slots * 16, None, slots, d_plen.data_ptr(), d_desc.data_ptr(), n, ifc, d_out.data_ptr(),
                             n * oslots * 16, d_ooff.data_ptr(), d_olen.data_ptr(), d_st.data_ptr(), st.cuda_stream, flags=flags)
        # warm up for at least 40 ms (the descriptor set-up above leaves the GPU idle long enough for its clocks to drop:
        # three launches measured 5-10 % low at some sizes), then the median of seven groups of four launches
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.04:
            run()
            torch.cuda.synchronize()
        groups = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run()
            e1.record()
            torch.cuda.synchronize()
            groups.append(e0.elapsed_time(e1) / 4)
        ms = statistics.median(groups)
        assert int(d_st.max().item()) == 0 and int(d_olen.min().item()) == flen
        print(json.dumps({"mode": "L2" if mode else "L3", "layout": "lines+fill" if flags else "packed", "payload": L, "packet": flen, "packets": n,
                          "GBps": round(n * (L + flen + 41) / ms / 1e6, 1), "Gpkt_s": round(n / ms / 1e6, 3)}), flush=True)
        del d_out
