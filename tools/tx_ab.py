#!/usr/bin/env python3
"""A/B of unet_tx_build_dev configurations inside one process, on the same buffers: every configuration is a set of
environment knobs read by unet_open (UNET_TX_WAVE, UNET_TXW_GROUP, UNET_TXW_BLOCK, UNET_TXW_POLICY, UNET_TX_TILE), so
one context per configuration; launches alternate between them for several rounds and the median per configuration is
reported -- runs of separate processes differ by a few per cent on one box (where the buffers land), which is more
than most knobs move.
usage: tx_ab.py '<name>=K1:V1,K2:V2;<name>=...' size [size ...]     (TX_AB_L2=1: Ethernet output, TX_AB_PACKED=1)"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import unet_amd
from unet_amd._ffi import TX_DESC_DTYPE, tx_frame_len

KNOBS = ("UNET_TX_WAVE", "UNET_TXW_GROUP", "UNET_TXW_BLOCK", "UNET_TXW_POLICY", "UNET_TX_TILE", "UNET_TX_STAGE")
configs = []
for item in sys.argv[1].split(";"):
    name, _, kv = item.partition("=")
    configs.append((name, dict(x.split(":") for x in kv.split(",") if x)))
sizes = [int(x) for x in sys.argv[2:]]
ctxs = []
for name, env in configs:
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    ctxs.append(unet_amd.Offload(0))
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
total = 2 << 30
buf = torch.randint(0, 2**31 - 1, (total // 4 + 16,), dtype=torch.int32, device="cuda").view(torch.uint8)
mode = 1 if os.environ.get("TX_AB_L2") else 0
packed = bool(os.environ.get("TX_AB_PACKED"))
ifc = unet_amd.Iface(mode=mode)
for L in sizes:
    slots = max(1, (L + 15) // 16)
    n = min(total // (slots * 16), 1 << 24)
    flen = int(tx_frame_len(L, mode))
    oslots = (flen + 15) // 16
    if not packed:
        oslots = (oslots + 7) // 8 * 8
    desc = np.zeros(n, dtype=TX_DESC_DTYPE)
    desc["src"], desc["dst"], desc["proto"], desc["l4a"], desc["l4b"] = ifc.unicast, 0xC6336407, 17, 1234, 4321
    desc["ip_id"] = np.arange(n) & 0xFFFF
    d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1)).cuda()
    d_plen = torch.full((n,), L, dtype=torch.int32, device="cuda")
    d_ooff = (torch.arange(n, dtype=torch.int64, device="cuda") * oslots).to(torch.int32)
    d_out = torch.empty(n * oslots * 16 + 16, dtype=torch.uint8, device="cuda")
    d_olen = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")
    flags = 0 if packed else 1

    def run(ctx):
        ctx.tx_build_dev(buf.data_ptr(), n * slots * 16, None, slots, d_plen.data_ptr(), d_desc.data_ptr(), n, ifc, d_out.data_ptr(),
                         n * oslots * 16, d_ooff.data_ptr(), d_olen.data_ptr(), d_st.data_ptr(), st.cuda_stream, flags=flags)
    times = [[] for _ in ctxs]
    for rnd in range(7):
        for ci, ctx in enumerate(ctxs):
            run(ctx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run(ctx)
            e1.record()
            torch.cuda.synchronize()
            times[ci].append(e0.elapsed_time(e1) / 4)
    assert int(d_st.max().item()) == 0 and int(d_olen.min().item()) == flen
    row = {"mode": "L2" if mode else "L3", "layout": "packed" if packed else "lines+fill", "payload": L, "packets": n}
    for (name, _), t in zip(configs, times):
        row[name] = round(n * (L + flen + 41) / statistics.median(t) / 1e6)
    print(json.dumps(row), flush=True)
    del d_out
