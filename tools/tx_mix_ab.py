#!/usr/bin/env python3
"""unet_tx_build_dev on a packed batch whose payload lengths are drawn from a weighted list, per configuration of the
environment knobs (see tx_ab.py), inside one process on the same buffers.
usage: tx_mix_ab.py '<name>=K:V,...;...' 'len:weight,len:weight,...' [packets]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import unet_amd
from unet_amd._ffi import TX_DESC_DTYPE, tx_frame_len, tx_out_layout

KNOBS = ("UNET_TX_WAVE", "UNET_TXW_GROUP", "UNET_TXW_BLOCK", "UNET_TXW_POLICY", "UNET_TX_TILE", "UNET_TX_STAGE")
configs = []
for item in sys.argv[1].split(";"):
    name, _, kv = item.partition("=")
    configs.append((name, dict(x.split(":") for x in kv.split(",") if x)))
mix = [(int(a), float(b)) for a, b in (x.split(":") for x in sys.argv[2].split(","))]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 22
ctxs = []
for name, env in configs:
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    ctxs.append(unet_amd.Offload(0))
rng = np.random.default_rng(5)
lens = rng.choice([m[0] for m in mix], size=n, p=np.array([m[1] for m in mix]) / sum(m[1] for m in mix)).astype(np.uint32)
slots = (lens.astype(np.int64) + 15) // 16
off16 = (np.cumsum(slots) - slots).astype(np.uint32)
arena_bytes = int(slots.sum()) * 16
ifc = unet_amd.Iface(mode=0)
ooff, obytes = tx_out_layout(lens, 0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
buf = torch.randint(0, 2**31 - 1, (arena_bytes // 4 + 16,), dtype=torch.int32, device="cuda").view(torch.uint8)
desc = np.zeros(n, dtype=TX_DESC_DTYPE)
desc["src"], desc["dst"], desc["proto"], desc["l4a"], desc["l4b"] = ifc.unicast, 0xC6336407, 17, 1234, 4321
d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1)).cuda()
d_plen, d_off, d_ooff = torch.from_numpy(lens.view(np.int32)).cuda(), torch.from_numpy(off16.view(np.int32)).cuda(), torch.from_numpy(ooff.view(np.int32)).cuda()
d_out = torch.empty(obytes + 16, dtype=torch.uint8, device="cuda")
d_olen = torch.zeros(n, dtype=torch.int32, device="cuda")
d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")


def run(ctx):
    ctx.tx_build_dev(buf.data_ptr(), arena_bytes, d_off.data_ptr(), 0, d_plen.data_ptr(), d_desc.data_ptr(), n, ifc, d_out.data_ptr(), obytes,
                     d_ooff.data_ptr(), d_olen.data_ptr(), d_st.data_ptr(), st.cuda_stream, flags=0)


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
assert int(d_st.max().item()) == 0
alg = float(lens.astype(np.int64).sum() + tx_frame_len(lens, 0).astype(np.int64).sum() + 45 * n)  # + length, offset, descriptor, out offset, out length, status
row = {"mix": sys.argv[2], "packets": n, "mean_payload": round(float(lens.mean()), 1)}
for (name, _), t in zip(configs, times):
    row[name] = round(alg / statistics.median(t) / 1e6)
print(json.dumps(row), flush=True)
