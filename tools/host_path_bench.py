#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer entry points on a large batch: unet_verify() (pageable host arena -> pinned
staging -> H2D -> kernel -> D2H of results) and unet_tx_build() (payloads up, whole packets back).  PCIe-inclusive
figures for the record; never the benchmark's `value`."""
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
                          "accepted": int((res["verdict"] == 20).sum())}), flush=True)
        # transmit side: n payloads of 1472 bytes from pageable memory, n packets of 1500 bytes back into pageable memory
        from unet_amd._ffi import TX_DESC_DTYPE
        rng = np.random.default_rng(1)
        plen = np.full(n, 1472, dtype=np.uint32)
        pay = rng.integers(0, 256, size=n * 1472, dtype=np.uint8)
        desc = np.zeros(n, dtype=TX_DESC_DTYPE)
        desc["src"] = ifc.unicast
        desc["dst"] = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
        desc["l4a"], desc["l4b"], desc["proto"] = 7, 9, 17
        desc["ip_id"] = np.arange(n) & 0xffff
        # (the C entry point itself, into buffers that exist already: a fresh 394 MB numpy array per call would be timed
        # for its page faults)
        import ctypes as C
        from unet_amd._ffi import _aligned_u8, _check, _np_ptr
        need = C.c_uint64(0)
        _check(ctx.lib.unet_tx_layout(_np_ptr(plen), n, int(ifc.mode), None, C.byref(need)))
        out = _aligned_u8(need.value)
        out[:] = 0
        ooff, olen, st = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint8)
        ic = ifc.c()

        def build():
            _check(ctx.lib.unet_tx_build(ctx.h, _np_ptr(pay), pay.nbytes, None, 92, _np_ptr(plen), desc.ctypes.data, n, C.byref(ic), _np_ptr(out),
                                         out.nbytes, _np_ptr(ooff), _np_ptr(olen), _np_ptr(st)))
        build()
        t0 = time.perf_counter()
        for _ in range(reps):
            build()
        dt = (time.perf_counter() - t0) / reps
        assert (st == 0).all() and (olen == 1500).all()
        # the same from and into pinned memory of the caller's own (torch's pinned tensors: hipHostMalloc)
        import torch
        unreg_dt = dt
        t_pay, t_out = torch.from_numpy(pay).pin_memory(), torch.empty(out.nbytes, dtype=torch.uint8).pin_memory()
        pay, out = t_pay.numpy(), t_out.numpy()
        build()
        t0 = time.perf_counter()
        for _ in range(reps):
            build()
        reg_dt = (time.perf_counter() - t0) / reps
        t_arena = torch.from_numpy(arena).pin_memory()
        p_arena = t_arena.numpy()
        ctx.verify(p_arena, off16, lens, ifc)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.verify(p_arena, off16, lens, ifc)
        reg_verify_dt = (time.perf_counter() - t0) / reps
        dt = unreg_dt
        print(json.dumps({"entry_point": "unet_tx_build (host pointers)", "packets": n, "payload_bytes": int(pay.nbytes), "packet_bytes": int(out.nbytes),
                          "seconds_per_call": dt, "Gpkt_s": n / dt / 1e9, "GB_s_incl_pcie_both_ways": (pay.nbytes + out.nbytes) / dt / 1e9}), flush=True)

        print(json.dumps({"entry_point": "unet_tx_build, both arenas pinned by the caller", "packets": n, "seconds_per_call": reg_dt,
                          "GB_s_incl_pcie_both_ways": (pay.nbytes + out.nbytes) / reg_dt / 1e9}), flush=True)
        print(json.dumps({"entry_point": "unet_verify, arena pinned by the caller", "records": n, "seconds_per_call": reg_verify_dt,
                          "GB_s_incl_pcie": arena.nbytes / reg_verify_dt / 1e9}), flush=True)


if __name__ == "__main__":
    main()
