#!/usr/bin/env python3
"""Cost of one call as a function of batch size: the caller this offload replaces (the drain loop of
src/protocols.rs:83-97) hands over a few to a few thousand frames at a time, not millions.

For n in {1, 16, 64, 256, 1024, 4096, 65536, 1048576} records of 64 / 1500 / 9000 bytes: microseconds per call of
  unet_verify         host pointers (staging, H2D, kernel, D2H, wait)
  unet_verify_static  the same through the static device scratch (n <= 65536, arena <= 8 MiB)
  unet_verify_async   two batches in flight, per-call cost in steady state
  unet_verify_dev     device pointers, launch + wait (no copies)
  oracle (1 core)     the CPU restatement of the reference on the same records
and the batch size from which each GPU path beats the single CPU core.  Also BASELINE configs[0]: one 98-byte
ICMP echo frame, CPU nanoseconds vs offload microseconds, and unet_csum (one buffer per call).
Prints one JSON object per line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import unet_amd
from oracle import oracle_ffi as O

SEED = 0x756E6574


def make_batch(n, size, ifc):
    """n valid IPv4/UDP packets of `size` bytes (the UDP-1500 trace's layout rules, other length)."""
    rng = np.random.default_rng(size * 1000003 + n)
    slots = (size + 15) // 16
    arena = rng.integers(0, 256, size=n * slots * 16, dtype=np.uint8).reshape(n, slots * 16)
    import struct
    hdr = bytearray(struct.pack(">BBHHHBBHII", 0x45, 0, size, 0, 0x4000, 64, 17, 0, 0x0A000001, ifc.unicast))
    from helpers import py_ck
    struct.pack_into(">H", hdr, 10, py_ck(bytes(hdr)))
    arena[:, :20] = np.frombuffer(bytes(hdr), dtype=np.uint8)
    arena[:, 20:28] = np.frombuffer(struct.pack(">HHHH", 7, 9, size - 20, 0), dtype=np.uint8)
    flat = arena.reshape(-1)
    lens = np.full(n, size, dtype=np.uint32)
    # the UDP checksum is left as 0: the verdict is DROP_UDP_CSUM for (almost) all -- the work is the same
    return flat, slots, lens


def timed(fn, min_s=0.25, min_reps=5):
    fn()
    reps, t0 = 0, time.perf_counter()
    while True:
        fn()
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= min_s and reps >= min_reps:
            return dt / reps * 1e6  # microseconds per call


def main():
    ifc = unet_amd.Iface()
    ctx = unet_amd.Offload(0)
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    sizes = [64, 1500, 9000]
    counts = [1, 16, 64, 256, 1024, 4096, 65536, 1 << 20]
    for size in sizes:
        cross = {}
        for n in counts:
            if n * ((size + 15) // 16 * 16) > (3 << 30):
                continue
            arena, stride16, lens = make_batch(n, size, ifc)
            row = {"record_bytes": size, "n": n}
            exp = O.recv_batch(arena, None, lens, ifc, stride16=stride16)
            got = ctx.verify(arena, None, lens, ifc, stride16=stride16)
            assert np.array_equal(got, exp)
            row["unet_verify_us"] = timed(lambda: ctx.verify(arena, None, lens, ifc, stride16=stride16))
            if n <= 65536 and arena.nbytes <= (8 << 20):
                row["unet_verify_static_us"] = timed(lambda: ctx.verify_static(arena, None, lens, ifc, stride16=stride16))
            if arena.nbytes <= (1 << 30):
                pend = []

                def piped():
                    pend.append(ctx.verify_async(arena, None, lens, ifc, stride16=stride16))
                    if len(pend) == 2:
                        t, out = pend.pop(0)  # `out` must stay alive until the wait has filled it
                        ctx.wait(t)
                row["unet_verify_async_us"] = timed(piped)
                while pend:
                    t, out = pend.pop(0)
                    ctx.wait(t)
            d_arena = torch.from_numpy(arena).cuda()
            d_len = torch.from_numpy(lens.view(np.int32)).cuda()
            d_out = torch.zeros(n, dtype=torch.int32, device="cuda")

            def dev():
                ctx.verify_dev(d_arena.data_ptr(), arena.nbytes, None, stride16, d_len.data_ptr(), n, ifc, d_out.data_ptr(), st.cuda_stream)
                st.synchronize()
            row["unet_verify_dev_us"] = timed(dev)
            row["oracle_1core_us"] = timed(lambda: O.recv_batch(arena, None, lens, ifc, stride16=stride16), min_s=0.25, min_reps=2)
            for k in ("unet_verify_us", "unet_verify_static_us", "unet_verify_async_us", "unet_verify_dev_us"):
                if k in row and row[k] < row["oracle_1core_us"] and k not in cross:
                    cross[k] = n
            print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
            del d_arena, d_len, d_out
        print(json.dumps({"record_bytes": size, "first_n_where_gpu_call_beats_one_cpu_core": cross}), flush=True)

    # BASELINE configs[0]: one 98-byte Ethernet frame carrying a 64-byte ICMP echo
    from helpers import config1_echo_frame
    frame, tap_mac, _ = config1_echo_frame()
    tap = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=tap_mac, mode=1)
    a, o, l = unet_amd.pack_arena([frame])
    cpu_us = timed(lambda: O.recv(frame, tap), min_s=0.2, min_reps=1000)
    batch_us = timed(lambda: O.recv_batch(a, o, l, tap), min_s=0.2, min_reps=1000)
    print(json.dumps({"config": "one 98-byte ICMP echo frame (BASELINE configs[0])",
                      "oracle_ns_per_frame_via_ctypes": round(min(cpu_us, batch_us) * 1e3, 1),
                      "unet_verify_us": round(timed(lambda: ctx.verify(a, o, l, tap)), 2),
                      "unet_verify_static_us": round(timed(lambda: ctx.verify_static(a, o, l, tap)), 2),
                      "note": "the CPU figure includes the ctypes call (about 1 us); the checksum arithmetic itself is tens of ns"}), flush=True)
    data = bytes(range(256)) * 5 + bytes(220)
    print(json.dumps({"config": "calculate_checksum on one 1500-byte buffer", "oracle_us_via_ctypes": round(timed(lambda: O.calculate_checksum(data, 0), min_reps=1000), 3),
                      "unet_csum_us": round(timed(lambda: ctx.calculate_checksum(data, 0)), 2)}), flush=True)
    # transmit side, host pointers: the echo replies / datagrams of one drained queue
    from unet_amd._ffi import TX_DESC_DTYPE, tx_out_layout
    lo = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=tap_mac, mode=1)
    for n, size in ((1, 56), (16, 56), (256, 56), (16, 1472), (256, 1472), (4096, 1472)):
        rng = np.random.default_rng(n + size)
        payloads = [rng.integers(0, 256, size=size, dtype=np.uint8).tobytes() for _ in range(n)]
        pa, po, pl = unet_amd.pack_arena(payloads)
        desc = np.zeros(n, dtype=TX_DESC_DTYPE)
        desc["src"] = lo.unicast
        desc["dst"] = 0xC0000201
        desc["proto"] = np.where(np.arange(n) % 2 == 0, 1, 17)
        desc["l4a"] = 0x0800
        ooff, obytes = tx_out_layout(pl, lo.mode)
        print(json.dumps({"tx_build": f"{n} payloads of {size} bytes -> Ethernet frames",
                          "unet_tx_build_us": round(timed(lambda: ctx.tx_build(pa, po, pl, desc, lo)), 2),
                          "oracle_1core_us": round(timed(lambda: O.tx_build_batch(pa, po, pl, desc, lo, ooff, obytes), min_s=0.2, min_reps=2), 2)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
