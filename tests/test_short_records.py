"""Batches made only of short records (0..112 bytes): the kernels stage such a tile's byte range in LDS and
sum every record out of LDS.  Every layout that can or cannot take that route -- packed, fixed stride, offsets
in descending order, offsets with small and with large gaps, a tile cut off by the end of the batch -- against
the CPU oracle, for verdicts (L3 and L2), calculate_checksum and the transmit-side fill."""
import struct

import numpy as np
import pytest

import unet_amd
from unet_amd import Verdict
from helpers import py_ck
from oracle import oracle_ffi as O

pytestmark = pytest.mark.gpu


def _packets(rng, n, ifc, max_total=112):
    """Valid UDP / ICMP / TCP packets of 28..max_total bytes (TCP from 40), a third damaged in one byte, plus
    runts of 0..27 bytes."""
    out = []
    for i in range(n):
        proto = (17, 1, 6)[i % 3]
        lo = 40 if proto == 6 else 28
        total = int(rng.integers(lo, max_total + 1))
        src = int(rng.integers(1, 2**32 - 1))
        body = rng.integers(0, 256, size=total - 20, dtype=np.uint8).tobytes()
        if proto == 17:
            seg = bytearray(struct.pack(">HHHH", 7, 9, total - 20, 0) + body[8:])
            struct.pack_into(">H", seg, 6, O.udp_send_checksum(src, ifc.unicast, 7, 9, bytes(seg[8:])))
        elif proto == 1:
            ty = 8 if i % 2 else 0
            seg = bytearray(struct.pack(">BBHI", ty, 0, 0, 0x11223344) + body[8:])
            struct.pack_into(">H", seg, 2, O.icmp_send_checksum(ty, 0, 0x11223344, bytes(seg[8:])))
        else:
            seg = bytearray(struct.pack(">HHIIBBHHH", 1, 2, 3, 4, 5 << 4, 0x10, 512, 0, 0) + body[20:])
            pseudo = struct.pack(">IIBBH", src, ifc.unicast, 0, 6, len(seg))
            struct.pack_into(">H", seg, 16, py_ck(bytes(seg), ~py_ck(pseudo, 0) & 0xFFFF))
        pkt = bytearray(O.create_ip_header(i & 0xFFFF, proto, src, ifc.unicast, len(seg)) + bytes(seg))
        if i % 3 == 2 - (i // 3) % 3:
            pkt[int(rng.integers(0, len(pkt)))] ^= 1 << int(rng.integers(0, 8))
        if i % 17 == 5:
            pkt = pkt[:int(rng.integers(0, 28))]
        l2 = (bytes(ifc.mac) + b"\x02\x00\x00\x00\x00\x09\x08\x00") if ifc.mode else b""
        out.append((l2 + bytes(pkt))[:max_total])
    return out


def _layouts(rng, frames):
    """(name, arena, off16 or None, stride16, lens) for the same frames in several arena layouts."""
    lens = np.array([len(f) for f in frames], dtype=np.uint32)
    n = len(frames)
    slots = (lens.astype(np.int64) + 15) // 16

    def place(off):
        size = int((off + slots).max()) * 16 if n else 16
        arena = rng.integers(0, 256, size=max(size, 16), dtype=np.uint8)  # pad and gap bytes are garbage
        for f, o in zip(frames, off):
            arena[int(o) * 16:int(o) * 16 + len(f)] = np.frombuffer(f, dtype=np.uint8)
        return arena

    packed = np.concatenate([[0], np.cumsum(slots)[:-1]]) if n else np.zeros(0, dtype=np.int64)
    yield "packed", place(packed), packed.astype(np.uint32), 0, lens
    stride = int(slots.max()) if n else 1
    fixed = np.arange(n, dtype=np.int64) * stride
    yield "stride", place(fixed), None, stride, lens
    yield "stride+1", place(np.arange(n, dtype=np.int64) * (stride + 1)), None, stride + 1, lens
    rev = (int(packed[-1] + slots[-1]) - packed - slots) if n else packed
    yield "descending", place(rev), rev.astype(np.uint32), 0, lens
    gaps = packed + np.arange(n) // 5  # one free slot after every fifth record
    yield "small gaps", place(gaps), gaps.astype(np.uint32), 0, lens
    far = packed + (np.arange(n) % 64 == 13) * 4096  # one record of every tile far away: the tile does not qualify
    yield "one far record per tile", place(far), far.astype(np.uint32), 0, lens
    perm = rng.permutation(n)  # packed, but in another order than the tables list them
    yield "shuffled", place_shuffled(rng, frames, perm, slots), off_shuffled(perm, slots), 0, lens


def off_shuffled(perm, slots):
    """Record perm[k] is the k-th record in memory."""
    off = np.zeros(len(perm), dtype=np.int64)
    at = 0
    for k in perm:
        off[k] = at
        at += int(slots[k])
    return off.astype(np.uint32)


def place_shuffled(rng, frames, perm, slots):
    off = off_shuffled(perm, slots)
    size = int(sum(int(s) for s in slots)) * 16
    arena = rng.integers(0, 256, size=max(size, 16), dtype=np.uint8)
    for f, o in zip(frames, off):
        arena[int(o) * 16:int(o) * 16 + len(f)] = np.frombuffer(f, dtype=np.uint8)
    return arena


@pytest.mark.parametrize("mode", [unet_amd.MODE_L3, unet_amd.MODE_L2])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4133])
def test_short_record_tiles_verify(offload, mode, n):
    rng = np.random.default_rng(1000 * mode + n)
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode, ext_tcp=True)
    frames = _packets(rng, n, ifc)
    seen = set()
    for name, arena, off16, stride16, lens in _layouts(rng, frames):
        got = offload.verify(arena, off16, lens, ifc, stride16=stride16)
        exp = O.recv_batch(arena, off16, lens, ifc, stride16=stride16)
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, (name, bad[:5], got[bad[:5]], exp[bad[:5]], lens[bad[:5]])
        seen |= set(got["verdict"].tolist())
    if n >= 1000:
        for v in (Verdict.ACCEPT_UDP, Verdict.ACCEPT_ICMP_ECHO, Verdict.ACCEPT_ICMP_ECHOREPLY, Verdict.ACCEPT_TCP, Verdict.DROP_UDP_CSUM,
                  Verdict.DROP_L3_TRUNCATED):
            assert int(v) in seen, v


@pytest.mark.parametrize("max_len", [16, 20, 64, 80, 112, 113, 128])
def test_short_record_tiles_checksum(offload, max_len):
    rng = np.random.default_rng(max_len)
    n = 3000
    frames = [rng.integers(0, 256, size=int(l), dtype=np.uint8).tobytes() for l in rng.integers(0, max_len + 1, size=n)]
    seeds = rng.integers(0, 65536, size=n).astype(np.uint16)
    for name, arena, off16, stride16, lens in _layouts(rng, frames):
        got = offload.csum_batch(arena, off16, lens, seeds, stride16=stride16)
        exp = O.csum_batch(arena, off16, lens, seeds, stride16=stride16)
        assert (got == exp).all(), (name, np.nonzero(got != exp)[0][:5])
    # uniform lengths, as a size sweep uses them
    for L in (1, 15, 16, 17, 20, 40, 64, 65, 81, 100, 112):
        arena = rng.integers(0, 256, size=n * 7 * 16, dtype=np.uint8)
        lens = np.full(n, L, dtype=np.uint32)
        s16 = (L + 15) // 16
        assert (offload.csum_batch(arena, None, lens, None, stride16=s16) == O.csum_batch(arena, None, lens, None, stride16=s16)).all(), L


def test_short_record_tiles_generate(offload):
    rng = np.random.default_rng(77)
    ifc = unet_amd.Iface()
    frames = _packets(rng, 3000, ifc)
    for name, arena, off16, stride16, lens in _layouts(rng, frames):
        got_arena, got_status = offload.generate(arena, off16, lens, stride16=stride16)
        exp_arena, exp_status = O.generate_batch(arena, off16, lens, stride16=stride16)
        assert (got_status == exp_status).all(), name
        assert (got_arena == exp_arena).all(), name


def test_short_record_tiles_inside_a_batch_of_long_records(offload):
    """A batch whose mean length makes the library take few records per wave (16, 8, 4, 2 or 1), with runs of short
    records in it: tiles made only of short records stage their byte range in an LDS area sized for that tile
    shape (112 chunks with 16 records per wave, where two full rows would be 128)."""
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=unet_amd.MODE_L3, ext_tcp=True)
    for long_len, per_block, blocks in ((30000, 2, 25), (65535, 4, 25), (9000, 30, 25), (3000, 60, 25), (20000, 3, 600)):
        rng = np.random.default_rng(long_len)
        frames = []
        short = _packets(rng, 48, ifc, max_total=112)
        for block in range(blocks):
            frames += short if blocks > 100 else _packets(rng, 48, ifc, max_total=112)
            frames += [rng.integers(0, 256, size=long_len, dtype=np.uint8).tobytes() for _ in range(per_block)]
        arena, off16, lens = unet_amd.pack_arena(frames)
        assert lens.mean() > 1024
        for _ in range(3):  # (a neighbouring wave's stores are a race: look more than once)
            got = offload.verify(arena, off16, lens, ifc)
            exp = O.recv_batch(arena, off16, lens, ifc)
            bad = np.nonzero(got != exp)[0]
            assert bad.size == 0, (long_len, bad[:5], got[bad[:5]], exp[bad[:5]], lens[bad[:5]])
            seeds = rng.integers(0, 65536, size=len(lens)).astype(np.uint16)
            assert (offload.csum_batch(arena, off16, lens, seeds) == O.csum_batch(arena, off16, lens, seeds)).all(), long_len
