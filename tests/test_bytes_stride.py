"""unet_verify_bytes_dev: records of one length at a byte stride (no 16-byte slots, no tables) against the CPU oracle,
through the C ABI: every length from 0 to 256 in L3 and L2 mode, strides equal to and above the length (so records
start on every byte alignment), batch sizes that end inside a wave's group of records, and the argument checks."""
import struct

import numpy as np
import pytest

import unet_amd
from unet_amd import Verdict
from helpers import py_ck
from oracle import oracle_ffi as O

pytestmark = pytest.mark.gpu


def _fixed_packets(rng, n, ifc, total):
    """n records of exactly `total` bytes: valid UDP / ICMP / TCP packets (frames in L2 mode) where the length allows
    one, a third of them damaged in one byte; random bytes where it does not."""
    l2 = (bytes(ifc.mac) + b"\x02\x00\x00\x00\x00\x09\x08\x00") if ifc.mode else b""
    l3 = total - len(l2)
    out = []
    for i in range(n):
        proto = (17, 1, 6)[i % 3]
        if l3 < (40 if proto == 6 else 28):
            out.append(rng.integers(0, 256, size=total, dtype=np.uint8).tobytes())
            continue
        src = int(rng.integers(1, 2**32 - 1))
        dst = ifc.unicast if i % 11 else 0x0A000001  # some for another host
        body = rng.integers(0, 256, size=l3 - 20, dtype=np.uint8).tobytes()
        if proto == 17:
            seg = bytearray(struct.pack(">HHHH", 7, 9, l3 - 20, 0) + body[8:])
            struct.pack_into(">H", seg, 6, O.udp_send_checksum(src, dst, 7, 9, bytes(seg[8:])))
        elif proto == 1:
            ty = 8 if i % 2 else 0
            seg = bytearray(struct.pack(">BBHI", ty, 0, 0, 0x11223344) + body[8:])
            struct.pack_into(">H", seg, 2, O.icmp_send_checksum(ty, 0, 0x11223344, bytes(seg[8:])))
        else:
            seg = bytearray(struct.pack(">HHIIBBHHH", 1, 2, 3, 4, 5 << 4, 0x10, 512, 0, 0) + body[20:])
            pseudo = struct.pack(">IIBBH", src, dst, 0, 6, len(seg))
            struct.pack_into(">H", seg, 16, py_ck(bytes(seg), ~py_ck(pseudo, 0) & 0xFFFF))
        pkt = bytearray(l2 + O.create_ip_header(i & 0xFFFF, proto, src, dst, len(seg)) + bytes(seg))
        if i % 3 == 2 - (i // 3) % 3:
            pkt[int(rng.integers(0, len(pkt)))] ^= 1 << int(rng.integers(0, 8))
        assert len(pkt) == total
        out.append(bytes(pkt))
    return out


def _run(offload, frames, length, stride, ifc, lead_garbage=True):
    import torch
    n = len(frames)
    rng = np.random.default_rng(length * 1000 + stride)
    nbytes = ((n - 1) * stride + length + 15) // 16 * 16 if n else 16
    buf = rng.integers(0, 256, size=max(nbytes, 16) + 64, dtype=np.uint8)  # gaps between records and the tail are garbage
    for i, f in enumerate(frames):
        buf[i * stride:i * stride + length] = np.frombuffer(f, dtype=np.uint8)
    d_buf = torch.from_numpy(buf).cuda()
    d_out = torch.full((max(n, 1),), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    offload.verify_bytes_dev(d_buf.data_ptr(), max(nbytes, 16), stride, length, n, ifc, d_out.data_ptr())
    offload.sync()
    got = d_out.cpu().numpy().view(unet_amd.RESULT_DTYPE).reshape(-1)[:n]
    # the oracle on the same bytes, in the slot layout it understands
    arena, off16, lens = unet_amd.pack_arena(frames)
    exp = O.recv_batch(arena, off16, lens, ifc)
    return got, exp


@pytest.mark.parametrize("mode", [unet_amd.MODE_L3, unet_amd.MODE_L2])
def test_every_length_at_its_own_stride(offload, mode):
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode, ext_tcp=True)
    rng = np.random.default_rng(7 + mode)
    seen = set()
    for length in list(range(0, 130)) + [143, 144, 145, 160, 200, 255, 256]:
        if length == 0:
            continue
        n = int(rng.integers(1, 700))
        frames = _fixed_packets(rng, n, ifc, length)
        got, exp = _run(offload, frames, length, length, ifc)
        bad = np.flatnonzero(got != exp)
        assert bad.size == 0, (length, n, bad[:5], got[bad[:5]], exp[bad[:5]])
        seen |= set(int(v) for v in got["verdict"])
    assert {Verdict.ACCEPT_UDP, Verdict.ACCEPT_TCP, Verdict.ACCEPT_ICMP_ECHO, Verdict.DROP_UDP_CSUM, Verdict.IGNORE_L3_NOT_FOR_ME,
            Verdict.DROP_L2_ETHERTYPE if mode else Verdict.DROP_L3_TRUNCATED} <= seen


@pytest.mark.parametrize("mode", [unet_amd.MODE_L3, unet_amd.MODE_L2])
def test_strides_above_the_length_and_batch_ends(offload, mode):
    """Strides that put records on every byte alignment (65, 67, 100, 113 ...), with gaps of garbage between them,
    and batches that end in the middle of a wave's group (a wave takes 4096 / stride records, its lanes several each)."""
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode, ext_tcp=False)
    rng = np.random.default_rng(11 + mode)
    for length, stride in [(20, 20), (28, 28), (40, 41), (46, 46), (60, 60), (60, 64), (65, 65), (65, 67), (74, 74), (81, 81), (100, 100), (113, 113),
                           (113, 128), (128, 129), (200, 255), (256, 256), (1, 1), (13, 256)]:
        per = 4096 // stride
        for n in (1, per - 1, per, per + 1, 4 * per, 4 * per + 3, 1000):
            if n <= 0:
                continue
            frames = _fixed_packets(rng, n, ifc, length)
            got, exp = _run(offload, frames, length, stride, ifc)
            bad = np.flatnonzero(got != exp)
            assert bad.size == 0, (length, stride, n, bad[:5], got[bad[:5]], exp[bad[:5]])


def test_a_large_batch_equals_the_slot_kernel(offload):
    """2,000,000 minimum-size frames (60 bytes, L2 mode): byte stride against the same frames in 16-byte slots."""
    import torch
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=1)
    rng = np.random.default_rng(3)
    base = _fixed_packets(rng, 4096, ifc, 60)
    n = 2_000_000
    pick = rng.integers(0, len(base), size=n)
    table = np.frombuffer(b"".join(base), dtype=np.uint8).reshape(len(base), 60)
    packed = table[pick]                                     # n x 60, byte stride 60
    slots = np.zeros((n, 64), dtype=np.uint8)
    slots[:, :60] = packed
    d_a = torch.from_numpy(np.concatenate([packed.reshape(-1), np.zeros(32, np.uint8)])).cuda()
    d_b = torch.from_numpy(slots.reshape(-1)).cuda()
    d_len = torch.full((n,), 60, dtype=torch.int32, device="cuda")
    o_a = torch.zeros(n, dtype=torch.int32, device="cuda")
    o_b = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    offload.verify_bytes_dev(d_a.data_ptr(), (n * 60 + 15) // 16 * 16, 60, 60, n, ifc, o_a.data_ptr())
    offload.verify_dev(d_b.data_ptr(), n * 64, None, 4, d_len.data_ptr(), n, ifc, o_b.data_ptr())
    offload.sync()
    assert torch.equal(o_a, o_b)
    exp = O.recv_batch(*unet_amd.pack_arena(base), ifc)
    assert np.array_equal(o_a.cpu().numpy().view(unet_amd.RESULT_DTYPE).reshape(-1)[:50000], exp[pick[:50000]])


def test_arguments(offload):
    import torch
    ifc = unet_amd.Iface()
    d = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    o = torch.zeros(64, dtype=torch.int32, device="cuda")
    for stride, length, n, nbytes, code in [(0, 0, 1, 4096, -5), (257, 20, 1, 4096, -5), (20, 21, 1, 4096, -5), (20, 20, 64, 1264, -5),
                                            (20, 20, 64, 1280, 0), (256, 256, 16, 4096, 0), (256, 256, 17, 4096, -5)]:
        if code:
            with pytest.raises(unet_amd.UnetError) as e:
                offload.verify_bytes_dev(d.data_ptr(), nbytes, stride, length, n, ifc, o.data_ptr())
            assert e.value.code == code, (stride, length, n)
        else:
            offload.verify_bytes_dev(d.data_ptr(), nbytes, stride, length, n, ifc, o.data_ptr())
    with pytest.raises(unet_amd.UnetError):
        offload.verify_bytes_dev(d.data_ptr() + 4, 2048, 20, 20, 10, ifc, o.data_ptr())
    offload.sync()
