"""A few long records judged one wavefront per 4 KiB segment (unet_verify_split_dev, and unet_verify on such a batch)
against the CPU oracle and against the ordinary kernel, through the C ABI."""
import struct

import numpy as np
import pytest

import unet_amd
from unet_amd import Verdict, _ffi
from helpers import py_ck
from oracle import oracle_ffi as O

pytestmark = pytest.mark.gpu


def _long_packets(rng, lengths, ifc):
    """UDP / ICMP / TCP packets (frames in L2 mode) of the given total lengths, every fourth damaged in one byte."""
    l2 = (bytes(ifc.mac) + b"\x02\x00\x00\x00\x00\x09\x08\x00") if ifc.mode else b""
    out = []
    for i, total in enumerate(lengths):
        l3 = total - len(l2)
        proto = (17, 6, 1)[i % 3]
        if l3 < 40:
            out.append(rng.integers(0, 256, size=total, dtype=np.uint8).tobytes())
            continue
        src = int(rng.integers(1, 2**32 - 1))
        body = rng.integers(0, 256, size=l3 - 20, dtype=np.uint8).tobytes()
        if proto == 17:
            seg = bytearray(struct.pack(">HHHH", 7, 9, (l3 - 20) & 0xFFFF, 0) + body[8:])
            struct.pack_into(">H", seg, 6, O.udp_send_checksum(src, ifc.unicast, 7, 9, bytes(seg[8:])))
        elif proto == 1:
            seg = bytearray(struct.pack(">BBHI", 8, 0, 0, 0x01020304) + body[8:])
            struct.pack_into(">H", seg, 2, O.icmp_send_checksum(8, 0, 0x01020304, bytes(seg[8:])))
        else:
            seg = bytearray(struct.pack(">HHIIBBHHH", 1, 2, 3, 4, 5 << 4, 0x10, 512, 0, 0) + body[20:])
            pseudo = struct.pack(">IIBBH", src, ifc.unicast, 0, 6, len(seg) & 0xFFFF)
            struct.pack_into(">H", seg, 16, py_ck(bytes(seg), ~py_ck(pseudo, 0) & 0xFFFF))
        pkt = bytearray(l2 + O.create_ip_header(i & 0xFFFF, proto, src, ifc.unicast, len(seg)) + bytes(seg))
        if i % 4 == 3:
            pkt[int(rng.integers(0, len(pkt)))] ^= 1 << int(rng.integers(0, 8))
        out.append(bytes(pkt))
    return out


def _split(offload, arena, off16, lens, ifc, max_len, stride16=0):
    import torch
    n = len(lens)
    d_arena = torch.from_numpy(arena).cuda()
    d_off = None if off16 is None else torch.from_numpy(off16.view(np.int32)).cuda()
    d_len = torch.from_numpy(lens.view(np.int32)).cuda()
    d_out = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    need = _ffi.verify_split_workspace(n, max_len)
    d_ws = torch.full((need // 4 + 1,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    offload.verify_split_dev(d_arena.data_ptr(), arena.nbytes, None if d_off is None else d_off.data_ptr(), stride16, d_len.data_ptr(), n, max_len, ifc,
                             d_out.data_ptr(), d_ws.data_ptr(), need)
    offload.sync()
    return d_out.cpu().numpy().view(unet_amd.RESULT_DTYPE).reshape(-1)


@pytest.mark.parametrize("mode", [unet_amd.MODE_L3, unet_amd.MODE_L2])
def test_segments_add_up_to_the_oracles_verdicts(offload, mode):
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode, ext_tcp=True)
    rng = np.random.default_rng(5 + mode)
    # every segment count from 1 to 16, lengths on and around the 4 KiB segment boundaries, odd lengths, runts
    lengths = [65535, 65534, 61441, 61440, 61439, 40960, 40961, 20000, 16385, 16384, 16383, 12289, 9000, 8193, 8192, 8191, 4097, 4096, 4095, 1500,
               64, 40, 21, 0] + [int(x) for x in rng.integers(0, 65536, size=40)]
    frames = _long_packets(rng, lengths, ifc)
    arena, off16, lens = unet_amd.pack_arena(frames, pad_byte=0x5C)
    exp = O.recv_batch(arena, off16, lens, ifc)
    got = _split(offload, arena, off16, lens, ifc, 65535)
    bad = np.flatnonzero(got != exp)
    assert bad.size == 0, (bad[:5], got[bad[:5]], exp[bad[:5]], lens[bad[:5]])
    assert {Verdict.ACCEPT_UDP, Verdict.ACCEPT_TCP, Verdict.ACCEPT_ICMP_ECHO} <= set(int(v) for v in got["verdict"])
    # the host-pointer call takes the same route for such a batch (few records, long ones) and must agree
    assert np.array_equal(offload.verify(arena, off16, lens, ifc), exp)
    sel = np.flatnonzero(lens >= 16384)[:16]
    a2, o2, l2 = unet_amd.pack_arena([frames[i] for i in sel])
    assert np.array_equal(offload.verify(a2, o2, l2, ifc), exp[sel])
    # fixed stride, one record
    one = unet_amd.pack_arena([frames[0]])
    assert np.array_equal(_split(offload, one[0], None, one[2], ifc, 65535, stride16=4096), exp[:1])


def test_records_longer_than_announced_are_left_alone_and_arguments(offload):
    ifc = unet_amd.Iface(ext_tcp=True)
    rng = np.random.default_rng(9)
    frames = _long_packets(rng, [9000, 4096, 30000, 100], ifc)
    arena, off16, lens = unet_amd.pack_arena(frames)
    exp = O.recv_batch(arena, off16, lens, ifc)
    got = _split(offload, arena, off16, lens, ifc, 9000)   # three segments per record
    assert np.array_equal(got[[0, 1, 3]], exp[[0, 1, 3]]) and int(got["verdict"][2]) == 0
    assert _ffi.verify_split_workspace(4, 9000) == 4 * 3 * 4 and _ffi.verify_split_workspace(0, 0) == 4
    import torch
    d = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    with pytest.raises(_ffi.UnetError) as e:
        offload.verify_split_dev(d.data_ptr(), 1 << 16, None, 4, d.data_ptr(), 4, 9000, ifc, d.data_ptr(), d.data_ptr(), 16)
    assert e.value.code == -5 and "unet_verify_split_workspace" in str(e.value)
    with pytest.raises(_ffi.UnetError):
        offload.verify_split_dev(d.data_ptr(), 1 << 16, None, 4, d.data_ptr(), 4, 70000, ifc, d.data_ptr(), d.data_ptr(), 1 << 16)
