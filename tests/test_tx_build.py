"""Transmit path, whole packets: udp::send / icmp::send -> ipv4::send -> Ethernet header + tap::send padding
(src/transport/udp.rs:168-216, src/transport/icmp.rs:66-95, src/protocols/ipv4.rs:300-391,
src/driver/tap.rs:132-166).  CPU part: the oracle against an independent Python model and the reference's
own vectors.  GPU part: unet_tx_build_dev against the oracle, byte for byte."""
import struct

import numpy as np
import pytest

import unet_amd
from unet_amd import Verdict
from unet_amd._ffi import TX_DESC_DTYPE, TX_E_PROTO, TX_E_RANGE, TX_E_TOO_LONG, tx_frame_len, tx_out_layout
from helpers import as_tuples, config1_echo_frame, load, py_ck
from oracle import oracle_ffi as O

KATS = load("reference_kats.json")
LENS = [0, 1, 2, 3, 4, 5, 6, 7, 15, 16, 17, 18, 19, 20, 31, 32, 33, 35, 36, 37, 100, 1000, 1023, 1024, 1025, 1472, 8972]


def py_tx_build(d, payload: bytes, ifc) -> bytes:
    """Independent model, written from the reference's send functions with struct.pack."""
    if d["proto"] == 17:
        length = 8 + len(payload)
        pseudo = struct.pack(">IIBBH", int(d["src"]), int(d["dst"]), 0, 17, length)
        seg = bytearray(struct.pack(">HHHH", int(d["l4a"]), int(d["l4b"]), length, 0) + payload)
        s = py_ck(pseudo, 0)                                    # udp.rs:200
        struct.pack_into(">H", seg, 6, py_ck(bytes(seg), ~s & 0xFFFF))  # udp.rs:201-203
    else:
        seg = bytearray(struct.pack(">BBHI", int(d["l4a"]) >> 8, int(d["l4a"]) & 0xFF, 0, int(d["values"])) + payload)
        struct.pack_into(">H", seg, 2, py_ck(bytes(seg), 0))    # icmp.rs:81-85
    hdr = bytearray(struct.pack(">BBHHHBBHII", 0x45, 0, 20 + len(seg), int(d["ip_id"]), 0, 64, int(d["proto"]), 0,
                                ifc.unicast, int(d["dst"])))
    struct.pack_into(">H", hdr, 10, py_ck(bytes(hdr), 0))       # ipv4.rs:387-389
    pkt = bytes(hdr) + bytes(seg)
    if ifc.mode == unet_amd.MODE_L3:
        return pkt
    pad = b"\0" * (60 - len(pkt)) if len(pkt) < 60 else b""     # tap.rs:147-152
    return bytes(d["dst_mac"]) + bytes(ifc.mac) + b"\x08\x00" + pkt + pad


def make_batch(rng, lens, ifc, any_src_every=5):
    n = len(lens)
    desc = np.zeros(n, dtype=TX_DESC_DTYPE)
    desc["src"] = np.where(np.arange(n) % any_src_every == 0, 0, ifc.unicast)
    desc["dst"] = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    desc["l4a"] = rng.integers(0, 65536, size=n)
    desc["l4b"] = rng.integers(0, 65536, size=n)
    desc["values"] = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    desc["ip_id"] = (128 + np.arange(n)) & 0xFFFF               # Ipv4IdGenerator starts at 128 (ipv4.rs:286-296)
    desc["proto"] = np.where(np.arange(n) % 3 == 0, 1, 17)
    desc["dst_mac"] = rng.integers(0, 256, size=(n, 6), dtype=np.uint8)
    desc["reserved0"] = 0xA5                                    # must be ignored
    desc["reserved1"] = 0x5A
    payloads = [rng.integers(0, 256, size=int(l), dtype=np.uint8).tobytes() for l in lens]
    return desc, payloads


def frames_of(out, off16, out_len):
    return [out[16 * int(o):16 * int(o) + int(l)].tobytes() for o, l in zip(off16, out_len)]


# ------------------------------------------------------------------------------------------ CPU

@pytest.mark.parametrize("mode", [unet_amd.MODE_L3, unet_amd.MODE_L2])
def test_oracle_tx_build_matches_python_model(mode):
    rng = np.random.default_rng(5)
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode)
    lens = LENS + [int(x) for x in rng.integers(0, 200, size=150)]
    desc, payloads = make_batch(rng, lens, ifc)
    arena, off16, plen = unet_amd.pack_arena(payloads, pad_byte=0x77)
    ooff, obytes = tx_out_layout(plen, mode)
    out, out_len, status = O.tx_build_batch(arena, off16, plen, desc, ifc, ooff, obytes, fill=0xCD)
    assert (status == 0).all()
    assert (out_len == tx_frame_len(plen, mode)).all()
    for i, f in enumerate(frames_of(out, ooff, out_len)):
        assert f == py_tx_build(desc[i], payloads[i], ifc), i


def test_oracle_tx_build_l2_padding_quirk():
    """tap::send pads the L3 packet, not the frame, to 60 bytes: the shortest frame it emits is 74 bytes."""
    ifc = unet_amd.Iface(mode=unet_amd.MODE_L2)
    assert tx_frame_len([0, 31, 32, 33, 1472], 1).tolist() == [74, 74, 74, 75, 1514]
    assert tx_frame_len([0, 31, 32, 33, 1472], 0).tolist() == [28, 59, 60, 61, 1500]
    desc = np.zeros(1, dtype=TX_DESC_DTYPE)
    desc["proto"], desc["dst"], desc["src"] = 17, 0xC0000201, ifc.unicast
    arena, off16, plen = unet_amd.pack_arena([b"hi"])
    out, out_len, status = O.tx_build_batch(arena, off16, plen, desc, ifc, np.zeros(1, np.uint32), 80, fill=0xCD)
    assert out_len.tolist() == [74] and out[44:74].tolist() == [0] * 30 and out[42:44].tobytes() == b"hi"
    assert out[74] == 0xCD


def test_oracle_tx_build_reference_vectors():
    """The reference's ICMP test message (src/transport/icmp.rs:128-139) rebuilt by the send path carries
    the checksum the message was captured with; the echo reply of configs[0] equals the reply assembled
    from the separate oracle functions."""
    k = KATS["icmp_header"]
    msg = bytes.fromhex(k["message"])
    lo = unet_amd.Iface("127.0.0.1", "255.0.0.0")
    desc = np.zeros(1, dtype=TX_DESC_DTYPE)
    desc["proto"], desc["dst"], desc["src"] = 1, lo.unicast, lo.unicast
    desc["l4a"] = (msg[0] << 8) | msg[1]
    desc["values"] = struct.unpack(">I", msg[4:8])[0]
    arena, off16, plen = unet_amd.pack_arena([msg[8:]])
    out, out_len, status = O.tx_build_batch(arena, off16, plen, desc, lo, np.zeros(1, np.uint32), 16 * 8)
    assert status.tolist() == [0] and out[20:20 + len(msg)].tobytes() == msg
    assert as_tuples(O.recv_batch(out, np.zeros(1, np.uint32), out_len, lo))[0][0] in (Verdict.ACCEPT_ICMP_ECHO, Verdict.ACCEPT_ICMP_ECHOREPLY)

    frame, tap_mac, peer_mac = config1_echo_frame()
    tap = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=tap_mac, mode=unet_amd.MODE_L2)
    m = frame[34:]
    desc["proto"], desc["dst"], desc["src"] = 1, 0xC0000201, tap.unicast
    desc["l4a"], desc["values"], desc["ip_id"] = m[1], struct.unpack(">I", m[4:8])[0], 128
    desc["dst_mac"][0] = np.frombuffer(peer_mac, dtype=np.uint8)
    arena, off16, plen = unet_amd.pack_arena([m[8:]])
    out, out_len, status = O.tx_build_batch(arena, off16, plen, desc, tap, np.zeros(1, np.uint32), 16 * 8)
    icmp = bytearray(struct.pack(">BBHI", 0, m[1], 0, int(desc["values"][0])) + m[8:])
    struct.pack_into(">H", icmp, 2, O.icmp_send_checksum(0, m[1], int(desc["values"][0]), m[8:]))
    want = peer_mac + tap_mac + b"\x08\x00" + O.create_ip_header(128, 1, tap.unicast, 0xC0000201, len(icmp)) + bytes(icmp)
    assert out[:int(out_len[0])].tobytes() == want and len(want) == 98


def test_oracle_tx_build_status_codes():
    ifc = unet_amd.Iface()
    desc = np.zeros(3, dtype=TX_DESC_DTYPE)
    desc["proto"] = [6, 17, 17]
    payloads = [b"x" * 10, b"y" * 65508, b"z" * 65507]
    arena, off16, plen = unet_amd.pack_arena(payloads)
    ooff, obytes = tx_out_layout(plen, 0)
    out, out_len, status = O.tx_build_batch(arena, off16, plen, desc, ifc, ooff, obytes)
    assert status.tolist() == [TX_E_PROTO, TX_E_TOO_LONG, 0] and out_len.tolist() == [0, 0, 65535]


# ------------------------------------------------------------------------------------------ GPU

TX_KERNEL_KNOBS = {
    "by_mean_payload": {},
    "tile_kernel": {"UNET_TX_WAVE": "0"},
    "one_packet_per_wave": {"UNET_TX_WAVE": "2", "UNET_TXW_GROUP": "1"},
    "two_waves_per_packet": {"UNET_TX_WAVE": "2", "UNET_TXW_GROUP": "2"},
    "four_waves_per_packet": {"UNET_TX_WAVE": "2", "UNET_TXW_GROUP": "4"},
    "sixteen_waves_per_packet": {"UNET_TX_WAVE": "2", "UNET_TXW_GROUP": "16"},
}


@pytest.fixture(scope="module", params=list(TX_KERNEL_KNOBS))
def offload(request):
    """Every GPU test of this file once per way of building a batch: the library's own choice between its two transmit
    kernels (by mean payload length), the tile kernel for every batch, and one packet per wave or per group of 2, 4 or
    16 waves for every batch (the knobs are read when the context is opened)."""
    import os
    knobs = TX_KERNEL_KNOBS[request.param]
    old = {k: os.environ.get(k) for k in ("UNET_TX_WAVE", "UNET_TXW_GROUP")}
    for k in old:
        os.environ.pop(k, None)
    os.environ.update(knobs)
    try:
        ctx = unet_amd.Offload(0)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    yield ctx
    from conftest import close_checked
    close_checked(ctx)


def _gpu_build(offload, arena, off16, plen, desc, ifc, ooff, obytes, fill=0xAB, stride16=0, flags=0):
    import torch
    assert torch.cuda.is_available()
    st = torch.cuda.Stream()
    d_arena = torch.from_numpy(arena).cuda()
    d_off = None if off16 is None else torch.from_numpy(off16).cuda()
    d_len = torch.from_numpy(plen).cuda()
    d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).cuda()
    d_ooff = torch.from_numpy(ooff).cuda()
    d_out = torch.full((obytes + 256,), fill, dtype=torch.uint8, device="cuda")
    d_olen = torch.full((len(plen),), 0x7FFFFFFF, dtype=torch.int32, device="cuda")
    d_status = torch.full((len(plen),), 0xEE, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    offload.tx_build_dev(d_arena.data_ptr(), arena.nbytes, 0 if d_off is None else d_off.data_ptr(), stride16, d_len.data_ptr(),
                         d_desc.data_ptr(), len(plen), ifc, d_out.data_ptr(), obytes, d_ooff.data_ptr(), d_olen.data_ptr(),
                         d_status.data_ptr(), stream=st.cuda_stream, flags=flags)
    st.synchronize()
    return d_out.cpu().numpy(), d_olen.cpu().numpy().view(np.uint32), d_status.cpu().numpy()


def _valid_mask(nbytes, ooff, out_len):
    """True for the bytes that belong to some built packet."""
    delta = np.zeros(nbytes + 1, dtype=np.int32)
    np.add.at(delta, ooff.astype(np.int64) * 16, 1)
    np.add.at(delta, ooff.astype(np.int64) * 16 + out_len.astype(np.int64), -1)
    return np.cumsum(delta[:-1]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [unet_amd.MODE_L3, unet_amd.MODE_L2])
def test_tx_build_matches_oracle(offload, mode):
    rng = np.random.default_rng(17 + mode)
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode)
    # every length around the hand-over points of the kernel's paths (128/129 bytes, 61-62 chunks, the end of
    # rows 0/1 at 125-126 chunks, the row steps of long payloads), plus a random mix
    edges = list(range(100, 160)) + list(range(950, 1050)) + list(range(1980, 2080)) + list(range(3000, 3100, 7)) + \
        [4000, 4079, 4080, 4096, 5000, 6100, 6128, 6129]
    lens = LENS * 3 + edges + [int(x) for x in rng.integers(0, 1600, size=6000)] + [65507, 65506, 40000, 9000]
    desc, payloads = make_batch(rng, lens, ifc)
    arena, off16, plen = unet_amd.pack_arena(payloads, pad_byte=0x77)
    ooff, obytes = tx_out_layout(plen, mode)
    exp, exp_len, exp_status = O.tx_build_batch(arena, off16, plen, desc, ifc, ooff, obytes)
    got, got_len, got_status = _gpu_build(offload, arena, off16, plen, desc, ifc, ooff, obytes)
    assert (got_status == exp_status).all() and (got_status == 0).all()
    assert (got_len == exp_len).all()
    m = _valid_mask(obytes, ooff, exp_len)
    assert (got[:obytes][m] == exp[m]).all()
    # pad bytes up to each packet's 16-byte boundary are zeros; nothing is written past the last packet
    assert (got[:obytes][~m] == 0).all()
    assert (got[obytes:] == 0xAB).all()
    # a receiver with the destination's address accepts every packet whose pseudo-header used the real source
    real_src = np.flatnonzero(desc["src"] == ifc.unicast)[:300]
    for i in real_src:
        rx = unet_amd.Iface(int(desc["dst"][i]), "255.255.255.255", mac=bytes(desc["dst_mac"][i]), mode=mode)
        f = got[16 * int(ooff[i]):16 * int(ooff[i]) + int(got_len[i])].tobytes()
        v = O.recv(f, rx)[0]
        if (int(desc["dst"][i]) >> 24) and not (mode == 1 and plen[i] < 32 and desc["proto"][i] == 17):
            assert v in (Verdict.ACCEPT_UDP, Verdict.ACCEPT_ICMP_ECHO, Verdict.ACCEPT_ICMP_ECHOREPLY, Verdict.DROP_ICMP_TYPE), (i, v)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [unet_amd.MODE_L3, unet_amd.MODE_L2])
def test_tx_build_every_row_count_of_long_payloads(offload, mode):
    """Payloads past their first two rows are copied in rows of 64 chunks, four rows to a step: every number of
    rows from 0 to 13 with the last one holding 1, 2, 33, 63 or 64 chunks and ending on or off a 16-byte boundary."""
    rng = np.random.default_rng(29 + mode)
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode)
    lens = []
    for first_two in (125, 126):  # chunks of payload in rows 0/1 (Ethernet, raw IP output)
        for rows in range(0, 14):
            for last_row in (1, 2, 33, 63, 64):
                chunks = first_two + 64 * rows + last_row
                lens += [16 * chunks - d for d in (0, 1, 15)]
    lens = [l for l in lens if l <= 65507] + list(range(1990, 12400, 97)) + [65507, 65493, 30000]
    desc, payloads = make_batch(rng, lens, ifc)
    arena, off16, plen = unet_amd.pack_arena(payloads, pad_byte=0x77)
    ooff, obytes = tx_out_layout(plen, mode)
    exp, exp_len, exp_status = O.tx_build_batch(arena, off16, plen, desc, ifc, ooff, obytes)
    got, got_len, got_status = _gpu_build(offload, arena, off16, plen, desc, ifc, ooff, obytes)
    assert (got_status == exp_status).all() and (got_status == 0).all() and (got_len == exp_len).all()
    m = _valid_mask(obytes, ooff, exp_len)
    bad = np.flatnonzero(got[:obytes][m] != exp[m])
    assert bad.size == 0, (bad[:4], len(lens))
    assert (got[:obytes][~m] == 0).all() and (got[obytes:] == 0xAB).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [unet_amd.MODE_L3, unet_amd.MODE_L2])
def test_tx_build_into_a_line_aligned_layout(offload, mode):
    """Packets placed on 128-byte lines (unet_tx_layout_aligned, align16 = 8): the same bytes as the oracle's, and the
    slots' spare chunks are left as they were."""
    rng = np.random.default_rng(41 + mode)
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode)
    lens = LENS + [int(x) for x in rng.integers(0, 3000, size=1500)] + [4000, 6000, 20000, 65507]
    desc, payloads = make_batch(rng, lens, ifc)
    arena, off16, plen = unet_amd.pack_arena(payloads, pad_byte=0x77)
    ooff, obytes = tx_out_layout(plen, mode, align16=8)
    assert (ooff % 8 == 0).all()
    exp, exp_len, exp_status = O.tx_build_batch(arena, off16, plen, desc, ifc, ooff, obytes)
    got, got_len, got_status = _gpu_build(offload, arena, off16, plen, desc, ifc, ooff, obytes)
    assert (got_status == exp_status).all() and (got_status == 0).all() and (got_len == exp_len).all()
    for o, l in zip(ooff, exp_len):
        a, b = 16 * int(o), 16 * int(o) + int(l)
        assert (got[a:b] == exp[a:b]).all(), (int(o), int(l))
    # chunks of a slot that hold no packet bytes keep the buffer's fill (0xAB); the packet's own last chunk is zero-padded
    slots = (ooff.astype(np.int64) + (exp_len.astype(np.int64) + 15) // 16)
    nxt = np.append(ooff[1:].astype(np.int64), obytes // 16)
    for s0, s1 in zip(slots, nxt):
        assert (got[16 * int(s0):16 * int(s1)] == 0xAB).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [unet_amd.MODE_L3, unet_amd.MODE_L2])
@pytest.mark.parametrize("align16", [8, 1, 2])
def test_tx_build_may_fill_each_packets_last_line(offload, mode, align16):
    """UNET_TX_FILL_LINES: the packet bytes are the oracle's; from a packet's end to the next 128-byte line of the
    output arena every byte is either untouched or zero (the kernel may, but need not, write the line whole); no
    byte past that line changes, whatever the layout (on lines, packed, on 32-byte boundaries), and nothing past
    out_arena_bytes."""
    from unet_amd._ffi import TX_FILL_LINES
    rng = np.random.default_rng(43 + mode + align16)
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode)
    lens = LENS + [int(x) for x in rng.integers(0, 3000, size=1500)] + [1472] * 40 + [2000] * 20 + [4000, 6000, 8972, 20000, 65507]
    desc, payloads = make_batch(rng, lens, ifc)
    arena, off16, plen = unet_amd.pack_arena(payloads, pad_byte=0x77)
    if align16 == 8:
        ooff, obytes = tx_out_layout(plen, mode, align16=8)
    else:
        # slots a line apart or more, starting off the lines: the filled range ends at the line, not at the slot
        slots = ((tx_frame_len(plen, mode) + 15) // 16 + 7) // 8 * 8 + 8
        ooff = (np.cumsum(slots) - slots + align16).astype(np.uint32)
        obytes = int(slots.sum() + 8) * 16
    exp, exp_len, exp_status = O.tx_build_batch(arena, off16, plen, desc, ifc, ooff, obytes)
    got, got_len, got_status = _gpu_build(offload, arena, off16, plen, desc, ifc, ooff, obytes, flags=TX_FILL_LINES)
    assert (got_status == exp_status).all() and (got_status == 0).all() and (got_len == exp_len).all()
    touched = np.zeros(obytes, dtype=bool)
    filled = 0
    for o, l in zip(ooff, exp_len):
        a, b = 16 * int(o), 16 * int(o) + int(l)
        assert (got[a:b] == exp[a:b]).all(), (int(o), int(l))
        line_end = min((b + 127) // 128 * 128, obytes)
        tail = got[b:line_end]
        chunk_end = (b + 15) // 16 * 16
        assert (got[b:chunk_end] == 0).all()                     # the packet's own last chunk is zero-padded
        rest = got[chunk_end:line_end].reshape(-1, 16)
        assert ((rest == 0).all(axis=1) | (rest == 0xAB).all(axis=1)).all(), (int(o), int(l))
        filled += int((rest == 0).all(axis=1).sum())
        touched[a:line_end] = True
    assert (got[:obytes][~touched] == 0xAB).all() and (got[obytes:] == 0xAB).all()
    assert filled > 0  # the permission is used


def _check_against_oracle(got, got_len, got_status, exp, exp_len, exp_status, ooff, obytes, fill=0xAB):
    """Packet bytes equal the oracle's, the pad up to each packet's 16-byte boundary is zero, every other byte of the
    buffer (slots of refused packets, gaps, what lies in front of the first packet and behind out_arena_bytes) keeps its fill."""
    assert (got_status == exp_status).all() and (got_len == exp_len).all()
    ok = exp_status == 0
    m = _valid_mask(obytes, ooff[ok], exp_len[ok])
    bad = np.flatnonzero(got[:obytes][m] != exp[m])
    assert bad.size == 0, bad[:4]
    slots = _valid_mask(obytes, ooff[ok], (exp_len[ok] + 15) // 16 * 16)
    assert (got[:obytes][slots & ~m] == 0).all()
    assert (got[:obytes][~slots] == fill).all() and (got[obytes:] == fill).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [unet_amd.MODE_L3, unet_amd.MODE_L2])
def test_tx_build_tiles_of_short_packets_staged_in_lds(offload, mode):
    """Tiles whose packets lie back to back in the output arena are assembled in LDS and leave as whole rows: uniform
    lengths on both sides of every limit (one lane up to 48 or 80 bytes, groups of 4 / 8 / 16 lanes, what the staging
    area of 64-, 32-, 16- and 8-packet tiles holds: a tile of short packets that is longer stages its first packets and
    stores the rest directly), output ranges that start on every chunk of a 128-byte line, batches
    that end inside a tile -- and the tiles that must not take that route: a refused packet, a gap or a step back in
    the output offsets, a long payload among the short ones."""
    rng = np.random.default_rng(101 + mode)
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode)

    def run(lens, front=0, mutate=None, fixed_stride=False):
        desc, payloads = make_batch(rng, lens, ifc)
        if fixed_stride:  # (payloads of up to 1008 bytes at a fixed stride are staged whatever their mean length)
            stride16 = max(1, (max(lens) + 15) // 16)
            arena = rng.integers(0, 256, size=len(lens) * stride16 * 16, dtype=np.uint8)
            off16, plen = None, np.asarray(lens, dtype=np.uint32)
        else:
            stride16 = 0
            arena, off16, plen = unet_amd.pack_arena(payloads, pad_byte=0x77)
        ooff, obytes = tx_out_layout(plen, mode)
        ooff = ooff + np.uint32(front)
        obytes += 16 * front
        if mutate:
            ooff, obytes = mutate(desc, ooff, obytes)
        exp, exp_len, exp_status = O.tx_build_batch(arena, off16, plen, desc, ifc, ooff, obytes, pay_stride16=stride16)
        got, got_len, got_status = _gpu_build(offload, arena, off16, plen, desc, ifc, ooff, obytes, stride16=stride16)
        _check_against_oracle(got, got_len, got_status, exp, exp_len, exp_status, ooff, obytes)
        return got_status

    for i, L in enumerate([0, 1, 15, 16, 17, 32, 33, 47, 48, 49, 54, 55, 64, 65, 80, 81, 84, 85, 100, 104, 116, 117, 128, 200,
                           228, 229, 244, 245, 256, 320, 336, 337, 400, 548, 549, 564, 565, 580, 1000]):
        run([L] * (64 * 3 + 7 + i % 5), front=i % 8, fixed_stride=L > 320 or i % 2 == 1)
    for lo, hi, n in ((0, 49, 517), (0, 81, 518), (0, 90, 400), (20, 130, 300), (0, 250, 301), (100, 260, 203), (200, 565, 155), (330, 600, 99)):
        run([int(x) for x in rng.integers(lo, hi, size=n)], front=int(rng.integers(0, 8)), fixed_stride=hi > 320)

    def refuse_some(desc, ooff, obytes):
        desc["proto"][5::37] = 6
        return ooff, obytes

    def gap_and_step_back(desc, ooff, obytes):
        ooff = ooff.copy()
        k = len(ooff) // 2
        ooff[k:] += 9                                     # a gap in front of packet k
        tail = ooff[-1] + 40 - ooff[10]                   # packet 10 moves behind everything else: a step forth and back
        ooff[10] += tail
        return ooff, obytes + 16 * int(tail + 9 + 40)

    for mut in (refuse_some, gap_and_step_back):
        for L in (18, 64, 200, 400):
            st = run([L] * 333, front=3, mutate=mut, fixed_stride=L > 320)
            assert (st != 0).any() == (mut is refuse_some)
    # a long payload among short ones: its tile takes the ordinary route, the others are staged
    lens = [40] * 500
    lens[70], lens[300] = 1500, 9000
    run(lens, front=1)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [unet_amd.MODE_L3, unet_amd.MODE_L2])
def test_tx_build_max_length_hint_changes_the_route_not_the_bytes(offload, mode):
    """unet_hint_max_len: packed batches of mid-size payloads (an offset table, no stride) are staged in LDS only on the
    caller's word that none is long; a payload that breaks the word is built correctly all the same."""
    rng = np.random.default_rng(211 + mode)
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode)
    for lens in ([400] * 300, [548] * 211, [int(x) for x in rng.integers(330, 600, size=257)],
                 [500] * 100 + [3000] + [500] * 100):
        desc, payloads = make_batch(rng, lens, ifc)
        arena, off16, plen = unet_amd.pack_arena(payloads, pad_byte=0x77)
        ooff, obytes = tx_out_layout(plen, mode)
        ooff = ooff + np.uint32(5)
        obytes += 80
        exp, exp_len, exp_status = O.tx_build_batch(arena, off16, plen, desc, ifc, ooff, obytes)
        try:
            for hint in (0, 900, 64):
                offload.hint_max_len(hint)
                got, got_len, got_status = _gpu_build(offload, arena, off16, plen, desc, ifc, ooff, obytes)
                _check_against_oracle(got, got_len, got_status, exp, exp_len, exp_status, ooff, obytes)
        finally:
            offload.hint_max_len(0)


@pytest.mark.gpu
def test_tx_build_fixed_stride_and_status(offload):
    rng = np.random.default_rng(3)
    ifc = unet_amd.Iface()
    n, stride16 = 4099, 94
    plen = rng.integers(0, 1473, size=n).astype(np.uint32)
    arena = rng.integers(0, 256, size=n * stride16 * 16, dtype=np.uint8)
    desc, _ = make_batch(rng, [0] * n, ifc)
    desc["proto"][::7] = 6
    ooff, obytes = tx_out_layout(plen, 0)
    ooff[-1] += 4000  # the last packet would land outside the output arena
    exp, exp_len, exp_status = O.tx_build_batch(arena, None, plen, desc, ifc, np.minimum(ooff, ooff[-2]), obytes, pay_stride16=stride16)
    got, got_len, got_status = _gpu_build(offload, arena, None, plen, desc, ifc, ooff, obytes, stride16=stride16)
    exp_status[-1], exp_len[-1] = TX_E_RANGE, 0
    assert (got_status == exp_status).all() and set(got_status.tolist()) == {0, TX_E_PROTO, TX_E_RANGE}
    assert (got_len == exp_len).all()
    ok = got_status == 0
    m = _valid_mask(obytes, ooff[ok], exp_len[ok])
    ref, _, _ = O.tx_build_batch(arena, None, plen[:-1], desc[:-1], ifc, ooff[:-1], obytes, pay_stride16=stride16)
    assert (got[:obytes][m] == ref[m]).all()
    # refused packets leave their slots untouched
    bad = np.flatnonzero(got_status == TX_E_PROTO)[:50]
    for i in bad:
        assert (got[16 * int(ooff[i]):16 * int(ooff[i]) + 28 + int(plen[i])] == 0xAB).all()


@pytest.mark.gpu
def test_tx_build_too_long_and_config1_reply(offload):
    ifc = unet_amd.Iface()
    desc = np.zeros(2, dtype=TX_DESC_DTYPE)
    desc["proto"] = 17
    arena, off16, plen = unet_amd.pack_arena([b"y" * 65508, b"z" * 65507])
    ooff, obytes = tx_out_layout(plen, 0)
    got, got_len, got_status = _gpu_build(offload, arena, off16, plen, desc, ifc, ooff, obytes)
    assert got_status.tolist() == [TX_E_TOO_LONG, 0] and got_len.tolist() == [0, 65535]
    frame, tap_mac, peer_mac = config1_echo_frame()
    tap = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=tap_mac, mode=unet_amd.MODE_L2)
    m = frame[34:]
    d = np.zeros(1, dtype=TX_DESC_DTYPE)
    d["proto"], d["dst"], d["src"], d["l4a"], d["ip_id"] = 1, 0xC0000201, tap.unicast, m[1], 128
    d["values"] = struct.unpack(">I", m[4:8])[0]
    d["dst_mac"][0] = np.frombuffer(peer_mac, dtype=np.uint8)
    arena, off16, plen = unet_amd.pack_arena([m[8:]])
    got, got_len, got_status = _gpu_build(offload, arena, off16, plen, d, tap, np.zeros(1, np.uint32), 112)
    exp, exp_len, _ = O.tx_build_batch(arena, off16, plen, d, tap, np.zeros(1, np.uint32), 112)
    assert got_len.tolist() == [98] and got[:98].tobytes() == exp[:98].tobytes()
    peer = unet_amd.Iface("192.0.2.1", "255.255.255.0", mac=peer_mac, mode=unet_amd.MODE_L2)
    assert O.recv(got[:98].tobytes(), peer)[0] == Verdict.ACCEPT_ICMP_ECHOREPLY


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [unet_amd.MODE_L3, unet_amd.MODE_L2])
def test_tx_build_host_pointer_entry(offload, mode):
    rng = np.random.default_rng(23)
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode)
    lens = LENS + [int(x) for x in rng.integers(0, 3000, size=800)]
    desc, payloads = make_batch(rng, lens, ifc)
    desc["proto"][5] = 99
    arena, off16, plen = unet_amd.pack_arena(payloads)
    got, got_off, got_len, got_status = offload.tx_build(arena, off16, plen, desc, ifc)
    ooff, obytes = tx_out_layout(plen, mode)
    assert (got_off == ooff).all() and got.nbytes == obytes
    exp, exp_len, exp_status = O.tx_build_batch(arena, off16, plen, desc, ifc, ooff, obytes)
    assert (got_status == exp_status).all() and got_status[5] == TX_E_PROTO and (got_len == exp_len).all()
    m = _valid_mask(obytes, ooff, exp_len)
    assert (got[m] == exp[m]).all() and (got[~m] == 0).all()
    assert offload.tx_build(arena[:16], off16[:0], plen[:0], desc[:0], ifc)[2].size == 0
