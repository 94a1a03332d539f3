"""Capture-file import/export around the arena (host code of the C ABI).  The independent model is the
struct-based pcap writer/reader below; limits follow the reference's devices: tap::read's 1514-byte buffer
(src/driver/tap.rs:168-173) and the loopback MTU of 65535 (src/devices/loopback.rs:16)."""
import struct

import numpy as np
import pytest

import unet_amd
from unet_amd import UnetError, Verdict
from unet_amd import _ffi as F
from helpers import as_tuples, config1_echo_frame
from oracle import oracle_ffi as O


def py_pcap(records, linktype=1, big_endian=False, nanos=False, snaplen=65535, origlens=None, ts=None):
    e = ">" if big_endian else "<"
    out = struct.pack(e + "IHHiIII", 0xA1B23C4D if nanos else 0xA1B2C3D4, 2, 4, 0, 0, snaplen, linktype)
    for i, r in enumerate(records):
        sec, frac = ts[i] if ts else (1700000000 + i, 7 * i)
        out += struct.pack(e + "IIII", sec, frac, len(r), origlens[i] if origlens else len(r)) + r
    return out


def records_of(arena, off16, lens):
    return [arena[16 * int(o):16 * int(o) + int(l)].tobytes() for o, l in zip(off16, lens)]


@pytest.mark.parametrize("big_endian", [False, True])
@pytest.mark.parametrize("nanos", [False, True])
def test_pack_matches_model(big_endian, nanos):
    rng = np.random.default_rng(9)
    recs = [rng.integers(0, 256, size=int(n), dtype=np.uint8).tobytes() for n in [0, 1, 15, 16, 17, 60, 98, 1514, 1515, 9000, 65535]]
    data = py_pcap(recs, big_endian=big_endian, nanos=nanos)
    arena, off16, lens, ts, info = F.pcap_pack(data, F.LOOPBACK_MTU)
    assert (info.linktype, info.mode, info.skip, info.n_records, info.nanosecond) == (1, unet_amd.MODE_L2, 0, len(recs), int(nanos))
    assert info.n_clipped == 0 and info.n_snapped == 0 and info.trailing_bytes == 0
    assert records_of(arena, off16, lens) == recs
    assert arena.ctypes.data % 16 == 0 and (off16[1:] == np.cumsum((lens[:-1] + 15) // 16)).all()
    mult = 1 if nanos else 1000
    assert ts.tolist() == [(1700000000 + i) * 10**9 + 7 * i * mult for i in range(len(recs))]
    # pad bytes between records are zero
    used = np.zeros(arena.nbytes, dtype=bool)
    for o, l in zip(off16, lens):
        used[16 * int(o):16 * int(o) + int(l)] = True
    assert (arena[~used] == 0).all()


def test_tap_limit_clips_like_the_read_buffer():
    recs = [b"\x11" * 1514, b"\x22" * 1515, b"\x33" * 4000, b"\x44" * 60]
    arena, off16, lens, _, info = F.pcap_pack(py_pcap(recs), F.TAP_FRAME_MAX)
    assert lens.tolist() == [1514, 1514, 1514, 60] and info.n_clipped == 2
    assert records_of(arena, off16, lens) == [r[:1514] for r in recs]


@pytest.mark.parametrize("linktype,mode,skip", [(0, 0, 4), (1, 1, 0), (101, 0, 0), (228, 0, 0)])
def test_link_types(linktype, mode, skip):
    pkt = O.create_ip_header(1, 17, 0x0A000001, 0xC0000202, 12) + b"\x00\x07\x00\x07\x00\x0c\x00\x00abcd"
    recs = [struct.pack("<I", 2) + pkt if linktype == 0 else pkt, b"\x02\x00"]  # the second is shorter than a NULL header
    arena, off16, lens, _, info = F.pcap_pack(py_pcap(recs, linktype=linktype))
    assert (info.mode, info.skip) == (mode, skip)
    got = records_of(arena, off16, lens)
    assert got[0] == pkt and got[1] == (b"" if linktype == 0 else b"\x02\x00")


def test_snapped_and_trailing_and_errors():
    recs = [b"a" * 100, b"b" * 50]
    data = py_pcap(recs, origlens=[1500, 50], snaplen=100)
    info = F.pcap_scan(data)
    assert (info.n_records, info.n_snapped, info.snaplen) == (2, 1, 100)
    cut = data[:-20]  # the second record is incomplete
    info = F.pcap_scan(cut)
    assert info.n_records == 1 and info.trailing_bytes == 16 + 30
    assert F.pcap_scan(data[:24]).n_records == 0 and F.pcap_scan(data[:24]).arena_bytes == 16
    arena, off16, lens, _, info = F.pcap_pack(data[:24])
    assert len(lens) == 0 and arena.nbytes == 16
    with pytest.raises(UnetError, match="global header"):
        F.pcap_scan(data[:23])
    with pytest.raises(UnetError, match="magic"):
        F.pcap_scan(b"\x0a\x0d\x0d\x0a" + bytes(40))  # pcapng section header
    with pytest.raises(UnetError, match="link type"):
        F.pcap_scan(py_pcap(recs, linktype=113))
    with pytest.raises(UnetError, match="max_len"):
        F.pcap_scan(data, 0)
    with pytest.raises(UnetError, match="max_len"):
        F.pcap_scan(data, 65536)


def test_write_roundtrip_and_skips_empty():
    rng = np.random.default_rng(2)
    recs = [rng.integers(0, 256, size=int(n), dtype=np.uint8).tobytes() for n in [74, 0, 98, 1514, 0, 33]]
    arena, off16, lens = unet_amd.pack_arena(recs)
    data = F.pcap_write(arena, off16, lens, F.LINKTYPE_ETHERNET)
    kept = [r for r in recs if r]
    assert data == py_pcap(kept, ts=[(0, i) for i, r in enumerate(recs) if r])
    a2, o2, l2, _, info = F.pcap_pack(data)
    assert records_of(a2, o2, l2) == kept and info.linktype == 1
    # fixed-stride arenas
    st = np.zeros(3 * 32, dtype=np.uint8)
    st[0:20], st[32:52], st[64:84] = 1, 2, 3
    d3 = F.pcap_write(st, None, np.array([20, 20, 20], dtype=np.uint32), F.LINKTYPE_RAW, stride16=2)
    assert records_of(*F.pcap_pack(d3)[:3]) == [b"\x01" * 20, b"\x02" * 20, b"\x03" * 20]
    with pytest.raises(UnetError, match="outside the arena"):
        F.pcap_write(st, np.array([6], dtype=np.uint32), np.array([20], dtype=np.uint32))


def test_capture_of_config1_through_the_oracle():
    """A capture holding the echo request of configs[0] judged after import: same verdict as the frame itself."""
    frame, tap_mac, _ = config1_echo_frame()
    arena, off16, lens, _, info = F.pcap_pack(py_pcap([frame, frame[:40]]), F.TAP_FRAME_MAX)
    tap = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=tap_mac, mode=info.mode)
    res = O.recv_batch(arena, off16, lens, tap)
    assert res["verdict"].tolist() == [Verdict.ACCEPT_ICMP_ECHO, Verdict.DROP_ICMP_TRUNCATED]


@pytest.mark.gpu
def test_capture_to_verdicts_and_built_packets_to_capture(offload):
    """pcap -> arena -> unet_verify; then payloads -> unet_tx_build_dev -> pcap -> arena -> unet_verify at the receiver."""
    import torch
    frame, tap_mac, peer_mac = config1_echo_frame()
    rng = np.random.default_rng(8)
    tap = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=tap_mac, mode=unet_amd.MODE_L2)
    h_arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_IMIX, 0x756E6574, 0, 600, tap.replace(mode=0))
    l3 = records_of(h_arena, off16, lens)
    frames = [frame] + [tap_mac + peer_mac + b"\x08\x00" + p for p in l3]
    arena, o2, l2, _, info = F.pcap_pack(py_pcap(frames), F.TAP_FRAME_MAX)
    assert info.n_clipped > 0 and int(l2.max()) == 1514
    got = offload.verify(arena, o2, l2, tap)
    exp = O.recv_batch(arena, o2, l2, tap)
    assert as_tuples(got) == as_tuples(exp) and got["verdict"][0] == Verdict.ACCEPT_ICMP_ECHO

    from unet_amd._ffi import TX_DESC_DTYPE, tx_out_layout
    n = 500
    plen = rng.integers(0, 1473, size=n).astype(np.uint32)
    payloads = [rng.integers(0, 256, size=int(x), dtype=np.uint8).tobytes() for x in plen]
    p_arena, p_off, p_len = unet_amd.pack_arena(payloads)
    desc = np.zeros(n, dtype=TX_DESC_DTYPE)
    desc["src"], desc["dst"], desc["proto"], desc["l4a"], desc["l4b"] = tap.unicast, 0xC0000201, 17, 7, 7
    desc["dst_mac"] = np.frombuffer(peer_mac, dtype=np.uint8)
    ooff, obytes = tx_out_layout(p_len, 1)
    st = torch.cuda.Stream()
    d = [torch.from_numpy(x).cuda() for x in (p_arena, p_off, p_len, desc.view(np.uint8).reshape(-1).copy(), ooff)]
    d_out = torch.zeros(obytes, dtype=torch.uint8, device="cuda")
    d_olen = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    offload.tx_build_dev(d[0].data_ptr(), p_arena.nbytes, d[1].data_ptr(), 0, d[2].data_ptr(), d[3].data_ptr(), n, tap,
                         d_out.data_ptr(), obytes, d[4].data_ptr(), d_olen.data_ptr(), None, stream=st.cuda_stream)
    st.synchronize()
    cap = F.pcap_write(d_out.cpu().numpy(), ooff, d_olen.cpu().numpy().view(np.uint32), F.LINKTYPE_ETHERNET)
    a3, o3, l3_, _, info3 = F.pcap_pack(cap, F.TAP_FRAME_MAX)
    assert info3.n_records == n and info3.n_clipped == 0
    peer = unet_amd.Iface("192.0.2.1", "255.255.255.0", mac=peer_mac, mode=unet_amd.MODE_L2)
    res = offload.verify(a3, o3, l3_, peer)
    # tap::send pads short packets, and udp::recv then sees a buffer longer than the length field
    short = p_len < 32
    assert (res["verdict"][~short] == Verdict.ACCEPT_UDP).all() and (res["verdict"][short] == Verdict.DROP_UDP_LEN).all()
