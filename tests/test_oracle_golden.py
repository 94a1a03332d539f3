"""CPU-only: pin the oracle (oracle/unet_oracle.c) to the reference's own known-answer vectors and to
the independent pure-Python model behind tests/golden/verdict_vectors.json."""
import struct

import numpy as np
import pytest

from helpers import as_tuples, config1_echo_frame, edge_data, load, py_ck, vector_iface, vector_record
from oracle import oracle_ffi as O

KATS = load("reference_kats.json")


# --- the reference's own tests, restated one for one -------------------------------------------

def test_calculate_checksum():
    """src/utils.rs:43-50"""
    data = bytes.fromhex(KATS["checksum_even"]["data"])
    s = O.calculate_checksum(data, 0)
    assert s == 0xBC58
    assert O.calculate_checksum(data, s) == 0


def test_calculate_checksum_odd_elements():
    """src/utils.rs:52-60"""
    data = bytes.fromhex(KATS["checksum_odd"]["data"])
    s = O.calculate_checksum(data, 0)
    assert s == 0x7C58
    assert O.calculate_checksum(data, s) == 0


def test_ipv4_header():
    """src/protocols/ipv4.rs:428-438: try_from + validate is Ok on the 48-byte packet."""
    k = KATS["ipv4_header"]
    pkt = bytes.fromhex(k["packet"])
    verdict, proto, csum = O.ipv4_header_validate(pkt[:20])
    assert (verdict, proto, csum) == (25, 1, 0)
    # the stored field is exactly what create_ip_header's arithmetic yields with the field zeroed
    z = bytearray(pkt[:20]); z[10:12] = b"\0\0"
    assert O.calculate_checksum(bytes(z), 0) == k["header_checksum_field"] == 0xBD4A


def test_parse_icmp_header():
    """src/transport/icmp.rs:128-139: type Echo, code 0, checksum 0x3564, values 0x00800001."""
    k = KATS["icmp_header"]
    msg = bytes.fromhex(k["message"])
    assert msg[0] == 8 and msg[1] == 0
    assert struct.unpack(">H", msg[2:4])[0] == 0x3564 and struct.unpack(">I", msg[4:8])[0] == 0x00800001
    assert O.calculate_checksum(msg, 0) == 0
    assert O.icmp_send_checksum(8, 0, 0x00800001, msg[8:]) == 0x3564


def test_udp_demo_derived_vector():
    k = KATS["udp_demo_derived"]
    payload = bytes.fromhex(k["payload"])
    assert O.udp_send_checksum(k["src"], k["dst"], k["sport"], k["dport"], payload) == 0x8019
    seg = bytes.fromhex(k["segment"])
    pkt = O.create_ip_header(0, 17, 0x7F000001, 0x7F000001, len(seg)) + seg
    # NB: udp::send computed the pseudo-header with src 0.0.0.0 (app.rs:86); ipv4::send then stamps the
    # interface address, so the datagram the stack itself emits fails its own verification.
    import unet_amd
    lo = unet_amd.Iface("127.0.0.1", "255.0.0.0")
    assert O.recv(pkt, lo)[0] == 19
    pkt0 = O.create_ip_header(0, 17, 0, 0x7F000001, len(seg)) + seg
    assert O.recv(pkt0, lo) == (20, 17, 0)


def test_reference_packet_through_ipv4_recv():
    k = KATS["ipv4_header"]
    import unet_amd
    lo = unet_amd.Iface("127.0.0.1", "255.0.0.0")
    assert lo.broadcast == 0x7F0000FF  # src/protocols/ipv4.rs:220 with the /8 of src/app.rs:33-36
    assert list(O.recv(bytes.fromhex(k["packet"]), lo)) == [15, 1, k["recv_verdict"][2]]
    assert O.lib().oracle_iface_broadcast(0xC0000202, 0xFFFFFF00) == 0xC00002FF


# --- edge vectors and the independent model ---------------------------------------------------

@pytest.mark.parametrize("e", load("checksum_edges.json")["vectors"], ids=lambda e: e["name"])
def test_checksum_edges(e):
    d = edge_data(e)
    assert O.calculate_checksum(d, e["seed"]) == e["ck"]
    assert O.calculate_checksum_fast(d, e["seed"]) == e["ck"]


@pytest.mark.parametrize("v", load("verdict_vectors.json")["vectors"], ids=lambda v: v["name"])
def test_verdict_vectors(v):
    got = O.recv(vector_record(v), vector_iface(v))
    assert got == (v["verdict_code"], v["proto"], v["csum"]), v["verdict"]


def test_verdict_codes_match_public_header():
    import re, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pub = dict(re.findall(r"UNET_V_(\w+)\s*=\s*(\d+)", open(os.path.join(root, "include", "unet_offload.h")).read()))
    ora = dict(re.findall(r"OV_(\w+)\s*=\s*(\d+)", open(os.path.join(root, "oracle", "unet_oracle.h")).read()))
    assert pub == ora
    gold = load("verdict_vectors.json")["verdict_codes"]
    assert {k: str(v) for k, v in gold.items()} == {k: v for k, v in pub.items() if k != "COUNT"}
    import unet_amd
    assert {m.name: str(int(m)) for m in unet_amd.Verdict} == {k: v for k, v in pub.items() if k != "COUNT"}


# --- properties of the checksum the reference's tests rely on ----------------------------------

def test_fast_equals_faithful_random():
    rng = np.random.default_rng(1)
    for _ in range(400):
        n = int(rng.integers(0, 3000))
        d = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        seed = int(rng.integers(0, 65536))
        a = O.calculate_checksum(d, seed)
        assert a == O.calculate_checksum_fast(d, seed)
        assert O.calculate_checksum(d, a if seed == 0 else 0) is not None


def test_chaining_properties():
    rng = np.random.default_rng(2)
    for _ in range(200):
        n = int(rng.integers(1, 2000))
        d = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        s = O.calculate_checksum(d, 0)
        assert O.calculate_checksum(d, s) == 0                         # the reference's own test idiom
        cut = int(rng.integers(0, n // 2 + 1)) * 2                     # even split point
        part = O.calculate_checksum(d[:cut], 0)
        assert O.calculate_checksum(d[cut:], (~part) & 0xFFFF) == s    # split-and-chain == one shot
        # udp::recv's two-step pseudo-header chaining == one pass over pseudo || segment
        pseudo = rng.integers(0, 256, size=12, dtype=np.uint8).tobytes()
        two_step = O.calculate_checksum(pseudo, (~s) & 0xFFFF)
        assert two_step == O.calculate_checksum(pseudo + d, 0)


def test_generate_then_verify_roundtrip():
    import unet_amd
    tap = unet_amd.Iface()
    rng = np.random.default_rng(3)
    frames = []
    for i in range(64):
        n = int(rng.integers(8, 1400))
        proto = 17 if i % 3 else 1
        hdr = bytearray(O.create_ip_header(i, proto, 0x0A000001, tap.unicast, n))
        hdr[10:12] = b"\xAA\x55"  # junk in the field: generate must overwrite it
        body = bytearray(rng.integers(0, 256, size=n, dtype=np.uint8).tobytes())
        if proto == 17:
            body[4:6] = struct.pack(">H", n)
        else:
            body[0] = 8
        frames.append(bytes(hdr) + bytes(body))
    arena, off16, lens = unet_amd.pack_arena(frames)
    filled, status = O.generate_batch(arena, off16, lens)
    assert not status.any()
    res = O.recv_batch(filled, off16, lens, tap)
    assert set(res["verdict"].tolist()) <= {20, 15}
    for f, o in zip(frames, off16):
        rec = filled[int(o) * 16:int(o) * 16 + len(f)].tobytes()
        if f[9] == 1:
            assert O.calculate_checksum(rec[20:], 0) == 0


def test_config1_single_icmp_echo_on_tap0():
    """The reference answering one 64-byte ICMP echo on tap0, restated on the CPU: receive verdict, then
    the exact bytes of the reply the stack emits (icmp::recv -> icmp::send -> ipv4::send -> tap::send:
    src/transport/icmp.rs:110-120,75-93, src/protocols/ipv4.rs:337-339,366-391, src/driver/tap.rs:138-152)."""
    import unet_amd
    frame, tap_mac, peer_mac = config1_echo_frame()
    tap = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=tap_mac, mode=unet_amd.MODE_L2)
    verdict, proto, reply_ck = O.recv(frame, tap)
    assert (verdict, proto) == (15, 1)
    msg = frame[34:]
    code, values, data = msg[1], struct.unpack(">I", msg[4:8])[0], msg[8:]
    reply_icmp = bytearray(struct.pack(">BBHI", 0, code, 0, values) + data)
    assert reply_ck == py_ck(bytes(reply_icmp)) == O.icmp_send_checksum(0, code, values, data)
    struct.pack_into(">H", reply_icmp, 2, reply_ck)
    reply_ip = O.create_ip_header(0, 1, 0xC0000202, 0xC0000201, len(reply_icmp))  # first id of Ipv4IdGenerator is 0
    assert reply_ip[:2] == b"\x45\x00" and struct.unpack(">H", reply_ip[2:4])[0] == 84 and reply_ip[8] == 64
    assert py_ck(reply_ip) == 0
    reply = peer_mac + tap_mac + b"\x08\x00" + reply_ip + bytes(reply_icmp)  # 98 bytes: no pad-to-60 needed
    assert len(reply) == 98
    peer = unet_amd.Iface("192.0.2.1", "255.255.255.0", mac=peer_mac, mode=unet_amd.MODE_L2)
    assert O.recv(reply, peer) == (16, 1, 0)
    assert py_ck(reply[34:]) == 0   # the peer's kernel does verify the ICMP checksum
