#!/usr/bin/env python3
"""Writes tests/golden/*.json.

Two kinds of vectors:

* ``reference_kats.json`` -- the byte arrays and asserted values of the reference's own unit tests
  for this path (src/utils.rs:43-60, src/protocols/ipv4.rs:428-438, src/transport/icmp.rs:128-139)
  plus the UDP demo datagram of src/app.rs:83-87.  The bytes are data transcribed from those tests;
  the derived numbers (0xbc58 ...) are recomputed below by a pure-Python model and must agree with
  what the reference asserts (chained checksum == 0, parsed fields).

* ``verdict_vectors.json`` / ``checksum_edges.json`` -- hand-built records, one or more per verdict
  code and per checksum edge case, with the outcome predicted by the pure-Python model in this file.
  That model is a second restatement of the reference's control flow, written independently of
  oracle/unet_oracle.c, so the C oracle, this model and the GPU kernels are checked against each
  other.  Nothing in the reference asserts these outcomes directly (it has no negative tests).

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- model
def ck(data: bytes, seed: int = 0) -> int:
    """src/utils.rs:1-17"""
    total = seed
    for i in range(0, len(data), 2):
        chunk = data[i:i + 2]
        total += (chunk[0] << 8) if len(chunk) == 1 else ((chunk[0] << 8) | chunk[1])
    total &= 0xFFFFFFFF
    while total >> 16:
        total = (total & 0xFFFF) + (total >> 16)
    return (~total) & 0xFFFF


V = dict(
    NONE=0, DROP_L2_TRUNCATED=1, DROP_L2_ETHERTYPE=2, DROP_L2_NOT_FOR_ME=3, PASS_ARP=4, DROP_L3_TRUNCATED=5,
    DROP_L3_PROTO=6, DROP_L3_VERSION=7, DROP_L3_IHL_SHORT=8, DROP_L3_IHL_LONG=9, DROP_L3_FRAGMENT=10,
    DROP_L3_CSUM=11, IGNORE_L3_NOT_FOR_ME=12, DROP_ICMP_TRUNCATED=13, DROP_ICMP_TYPE=14, ACCEPT_ICMP_ECHO=15,
    ACCEPT_ICMP_ECHOREPLY=16, DROP_UDP_SHORT=17, DROP_UDP_LEN=18, DROP_UDP_CSUM=19, ACCEPT_UDP=20,
    DROP_TCP_SHORT=21, DROP_TCP_DOFF=22, DROP_TCP_CSUM=23, ACCEPT_TCP=24, L3_HEADER_OK=25, DROP_TCP_OPTION=32,
    DROP_ARP_TRUNCATED=26, DROP_ARP_HTYPE=27, DROP_ARP_PTYPE=28, DROP_ARP_HLEN=29, DROP_ARP_PLEN=30,
    IGNORE_ARP_NOT_FOR_ME=31)


def hdr_validate(b: bytes, ext_tcp: bool):
    proto = b[9]
    if proto not in (1, 17) and not (ext_tcp and proto == 6):
        return ("DROP_L3_PROTO", proto, 0)
    if b[0] >> 4 != 4:
        return ("DROP_L3_VERSION", proto, 0)
    hl = (b[0] & 0xF) * 4
    if hl < 20:
        return ("DROP_L3_IHL_SHORT", proto, 0)
    if not hl < 60:
        return ("DROP_L3_IHL_LONG", proto, 0)
    ff = (b[6] << 8) | b[7]
    if ((ff >> 13) & 1) or (ff & 0x1FFF):
        return ("DROP_L3_FRAGMENT", proto, 0)
    c = ck(b[:20])
    if c:
        return ("DROP_L3_CSUM", proto, c)
    return ("L3_HEADER_OK", proto, 0)


def ipv4_recv(d: bytes, ifc):
    if len(d) < 20:
        return ("DROP_L3_TRUNCATED", 0, 0)
    r = hdr_validate(d, ifc["ext_tcp"])
    if r[0] != "L3_HEADER_OK":
        return r
    proto = d[9]
    src, dst = struct.unpack(">II", d[12:20])
    if dst not in (ifc["unicast"], ifc["broadcast"], 0xFFFFFFFF):
        return ("IGNORE_L3_NOT_FOR_ME", proto, 0)
    hl = (d[0] & 0xF) * 4
    if hl > len(d):
        return ("DROP_L3_TRUNCATED", proto, 0)
    p = d[hl:]
    if proto == 1:
        if len(p) < 1:
            return ("DROP_ICMP_TRUNCATED", 1, 0)
        if p[0] not in (0, 8):
            return ("DROP_ICMP_TYPE", 1, 0)
        if len(p) < 8:
            return ("DROP_ICMP_TRUNCATED", 1, 0)
        if p[0] == 8:
            reply = bytes([0, p[1], 0, 0]) + p[4:8] + p[8:]
            return ("ACCEPT_ICMP_ECHO", 1, ck(reply))
        return ("ACCEPT_ICMP_ECHOREPLY", 1, 0)
    if proto == 17:
        if len(p) < 8:
            return ("DROP_UDP_SHORT", 17, 0)
        s = ck(p)
        length = (p[4] << 8) | p[5]
        if len(p) != length:
            return ("DROP_UDP_LEN", 17, 0)
        pseudo = d[12:20] + bytes([0, 17]) + struct.pack(">H", length)
        f = ck(pseudo, (~s) & 0xFFFF)
        return ("DROP_UDP_CSUM", 17, f) if f else ("ACCEPT_UDP", 17, 0)
    if len(p) < 20:
        return ("DROP_TCP_SHORT", 6, 0)
    doff = p[12] >> 4
    if doff < 5 or doff * 4 > len(p):
        return ("DROP_TCP_DOFF", 6, 0)
    pseudo = d[12:20] + bytes([0, 6]) + struct.pack(">H", len(p))
    f = ck(pseudo, (~ck(p)) & 0xFFFF)
    if f:
        return ("DROP_TCP_CSUM", 6, f)
    return ("DROP_TCP_OPTION", 6, 0) if tcp_options_bad(p[20:doff * 4]) else ("ACCEPT_TCP", 6, 0)


FIXED_OPTION_LENGTH = {2: 4, 3: 3, 4: 2, 8: 10}  # MSS, window scale, SACK permitted, timestamps


def tcp_options_bad(opts: bytes) -> bool:
    """Extension, no counterpart in the reference: is the option area of a TCP header malformed?"""
    while opts:
        kind = opts[0]
        if kind == 0:
            return False
        if kind == 1:
            opts = opts[1:]
            continue
        if len(opts) < 2 or opts[1] < 2 or opts[1] > len(opts):
            return True
        n = opts[1]
        if kind in FIXED_OPTION_LENGTH and n != FIXED_OPTION_LENGTH[kind]:
            return True
        if kind == 5 and (n < 10 or (n - 2) % 8):
            return True
        opts = opts[n:]
    return False


def arp_recv(d: bytes, ifc):
    """src/protocols/arp.rs:206-246"""
    if len(d) < 8:
        return ("DROP_ARP_TRUNCATED", 0, 0)
    htype, ptype, hlen, plen = struct.unpack(">HHBB", d[:6])
    if htype != 1:
        return ("DROP_ARP_HTYPE", 0, 0)
    if ptype != 0x0800:
        return ("DROP_ARP_PTYPE", 0, 0)
    if hlen != 6:
        return ("DROP_ARP_HLEN", 0, 0)
    if plen != 4:
        return ("DROP_ARP_PLEN", 0, 0)
    if len(d) < 28:
        return ("DROP_ARP_TRUNCATED", 0, 0)
    tpa = struct.unpack(">I", d[24:28])[0]
    return ("PASS_ARP", 0, 0) if tpa == ifc["unicast"] else ("IGNORE_ARP_NOT_FOR_ME", 0, 0)


def eth_recv(fr: bytes, ifc):
    if len(fr) < 14:
        return ("DROP_L2_TRUNCATED", 0, 0)
    et = (fr[12] << 8) | fr[13]
    if et not in (0x0800, 0x0806):
        return ("DROP_L2_ETHERTYPE", 0, 0)
    if fr[:6] != bytes(ifc["mac"]) and fr[:6] != b"\xff" * 6:
        return ("DROP_L2_NOT_FOR_ME", 0, 0)
    if et == 0x0806:
        return arp_recv(fr[14:], ifc)
    return ipv4_recv(fr[14:], ifc)


# ----------------------------------------------------------------------------- builders
def ip_header(proto, src, dst, payload_len, ihl=5, ident=0x1234, flags_frag=0, ttl=64, version=4, options=b"",
              total_length=None, fix_csum=True):
    hl = ihl * 4
    tl = hl + payload_len if total_length is None else total_length
    h = bytearray(struct.pack(">BBHHHBBHII", (version << 4) | ihl, 0, tl & 0xFFFF, ident, flags_frag, ttl, proto, 0, src, dst))
    if fix_csum:
        struct.pack_into(">H", h, 10, ck(bytes(h[:20])))
    return bytes(h) + options


def udp_segment(src, dst, sport, dport, payload, length=None, csum=None):
    length = 8 + len(payload) if length is None else length
    seg = bytearray(struct.pack(">HHHH", sport, dport, length, 0) + payload)
    pseudo = struct.pack(">IIBBH", src, dst, 0, 17, length)
    c = ck(bytes(seg), (~ck(pseudo)) & 0xFFFF) if csum is None else csum
    struct.pack_into(">H", seg, 6, c)
    return bytes(seg)


def tcp_segment(src, dst, sport, dport, payload, doff=5, opts=b""):
    hdr = bytearray(struct.pack(">HHIIBBHHH", sport, dport, 0x01020304, 0x0A0B0C0D, doff << 4, 0x18, 4096, 0, 0) + opts)
    seg = bytearray(hdr + payload)
    pseudo = struct.pack(">IIBBH", src, dst, 0, 6, len(seg))
    struct.pack_into(">H", seg, 16, ck(bytes(seg), (~ck(pseudo)) & 0xFFFF))
    return bytes(seg)


def icmp_message(ty, code, values, data, csum=None):
    m = bytearray(struct.pack(">BBHI", ty, code, 0, values) + data)
    struct.pack_into(">H", m, 2, ck(bytes(m)) if csum is None else csum)
    return bytes(m)


def prng(tag: str, n: int) -> bytes:
    out = b""
    i = 0
    while len(out) < n:
        out += hashlib.sha256(f"{tag}:{i}".encode()).digest()
        i += 1
    return out[:n]


# ----------------------------------------------------------------------------- reference KATs
KAT_EVEN = bytes([0x1f, 0x40, 0x1f, 0x41, 0x0, 0x1c, 0x0, 0x0, 0x30, 0x31, 0x32, 0x33, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39])
KAT_ODD = KAT_EVEN + bytes([0x40])
KAT_IPV4 = bytes([
    0x45, 0x00, 0x00, 0x30, 0x00, 0x80, 0x00, 0x00, 0xff, 0x01, 0xbd, 0x4a, 0x7f, 0x00, 0x00, 0x01, 0x7f, 0x00, 0x00, 0x01,
    0x08, 0x00, 0x35, 0x64, 0x00, 0x80, 0x00, 0x01, 0x31, 0x32, 0x33, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x30, 0x21, 0x40,
    0x23, 0x24, 0x25, 0x5e, 0x26, 0x2a, 0x28, 0x29])
KAT_ICMP = KAT_IPV4[20:]
UDP_DEMO_PAYLOAD = KAT_IPV4[28:]


def main():
    lo = dict(unicast=0x7F000001, netmask=0xFF000000, broadcast=0x7F0000FF, mac=[0, 0, 0, 0, 0, 0], mode=0, ext_tcp=False)
    tap = dict(unicast=0xC0000202, netmask=0xFFFFFF00, broadcast=0xC00002FF, mac=[0x02, 0x00, 0x5e, 0x10, 0x20, 0x30], mode=0,
               ext_tcp=False)

    # ---- reference KATs
    s_even, s_odd = ck(KAT_EVEN), ck(KAT_ODD)
    assert ck(KAT_EVEN, s_even) == 0 and ck(KAT_ODD, s_odd) == 0      # what utils.rs:48-49,58-59 assert
    assert hdr_validate(KAT_IPV4, False)[0] == "L3_HEADER_OK"         # ipv4.rs:436-437
    assert KAT_ICMP[0] == 8 and KAT_ICMP[1] == 0 and KAT_ICMP[2:4] == b"\x35\x64" and KAT_ICMP[4:8] == b"\x00\x80\x00\x01"
    hdr_zeroed = bytearray(KAT_IPV4[:20]); hdr_zeroed[10:12] = b"\0\0"
    icmp_zeroed = bytearray(KAT_ICMP); icmp_zeroed[2:4] = b"\0\0"
    udp_demo = udp_segment(0x00000000, 0x7F000001, 8000, 8001, UDP_DEMO_PAYLOAD)
    kats = {
        "source": "ikanago/unet @ 2024_10_08 unit tests; derived numbers recomputed by tests/golden/make_golden.py",
        "checksum_even": {"ref": "src/utils.rs:43-50", "data": KAT_EVEN.hex(), "ck": s_even, "chained": 0},
        "checksum_odd": {"ref": "src/utils.rs:52-60", "data": KAT_ODD.hex(), "ck": s_odd, "chained": 0},
        "ipv4_header": {"ref": "src/protocols/ipv4.rs:428-438", "packet": KAT_IPV4.hex(), "validate_ok": True,
                        "header_checksum_field": 0xBD4A, "generated_header_checksum": ck(bytes(hdr_zeroed)),
                        "protocol": 1, "src": 0x7F000001, "dst": 0x7F000001, "ttl": 255, "total_length": 48,
                        "lo_iface": lo, "recv_verdict": list(ipv4_recv(KAT_IPV4, lo))},
        "icmp_header": {"ref": "src/transport/icmp.rs:128-139", "message": KAT_ICMP.hex(), "type": 8, "code": 0,
                        "checksum": 0x3564, "values": 0x00800001, "whole_message_ck": ck(KAT_ICMP),
                        "generated_checksum": ck(bytes(icmp_zeroed))},
        "udp_demo_derived": {"ref": "src/app.rs:83-87 + src/transport/udp.rs:186-203 (derived, not asserted by the reference)",
                             "src": 0, "sport": 8000, "dst": 0x7F000001, "dport": 8001, "payload": UDP_DEMO_PAYLOAD.hex(),
                             "segment": udp_demo.hex(), "checksum": struct.unpack(">H", udp_demo[6:8])[0]},
    }
    assert kats["checksum_even"]["ck"] == 0xBC58 and kats["checksum_odd"]["ck"] == 0x7C58
    assert kats["ipv4_header"]["generated_header_checksum"] == 0xBD4A
    assert kats["icmp_header"]["generated_checksum"] == 0x3564 and kats["icmp_header"]["whole_message_ck"] == 0
    assert kats["udp_demo_derived"]["checksum"] == 0x8019
    with open(os.path.join(HERE, "reference_kats.json"), "w") as f:
        json.dump(kats, f, indent=1)

    # ---- checksum edge cases
    edges = []

    def edge(name, data=None, gen=None, fill=None, seed=0):
        d = data if data is not None else (prng(gen[0], gen[1]) if gen else bytes([fill[0]]) * fill[1])
        e = {"name": name, "seed": seed, "ck": ck(d, seed)}
        if data is not None:
            e["data"] = data.hex()
        elif gen:
            e["prng"] = {"tag": gen[0], "len": gen[1]}
        else:
            e["fill"] = {"byte": fill[0], "len": fill[1]}
        edges.append(e)

    edge("empty", b"")
    edge("empty_seeded", b"", seed=0x1234)
    edge("one_byte", b"\xab")
    edge("two_bytes", b"\xab\xcd")
    edge("three_bytes", b"\xab\xcd\xef")
    edge("zeros_64", bytes(64))
    edge("ones_64", b"\xff" * 64)
    edge("ones_65535", fill=(0xFF, 65535))
    edge("zeros_65535", fill=(0, 65535), seed=0xFFFF)
    edge("sums_to_ffff", b"\xff\xff")
    edge("sums_to_ffff_seed", b"\xff\xfe", seed=1)
    edge("len_15", prng("l15", 15))
    edge("len_16", prng("l16", 16))
    edge("len_17", prng("l17", 17))
    edge("len_1023", gen=("l1023", 1023))
    edge("len_1024", gen=("l1024", 1024))
    edge("len_1025", gen=("l1025", 1025))
    edge("len_1500", gen=("l1500", 1500), seed=0xBEEF)
    edge("len_4096", gen=("l4096", 4096))
    edge("len_5121", gen=("l5121", 5121))
    edge("len_9000", gen=("l9000", 9000))
    edge("len_65535", gen=("l65535", 65535), seed=0xFFFF)
    with open(os.path.join(HERE, "checksum_edges.json"), "w") as f:
        json.dump({"note": "prng(tag,len) = concatenated sha256(f'{tag}:{i}') digests, truncated", "vectors": edges}, f, indent=1)

    # ---- verdict vectors
    vecs = []
    me, other = tap["unicast"], 0x0A000001
    mac = bytes(tap["mac"])

    def add(name, rec, ifc, l2=False, tail=None):
        """tail=(tag, n): the last n bytes of rec are prng(tag, n) and are stored as that recipe."""
        i = dict(ifc); i["mode"] = 1 if l2 else 0
        v = eth_recv(rec, i) if l2 else ipv4_recv(rec, i)
        e = {"name": name, "record": rec.hex(), "iface": i, "verdict": v[0], "verdict_code": V[v[0]], "proto": v[1], "csum": v[2]}
        if tail is not None:
            assert rec[len(rec) - tail[1]:] == prng(*tail)
            e["record"] = rec[:len(rec) - tail[1]].hex()
            e["tail_prng"] = {"tag": tail[0], "len": tail[1]}
        vecs.append(e)
        return v[0]

    pay = prng("pay", 100)
    good_udp = ip_header(17, other, me, 108) + udp_segment(other, me, 5000, 8000, pay)
    assert add("udp_ok", good_udp, tap) == "ACCEPT_UDP"
    assert add("udp_ok_odd_payload", ip_header(17, other, me, 8 + 33) + udp_segment(other, me, 1, 2, pay[:33]), tap) == "ACCEPT_UDP"
    assert add("udp_ok_broadcast_dst", ip_header(17, other, tap["broadcast"], 108) + udp_segment(other, tap["broadcast"], 5000, 8000, pay), tap) == "ACCEPT_UDP"
    assert add("udp_ok_limited_broadcast", ip_header(17, other, 0xFFFFFFFF, 108) + udp_segment(other, 0xFFFFFFFF, 5000, 8000, pay), tap) == "ACCEPT_UDP"
    assert add("udp_ok_empty_payload", ip_header(17, other, me, 8) + udp_segment(other, me, 7, 9, b""), tap) == "ACCEPT_UDP"
    assert add("udp_ok_ip_options", ip_header(17, other, me, 108, ihl=7, options=prng("opt", 8)) + udp_segment(other, me, 5000, 8000, pay), tap) == "ACCEPT_UDP"
    assert add("udp_ok_ihl14", ip_header(17, other, me, 108, ihl=14, options=prng("opt14", 36)) + udp_segment(other, me, 5000, 8000, pay), tap) == "ACCEPT_UDP"
    assert add("udp_ok_df_and_reserved_flag", ip_header(17, other, me, 108, flags_frag=0xC000) + udp_segment(other, me, 5000, 8000, pay), tap) == "ACCEPT_UDP"
    assert add("udp_ok_total_length_wrong", ip_header(17, other, me, 108, total_length=999) + udp_segment(other, me, 5000, 8000, pay), tap) == "ACCEPT_UDP"
    b = bytearray(good_udp); b[60] ^= 0x04
    assert add("udp_payload_bitflip", bytes(b), tap) == "DROP_UDP_CSUM"
    assert add("udp_transmitted_csum_zero", ip_header(17, other, me, 108) + udp_segment(other, me, 5000, 8000, pay, csum=0), tap) == "DROP_UDP_CSUM"
    assert add("udp_len_field_short", ip_header(17, other, me, 108) + udp_segment(other, me, 5000, 8000, pay, length=100), tap) == "DROP_UDP_LEN"
    assert add("udp_ethernet_padding", good_udp + bytes(6), tap) == "DROP_UDP_LEN"
    assert add("udp_short_7", ip_header(17, other, me, 7) + bytes(7), tap) == "DROP_UDP_SHORT"
    assert add("udp_short_0", ip_header(17, other, me, 0), tap) == "DROP_UDP_SHORT"
    # a datagram whose correct checksum is 0x0000 (stored as is by udp::send) verifies
    for k in range(70000):
        p2 = struct.pack(">I", k) + b"\0\0"
        seg = udp_segment(other, me, 5000, 8000, p2)
        if seg[6:8] == b"\0\0":
            assert add("udp_computed_csum_zero_ok", ip_header(17, other, me, len(seg)) + seg, tap) == "ACCEPT_UDP"
            break
    else:
        raise SystemExit("no zero-checksum datagram found")

    echo = icmp_message(8, 0, 0x00800001, prng("echo", 56))
    assert add("icmp_echo_64", ip_header(1, other, me, 64) + echo, tap) == "ACCEPT_ICMP_ECHO"
    assert add("icmp_echo_bad_checksum_still_accepted", ip_header(1, other, me, 64) + icmp_message(8, 0, 1, prng("echo", 56), csum=0xDEAD), tap) == "ACCEPT_ICMP_ECHO"
    assert add("icmp_echo_code_nonzero_odd", ip_header(1, other, me, 8 + 13) + icmp_message(8, 3, 0x01020304, prng("e13", 13)), tap) == "ACCEPT_ICMP_ECHO"
    assert add("icmp_echo_zero_body", ip_header(1, other, me, 8) + icmp_message(8, 0, 0, b""), tap) == "ACCEPT_ICMP_ECHO"
    assert add("icmp_echoreply", ip_header(1, other, me, 64) + icmp_message(0, 0, 7, prng("echo", 56)), tap) == "ACCEPT_ICMP_ECHOREPLY"
    assert add("icmp_type3", ip_header(1, other, me, 64) + icmp_message(3, 1, 0, prng("echo", 56)), tap) == "DROP_ICMP_TYPE"
    assert add("icmp_type3_short", ip_header(1, other, me, 2) + bytes([3, 1]), tap) == "DROP_ICMP_TYPE"
    assert add("icmp_empty", ip_header(1, other, me, 0), tap) == "DROP_ICMP_TRUNCATED"
    assert add("icmp_valid_type_short", ip_header(1, other, me, 5) + bytes([8, 0, 1, 2, 3]), tap) == "DROP_ICMP_TRUNCATED"
    assert add("reference_ipv4_test_packet_on_lo", KAT_IPV4, lo) == "ACCEPT_ICMP_ECHO"

    assert add("l3_truncated_19", good_udp[:19], tap) == "DROP_L3_TRUNCATED"
    assert add("l3_truncated_0", b"", tap) == "DROP_L3_TRUNCATED"
    assert add("l3_ihl_beyond_buffer", ip_header(17, other, me, 0, ihl=8, options=b"")[:24], tap) == "DROP_L3_TRUNCATED"
    assert add("l3_proto_tcp_strict", ip_header(6, other, me, 20) + tcp_segment(other, me, 1, 2, b""), tap) == "DROP_L3_PROTO"
    assert add("l3_proto_47", ip_header(47, other, me, 8) + bytes(8), tap) == "DROP_L3_PROTO"
    assert add("l3_proto_checked_before_version", ip_header(47, other, me, 8, version=6) + bytes(8), tap) == "DROP_L3_PROTO"
    assert add("l3_version_6", ip_header(17, other, me, 108, version=6) + udp_segment(other, me, 5000, 8000, pay), tap) == "DROP_L3_VERSION"
    assert add("l3_ihl_4", ip_header(17, other, me, 108, ihl=4)[:20] + udp_segment(other, me, 5000, 8000, pay), tap) == "DROP_L3_IHL_SHORT"
    assert add("l3_ihl_15", ip_header(17, other, me, 108, ihl=15, options=bytes(40)) + udp_segment(other, me, 5000, 8000, pay), tap) == "DROP_L3_IHL_LONG"
    assert add("l3_more_fragments", ip_header(17, other, me, 108, flags_frag=0x2000) + udp_segment(other, me, 5000, 8000, pay), tap) == "DROP_L3_FRAGMENT"
    assert add("l3_fragment_offset", ip_header(17, other, me, 108, flags_frag=0x0001) + udp_segment(other, me, 5000, 8000, pay), tap) == "DROP_L3_FRAGMENT"
    b = bytearray(good_udp); b[13] ^= 0x40
    assert add("l3_header_bitflip", bytes(b), tap) == "DROP_L3_CSUM"
    b = bytearray(ip_header(17, other, me, 108, ihl=6, options=b"\x01\x01\x01\x00") + udp_segment(other, me, 5000, 8000, pay))
    struct.pack_into(">H", b, 10, 0); struct.pack_into(">H", b, 10, ck(bytes(b[:24])))  # checksum over 24 bytes, the RFC way
    assert add("l3_options_summed_rfc_way_is_rejected", bytes(b), tap) == "DROP_L3_CSUM"
    assert add("l3_not_for_me", ip_header(17, other, 0xC0000203, 108) + udp_segment(other, 0xC0000203, 5000, 8000, pay), tap) == "IGNORE_L3_NOT_FOR_ME"
    assert add("l3_not_for_me_checked_after_checksum", ip_header(17, other, 0xC0000203, 108, fix_csum=False) + udp_segment(other, 0xC0000203, 5000, 8000, pay), tap) == "DROP_L3_CSUM"

    tcp_ifc = dict(tap); tcp_ifc["ext_tcp"] = True
    tcp_ok = ip_header(6, other, me, 20 + 100) + tcp_segment(other, me, 443, 50000, pay)
    assert add("tcp_ext_ok", tcp_ok, tcp_ifc) == "ACCEPT_TCP"
    ts = bytes([1, 1, 8, 10]) + prng("to", 8)
    assert add("tcp_ext_ok_options_odd", ip_header(6, other, me, 32 + 33) + tcp_segment(other, me, 443, 50000, pay[:33], doff=8, opts=ts), tcp_ifc) == "ACCEPT_TCP"

    def tcp_opt(name, opts, want, payload=pay[:21]):
        assert len(opts) % 4 == 0
        pkt = ip_header(6, other, me, 20 + len(opts) + len(payload)) + tcp_segment(other, me, 443, 50000, payload, doff=5 + len(opts) // 4, opts=opts)
        assert add(name, pkt, tcp_ifc) == want, name
        return pkt
    tcp_opt("tcp_opt_syn_typical", bytes([2, 4, 5, 0xb4, 4, 2, 8, 10]) + prng("t1", 8) + bytes([1, 3, 3, 7]), "ACCEPT_TCP")
    tcp_opt("tcp_opt_all_nop", bytes([1] * 40), "ACCEPT_TCP")
    tcp_opt("tcp_opt_eol_then_garbage", bytes([1, 0, 0xff, 0x01]), "ACCEPT_TCP")
    tcp_opt("tcp_opt_sack_two_blocks", bytes([1, 1, 5, 18]) + prng("t2", 16), "ACCEPT_TCP")
    tcp_opt("tcp_opt_unknown_kind_skipped", bytes([0xfd, 6, 1, 2, 3, 4, 1, 1]), "ACCEPT_TCP")
    tcp_opt("tcp_opt_length_zero", bytes([8, 0, 0, 0]), "DROP_TCP_OPTION")
    tcp_opt("tcp_opt_length_one", bytes([0x22, 1, 1, 1]), "DROP_TCP_OPTION")
    tcp_opt("tcp_opt_no_room_for_length", bytes([1, 1, 1, 8]), "DROP_TCP_OPTION")
    tcp_opt("tcp_opt_runs_past_data_offset", bytes([1, 1, 8, 10]) + prng("t3", 4), "DROP_TCP_OPTION")
    tcp_opt("tcp_opt_mss_wrong_length", bytes([2, 6, 5, 0xb4, 0, 0, 1, 1]), "DROP_TCP_OPTION")
    tcp_opt("tcp_opt_sack_wrong_length", bytes([5, 12]) + prng("t4", 10), "DROP_TCP_OPTION")
    tcp_opt("tcp_opt_last_byte_kind", bytes([1] * 39 + [3]), "DROP_TCP_OPTION")
    bad = bytearray(tcp_opt("tcp_opt_bad_and_bitflip_base", bytes([8, 0, 0, 0]), "DROP_TCP_OPTION")); bad[-1] ^= 1
    assert add("tcp_opt_checksum_is_judged_first", bytes(bad), tcp_ifc) == "DROP_TCP_CSUM"
    b = bytearray(tcp_ok); b[77] ^= 0x80
    assert add("tcp_ext_bitflip", bytes(b), tcp_ifc) == "DROP_TCP_CSUM"
    assert add("tcp_ext_short", ip_header(6, other, me, 19) + bytes(19), tcp_ifc) == "DROP_TCP_SHORT"
    b = bytearray(tcp_ok); b[32] = 0x40
    assert add("tcp_ext_doff_4", bytes(b), tcp_ifc) == "DROP_TCP_DOFF"
    b = bytearray(ip_header(6, other, me, 24) + tcp_segment(other, me, 1, 2, b"abcd")); b[32] = 0x70
    assert add("tcp_ext_doff_beyond_segment", bytes(b), tcp_ifc) == "DROP_TCP_DOFF"

    def eth(dst, et, l3):
        return dst + b"\x02\xaa\xbb\xcc\xdd\xee" + struct.pack(">H", et) + l3

    assert add("l2_udp_ok", eth(mac, 0x0800, good_udp), tap, l2=True) == "ACCEPT_UDP"
    assert add("l2_udp_ok_ip_options", eth(mac, 0x0800, ip_header(17, other, me, 108, ihl=9, options=prng("o9", 16)) + udp_segment(other, me, 5000, 8000, pay)), tap, l2=True) == "ACCEPT_UDP"
    assert add("l2_broadcast_mac_udp_ok", eth(b"\xff" * 6, 0x0800, good_udp), tap, l2=True) == "ACCEPT_UDP"
    short_udp = ip_header(17, other, me, 8 + 4) + udp_segment(other, me, 5, 6, b"ping")
    assert add("l2_min_frame_padding_breaks_udp_len", eth(mac, 0x0800, short_udp + bytes(46 - len(short_udp))), tap, l2=True) == "DROP_UDP_LEN"
    assert add("l2_icmp_echo_98", eth(mac, 0x0800, ip_header(1, 0xC0000201, me, 64) + echo), tap, l2=True) == "ACCEPT_ICMP_ECHO"
    assert add("l2_tcp_ext_ok", eth(mac, 0x0800, tcp_ok), tcp_ifc, l2=True) == "ACCEPT_TCP"
    # TCP options that lie beyond the first 96 bytes of the record: IPv4 options (IHL 14) in front, 40 option bytes
    deep_ok = bytes([1] * 26 + [8, 10]) + prng("d1", 8) + bytes([1, 1, 1, 0])
    deep_bad = bytes([1] * 36 + [3, 4, 0, 0])  # window scale with length 4, at option bytes 36..39
    for nm, opts, want in (("deep_ok", deep_ok, "ACCEPT_TCP"), ("deep_bad", deep_bad, "DROP_TCP_OPTION")):
        seg = tcp_segment(other, me, 443, 50000, pay + pay, doff=15, opts=opts)
        pkt = ip_header(6, other, me, len(seg), ihl=14, options=prng("opt14", 36)) + seg
        assert add("tcp_opt_" + nm + "_l3", pkt, tcp_ifc) == want
        assert add("tcp_opt_" + nm + "_l2", eth(mac, 0x0800, pkt), tcp_ifc, l2=True) == want
    assert add("l2_truncated_13", eth(mac, 0x0800, b"")[:13], tap, l2=True) == "DROP_L2_TRUNCATED"
    assert add("l2_ethertype_ipv6", eth(mac, 0x86DD, good_udp), tap, l2=True) == "DROP_L2_ETHERTYPE"
    assert add("l2_ethertype_checked_before_mac", eth(b"\x02\x01\x02\x03\x04\x05", 0x86DD, good_udp), tap, l2=True) == "DROP_L2_ETHERTYPE"
    assert add("l2_other_mac", eth(b"\x02\x01\x02\x03\x04\x05", 0x0800, good_udp), tap, l2=True) == "DROP_L2_NOT_FOR_ME"
    assert add("l2_mac_differs_in_last_byte", eth(mac[:5] + b"\x31", 0x0800, good_udp), tap, l2=True) == "DROP_L2_NOT_FOR_ME"
    def arp(oper, spa, tpa, htype=1, ptype=0x0800, hlen=6, plen=4):
        return struct.pack(">HHBBH", htype, ptype, hlen, plen, oper) + b"\x02\xaa\xbb\xcc\xdd\xee" + struct.pack(">I", spa) + bytes(6) + struct.pack(">I", tpa)

    assert add("l2_arp_request_for_me", eth(b"\xff" * 6, 0x0806, arp(1, 0xC0000201, me)), tap, l2=True) == "PASS_ARP"
    assert add("l2_arp_reply_for_me_padded_to_60", eth(mac, 0x0806, arp(2, 0xC0000201, me) + bytes(18)), tap, l2=True) == "PASS_ARP"
    assert add("l2_arp_request_for_other_host", eth(b"\xff" * 6, 0x0806, arp(1, 0xC0000201, 0xC0000207)), tap, l2=True) == "IGNORE_ARP_NOT_FOR_ME"
    assert add("l2_arp_other_mac", eth(b"\x02\x01\x02\x03\x04\x05", 0x0806, arp(1, 0xC0000201, me)), tap, l2=True) == "DROP_L2_NOT_FOR_ME"
    assert add("l2_arp_htype_6", eth(mac, 0x0806, arp(1, 0xC0000201, me, htype=6)), tap, l2=True) == "DROP_ARP_HTYPE"
    assert add("l2_arp_ptype_ipv6", eth(mac, 0x0806, arp(1, 0xC0000201, me, ptype=0x86DD)), tap, l2=True) == "DROP_ARP_PTYPE"
    assert add("l2_arp_hlen_8", eth(mac, 0x0806, arp(1, 0xC0000201, me, hlen=8)), tap, l2=True) == "DROP_ARP_HLEN"
    assert add("l2_arp_plen_16", eth(mac, 0x0806, arp(1, 0xC0000201, me, plen=16)), tap, l2=True) == "DROP_ARP_PLEN"
    assert add("l2_arp_htype_checked_before_ptype", eth(mac, 0x0806, arp(1, 0xC0000201, me, htype=2, ptype=1, hlen=1, plen=1)), tap, l2=True) == "DROP_ARP_HTYPE"
    assert add("l2_arp_truncated_7", eth(mac, 0x0806, arp(1, 0xC0000201, me)[:7]), tap, l2=True) == "DROP_ARP_TRUNCATED"
    assert add("l2_arp_truncated_27", eth(mac, 0x0806, arp(1, 0xC0000201, me)[:27]), tap, l2=True) == "DROP_ARP_TRUNCATED"
    assert add("l2_arp_bad_plen_short_buffer", eth(mac, 0x0806, arp(1, 0xC0000201, me, plen=5)[:12]), tap, l2=True) == "DROP_ARP_PLEN"
    assert add("l2_arp_empty", eth(mac, 0x0806, b""), tap, l2=True) == "DROP_ARP_TRUNCATED"
    assert add("l2_header_only", eth(mac, 0x0800, b""), tap, l2=True) == "DROP_L3_TRUNCATED"
    assert add("l2_jumbo_9000_udp", eth(mac, 0x0800, ip_header(17, other, me, 8 + 8958) + udp_segment(other, me, 9, 10, prng("jumbo", 8958))), tap, l2=True, tail=("jumbo", 8958)) == "ACCEPT_UDP"

    big = prng("big", 65535 - 28)
    assert add("udp_max_65535", ip_header(17, other, me, 65535 - 20) + udp_segment(other, me, 1, 2, big), tap, tail=("big", 65535 - 28)) == "ACCEPT_UDP"
    assert add("udp_4096", ip_header(17, other, me, 4096 - 20) + udp_segment(other, me, 1, 2, big[:4096 - 28]), tap, tail=("big", 4096 - 28)) == "ACCEPT_UDP"
    assert add("udp_1500", ip_header(17, other, me, 1480) + udp_segment(other, me, 1, 2, big[:1472]), tap, tail=("big", 1472)) == "ACCEPT_UDP"

    seen = {v["verdict"] for v in vecs}
    missing = set(V) - seen - {"NONE", "L3_HEADER_OK"}
    assert not missing, missing
    with open(os.path.join(HERE, "verdict_vectors.json"), "w") as f:
        json.dump({"verdict_codes": V, "vectors": vecs}, f, indent=1)
    print(f"wrote {len(vecs)} verdict vectors, {len(edges)} checksum edges, 5 reference KATs")


if __name__ == "__main__":
    main()
