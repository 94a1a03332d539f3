"""Property tests (hypothesis).  CPU: the C oracle against the independent pure-Python model of
tests/golden/make_golden.py on generated inputs, and the algebraic facts of calculate_checksum
(src/utils.rs:1-17) that udp::recv / udp::send rely on.  GPU: the generated cases judged in one batch."""
import importlib.util
import os
import struct

import numpy as np
import pytest
from hypothesis import HealthCheck, given, seed, settings
from hypothesis import strategies as st

import unet_amd
from unet_amd import Verdict
from helpers import as_tuples
from oracle import oracle_ffi as O

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
M = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(M)

PROFILE = dict(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
UNICAST, MAC = 0xC0000202, bytes.fromhex("02aabbccddee")


def model_iface(mode, ext_tcp):
    return {"unicast": UNICAST, "broadcast": 0xC00002FF, "mac": MAC, "ext_tcp": ext_tcp, "mode": mode}


def build_iface(mode, ext_tcp):
    return unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=MAC, mode=mode, ext_tcp=ext_tcp)


def model_recv(rec: bytes, mode, ext_tcp):
    ifc = model_iface(mode, ext_tcp)
    name, proto, csum = (M.eth_recv if mode else M.ipv4_recv)(rec, ifc)
    return (M.V[name], proto, csum)


# ------------------------------------------------------------------ calculate_checksum

@settings(**PROFILE)
@given(st.binary(max_size=4096), st.integers(0, 0xFFFF))
def test_checksum_three_implementations_agree(data, sd):
    a = O.calculate_checksum(data, sd)
    assert a == O.calculate_checksum_fast(data, sd) == M.ck(data, sd)


@settings(**PROFILE)
@given(st.binary(max_size=2048))
def test_checksum_of_data_seeded_with_its_checksum_is_zero(data):
    """The idiom of the reference's own tests (src/utils.rs:49,59)."""
    assert O.calculate_checksum(data, O.calculate_checksum(data, 0)) == 0


@settings(**PROFILE)
@given(st.binary(min_size=2, max_size=2048), st.data())
def test_split_and_chain_equals_one_shot(data, d):
    cut = 2 * d.draw(st.integers(0, len(data) // 2))
    part = O.calculate_checksum(data[:cut], 0)
    assert O.calculate_checksum(data[cut:], ~part & 0xFFFF) == O.calculate_checksum(data, 0)


@settings(**PROFILE)
@given(st.binary(max_size=1024), st.binary(min_size=12, max_size=12))
def test_pseudo_header_chaining_commutes(seg, pseudo):
    """udp::recv sums the segment first and the pseudo-header second (udp.rs:229,247); udp::send the other
    way round (udp.rs:200-202).  Both equal one pass over pseudo || segment when the segment is even."""
    s = O.calculate_checksum(seg, 0)
    recv_order = O.calculate_checksum(pseudo, ~s & 0xFFFF)
    p = O.calculate_checksum(pseudo, 0)
    send_order = O.calculate_checksum(seg, ~p & 0xFFFF)
    assert recv_order == send_order == O.calculate_checksum(pseudo + seg, 0)


@settings(max_examples=25, deadline=None)
@given(st.integers(60000, 65535), st.integers(0, 255), st.integers(0, 0xFFFF))
def test_checksum_never_overflows_at_maximum_length(n, fill, sd):
    data = bytes([fill]) * n
    assert O.calculate_checksum(data, sd) == O.calculate_checksum_fast(data, sd) == M.ck(data, sd)


# ------------------------------------------------------------------ verdicts on generated packets

@st.composite
def l3_packets(draw):
    proto = draw(st.sampled_from([1, 17, 17, 6, 47]))
    ihl = draw(st.sampled_from([5, 5, 5, 6, 8, 14, 15, 4, 0]))
    version = draw(st.sampled_from([4, 4, 4, 4, 6, 0]))
    ff = draw(st.sampled_from([0, 0, 0, 0x4000, 0x8000, 0x2000, 0x0001, 0x1FFF]))
    dst = draw(st.sampled_from([UNICAST, UNICAST, 0xC00002FF, 0xFFFFFFFF, 0xC0000203, 0]))
    src = draw(st.integers(0, 0xFFFFFFFF))
    body = draw(st.binary(max_size=200))
    opts = bytes(max(ihl - 5, 0) * 4)
    if proto == 17:
        l4 = M.udp_segment(src, dst, draw(st.integers(0, 65535)), draw(st.integers(0, 65535)), body,
                           length=draw(st.sampled_from([None, None, None, 8, 0, len(body)])),
                           csum=draw(st.sampled_from([None, None, None, 0, 0xFFFF])))
    elif proto == 1:
        l4 = M.icmp_message(draw(st.sampled_from([8, 0, 8, 3, 11])), draw(st.integers(0, 255)), draw(st.integers(0, 0xFFFFFFFF)), body)
    elif proto == 6:
        l4 = M.tcp_segment(src, dst, 1, 2, body, doff=draw(st.sampled_from([5, 5, 6, 4, 15])), opts=bytes(4) * draw(st.integers(0, 2)))
    else:
        l4 = body
    hdr = M.ip_header(proto, src, dst, len(l4), ihl=ihl, version=version, flags_frag=ff, options=opts,
                      fix_csum=draw(st.sampled_from([True, True, True, False])))
    pkt = bytearray(hdr + l4)
    if draw(st.booleans()) and len(pkt):
        k = draw(st.integers(0, len(pkt) - 1))
        if draw(st.integers(0, 3)) == 0:
            pkt[k] ^= 1 << draw(st.integers(0, 7))   # single bit error anywhere
        elif draw(st.integers(0, 3)) == 0:
            pkt = pkt[:k]                            # truncation anywhere
    return bytes(pkt)


@st.composite
def frames(draw):
    kind = draw(st.sampled_from(["ip", "ip", "ip", "arp", "other", "short"]))
    dstmac = draw(st.sampled_from([MAC, MAC, b"\xff" * 6, bytes.fromhex("02aabbccddef")]))
    srcmac = draw(st.binary(min_size=6, max_size=6))
    if kind == "ip":
        return dstmac + srcmac + b"\x08\x00" + draw(l3_packets())
    if kind == "arp":
        arp = bytearray(struct.pack(">HHBBH", 1, 0x0800, 6, 4, draw(st.sampled_from([1, 2]))) + srcmac + struct.pack(">I", 0xC0000201) +
                        bytes(6) + struct.pack(">I", draw(st.sampled_from([UNICAST, UNICAST, 0xC0000209]))))
        if draw(st.booleans()):
            arp[draw(st.integers(0, 5))] ^= draw(st.sampled_from([1, 4, 0x80]))
        if draw(st.integers(0, 4)) == 0:
            arp = arp[:draw(st.integers(0, 27))]
        return dstmac + srcmac + b"\x08\x06" + bytes(arp) + bytes(draw(st.sampled_from([0, 18])))
    if kind == "other":
        return dstmac + srcmac + draw(st.sampled_from([b"\x86\xdd", b"\x08\x01", b"\x00\x00"])) + draw(st.binary(max_size=64))
    return draw(st.binary(max_size=13))


@settings(**PROFILE)
@given(l3_packets(), st.booleans())
def test_oracle_equals_python_model_l3(pkt, ext_tcp):
    assert O.recv(pkt, build_iface(0, ext_tcp)) == model_recv(pkt, 0, ext_tcp)


@settings(**PROFILE)
@given(frames(), st.booleans())
def test_oracle_equals_python_model_l2(frame, ext_tcp):
    assert O.recv(frame, build_iface(1, ext_tcp)) == model_recv(frame, 1, ext_tcp)


@settings(**PROFILE)
@given(l3_packets())
def test_generate_makes_well_formed_packets_verify(pkt):
    """Whatever generate accepts (status 0) is afterwards judged without any checksum verdict."""
    arena, off16, lens = unet_amd.pack_arena([pkt])
    filled, status = O.generate_batch(arena, off16, lens)
    if status[0] == 0:
        v = O.recv_batch(filled, off16, lens, build_iface(0, False))["verdict"][0]
        assert v not in (Verdict.DROP_L3_CSUM, Verdict.DROP_UDP_CSUM)
        again, status2 = O.generate_batch(filled, off16, lens)
        assert status2[0] == 0 and (again == filled).all()  # idempotent


# ------------------------------------------------------------------ the same generators on the GPU

@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("ext_tcp", [False, True])
def test_generated_cases_on_gpu(offload, mode, ext_tcp):
    collected = []

    @seed(20240 + mode * 2 + int(ext_tcp))
    @settings(max_examples=1500, deadline=None, database=None, suppress_health_check=list(HealthCheck))
    @given(frames() if mode else l3_packets())
    def collect(rec):
        collected.append(rec)

    collect()
    assert len(collected) >= 1000
    ifc = build_iface(mode, ext_tcp)
    arena, off16, lens = unet_amd.pack_arena(collected, pad_byte=0x5C)
    got = as_tuples(offload.verify(arena, off16, lens, ifc))
    exp = as_tuples(O.recv_batch(arena, off16, lens, ifc))
    assert got == exp
    assert got == [model_recv(r, mode, ext_tcp) for r in collected]
    assert len({g[0] for g in got}) >= (12 if mode else 10)  # the generators reach most verdict codes
