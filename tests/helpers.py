"""Shared helpers for the tests: golden-vector loading and small record builders."""
import hashlib
import json
import os
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def prng(tag: str, n: int) -> bytes:
    out = b""
    i = 0
    while len(out) < n:
        out += hashlib.sha256(f"{tag}:{i}".encode()).digest()
        i += 1
    return out[:n]


def edge_data(e) -> bytes:
    if "data" in e:
        return bytes.fromhex(e["data"])
    if "prng" in e:
        return prng(e["prng"]["tag"], e["prng"]["len"])
    return bytes([e["fill"]["byte"]]) * e["fill"]["len"]


def vector_record(v) -> bytes:
    rec = bytes.fromhex(v["record"])
    if "tail_prng" in v:
        rec += prng(v["tail_prng"]["tag"], v["tail_prng"]["len"])
    return rec


def vector_iface(v):
    """Golden iface dict -> object usable by both unet_amd (has .c()) and the oracle binding."""
    import unet_amd

    i = v["iface"]
    o = unet_amd.Iface(unicast=i["unicast"], netmask=i["netmask"], mac=bytes(i["mac"]), mode=i["mode"], ext_tcp=i["ext_tcp"])
    assert o.broadcast == i["broadcast"]
    return o


def iface_key(i):
    return (i["unicast"], i["netmask"], tuple(i["mac"]), i["mode"], i["ext_tcp"])


def as_tuples(res: np.ndarray):
    return [(int(r["verdict"]), int(r["proto"]), int(r["csum"])) for r in res]


def random_lengths(rng, n, max_len=9000):
    """Length mix that hits the kernels' edges: empty, sub-chunk, the hand-over points between the per-lane,
    16-lane-row and whole-wave paths, row and loop-step boundaries of long records, the maximum."""
    special = np.array([0, 1, 2, 3, 15, 16, 17, 19, 20, 21, 27, 28, 29, 63, 64, 65, 127, 128, 129, 143, 144, 145, 175, 176, 177,
                        191, 192, 193, 207, 208, 209, 255, 256, 257, 271, 272, 273, 511, 512, 513, 527, 528, 529, 767, 768, 769,
                        1007, 1008, 1009, 1023, 1024, 1025, 1039, 1040, 1041, 1499, 1500, 1501, 2031, 2032, 2033, 2047, 2048, 2049,
                        2063, 2064, 2065, 3071, 3072, 3073, 4095, 4096, 4097, 5119, 5120, 5121, 6143, 6144, 6145, 6159, 6160,
                        6161, 7167, 7168, 7169, 8999, 9000, 10239, 10240, 10241, 65519, 65520, 65521, 65534, 65535])
    special = special[special <= max_len]
    lens = rng.integers(0, max_len + 1, size=n)
    k = min(n, len(special))
    lens[:k] = special[:k]
    small = rng.integers(0, 200, size=n)
    pick = rng.random(n) < 0.4
    lens[k:] = np.where(pick[k:], small[k:], lens[k:])
    return lens.astype(np.uint32)


def py_ck(data: bytes, seed: int = 0) -> int:
    """src/utils.rs:1-17 in plain Python (third, independent implementation used by a few tests)."""
    total = seed
    for i in range(0, len(data), 2):
        total += (data[i] << 8) | (data[i + 1] if i + 1 < len(data) else 0)
    while total >> 16:
        total = (total & 0xFFFF) + (total >> 16)
    return (~total) & 0xFFFF


def config1_echo_frame():
    """BASELINE configs[0] stand-in: the 98-byte Ethernet frame of a 64-byte ICMP echo request sent by
    the tap peer 192.0.2.1 to the stack's 192.0.2.2 (src/app.rs:42-54, setup_tap.sh)."""
    import struct
    tap_mac, peer_mac = b"\x02\x00\x5e\x10\x20\x30", b"\x02\xaa\xbb\xcc\xdd\xee"
    data = bytes(range(56))
    icmp = bytearray(struct.pack(">BBHI", 8, 0, 0, 0x12340001) + data)
    struct.pack_into(">H", icmp, 2, py_ck(bytes(icmp)))
    ip = bytearray(struct.pack(">BBHHHBBHII", 0x45, 0, 84, 0xBEEF, 0x4000, 64, 1, 0, 0xC0000201, 0xC0000202))
    struct.pack_into(">H", ip, 10, py_ck(bytes(ip)))
    frame = tap_mac + peer_mac + b"\x08\x00" + bytes(ip) + bytes(icmp)
    assert len(frame) == 98
    return frame, tap_mac, peer_mac
