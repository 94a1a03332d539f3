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
    """Length mix that hits the kernel's edges: empty, sub-chunk, around 16/1024-byte boundaries, jumbo."""
    special = np.array([0, 1, 2, 3, 15, 16, 17, 19, 20, 21, 27, 28, 29, 63, 64, 65, 1007, 1008, 1009, 1023, 1024, 1025,
                        1039, 1040, 1041, 1499, 1500, 1501, 2047, 2048, 2049, 4095, 4096, 4097, 5119, 5120, 5121, 8999, 9000])
    special = special[special <= max_len]
    lens = rng.integers(0, max_len + 1, size=n)
    k = min(n, len(special))
    lens[:k] = special[:k]
    small = rng.integers(0, 200, size=n)
    pick = rng.random(n) < 0.4
    lens[k:] = np.where(pick[k:], small[k:], lens[k:])
    return lens.astype(np.uint32)
