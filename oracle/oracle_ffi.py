"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

RESULT_DTYPE = np.dtype([("csum", "<u2"), ("verdict", "u1"), ("proto", "u1")])


class IfaceC(C.Structure):
    _fields_ = [("unicast", C.c_uint32), ("broadcast", C.c_uint32), ("mac", C.c_uint8 * 6), ("mode", C.c_uint8),
                ("ext_tcp", C.c_uint8)]


class ResultC(C.Structure):
    _fields_ = [("csum", C.c_uint16), ("verdict", C.c_uint8), ("proto", C.c_uint8)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.oracle_calculate_checksum.restype = C.c_uint16
        L.oracle_calculate_checksum.argtypes = [C.c_char_p, C.c_size_t, C.c_uint16]
        L.oracle_calculate_checksum_fast.restype = C.c_uint16
        L.oracle_calculate_checksum_fast.argtypes = [C.c_char_p, C.c_size_t, C.c_uint16]
        L.oracle_iface_broadcast.restype = C.c_uint32
        L.oracle_iface_broadcast.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_ipv4_header_validate.restype = ResultC
        L.oracle_ipv4_header_validate.argtypes = [C.c_char_p, C.c_int]
        for name in ("oracle_ipv4_recv", "oracle_ethernet_recv", "oracle_recv"):
            f = getattr(L, name)
            f.restype = ResultC
            f.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(IfaceC)]
        L.oracle_udp_send_checksum.restype = C.c_uint16
        L.oracle_udp_send_checksum.argtypes = [C.c_uint32, C.c_uint32, C.c_uint16, C.c_uint16, C.c_char_p, C.c_size_t]
        L.oracle_icmp_send_checksum.restype = C.c_uint16
        L.oracle_icmp_send_checksum.argtypes = [C.c_uint8, C.c_uint8, C.c_uint32, C.c_char_p, C.c_size_t]
        L.oracle_create_ip_header.restype = None
        L.oracle_create_ip_header.argtypes = [C.c_uint16, C.c_uint8, C.c_uint32, C.c_uint32, C.c_size_t, C.c_char_p]
        L.oracle_generate.restype = C.c_uint8
        L.oracle_generate.argtypes = [vp, C.c_size_t]
        L.oracle_recv_batch.restype = None
        L.oracle_recv_batch.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.POINTER(IfaceC), vp]
        L.oracle_recv_batch_fast.restype = None
        L.oracle_recv_batch_fast.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.POINTER(IfaceC), vp]
        L.oracle_csum_batch.restype = None
        L.oracle_csum_batch.argtypes = [vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp]
        L.oracle_ipv4_validate_batch.restype = None
        L.oracle_ipv4_validate_batch.argtypes = [vp, C.c_uint32, C.c_int, vp]
        L.oracle_generate_batch.restype = None
        L.oracle_generate_batch.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, vp]
        L.oracle_tx_build_batch.restype = None
        L.oracle_tx_build_batch.argtypes = [vp, vp, C.c_uint32, vp, vp, C.c_uint32, C.POINTER(IfaceC), vp, vp, vp, vp]
        L.oracle_flow_hash5.restype = C.c_uint32
        L.oracle_flow_hash5.argtypes = [C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint16, C.c_uint16]
        L.oracle_flow_hash_record.restype = C.c_uint32
        L.oracle_flow_hash_record.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(IfaceC)]
        L.oracle_tcp_track.restype = None
        L.oracle_tcp_track.argtypes = [vp, vp, C.c_uint32, vp, vp, C.c_uint32, C.POINTER(IfaceC), vp, C.c_uint32, vp]
        L.oracle_tcp_slot.restype = C.c_uint32
        L.oracle_tcp_slot.argtypes = [C.c_uint32, C.c_uint32, C.c_uint16, C.c_uint16, C.c_uint32]
        _lib = L
    return _lib


def iface_c(iface) -> IfaceC:
    """Accepts a unet_amd.Iface-like object (unicast, broadcast, mac, mode, ext_tcp)."""
    s = IfaceC()
    s.unicast = iface.unicast
    s.broadcast = iface.broadcast
    for i, b in enumerate(bytes(iface.mac)):
        s.mac[i] = b
    s.mode = int(iface.mode)
    s.ext_tcp = 1 if iface.ext_tcp else 0
    return s


def calculate_checksum(data: bytes, sum: int = 0) -> int:
    data = bytes(data)
    return int(lib().oracle_calculate_checksum(data, len(data), sum & 0xFFFF))


def calculate_checksum_fast(data: bytes, sum: int = 0) -> int:
    data = bytes(data)
    return int(lib().oracle_calculate_checksum_fast(data, len(data), sum & 0xFFFF))


def _res(r: ResultC):
    return (int(r.verdict), int(r.proto), int(r.csum))


def ipv4_header_validate(hdr20: bytes, ext_tcp: bool = False):
    assert len(hdr20) >= 20
    return _res(lib().oracle_ipv4_header_validate(bytes(hdr20), 1 if ext_tcp else 0))


def recv(rec: bytes, iface):
    rec = bytes(rec)
    ic = iface_c(iface)
    return _res(lib().oracle_recv(rec, len(rec), C.byref(ic)))


def recv_batch(arena: np.ndarray, off16, lens: np.ndarray, iface, stride16: int = 0, fast: bool = False) -> np.ndarray:
    """fast=True computes the same verdicts with the 64-bit-accumulator checksum."""
    arena = np.ascontiguousarray(arena, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    if off16 is not None:
        off16 = np.ascontiguousarray(off16, dtype=np.uint32)
    out = np.zeros(len(lens), dtype=RESULT_DTYPE)
    ic = iface_c(iface)
    fn = lib().oracle_recv_batch_fast if fast else lib().oracle_recv_batch
    fn(arena.ctypes.data, None if off16 is None else off16.ctypes.data, stride16, lens.ctypes.data,
       len(lens), C.byref(ic), out.ctypes.data)
    return out


def csum_batch(arena: np.ndarray, off16, lens: np.ndarray, seeds=None, stride16: int = 0) -> np.ndarray:
    arena = np.ascontiguousarray(arena, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    if off16 is not None:
        off16 = np.ascontiguousarray(off16, dtype=np.uint32)
    if seeds is not None:
        seeds = np.ascontiguousarray(seeds, dtype=np.uint16)
    out = np.zeros(len(lens), dtype=np.uint16)
    lib().oracle_csum_batch(arena.ctypes.data, None if off16 is None else off16.ctypes.data, stride16, lens.ctypes.data,
                            None if seeds is None else seeds.ctypes.data, len(lens), out.ctypes.data)
    return out


def ipv4_validate_batch(hdrs: np.ndarray, ext_tcp: bool = False) -> np.ndarray:
    hdrs = np.ascontiguousarray(hdrs, dtype=np.uint8).reshape(-1)
    n = hdrs.size // 20
    out = np.zeros(n, dtype=RESULT_DTYPE)
    lib().oracle_ipv4_validate_batch(hdrs.ctypes.data, n, 1 if ext_tcp else 0, out.ctypes.data)
    return out


def generate_batch(arena: np.ndarray, off16, lens: np.ndarray, stride16: int = 0):
    arena = np.ascontiguousarray(arena, dtype=np.uint8).copy()
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    if off16 is not None:
        off16 = np.ascontiguousarray(off16, dtype=np.uint32)
    status = np.zeros(len(lens), dtype=np.uint8)
    lib().oracle_generate_batch(arena.ctypes.data, None if off16 is None else off16.ctypes.data, stride16,
                                lens.ctypes.data, len(lens), status.ctypes.data)
    return arena, status


def udp_send_checksum(src: int, dst: int, sport: int, dport: int, payload: bytes) -> int:
    payload = bytes(payload)
    return int(lib().oracle_udp_send_checksum(src, dst, sport, dport, payload, len(payload)))


def icmp_send_checksum(ty: int, code: int, values: int, data: bytes) -> int:
    data = bytes(data)
    return int(lib().oracle_icmp_send_checksum(ty, code, values, data, len(data)))


def create_ip_header(ident: int, protocol: int, src: int, dst: int, data_len: int) -> bytes:
    buf = C.create_string_buffer(20)
    lib().oracle_create_ip_header(ident, protocol, src, dst, data_len, buf)
    return buf.raw


def tx_build_batch(pay_arena: np.ndarray, pay_off16, pay_lens: np.ndarray, desc: np.ndarray, iface, out_off16: np.ndarray,
                   out_bytes: int, pay_stride16: int = 0, fill: int = 0):
    """Returns (out_arena, out_len, status); bytes the oracle does not write keep `fill`."""
    pay_arena = np.ascontiguousarray(pay_arena, dtype=np.uint8)
    pay_lens = np.ascontiguousarray(pay_lens, dtype=np.uint32)
    if pay_off16 is not None:
        pay_off16 = np.ascontiguousarray(pay_off16, dtype=np.uint32)
    out_off16 = np.ascontiguousarray(out_off16, dtype=np.uint32)
    desc = np.ascontiguousarray(desc)
    assert desc.dtype.itemsize == 28
    n = len(pay_lens)
    out = np.full(out_bytes + 64, fill, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.uint8)
    ic = iface_c(iface)
    lib().oracle_tx_build_batch(pay_arena.ctypes.data, None if pay_off16 is None else pay_off16.ctypes.data, pay_stride16,
                                pay_lens.ctypes.data, desc.ctypes.data, n, C.byref(ic), out.ctypes.data, out_off16.ctypes.data,
                                out_len.ctypes.data, status.ctypes.data)
    return out[:out_bytes], out_len, status


def flow_hash5(src, dst, proto, sport, dport) -> int:
    return int(lib().oracle_flow_hash5(src, dst, proto, sport, dport))


def flow_hash_record(rec: bytes, iface) -> int:
    rec = bytes(rec)
    ic = iface_c(iface)
    return int(lib().oracle_flow_hash_record(rec, len(rec), C.byref(ic)))


TCP_FLOW_DTYPE = np.dtype([("src", "<u4"), ("dst", "<u4"), ("ports", "<u4"), ("next_seq", "<u4"), ("state", "<u4"), ("segments", "<u4"),
                           ("out_of_order", "<u4"), ("after_close", "<u4")])


def tcp_track(arena: np.ndarray, off16, lens: np.ndarray, results: np.ndarray, iface, table: np.ndarray, stride16: int = 0) -> np.ndarray:
    """Sequential per-flow TCP tracking (extension): updates `table` (TCP_FLOW_DTYPE, power-of-two length) in place and
    returns the per-record segment states."""
    arena = np.ascontiguousarray(arena, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    if off16 is not None:
        off16 = np.ascontiguousarray(off16, dtype=np.uint32)
    results = np.ascontiguousarray(results)
    assert table.dtype == TCP_FLOW_DTYPE and table.flags.c_contiguous
    out = np.zeros(len(lens), dtype=np.uint8)
    ic = iface_c(iface)
    lib().oracle_tcp_track(arena.ctypes.data, None if off16 is None else off16.ctypes.data, stride16, lens.ctypes.data,
                           results.ctypes.data, len(lens), C.byref(ic), table.ctypes.data, len(table), out.ctypes.data)
    return out


def tcp_slot(src, dst, sport, dport, slots) -> int:
    return int(lib().oracle_tcp_slot(src, dst, sport, dport, slots))
