"""ctypes binding of libunet_offload.so (include/unet_offload.h)."""
from __future__ import annotations

import ctypes as C
import enum
import os
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libunet_offload.so")

RESULT_DTYPE = np.dtype([("csum", "<u2"), ("verdict", "u1"), ("proto", "u1")])


class UnetError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"unet_offload error {code}: {msg}")
        self.code = code


class Verdict(enum.IntEnum):
    NONE = 0
    DROP_L2_TRUNCATED = 1
    DROP_L2_ETHERTYPE = 2
    DROP_L2_NOT_FOR_ME = 3
    PASS_ARP = 4
    DROP_L3_TRUNCATED = 5
    DROP_L3_PROTO = 6
    DROP_L3_VERSION = 7
    DROP_L3_IHL_SHORT = 8
    DROP_L3_IHL_LONG = 9
    DROP_L3_FRAGMENT = 10
    DROP_L3_CSUM = 11
    IGNORE_L3_NOT_FOR_ME = 12
    DROP_ICMP_TRUNCATED = 13
    DROP_ICMP_TYPE = 14
    ACCEPT_ICMP_ECHO = 15
    ACCEPT_ICMP_ECHOREPLY = 16
    DROP_UDP_SHORT = 17
    DROP_UDP_LEN = 18
    DROP_UDP_CSUM = 19
    ACCEPT_UDP = 20
    DROP_TCP_SHORT = 21
    DROP_TCP_DOFF = 22
    DROP_TCP_CSUM = 23
    ACCEPT_TCP = 24
    L3_HEADER_OK = 25
    DROP_ARP_TRUNCATED = 26
    DROP_ARP_HTYPE = 27
    DROP_ARP_PTYPE = 28
    DROP_ARP_HLEN = 29
    DROP_ARP_PLEN = 30
    IGNORE_ARP_NOT_FOR_ME = 31
    DROP_TCP_OPTION = 32


V_COUNT = 40


class _IfaceC(C.Structure):
    _fields_ = [
        ("unicast", C.c_uint32),
        ("broadcast", C.c_uint32),
        ("mac", C.c_uint8 * 6),
        ("mode", C.c_uint8),
        ("ext_tcp", C.c_uint8),
    ]


def _ip(addr) -> int:
    if isinstance(addr, int):
        return addr & 0xFFFFFFFF
    a, b, c, d = (int(x) for x in str(addr).split("."))
    return (a << 24) | (b << 16) | (c << 8) | d


class Iface:
    """Receiving interface constants (Ipv4Interface::new, src/protocols/ipv4.rs:218-229)."""

    def __init__(self, unicast="192.0.2.2", netmask="255.255.255.0", mac=b"\x02\x00\x00\x00\x00\x01",
                 mode: int = 0, ext_tcp: bool = False):
        self.unicast = _ip(unicast)
        self.netmask = _ip(netmask)
        self.broadcast = (self.unicast & 0xFFFFFF00) | (~self.netmask & 0xFF)
        self.mac = bytes(mac)
        assert len(self.mac) == 6
        self.mode = int(mode)
        self.ext_tcp = bool(ext_tcp)

    def c(self) -> _IfaceC:
        s = _IfaceC()
        s.unicast = self.unicast
        s.broadcast = self.broadcast
        for i, b in enumerate(self.mac):
            s.mac[i] = b
        s.mode = self.mode
        s.ext_tcp = 1 if self.ext_tcp else 0
        return s

    def replace(self, **kw) -> "Iface":
        o = Iface.__new__(Iface)
        o.__dict__.update(self.__dict__)
        o.__dict__.update(kw)
        return o


# unet_tx_desc, 28 bytes
TX_DESC_DTYPE = np.dtype([("src", "<u4"), ("dst", "<u4"), ("l4a", "<u2"), ("l4b", "<u2"), ("values", "<u4"), ("ip_id", "<u2"),
                          ("proto", "u1"), ("reserved0", "u1"), ("dst_mac", "u1", (6,)), ("reserved1", "u1", (2,))])
TX_OK, TX_E_PROTO, TX_E_TOO_LONG, TX_E_RANGE = 0, 1, 2, 3
TX_FILL_LINES = 1  # unet_tx_build_opts_dev: packets' last 128-byte lines may be written whole (zeros behind the packet)

# unet_tcp_flow, 32 bytes
TCP_FLOW_DTYPE = np.dtype([("src", "<u4"), ("dst", "<u4"), ("ports", "<u4"), ("next_seq", "<u4"), ("state", "<u4"), ("segments", "<u4"),
                           ("out_of_order", "<u4"), ("after_close", "<u4")])
TCP_FREE, TCP_SYN, TCP_ESTABLISHED, TCP_FIN, TCP_RESET = 0, 1, 2, 3, 4
TCPS_NOT_TCP, TCPS_NEW, TCPS_IN_ORDER, TCPS_OUT_OF_ORDER, TCPS_AFTER_CLOSE = 0, 1, 2, 3, 4


def tx_frame_len(payload_len, mode: int):
    """Length of the packet unet_tx_build_dev writes for a payload: 28 + payload in L3 mode; in L2 mode the
    Ethernet header plus the L3 packet padded to 60 bytes (src/driver/tap.rs:147-152)."""
    l3 = np.asarray(payload_len, dtype=np.int64) + 28
    return l3 if mode == 0 else 14 + np.maximum(l3, 60)


def tx_out_layout(payload_len, mode: int, align16: int = 1):
    """(out_off16, out_arena_bytes) for packing the built packets back to back on 16-byte boundaries, or on
    multiples of align16 such slots (8 = 128-byte lines)."""
    slots = ((tx_frame_len(payload_len, mode) + 15) // 16 + align16 - 1) // align16 * align16
    off16 = np.zeros(len(slots), dtype=np.int64)
    np.cumsum(slots[:-1], out=off16[1:])
    return off16.astype(np.uint32), int(max(16, 16 * int(slots.sum())))


_lib = None

_vp, _u8p, _u16p, _u32p, _u64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p

_SIGS = {
    "unet_open": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "unet_close": (None, [C.c_void_p]),
    "unet_last_error": (C.c_char_p, []),
    "unet_device_count": (C.c_int, []),
    "unet_sync": (C.c_int, [C.c_void_p]),
    "unet_csum_batch_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, _u16p, C.c_uint32, _u16p]),
    "unet_csum_batch": (C.c_int, [_vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, _u16p, C.c_uint32, _u16p]),
    "unet_csum": (C.c_int, [_vp, C.c_char_p, C.c_size_t, C.c_uint16, C.POINTER(C.c_uint16)]),
    "unet_ipv4_validate_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint32, C.POINTER(_IfaceC), _vp]),
    "unet_ipv4_validate": (C.c_int, [_vp, _u8p, C.c_uint32, C.POINTER(_IfaceC), _vp]),
    "unet_verify_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, C.c_uint32, C.POINTER(_IfaceC), _vp]),
    "unet_verify_split_workspace": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "unet_verify_split_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, C.c_uint32, C.c_uint32, C.POINTER(_IfaceC), _vp, _vp,
                                        C.c_uint64]),
    "unet_verify_bytes_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_IfaceC), _vp]),
    "unet_verify": (C.c_int, [_vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, C.c_uint32, C.POINTER(_IfaceC), _vp]),
    "unet_verify_async": (C.c_int, [_vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, C.c_uint32, C.POINTER(_IfaceC), _vp, C.POINTER(C.c_uint32)]),
    "unet_wait": (C.c_int, [_vp, C.c_uint32]),
    "unet_counters": (C.c_int, [_vp, _u64p, C.c_int]),
    "unet_counters_sum": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, _u64p, C.c_int]),
    "unet_verify_static": (C.c_int, [_vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, C.c_uint32, C.POINTER(_IfaceC), _vp]),
    "unet_hint_mean_len": (C.c_int, [_vp, C.c_uint32]),
    "unet_hint_max_len": (C.c_int, [_vp, C.c_uint32]),
    "unet_records_per_wave": (C.c_int, [C.c_uint32, C.c_uint32]),
    "unet_tx_packets_per_wave": (C.c_int, [C.c_uint32]),
    "unet_tx_waves_per_packet": (C.c_int, [C.c_uint32, C.c_int]),
    "unet_generate_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, C.c_uint32, _u8p]),
    "unet_checksum_fields_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, C.c_uint32, _vp, _u8p]),
    "unet_generate": (C.c_int, [_vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, C.c_uint32, _u8p]),
    "unet_tx_build_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, _vp, C.c_uint32, C.POINTER(_IfaceC), _u8p,
                                    C.c_uint64, _u32p, _u32p, _u8p]),
    "unet_tx_build_opts_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, _vp, C.c_uint32, C.POINTER(_IfaceC), _u8p,
                                         C.c_uint64, _u32p, _u32p, _u8p, C.c_uint32]),
    "unet_tx_layout": (C.c_int, [_u32p, C.c_uint32, C.c_int, _u32p, C.POINTER(C.c_uint64)]),
    "unet_tx_layout_aligned": (C.c_int, [_u32p, C.c_uint32, C.c_int, C.c_uint32, _u32p, C.POINTER(C.c_uint64)]),
    "unet_tx_build": (C.c_int, [_vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, _vp, C.c_uint32, C.POINTER(_IfaceC), _u8p, C.c_uint64,
                                _u32p, _u32p, _u8p]),
    "unet_count_verdicts_dev": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _u64p]),
    "unet_tcp_track_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, _vp, C.c_uint32, C.POINTER(_IfaceC), _vp, C.c_uint32, _u8p]),
    "unet_tcp_track_workspace": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "unet_tcp_track_ws_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, _vp, C.c_uint32, C.POINTER(_IfaceC), _vp, C.c_uint32, _u8p,
                                        _vp, C.c_uint64, _vp]),
    "unet_tcp_slot": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint16, C.c_uint16, C.c_uint32]),
    "unet_flow_hash5": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint16, C.c_uint16]),
    "unet_flow_hash_record": (C.c_uint32, [C.c_char_p, C.c_size_t, C.c_int]),
    "unet_flow_hash_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, C.c_uint32, C.POINTER(_IfaceC), _u32p]),
    "unet_synth_len": (C.c_uint32, [C.c_int, C.c_uint64, C.c_uint64]),
    "unet_synth_layout": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, _u32p, _u32p, C.POINTER(C.c_uint64)]),
    "unet_synth_host": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(_IfaceC), _u32p, _u32p, _u8p, C.c_uint64]),
    "unet_synth_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(_IfaceC), _u32p, _u32p, _u8p, C.c_uint64]),
    "unet_synth_indexed_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_uint64, C.c_uint64, _u32p, C.c_uint32, C.POINTER(_IfaceC), _u32p, _u32p, _u8p, C.c_uint64]),
    "unet_synth_len_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_uint64, C.c_uint64, _u32p, C.c_uint32, _u32p]),
    "unet_synth_shard_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(_IfaceC), C.c_uint32, _u8p]),
    "unet_synth_flow_hash": (C.c_uint32, [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(_IfaceC)]),
    "unet_pack_size": (C.c_int, [_u32p, C.c_uint32, C.POINTER(C.c_uint64)]),
    "unet_pack_frames": (C.c_int, [C.POINTER(C.c_char_p), _u32p, C.c_uint32, _u8p, C.c_uint64, _u32p]),
    "unet_pack_length_prefixed": (C.c_int, [_vp, C.c_uint64, _u8p, C.POINTER(C.c_uint64), _u32p, _u32p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "unet_pcap_scan": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp]),
    "unet_pcap_pack": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _u8p, C.c_uint64, _u32p, _u32p, _u64p, C.c_uint32]),
    "unet_pcap_write": (C.c_int, [_u8p, C.c_uint64, _u32p, C.c_uint32, _u32p, C.c_uint32, C.c_uint32, _u8p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "unet_stream_read_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, _u32p]),
    "unet_stream_read_pieces_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, _u32p]),
    "unet_stream_read_pieces_full_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, _u32p]),
    "unet_stream_read_records_dev": (C.c_int, [_vp, _vp, _u8p, C.c_uint64, C.c_uint32, _u32p]),
    "unet_stream_copy_dev": (C.c_int, [_vp, _vp, _u8p, _u8p, C.c_uint64]),
    "unet_debug_lds_violations": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.c_int]),
}

EXPORTS = tuple(_SIGS)


def load_library(path: Optional[str] = None):
    """dlopen the offload library; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("UNET_OFFLOAD_LIB") or LIB_PATH
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  A process must hold ONE HIP runtime: if
    # torch is going to be used beside this library (device buffers, torch.distributed), its runtime has
    # to be the one already mapped when ours is dlopen'ed, so that the loader resolves our DT_NEEDED to it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: build it with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950); "
            "there is no CPU fallback for the offload")
    lib = C.CDLL(p)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError here means the ABI and the header disagree
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _check(rc: int):
    if rc != 0:
        raise UnetError(rc, load_library().unet_last_error().decode(errors="replace"))


def _np_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


def pack_arena(frames: Iterable[bytes], pad_byte: int = 0xEE) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Arena packer: every record starts on a 16-byte boundary; pad bytes are arbitrary (0xEE)."""
    frames = [bytes(f) for f in frames]
    lens = np.array([len(f) for f in frames], dtype=np.uint32)
    slots = (lens.astype(np.uint64) + 15) // 16
    off16 = np.zeros(len(frames), dtype=np.uint32)
    if len(frames):
        off16[1:] = np.cumsum(slots)[:-1].astype(np.uint32)
    total = int(slots.sum()) * 16
    arena = np.full(max(total, 16), pad_byte, dtype=np.uint8)
    for f, o in zip(frames, off16):
        arena[int(o) * 16:int(o) * 16 + len(f)] = np.frombuffer(f, dtype=np.uint8)
    return arena, off16, lens


def _aligned_u8(nbytes: int) -> np.ndarray:
    raw = np.empty(nbytes + 16, dtype=np.uint8)
    shift = (-raw.ctypes.data) % 16
    return raw[shift:shift + nbytes]


def records_per_wave(mean_len: int, max_len: int = 0) -> int:
    """Records one wavefront is given for a batch of this mean (and, if known, longest) record length."""
    return int(load_library().unet_records_per_wave(int(mean_len), int(max_len)))


def tx_packets_per_wave(mean_payload_len: int) -> int:
    """Packets one wavefront builds for a batch of this mean payload length."""
    return int(load_library().unet_tx_packets_per_wave(int(mean_payload_len)))


def tx_waves_per_packet(mean_payload_len: int, mode: int = 0) -> int:
    """Wavefronts that share one packet of a batch of this mean payload length (1 up to 1 KiB and for packets of up to three rows)."""
    return int(load_library().unet_tx_waves_per_packet(int(mean_payload_len), int(mode)))


def synth_len(kind: int, seed: int, idx: int) -> int:
    return int(load_library().unet_synth_len(kind, seed, idx))


def synth_layout(kind: int, seed: int, first_idx: int, n: int):
    off16 = np.zeros(n, dtype=np.uint32)
    lens = np.zeros(n, dtype=np.uint32)
    nbytes = C.c_uint64(0)
    _check(load_library().unet_synth_layout(kind, seed, first_idx, n, _np_ptr(off16), _np_ptr(lens), C.byref(nbytes)))
    return off16, lens, int(nbytes.value)


def synth_host(kind: int, seed: int, first_idx: int, n: int, iface: Iface):
    """Host twin of the device trace generator -> (arena, off16, lens)."""
    off16, lens, nbytes = synth_layout(kind, seed, first_idx, n)
    arena = _aligned_u8(max(nbytes, 16))
    arena[:] = 0xEE
    ic = iface.c()
    _check(load_library().unet_synth_host(kind, seed, first_idx, n, C.byref(ic), _np_ptr(off16), _np_ptr(lens),
                                          _np_ptr(arena), nbytes))
    return arena, off16, lens


def synth_flow_hash(kind: int, seed: int, idx: int, iface: Iface) -> int:
    """Flow hash of the 5-tuple that record `idx` of the trace carries (before fault injection)."""
    ic = iface.c()
    return int(load_library().unet_synth_flow_hash(kind, seed, idx, C.byref(ic)))


def pack_frames_native(frames: Sequence[bytes]):
    """unet_pack_frames: separately owned buffers -> (arena, off16, lens), done by the library."""
    lib = load_library()
    n = len(frames)
    lens = np.array([len(f) for f in frames], dtype=np.uint32)
    need = C.c_uint64(0)
    _check(lib.unet_pack_size(_np_ptr(lens), n, C.byref(need)))
    arena = _aligned_u8(need.value)
    off16 = np.zeros(n, dtype=np.uint32)
    ptrs = (C.c_char_p * max(n, 1))(*[bytes(f) for f in frames])
    _check(lib.unet_pack_frames(ptrs, _np_ptr(lens), n, _np_ptr(arena), arena.nbytes, _np_ptr(off16)))
    return arena, off16, lens


def pack_length_prefixed(stream: bytes):
    """unet_pack_length_prefixed: [u32 length][bytes]... -> (arena, off16, lens)."""
    lib = load_library()
    buf = np.frombuffer(stream, dtype=np.uint8)
    need, n = C.c_uint64(0), C.c_uint32(0)
    _check(lib.unet_pack_length_prefixed(_np_ptr(buf) if len(stream) else None, len(stream), None, C.byref(need), None, None, 0, C.byref(n)))
    arena = _aligned_u8(need.value)
    off16 = np.zeros(n.value, dtype=np.uint32)
    lens = np.zeros(n.value, dtype=np.uint32)
    cap = C.c_uint64(arena.nbytes)
    _check(lib.unet_pack_length_prefixed(_np_ptr(buf) if len(stream) else None, len(stream), _np_ptr(arena), C.byref(cap), _np_ptr(off16),
                                         _np_ptr(lens), n.value, C.byref(n)))
    return arena, off16, lens


TAP_FRAME_MAX = 1514      # src/driver/tap.rs:170
LOOPBACK_MTU = 65535      # src/devices/loopback.rs:16
LINKTYPE_NULL, LINKTYPE_ETHERNET, LINKTYPE_RAW, LINKTYPE_IPV4 = 0, 1, 101, 228


class PcapInfo(C.Structure):
    _fields_ = [("linktype", C.c_uint32), ("snaplen", C.c_uint32), ("mode", C.c_uint32), ("skip", C.c_uint32),
                ("n_records", C.c_uint32), ("n_snapped", C.c_uint32), ("n_clipped", C.c_uint32), ("nanosecond", C.c_uint32),
                ("arena_bytes", C.c_uint64), ("trailing_bytes", C.c_uint64)]


def pcap_scan(data: bytes, max_len: int = LOOPBACK_MTU) -> PcapInfo:
    info = PcapInfo()
    buf = np.frombuffer(data, dtype=np.uint8)
    _check(load_library().unet_pcap_scan(_np_ptr(buf), len(data), max_len, C.addressof(info)))
    return info


def pcap_pack(data: bytes, max_len: int = LOOPBACK_MTU):
    """Classic pcap file bytes -> (arena, off16, lens, ts_ns, info); records cut at max_len the way the device
    would have delivered them (TAP_FRAME_MAX for tap0, LOOPBACK_MTU for lo)."""
    info = pcap_scan(data, max_len)
    buf = np.frombuffer(data, dtype=np.uint8)
    n = info.n_records
    arena = _aligned_u8(info.arena_bytes)
    off16 = np.zeros(n, dtype=np.uint32)
    lens = np.zeros(n, dtype=np.uint32)
    ts = np.zeros(n, dtype=np.uint64)
    _check(load_library().unet_pcap_pack(_np_ptr(buf), len(data), max_len, _np_ptr(arena), arena.nbytes, _np_ptr(off16),
                                         _np_ptr(lens), _np_ptr(ts), n))
    return arena, off16, lens, ts, info


def pcap_write(arena: np.ndarray, off16, lens: np.ndarray, linktype: int = LINKTYPE_ETHERNET, stride16: int = 0) -> bytes:
    arena = np.ascontiguousarray(arena, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    if off16 is not None:
        off16 = np.ascontiguousarray(off16, dtype=np.uint32)
    need = C.c_uint64(0)
    lib = load_library()
    _check(lib.unet_pcap_write(_np_ptr(arena), arena.nbytes, _np_ptr(off16), stride16, _np_ptr(lens), len(lens), linktype, None, 0,
                               C.byref(need)))
    out = np.zeros(need.value, dtype=np.uint8)
    _check(lib.unet_pcap_write(_np_ptr(arena), arena.nbytes, _np_ptr(off16), stride16, _np_ptr(lens), len(lens), linktype,
                               _np_ptr(out), out.nbytes, C.byref(need)))
    return out.tobytes()


def flow_hash_record(rec: bytes, mode: int = 0) -> int:
    """Flow hash of one packet (mode 0) or Ethernet frame (mode 1) from its bytes, on the host."""
    rec = bytes(rec)
    return int(load_library().unet_flow_hash_record(rec, len(rec), mode))


def verify_split_workspace(n: int, max_len: int) -> int:
    need = C.c_uint64(0)
    _check(load_library().unet_verify_split_workspace(int(n), int(max_len), C.byref(need)))
    return int(need.value)


def tcp_track_workspace(n: int, table_slots: int) -> int:
    """Bytes of device workspace unet_tcp_track_ws_dev needs for n records."""
    need = C.c_uint64(0)
    _check(load_library().unet_tcp_track_workspace(int(n), int(table_slots), C.byref(need)))
    return int(need.value)


def flow_hash5(src: int, dst: int, proto: int, sport: int, dport: int) -> int:
    return int(load_library().unet_flow_hash5(src, dst, proto, sport, dport))


class Offload:
    """One context on one GPU.  Host-pointer methods take numpy arrays; ``*_dev`` methods take raw
    device pointers (ints, e.g. ``tensor.data_ptr()``) and enqueue on ``stream`` without waiting."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib.unet_open(device, C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.unet_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(self.lib.unet_sync(self.h))

    # -- a1 ---------------------------------------------------------------------------------
    def calculate_checksum(self, data: bytes, sum: int = 0) -> int:
        """`calculate_checksum(data, sum)` of src/utils.rs:1, computed on the GPU."""
        out = C.c_uint16(0)
        data = bytes(data)
        _check(self.lib.unet_csum(self.h, data, len(data), sum & 0xFFFF, C.byref(out)))
        return int(out.value)

    def csum_batch(self, arena: np.ndarray, off16, lens: np.ndarray, seeds=None, stride16: int = 0) -> np.ndarray:
        arena, off16, lens = self._prep(arena, off16, lens)
        n = len(lens)
        out = np.zeros(n, dtype=np.uint16)
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, dtype=np.uint16)
        _check(self.lib.unet_csum_batch(self.h, _np_ptr(arena), arena.nbytes, _np_ptr(off16), stride16, _np_ptr(lens),
                                        _np_ptr(seeds), n, _np_ptr(out)))
        return out

    # -- a2-a4 ------------------------------------------------------------------------------
    def ipv4_validate(self, hdrs: np.ndarray, iface: Iface) -> np.ndarray:
        hdrs = np.ascontiguousarray(hdrs, dtype=np.uint8).reshape(-1)
        assert hdrs.size % 20 == 0
        n = hdrs.size // 20
        out = np.zeros(n, dtype=RESULT_DTYPE)
        ic = iface.c()
        _check(self.lib.unet_ipv4_validate(self.h, _np_ptr(hdrs), n, C.byref(ic), _np_ptr(out)))
        return out

    # -- receive path -----------------------------------------------------------------------
    def verify(self, arena: np.ndarray, off16, lens: np.ndarray, iface: Iface, stride16: int = 0) -> np.ndarray:
        arena, off16, lens = self._prep(arena, off16, lens)
        n = len(lens)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        ic = iface.c()
        _check(self.lib.unet_verify(self.h, _np_ptr(arena), arena.nbytes, _np_ptr(off16), stride16, _np_ptr(lens), n,
                                    C.byref(ic), _np_ptr(out)))
        return out

    def verify_static(self, arena: np.ndarray, off16, lens: np.ndarray, iface: Iface, stride16: int = 0) -> np.ndarray:
        """unet_verify_static: the same verdicts through the library's static device scratch (no allocation)."""
        arena, off16, lens = self._prep(arena, off16, lens)
        n = len(lens)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        ic = iface.c()
        _check(self.lib.unet_verify_static(self.h, _np_ptr(arena), arena.nbytes, _np_ptr(off16), stride16, _np_ptr(lens), n,
                                           C.byref(ic), _np_ptr(out)))
        return out

    def verify_async(self, arena: np.ndarray, off16, lens: np.ndarray, iface: Iface, stride16: int = 0):
        """unet_verify_async -> (ticket, out): `out` holds the results once wait(ticket) has returned."""
        arena, off16, lens = self._prep(arena, off16, lens)
        n = len(lens)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        ic = iface.c()
        t = C.c_uint32(0)
        _check(self.lib.unet_verify_async(self.h, _np_ptr(arena), arena.nbytes, _np_ptr(off16), stride16, _np_ptr(lens), n,
                                          C.byref(ic), _np_ptr(out), C.byref(t)))
        # the library writes into `out` during wait(): keep it alive until then even if the caller drops it
        if not hasattr(self, "_inflight"):
            self._inflight = {}
        self._inflight[int(t.value)] = out
        return int(t.value), out

    def wait(self, ticket: int):
        try:
            _check(self.lib.unet_wait(self.h, ticket))
        finally:
            getattr(self, "_inflight", {}).pop(int(ticket), None)

    def counters(self, reset: bool = False) -> np.ndarray:
        out = np.zeros(V_COUNT, dtype=np.uint64)
        _check(self.lib.unet_counters(self.h, _np_ptr(out), 1 if reset else 0))
        return out

    def hint_mean_len(self, mean_len: int):
        _check(self.lib.unet_hint_mean_len(self.h, int(mean_len)))

    def hint_max_len(self, max_len: int):
        _check(self.lib.unet_hint_max_len(self.h, int(max_len)))

    def verify_frames(self, frames: Sequence[bytes], iface: Iface) -> np.ndarray:
        arena, off16, lens = pack_arena(frames)
        return self.verify(arena, off16, lens, iface)

    # -- transmit path ----------------------------------------------------------------------
    def generate(self, arena: np.ndarray, off16, lens: np.ndarray, stride16: int = 0):
        """Returns (filled copy of the arena, status)."""
        arena, off16, lens = self._prep(arena, off16, lens)
        arena = arena.copy()
        n = len(lens)
        status = np.zeros(n, dtype=np.uint8)
        _check(self.lib.unet_generate(self.h, _np_ptr(arena), arena.nbytes, _np_ptr(off16), stride16, _np_ptr(lens), n,
                                      _np_ptr(status)))
        return arena, status

    def tx_build(self, payload_arena: np.ndarray, payload_off16, payload_lens: np.ndarray, desc: np.ndarray, iface: Iface,
                 payload_stride16: int = 0, out: Optional[np.ndarray] = None):
        """Returns (packet arena, out_off16, out_len, status).  `out`: the caller's packet arena (at least as large as
        unet_tx_layout says) instead of a fresh one."""
        arena, off16, lens = self._prep(payload_arena, payload_off16, payload_lens)
        desc = np.ascontiguousarray(desc)
        assert desc.dtype.itemsize == 28 and len(desc) == len(lens)
        n = len(lens)
        need = C.c_uint64(0)
        _check(self.lib.unet_tx_layout(_np_ptr(lens), n, int(iface.mode), None, C.byref(need)))
        if out is None:
            out = _aligned_u8(need.value)
        else:
            assert out.dtype == np.uint8 and out.flags.c_contiguous and out.nbytes >= need.value and out.ctypes.data % 16 == 0
            out = out[:need.value]
        out_off = np.zeros(n, dtype=np.uint32)
        out_len = np.zeros(n, dtype=np.uint32)
        status = np.zeros(n, dtype=np.uint8)
        ic = iface.c()
        _check(self.lib.unet_tx_build(self.h, _np_ptr(arena), arena.nbytes, _np_ptr(off16), payload_stride16, _np_ptr(lens),
                                      desc.ctypes.data, n, C.byref(ic), _np_ptr(out), out.nbytes, _np_ptr(out_off), _np_ptr(out_len),
                                      _np_ptr(status)))
        return out, out_off, out_len, status

    # -- device-pointer entry points ----------------------------------------------------------
    def csum_batch_dev(self, d_arena, arena_bytes, d_off16, stride16, d_len, d_seeds, n, d_out, stream=None):
        _check(self.lib.unet_csum_batch_dev(self.h, stream, d_arena, arena_bytes, d_off16, stride16, d_len, d_seeds, n, d_out))

    def verify_dev(self, d_arena, arena_bytes, d_off16, stride16, d_len, n, iface: Iface, d_out, stream=None):
        ic = iface.c()
        _check(self.lib.unet_verify_dev(self.h, stream, d_arena, arena_bytes, d_off16, stride16, d_len, n, C.byref(ic), d_out))

    def verify_split_dev(self, d_arena, arena_bytes, d_off16, stride16, d_len, n, max_len, iface: Iface, d_out, d_workspace, workspace_bytes,
                         stream=None):
        """unet_verify_split_dev: a few long records, one wavefront per 4 KiB segment."""
        ic = iface.c()
        _check(self.lib.unet_verify_split_dev(self.h, stream, d_arena, arena_bytes, d_off16, stride16, d_len, n, max_len, C.byref(ic), d_out,
                                              d_workspace, workspace_bytes))

    def verify_bytes_dev(self, d_buf, buf_bytes, stride_bytes, length, n, iface: Iface, d_out, stream=None):
        """unet_verify_bytes_dev: n records of `length` bytes at a byte stride (no 16-byte slots, no tables)."""
        ic = iface.c()
        _check(self.lib.unet_verify_bytes_dev(self.h, stream, d_buf, buf_bytes, stride_bytes, length, n, C.byref(ic), d_out))

    def ipv4_validate_dev(self, d_hdrs, n, iface: Iface, d_out, stream=None):
        ic = iface.c()
        _check(self.lib.unet_ipv4_validate_dev(self.h, stream, d_hdrs, n, C.byref(ic), d_out))

    def generate_dev(self, d_arena, arena_bytes, d_off16, stride16, d_len, n, d_status, stream=None):
        _check(self.lib.unet_generate_dev(self.h, stream, d_arena, arena_bytes, d_off16, stride16, d_len, n, d_status))

    def tx_build_dev(self, d_pay_arena, pay_bytes, d_pay_off16, pay_stride16, d_pay_len, d_desc, n, iface: Iface, d_out_arena,
                     out_bytes, d_out_off16, d_out_len, d_status, stream=None, flags: int = 0):
        ic = iface.c()
        if flags:
            _check(self.lib.unet_tx_build_opts_dev(self.h, stream, d_pay_arena, pay_bytes, d_pay_off16, pay_stride16, d_pay_len, d_desc, n,
                                                   C.byref(ic), d_out_arena, out_bytes, d_out_off16, d_out_len, d_status, flags))
            return
        _check(self.lib.unet_tx_build_dev(self.h, stream, d_pay_arena, pay_bytes, d_pay_off16, pay_stride16, d_pay_len, d_desc, n,
                                          C.byref(ic), d_out_arena, out_bytes, d_out_off16, d_out_len, d_status))

    def checksum_fields_dev(self, d_arena, arena_bytes, d_off16, stride16, d_len, n, d_fields, d_status, stream=None):
        _check(self.lib.unet_checksum_fields_dev(self.h, stream, d_arena, arena_bytes, d_off16, stride16, d_len, n, d_fields, d_status))

    def count_verdicts_dev(self, d_results, n, d_counts, stream=None):
        _check(self.lib.unet_count_verdicts_dev(self.h, stream, d_results, n, d_counts))

    def tcp_track_dev(self, d_arena, arena_bytes, d_off16, stride16, d_len, d_results, n, iface: Iface, d_table, table_slots, d_seg_state,
                      stream=None):
        ic = iface.c()
        _check(self.lib.unet_tcp_track_dev(self.h, stream, d_arena, arena_bytes, d_off16, stride16, d_len, d_results, n, C.byref(ic), d_table,
                                           table_slots, d_seg_state))

    def tcp_track_ws_dev(self, d_arena, arena_bytes, d_off16, stride16, d_len, d_results, n, iface: Iface, d_table, table_slots, d_seg_state,
                         d_workspace, workspace_bytes, d_stats=None, stream=None):
        """unet_tcp_track_ws_dev: the caller's workspace (tcp_track_workspace bytes, 256-byte aligned), optional stats."""
        ic = iface.c()
        _check(self.lib.unet_tcp_track_ws_dev(self.h, stream, d_arena, arena_bytes, d_off16, stride16, d_len, d_results, n, C.byref(ic), d_table,
                                              table_slots, d_seg_state, d_workspace, workspace_bytes, d_stats))

    def flow_hash_dev(self, d_arena, arena_bytes, d_off16, stride16, d_len, n, iface: Iface, d_hash, stream=None):
        ic = iface.c()
        _check(self.lib.unet_flow_hash_dev(self.h, stream, d_arena, arena_bytes, d_off16, stride16, d_len, n, C.byref(ic), d_hash))

    def synth_dev(self, kind, seed, first_idx, n, iface: Iface, d_off16, d_len, d_arena, arena_bytes, stream=None):
        ic = iface.c()
        _check(self.lib.unet_synth_dev(self.h, stream, kind, seed, first_idx, n, C.byref(ic), d_off16, d_len, d_arena, arena_bytes))

    def synth_indexed_dev(self, kind, seed, first_idx, d_rel_idx, n, iface: Iface, d_off16, d_len, d_arena, arena_bytes, stream=None):
        ic = iface.c()
        _check(self.lib.unet_synth_indexed_dev(self.h, stream, kind, seed, first_idx, d_rel_idx, n, C.byref(ic), d_off16, d_len, d_arena, arena_bytes))

    def synth_len_dev(self, kind, seed, first_idx, d_rel_idx, n, d_len, stream=None):
        _check(self.lib.unet_synth_len_dev(self.h, stream, kind, seed, first_idx, d_rel_idx, n, d_len))

    def synth_shard_dev(self, kind, seed, first_idx, n, iface: Iface, world, d_shard, stream=None):
        ic = iface.c()
        _check(self.lib.unet_synth_shard_dev(self.h, stream, kind, seed, first_idx, n, C.byref(ic), world, d_shard))

    def stream_read_dev(self, d_buf, nbytes, d_sink, stream=None):
        _check(self.lib.unet_stream_read_dev(self.h, stream, d_buf, nbytes, d_sink))

    def stream_read_pieces_dev(self, d_buf, nbytes, d_sink, stream=None):
        _check(self.lib.unet_stream_read_pieces_dev(self.h, stream, d_buf, nbytes, d_sink))

    def stream_read_pieces_full_dev(self, d_buf, nbytes, d_sink, stream=None):
        _check(self.lib.unet_stream_read_pieces_full_dev(self.h, stream, d_buf, nbytes, d_sink))

    def stream_read_records_dev(self, d_buf, nbytes, record_bytes, d_sink, stream=None):
        _check(self.lib.unet_stream_read_records_dev(self.h, stream, d_buf, nbytes, record_bytes, d_sink))

    def debug_lds_violations(self, reset: bool = False):
        """Out-of-area LDS stores counted by a -DUNET_DEBUG build, or None when the library is a release build."""
        v = C.c_uint32(0)
        if self.lib.unet_debug_lds_violations(self.h, C.byref(v), 1 if reset else 0) != 0:
            return None
        return int(v.value)

    def stream_copy_dev(self, d_src, d_dst, nbytes, stream=None):
        _check(self.lib.unet_stream_copy_dev(self.h, stream, d_src, d_dst, nbytes))

    # -- helpers --------------------------------------------------------------------------------
    @staticmethod
    def _prep(arena, off16, lens):
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        if arena.ctypes.data % 16:
            a2 = _aligned_u8(arena.nbytes)
            a2[:] = arena
            arena = a2
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        if off16 is not None:
            off16 = np.ascontiguousarray(off16, dtype=np.uint32)
        return arena, off16, lens
