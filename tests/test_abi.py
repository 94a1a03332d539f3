"""CPU-only: the C-ABI library loads, exports every symbol include/unet_offload.h declares, rejects bad
arguments without a GPU, and the host-side pieces (arena packer, trace layout, host trace twin) behave."""
import ctypes
import os
import re

import numpy as np
import pytest

import unet_amd
from unet_amd import _ffi
from helpers import as_tuples
from oracle import oracle_ffi as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x756E6574


def declared_functions():
    text = open(os.path.join(ROOT, "include", "unet_offload.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(unet_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/unet_offload.h but not exported"
    assert sorted(_ffi.EXPORTS) == names, "python binding and header disagree"


def test_no_context_without_gpu_and_errors_are_reported():
    lib = unet_amd.load_library()
    if lib.unet_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(unet_amd.UnetError) as ei:
        unet_amd.Offload(0)
    assert ei.value.code == -2 and "HIP device" in str(ei.value)


def test_missing_library_is_loud(tmp_path):
    with pytest.raises(FileNotFoundError):
        unet_amd.load_library(str(tmp_path / "nope.so"))


def test_records_per_wave_rule():
    """The host's choice of records per wavefront, at every boundary of the rule (no GPU involved)."""
    rpw = _ffi.records_per_wave
    # shortest records: 64 to a wave when none is known to be longer than 64 bytes, or -- lengths unseen -- by the mean
    assert [rpw(0), rpw(64), rpw(65), rpw(1024)] == [64, 64, 32, 32]
    assert [rpw(40, 64), rpw(40, 65), rpw(40, 80), rpw(40, 81), rpw(40, 96), rpw(40, 97), rpw(40, 1500)] == [64, 48, 48, 40, 40, 32, 32]
    assert rpw(64, 64) == 64 and rpw(70, 96) == 40 and rpw(1024, 1024) == 32 and rpw(600, 9000) == 32
    # long records: sixteen up to 1536 bytes, then about 16 KiB per wave, and eight of the very longest
    assert [rpw(1025), rpw(1500, 1500), rpw(1536), rpw(1537)] == [16, 16, 16, 8]
    assert [rpw(2048), rpw(2049), rpw(4096), rpw(4097), rpw(8192), rpw(8193), rpw(16384), rpw(49151), rpw(49152), rpw(65535)] == [8, 4, 4, 2, 2, 1, 1, 1, 8, 8]
    # the longest record does not matter above 1 KiB of mean
    assert rpw(3000, 65535) == rpw(3000) == 4
    for mean in range(0, 65536, 7):
        assert rpw(mean) in (64, 32, 16, 8, 4, 2, 1)


def test_tx_packets_per_wave_rule():
    tpw = _ffi.tx_packets_per_wave
    assert [tpw(0), tpw(18), tpw(128), tpw(129), tpw(320), tpw(321), tpw(720), tpw(721), tpw(1024), tpw(1025), tpw(1152), tpw(1153)] == [64, 64, 64, 32, 32, 16, 16, 8, 8, 8, 8, 1]
    assert [tpw(1472), tpw(2500), tpw(8972), tpw(65507)] == [1, 1, 1, 1]
    # above 1152 bytes of mean payload: one packet per wavefront, or per group of wavefronts -- about three 1 KiB rows of output each
    wpp = _ffi.tx_waves_per_packet
    assert [wpp(0), wpp(1024), wpp(1025), wpp(1472), wpp(3044), wpp(3045), wpp(6116), wpp(6117)] == [1, 1, 1, 1, 1, 2, 2, 4]
    assert [wpp(12260), wpp(12261), wpp(24548), wpp(24549), wpp(65507)] == [4, 8, 8, 16, 16]
    assert [wpp(3030, 1), wpp(3031, 1), wpp(6102, 1), wpp(6103, 1)] == [1, 2, 2, 4]   # Ethernet output: 14 bytes more per packet


def test_tx_layout_aligned_matches_python_and_rejects_bad_alignment():
    lib = unet_amd.load_library()
    rng = np.random.default_rng(8)
    plen = rng.integers(0, 9000, size=500).astype(np.uint32)
    for mode in (0, 1):
        for align16 in (1, 2, 8, 64):
            off = np.zeros(500, dtype=np.uint32)
            need = ctypes.c_uint64(0)
            rc = lib.unet_tx_layout_aligned(_ffi._np_ptr(plen), 500, mode, align16, _ffi._np_ptr(off), ctypes.byref(need))
            assert rc == 0
            eoff, ebytes = _ffi.tx_out_layout(plen, mode, align16)
            assert (off == eoff).all() and need.value == ebytes and (off % align16 == 0).all()
    need = ctypes.c_uint64(0)
    for bad in (0, 3, 128):
        assert lib.unet_tx_layout_aligned(_ffi._np_ptr(plen), 500, 0, bad, None, ctypes.byref(need)) == -1


def test_pack_arena_layout():
    frames = [b"", b"a", b"b" * 16, b"c" * 17, b"d" * 1500]
    arena, off16, lens = unet_amd.pack_arena(frames)
    assert lens.tolist() == [0, 1, 16, 17, 1500]
    assert off16.tolist() == [0, 0, 1, 2, 4]
    assert arena.nbytes == (0 + 1 + 1 + 2 + 94) * 16
    for f, o in zip(frames, off16):
        assert arena[int(o) * 16:int(o) * 16 + len(f)].tobytes() == f


def test_flow_hash_matches_oracle_definition():
    rng = np.random.default_rng(5)
    for _ in range(200):
        a = [int(x) for x in rng.integers(0, 2**32, size=2)] + [int(rng.integers(0, 256))] + [int(x) for x in rng.integers(0, 65536, size=2)]
        assert unet_amd.flow_hash5(*a) == O.flow_hash5(*a)


@pytest.mark.parametrize("kind", [unet_amd.SYNTH_UDP1500, unet_amd.SYNTH_IMIX])
def test_trace_layout_is_index_addressable(kind):
    off_a, len_a, bytes_a = unet_amd.synth_layout(kind, SEED, 0, 300)
    off_b, len_b, _ = unet_amd.synth_layout(kind, SEED, 100, 200)
    assert (len_a[100:] == len_b).all()
    assert [unet_amd.synth_len(kind, SEED, i) for i in range(5)] == len_a[:5].tolist()
    slots = (len_a.astype(np.int64) + 15) // 16
    assert (off_a[1:] == np.cumsum(slots)[:-1]).all() and bytes_a == int(slots.sum()) * 16
    ifc = unet_amd.Iface(ext_tcp=True)
    a, _, _ = unet_amd.synth_host(kind, SEED, 0, 300, ifc)
    b, _, _ = unet_amd.synth_host(kind, SEED, 100, 200, ifc)
    start = int(off_a[100]) * 16
    assert (a[start:] == b[:len(a) - start]).all(), "record idx must not depend on the batch it is generated in"


def test_udp1500_trace_faults_are_what_the_oracle_sees():
    ifc = unet_amd.Iface()
    n = 1024
    arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_UDP1500, SEED, 0, n, ifc)
    assert (lens == 1500).all() and (np.diff(off16) == 94).all()
    res = O.recv_batch(arena, off16, lens, ifc)
    idx = np.arange(n)
    exp = np.full(n, 20)
    exp[idx % 64 == 7] = 19      # payload bit flip
    exp[idx % 256 == 33] = 18    # UDP length field mismatch
    exp[idx % 256 == 97] = 19    # transmitted checksum 0
    assert (res["verdict"] == exp).all()
    assert (res["csum"][exp == 20] == 0).all() and (res["csum"][exp == 19] != 0).all()


def test_imix_trace_exercises_every_l3_l4_verdict():
    n = 1 << 15
    for ext in (True, False):
        ifc = unet_amd.Iface(ext_tcp=ext)
        arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_IMIX, SEED, 0, n, ifc)
        res = O.recv_batch(arena, off16, lens, ifc)
        seen = set(res["verdict"].tolist())
        want = {6, 7, 8, 9, 10, 11, 12, 14, 15, 16, 18, 19, 20}
        if ext:
            want |= {23, 24}
        assert want <= seen, sorted(want - seen)
        assert set(np.unique(lens).tolist()) == {64, 576, 1500, 4096, 9000}
        assert 560 < lens.mean() < 650


def test_hdr20_trace_verdicts():
    n = 4096
    ifc = unet_amd.Iface()
    arena, _, lens = unet_amd.synth_host(unet_amd.SYNTH_IPV4_HDR20, SEED, 0, n, ifc)
    assert (lens == 20).all()
    res = O.ipv4_validate_batch(arena[:n * 20])
    seen = set(res["verdict"].tolist())
    assert {6, 7, 8, 9, 10, 11, 25} <= seen
    idx = np.arange(n)
    assert (res["verdict"][idx % 64 == 27] != 25).all()


def test_flow_trace_tuples_match_the_bytes_and_are_skewed():
    """UNET_SYNTH_FLOWS: the 5-tuple announced by unet_synth_flow_hash is the one in the record's bytes
    (records whose fault does not touch addressing), and flow popularity is long-tailed."""
    ifc = unet_amd.Iface(ext_tcp=True)
    n = 20000
    for kind in (unet_amd.SYNTH_FLOWS, unet_amd.SYNTH_IMIX, unet_amd.SYNTH_UDP1500):
        arena, off16, lens = unet_amd.synth_host(kind, SEED, 1000, n, ifc)
        checked = 0
        hashes = []
        for r in range(0, n, 7):
            idx = 1000 + r
            h = unet_amd.synth_flow_hash(kind, SEED, idx, ifc)
            hashes.append(h)
            if kind != unet_amd.SYNTH_UDP1500 and idx % 64 in (3, 19, 21, 27, 51):
                continue  # header checksum flip in src, IHL nibble, foreign dst, protocol 47
            rec = arena[int(off16[r]) * 16:int(off16[r]) * 16 + int(lens[r])].tobytes()
            assert O.flow_hash_record(rec, ifc) == h, (kind, idx)
            checked += 1
        assert checked > 2000
    # popularity is skewed: 60000 uniform draws from 2^20 flows would leave ~58280 distinct flows
    ids = {}
    for idx in range(0, 60000):
        h = unet_amd.synth_flow_hash(unet_amd.SYNTH_FLOWS, SEED, idx, ifc)
        ids[h] = ids.get(h, 0) + 1
    counts = np.sort(np.array(list(ids.values())))[::-1]
    assert len(ids) < 57600 and counts[0] >= 4
    res = O.recv_batch(*unet_amd.synth_host(unet_amd.SYNTH_FLOWS, SEED, 0, 4096, ifc), ifc)
    assert {20, 24, 15, 16} & set(res["verdict"].tolist())


def test_flow_hash_record_on_the_host_matches_the_oracle():
    """unet_flow_hash_record (what a multi-GPU dispatcher calls before any GPU has seen the frame) against the
    oracle's rule, L3 and L2 forms, including buffers too short to reach the ports or the header."""
    from unet_amd import _ffi as F
    ifc = unet_amd.Iface(ext_tcp=True)
    arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_IMIX, SEED, 77, 3000, ifc)
    l2 = ifc.replace(mode=1)
    for r in range(3000):
        rec = arena[int(off16[r]) * 16:int(off16[r]) * 16 + int(lens[r])].tobytes()
        assert F.flow_hash_record(rec, 0) == O.flow_hash_record(rec, ifc)
        if r % 5 == 0:
            frame = b"\x02" * 12 + (b"\x08\x00" if r % 10 else b"\x08\x06") + rec
            assert F.flow_hash_record(frame, 1) == O.flow_hash_record(frame, l2)
            for cut in (0, 13, 14, 33, 34, 37, 38):
                assert F.flow_hash_record(frame[:cut], 1) == O.flow_hash_record(frame[:cut], l2)
                assert F.flow_hash_record(rec[:cut], 0) == O.flow_hash_record(rec[:cut], ifc)


def test_arena_packers():
    """unet_pack_frames / unet_pack_length_prefixed against the Python packer used by the other tests."""
    import struct
    from unet_amd import _ffi as F
    rng = np.random.default_rng(4)
    frames = [rng.integers(0, 256, size=int(n), dtype=np.uint8).tobytes() for n in [0, 1, 15, 16, 17, 98, 1514, 0, 65535, 33]]
    ref_arena, ref_off, ref_len = unet_amd.pack_arena(frames, pad_byte=0)
    for arena, off16, lens in (F.pack_frames_native(frames),
                               F.pack_length_prefixed(b"".join(struct.pack("<I", len(f)) + f for f in frames))):
        assert arena.ctypes.data % 16 == 0
        assert (off16 == ref_off).all() and (lens == ref_len).all() and (arena == ref_arena[:arena.nbytes]).all()
    a, o, l = F.pack_frames_native([])
    assert a.nbytes == 16 and len(o) == 0
    a, o, l = F.pack_length_prefixed(b"")
    assert a.nbytes == 16 and len(l) == 0
    with pytest.raises(unet_amd.UnetError, match="truncated length prefix"):
        F.pack_length_prefixed(struct.pack("<I", 2) + b"ab" + b"\x01\x00")
    with pytest.raises(unet_amd.UnetError, match="announced"):
        F.pack_length_prefixed(struct.pack("<I", 9) + b"abc")
    with pytest.raises(unet_amd.UnetError, match="65535"):
        F.pack_length_prefixed(struct.pack("<I", 70000) + bytes(70000))


def test_public_header_compiles_as_c99_c11_and_cpp11(tmp_path):
    """include/unet_offload.h is what a C, C++ or (through bindgen-style declarations) Rust caller binds against: it must
    be a valid header in each dialect by itself, pedantic warnings as errors."""
    import subprocess
    src = tmp_path / "hdr.c"
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "unet_offload.h")
    src.write_text(f'#include "{hdr}"\nint main(void) {{ unet_iface i; unet_result r; unet_tx_desc d; (void)i; (void)r; (void)d; '
                   'return (int)sizeof(unet_tx_desc) - 28 + (int)sizeof(unet_result) - 4; }\n')
    for cc, flags in (("gcc", ["-std=c99"]), ("gcc", ["-std=c11", "-Wstrict-prototypes"]), ("g++", ["-std=c++11", "-x", "c++"])):
        p = subprocess.run([cc, *flags, "-Wall", "-Wextra", "-pedantic", "-Werror", "-c", str(src), "-o", str(tmp_path / "hdr.o")],
                           capture_output=True, text=True)
        assert p.returncode == 0, (cc, flags, p.stderr)
