"""32-record tiles whose records lie back to back are read as one byte stream (whole 1 KiB rows, record boundaries
from bitmaps, per-record sums from prefix sums).  Batches built to hit every corner of that route -- and the
layouts that must NOT take it (gaps, empty records, descending or shuffled offsets) -- against the CPU oracle, for
verdicts, calculate_checksum and the transmit-side fill.  The host entry points pick 32-record tiles for batches
whose mean record length is below 1 KiB and whose longest record is above 64 bytes."""
import zlib

import numpy as np
import pytest

import unet_amd
from unet_amd import Verdict
from oracle import oracle_ffi as O
from test_short_records import _layouts, _packets

pytestmark = pytest.mark.gpu


def _packed(rng, frames, lead_slots=0):
    """Arena with the frames back to back (garbage in every pad byte) after `lead_slots` unused 16-byte slots."""
    lens = np.array([len(f) for f in frames], dtype=np.uint32)
    slots = (lens.astype(np.int64) + 15) // 16
    off = lead_slots + np.concatenate([[0], np.cumsum(slots)[:-1]])
    arena = rng.integers(0, 256, size=max(int(off[-1] + slots[-1]) * 16, 16), dtype=np.uint8)
    for f, o in zip(frames, off):
        arena[int(o) * 16:int(o) * 16 + len(f)] = np.frombuffer(f, dtype=np.uint8)
    return arena, off.astype(np.uint32), lens


def _random_frames(rng, lengths):
    return [rng.integers(0, 256, size=int(l), dtype=np.uint8).tobytes() for l in lengths]


LENGTH_MIXES = {
    # name: (lengths drawn per record, records)
    "1..1900": (lambda rng, n: rng.integers(1, 1901, size=n), 1500),
    "chunk edges": (lambda rng, n: rng.choice([16, 17, 31, 32, 33, 95, 96, 97, 1008, 1023, 1024, 1025, 1039, 1040], size=n), 2000),
    "rows of exactly 1 KiB": (lambda rng, n: np.where(np.arange(n) % 16 == 5, 512, 1024), 700),  # most records start where a row starts
    "row minus one chunk": (lambda rng, n: np.full(n, 1008), 700),          # starts walk backwards through the lanes
    "header across two rows": (lambda rng, n: np.full(n, 1024 - 48), 700),  # chunks 0..5 of many records straddle a row end
    "cut chunk in lane 63 and 0": (lambda rng, n: rng.choice([1009, 1025, 2033, 2049, 17, 100, 200, 300], size=n), 900),
    "long among short": (lambda rng, n: np.where(np.arange(n) % 97 == 7, 65535, rng.integers(65, 200, size=n)), 1000),  # > 32 rows: bitmaps rebuilt
    "jumbo runs": (lambda rng, n: np.where(np.arange(n) % 30 < 2, 9000, rng.integers(20, 600, size=n)), 900),
    "one byte records between": (lambda rng, n: np.where(np.arange(n) % 3 == 0, 1, rng.integers(65, 900, size=n)), 1500),
}


@pytest.mark.parametrize("mix", sorted(LENGTH_MIXES))
@pytest.mark.parametrize("lead", [0, 3, 7])
def test_stream_tiles_checksum(offload, mix, lead):
    draw, n = LENGTH_MIXES[mix]
    rng = np.random.default_rng(zlib.crc32(mix.encode()) + lead)
    lens = np.asarray(draw(rng, n), dtype=np.int64)
    arena, off16, lens = _packed(rng, _random_frames(rng, lens), lead_slots=lead)
    assert lens.mean() < 1024 and lens.max() > 64  # what makes the host pick 32-record tiles
    seeds = rng.integers(0, 65536, size=n).astype(np.uint16)
    got = offload.csum_batch(arena, off16, lens, seeds)
    exp = O.csum_batch(arena, off16, lens, seeds)
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, (mix, bad[:5], lens[bad[:5]], off16[bad[:5]])
    # every prefix that ends inside a tile: the last tile is cut off at another record each time
    for m in (n - 1, n - 17, n - 31, 33):
        assert (offload.csum_batch(arena, off16[:m], lens[:m], seeds[:m]) == exp[:m]).all(), (mix, m)


@pytest.mark.parametrize("mode", [unet_amd.MODE_L3, unet_amd.MODE_L2])
def test_stream_tiles_verify_and_generate(offload, mode):
    """Valid and damaged UDP / ICMP / TCP packets of 28..1400 bytes in every layout: packed takes the stream route,
    the others fall back."""
    rng = np.random.default_rng(4242 + mode)
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode, ext_tcp=True)
    frames = _packets(rng, 2500, ifc, max_total=1400)
    seen = set()
    for name, arena, off16, stride16, lens in _layouts(rng, frames):
        got = offload.verify(arena, off16, lens, ifc, stride16=stride16)
        exp = O.recv_batch(arena, off16, lens, ifc, stride16=stride16)
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, (name, bad[:5], got[bad[:5]], exp[bad[:5]], lens[bad[:5]])
        seen |= set(got["verdict"].tolist())
        if mode == unet_amd.MODE_L3:
            got_arena, got_status = offload.generate(arena, off16, lens, stride16=stride16)
            exp_arena, exp_status = O.generate_batch(arena, off16, lens, stride16=stride16)
            assert (got_status == exp_status).all(), name
            assert (got_arena == exp_arena).all(), name
    for v in (Verdict.ACCEPT_UDP, Verdict.ACCEPT_ICMP_ECHO, Verdict.ACCEPT_TCP, Verdict.DROP_UDP_CSUM, Verdict.DROP_TCP_CSUM):
        assert int(v) in seen, v


def test_stream_tiles_of_the_synthetic_traces(offload):
    """IMIX and its odd-length twin, packed by the generator: tiles mix 64-byte records with jumbo frames."""
    ifc = unet_amd.Iface(ext_tcp=True)
    for kind in (unet_amd.SYNTH_IMIX, unet_amd.SYNTH_ODD):
        a, o, l = unet_amd.synth_host(kind, 0x756E6574, 12345, 6000, ifc)
        assert np.array_equal(offload.verify(a, o, l, ifc), O.recv_batch(a, o, l, ifc))
        # an empty record in the middle of a tile: that tile falls back, its neighbours do not
        l2 = l.copy()
        l2[100] = 0
        assert np.array_equal(offload.verify(a, o, l2, ifc), O.recv_batch(a, o, l2, ifc))
        # a record cut short (its slot keeps its size, so a gap follows it)
        l3 = l.copy()
        l3[200] = max(int(l3[200]) - 40, 1)
        assert np.array_equal(offload.verify(a, o, l3, ifc), O.recv_batch(a, o, l3, ifc))
