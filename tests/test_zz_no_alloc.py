"""unet_verify_static (the entry point that allocates nothing) against unet_verify, through the C ABI: same
kernel, staging in the library's static device array.  Two calls in a row, the second on a smaller batch, must
each equal unet_verify's answer bit for bit -- the array keeps the first call's bytes, so this is what shows
that nothing stale is read."""
import ctypes as C

import numpy as np
import pytest

import unet_amd
from unet_amd import _ffi

pytestmark = pytest.mark.gpu

SEED = 0x756E6574


def _both(ctx, arena, off16, lens, ifc, stride16=0):
    a = ctx.verify(arena, off16, lens, ifc, stride16=stride16)
    b = ctx.verify_static(arena, off16, lens, ifc, stride16=stride16)
    return a, b


def test_static_entry_point_equals_unet_verify_twice_in_a_row():
    from oracle import oracle_ffi as O

    ifc = unet_amd.Iface(ext_tcp=True)
    with unet_amd.Offload(0) as ctx:
        # first call: 6000 IMIX records (64..9000 bytes, every fault class), about 3.6 MB of arena
        arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_IMIX, SEED, 0, 6000, ifc)
        assert arena.nbytes <= (8 << 20)
        want, got = _both(ctx, arena, off16, lens, ifc)
        assert np.array_equal(want, got)
        assert np.array_equal(got, O.recv_batch(arena, off16, lens, ifc))
        # second call: a smaller batch of other records; the scratch still holds the first call's bytes behind it
        arena2, off2, lens2 = unet_amd.synth_host(unet_amd.SYNTH_IMIX, SEED, 777777, 157, ifc)
        want2, got2 = _both(ctx, arena2, off2, lens2, ifc)
        assert np.array_equal(want2, got2)
        assert np.array_equal(got2, O.recv_batch(arena2, off2, lens2, ifc))
        # third: fixed stride, L2 mode, one frame
        from helpers import config1_echo_frame
        frame, tap_mac, _ = config1_echo_frame()
        l2 = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=tap_mac, mode=1)
        a3, o3, l3 = unet_amd.pack_arena([frame])
        want3, got3 = _both(ctx, a3, None, l3, l2, stride16=7)
        assert np.array_equal(want3, got3) and int(got3["verdict"][0]) == unet_amd.Verdict.ACCEPT_ICMP_ECHO


def test_static_entry_point_limits_and_second_context():
    ifc = unet_amd.Iface()
    lib = _ffi.load_library()
    with unet_amd.Offload(0) as ctx, unet_amd.Offload(0) as ctx2:
        n = 65536 + 1
        lens = np.full(n, 20, dtype=np.uint32)
        arena = np.zeros(n * 32, dtype=np.uint8)
        with pytest.raises(_ffi.UnetError) as e:
            ctx.verify_static(arena, None, lens, ifc, stride16=2)
        assert e.value.code == -5 and "at most 65536 records" in str(e.value)
        big = np.zeros((8 << 20) + 16, dtype=np.uint8)
        with pytest.raises(_ffi.UnetError) as e:
            ctx.verify_static(big, np.array([(8 << 20) // 16], dtype=np.uint32), np.array([16], dtype=np.uint32), ifc)
        assert e.value.code == -5 and "first 8388608 bytes" in str(e.value)
        # the largest batch that fits: 65536 records filling exactly 8 MiB
        arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_UDP1500, SEED, 0, 5577, ifc)
        assert arena.nbytes <= (8 << 20)
        assert np.array_equal(ctx.verify(arena, off16, lens, ifc), ctx.verify_static(arena, off16, lens, ifc))
        rng = np.random.default_rng(5)
        small = rng.integers(0, 256, size=65536 * 128, dtype=np.uint8)
        l = np.full(65536, 113, dtype=np.uint32)
        assert np.array_equal(ctx.verify(small, None, l, ifc, stride16=8), ctx.verify_static(small, None, l, ifc, stride16=8))
        # a second context on the same GPU uses the same array (one after the other)
        assert np.array_equal(ctx2.verify_static(arena, off16, lens, ifc), ctx.verify(arena, off16, lens, ifc))
        # n = 0 is a no-op
        assert lib.unet_verify_static(ctx.h, None, 0, None, 0, None, 0, None, None) == 0
