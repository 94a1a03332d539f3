"""unet_verify_async / unet_wait (two batches in flight, results identical to unet_verify) and the per-context
verdict counters (unet_counters, unet_counters_sum), through the C ABI on the GPU."""
import ctypes as C

import numpy as np
import pytest

import unet_amd
from unet_amd import _ffi
from oracle import oracle_ffi as O

pytestmark = pytest.mark.gpu

SEED = 0x756E6574


def test_two_batches_in_flight_match_the_synchronous_call():
    ifc = unet_amd.Iface(ext_tcp=True)
    with unet_amd.Offload(0) as ctx:
        batches = [unet_amd.synth_host(unet_amd.SYNTH_IMIX, SEED, 1000 * k, n, ifc) for k, n in enumerate((5000, 300, 1, 7000, 64))]
        want = [O.recv_batch(a, o, l, ifc) for a, o, l in batches]
        # pipeline of depth two: submit k+1 before waiting for k
        pending, got = [], []
        for a, o, l in batches:
            # the arena may be reused right after the submit: hand over a scratch copy and wreck it afterwards
            a2 = a.copy()
            t, out = ctx.verify_async(a2, o, l, ifc)
            a2[:] = 0xFF
            pending.append((t, out))
            if len(pending) == 2:
                t0, out0 = pending.pop(0)
                ctx.wait(t0)
                got.append(out0)
        for t0, out0 in pending:
            ctx.wait(t0)
            got.append(out0)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
        # a third submit while two are in flight is refused, and a ticket cannot be waited for twice
        (a, o, l) = batches[0]
        t1, _ = ctx.verify_async(a, o, l, ifc)
        t2, _ = ctx.verify_async(a, o, l, ifc)
        with pytest.raises(_ffi.UnetError) as e:
            ctx.verify_async(a, o, l, ifc)
        assert e.value.code == -6
        ctx.wait(t1)
        with pytest.raises(_ffi.UnetError):
            ctx.wait(t1)
        t3, out3 = ctx.verify_async(a, o, l, ifc)  # the freed slot is usable again
        ctx.wait(t2)
        ctx.wait(t3)
        assert np.array_equal(out3, want[0])
        # an empty batch gives the "nothing to wait for" ticket
        t, out = ctx.verify_async(a, o[:0], l[:0], ifc)
        assert t == 0xFFFFFFFF
        ctx.wait(t)


def test_context_counters_follow_every_host_pointer_call():
    ifc = unet_amd.Iface(ext_tcp=True)
    lib = _ffi.load_library()
    with unet_amd.Offload(0) as ctx, unet_amd.Offload(0) as ctx2:
        assert not ctx.counters().any()
        total = np.zeros(unet_amd.V_COUNT, dtype=np.uint64)
        a, o, l = unet_amd.synth_host(unet_amd.SYNTH_IMIX, SEED, 0, 4000, ifc)
        r = ctx.verify(a, o, l, ifc)
        total += np.bincount(r["verdict"], minlength=unet_amd.V_COUNT).astype(np.uint64)
        r = ctx.verify_static(a, o, l, ifc)
        total += np.bincount(r["verdict"], minlength=unet_amd.V_COUNT).astype(np.uint64)
        t, out = ctx.verify_async(a, o, l, ifc)
        ctx.wait(t)
        total += np.bincount(out["verdict"], minlength=unet_amd.V_COUNT).astype(np.uint64)
        hdrs, _, _ = unet_amd.synth_host(unet_amd.SYNTH_IPV4_HDR20, SEED, 0, 5000, ifc)
        r = ctx.ipv4_validate(hdrs[:5000 * 20], ifc)
        total += np.bincount(r["verdict"], minlength=unet_amd.V_COUNT).astype(np.uint64)
        # a batch too large for one staging buffer takes the chunked path
        big_a, big_o, big_l = unet_amd.synth_host(unet_amd.SYNTH_UDP1500, SEED, 0, 30000, ifc)
        assert big_a.nbytes > (32 << 20)
        r = ctx.verify(big_a, big_o, big_l, ifc)
        assert np.array_equal(r, O.recv_batch(big_a, big_o, big_l, ifc))
        total += np.bincount(r["verdict"], minlength=unet_amd.V_COUNT).astype(np.uint64)
        assert np.array_equal(ctx.counters(), total)
        # a second context counts separately; the sum is the sum
        r2 = ctx2.verify(a, o, l, ifc)
        c2 = np.bincount(r2["verdict"], minlength=unet_amd.V_COUNT).astype(np.uint64)
        assert np.array_equal(ctx2.counters(), c2)
        both = np.zeros(unet_amd.V_COUNT, dtype=np.uint64)
        handles = (C.c_void_p * 2)(ctx.h.value, ctx2.h.value)
        assert lib.unet_counters_sum(handles, 2, both.ctypes.data, 1) == 0
        assert np.array_equal(both, total + c2)
        assert not ctx.counters().any() and not ctx2.counters().any()  # reset by the summing call
