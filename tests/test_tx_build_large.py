"""unet_tx_build (host pointers) on batches too large for one staging image: the library cuts them into runs of packets
and keeps payloads going up, packets being built and packets coming back at the same time.  Every packet byte against
the CPU oracle, with the pipeline on and off, payloads packed, at a fixed stride, with gaps and in no order at all."""
import os

import numpy as np
import pytest

import unet_amd
from oracle import oracle_ffi as O
from unet_amd._ffi import TX_DESC_DTYPE, TX_E_PROTO, tx_out_layout

pytestmark = pytest.mark.gpu


def _open(pipeline: bool):
    old = os.environ.pop("UNET_TX_PIPELINE", None)
    if not pipeline:
        os.environ["UNET_TX_PIPELINE"] = "0"
    try:
        return unet_amd.Offload(0)
    finally:
        os.environ.pop("UNET_TX_PIPELINE", None)
        if old is not None:
            os.environ["UNET_TX_PIPELINE"] = old


def _descs(rng, n, ifc):
    desc = np.zeros(n, dtype=TX_DESC_DTYPE)
    desc["src"] = np.where(rng.random(n) < 0.2, 0, ifc.unicast)
    desc["dst"] = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    desc["l4a"] = rng.integers(0, 65536, size=n)
    desc["l4b"] = rng.integers(0, 65536, size=n)
    desc["values"] = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    desc["ip_id"] = rng.integers(0, 65536, size=n)
    desc["proto"] = np.where(rng.random(n) < 0.3, 1, 17)
    desc["proto"][rng.integers(0, n, size=7)] = 99  # refused: their slots come back as zeros
    desc["dst_mac"] = rng.integers(0, 256, size=(n, 6), dtype=np.uint8)
    return desc


def _layout(rng, lens, how):
    """(arena, off16 or None, stride16): payload i at 16 * off16[i] (packed, with gaps, or shuffled) or at i * 16 * stride16."""
    n = len(lens)
    slots = (lens.astype(np.int64) + 15) // 16
    if how == "stride":
        stride16 = int(slots.max())
        off = np.arange(n, dtype=np.int64) * stride16
        size = n * stride16 * 16
    else:
        stride16 = 0
        if how == "gaps":
            slots_g = slots + (rng.random(n) < 0.01) * rng.integers(1, 3000, size=n)
        else:
            slots_g = slots
        packed = np.concatenate([[0], np.cumsum(slots_g)[:-1]])
        if how == "shuffled":
            perm = rng.permutation(n)
            off = np.empty(n, dtype=np.int64)
            off[perm] = np.concatenate([[0], np.cumsum(slots[perm])[:-1]])
        else:
            off = packed
        size = int((off + slots).max()) * 16
    arena = rng.integers(0, 256, size=max(size, 16), dtype=np.uint8)
    return arena, (None if stride16 else off.astype(np.uint32)), stride16


CASES = [
    # (mode, layout, lengths)
    (unet_amd.MODE_L3, "packed", lambda r: r.choice([36, 548, 1472, 4068, 8972, 65507], p=[.3, .3, .3, .05, .04, .01], size=30000)),
    (unet_amd.MODE_L2, "packed", lambda r: r.integers(0, 4000, size=30000)),
    (unet_amd.MODE_L3, "stride", lambda r: np.where(r.random(40000) < 0.9, 1472, r.integers(0, 1473, size=40000))),
    (unet_amd.MODE_L2, "gaps", lambda r: r.choice([18, 1472, 9000], p=[.3, .6, .1], size=36000)),
    (unet_amd.MODE_L3, "shuffled", lambda r: r.integers(1000, 3000, size=30000)),  # not in index order: uploaded whole, then built
    (unet_amd.MODE_L2, "packed", lambda r: r.integers(1000, 1900, size=14000)),    # 20 MiB: would fit one image; runs of 4 MiB
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_large_batches_pipelined_and_not(case):
    from conftest import close_checked
    mode, how, draw = CASES[case]
    rng = np.random.default_rng(900 + case)
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode)
    plen = np.asarray(draw(rng), dtype=np.uint32)
    n = len(plen)
    arena, off16, stride16 = _layout(rng, plen, how)
    assert arena.nbytes > (17 << 20)  # several runs
    desc = _descs(rng, n, ifc)
    ooff, obytes = tx_out_layout(plen, mode)
    exp, exp_len, exp_status = O.tx_build_batch(arena, off16, plen, desc, ifc, ooff, obytes, pay_stride16=stride16)
    assert (exp_status == TX_E_PROTO).sum() >= 1
    ok = exp_status == 0
    built = np.zeros(obytes, dtype=bool)
    for o, l in zip(ooff[ok], exp_len[ok]):
        built[int(o) * 16:int(o) * 16 + int(l)] = True
    for pipeline in (True, False):
        ctx = _open(pipeline)
        try:
            for _ in range(2):  # (the second call meets buffers, events and pinned halves the first one left behind)
                got, got_off, got_len, got_status = ctx.tx_build(arena, off16, plen, desc, ifc, payload_stride16=stride16)
                assert (got_off == ooff).all() and got.nbytes == obytes
                assert (got_status == exp_status).all() and (got_len == exp_len).all()
                bad = np.flatnonzero(got[built] != exp[built])
                assert bad.size == 0, (pipeline, bad[:4])
                assert (got[~built] == 0).all()  # pads, and the slots of refused packets
        finally:
            close_checked(ctx)


def test_large_batches_from_and_into_pinned_memory():
    """Payload arena and packet arena allocated as pinned host memory by the caller (here: torch's pinned tensors):
    unet_tx_build and unet_verify copy straight from and into them.  Same bytes as through the staging buffers; one
    arena pinned and the other not works too."""
    import torch
    from conftest import close_checked
    rng = np.random.default_rng(77)
    mode = unet_amd.MODE_L3
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", mac=bytes.fromhex("02aabbccddee"), mode=mode)
    plen = np.asarray(rng.choice([36, 548, 1472, 8972], p=[.3, .3, .3, .1], size=30000), dtype=np.uint32)
    arena, off16, stride16 = _layout(rng, plen, "packed")
    desc = _descs(rng, len(plen), ifc)
    ooff, obytes = tx_out_layout(plen, mode)
    exp, exp_len, exp_status = O.tx_build_batch(arena, off16, plen, desc, ifc, ooff, obytes)
    built = np.zeros(obytes, dtype=bool)
    for o, l in zip(ooff[exp_status == 0], exp_len[exp_status == 0]):
        built[int(o) * 16:int(o) * 16 + int(l)] = True
    t_in = torch.empty(arena.nbytes, dtype=torch.uint8).pin_memory()
    t_out = torch.empty(obytes + 4096, dtype=torch.uint8).pin_memory()
    pinned_in, pinned_out = t_in.numpy(), t_out.numpy()
    pinned_in[:] = arena
    plain_out = np.empty(obytes + 4096, dtype=np.uint8)
    ctx = unet_amd.Offload(0)
    try:
        for src, dst in ((pinned_in, pinned_out), (pinned_in, plain_out), (arena, pinned_out)):
            dst[:] = 0xAB
            got, got_off, got_len, got_status = ctx.tx_build(src, off16, plen, desc, ifc, out=dst)
            assert (got_status == exp_status).all() and (got_len == exp_len).all() and (got_off == ooff).all()
            assert (got[built] == exp[built]).all() and (got[~built] == 0).all()
            assert (dst[obytes:] == 0xAB).all()
            # the packets just built, judged where they lie (a pinned arena read by unet_verify)
            res = ctx.verify(dst[:obytes], ooff, exp_len, ifc)
            assert (res == O.recv_batch(got.copy(), ooff, exp_len, ifc)).all()
    finally:
        close_checked(ctx)


def test_large_host_batches_through_the_receive_and_fill_entry_points():
    """unet_verify, unet_csum_batch and unet_generate on a 60 MB batch from pageable memory (uploaded through the pinned
    staging buffers, two at a time) and from pinned memory of the caller's own (uploaded as it lies), against the oracle."""
    import torch
    from conftest import close_checked
    ifc = unet_amd.Iface("192.0.2.2", "255.255.255.0", ext_tcp=True)
    arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_IMIX, 4242, 0, 100000, ifc)
    assert arena.nbytes > (40 << 20)
    t_pinned = torch.from_numpy(arena).pin_memory()
    rng = np.random.default_rng(3)
    seeds = rng.integers(0, 65536, size=len(lens)).astype(np.uint16)
    want = O.recv_batch(arena, off16, lens, ifc)
    want_sums = O.csum_batch(arena, off16, lens, seeds)
    want_filled, want_status = O.generate_batch(arena, off16, lens)
    ctx = unet_amd.Offload(0)
    try:
        for src in (arena, t_pinned.numpy()):
            assert np.array_equal(ctx.verify(src, off16, lens, ifc), want)
            assert np.array_equal(ctx.csum_batch(src, off16, lens, seeds), want_sums)
            filled, status = ctx.generate(src, off16, lens)
            assert np.array_equal(status, want_status) and np.array_equal(filled, want_filled)
    finally:
        close_checked(ctx)
