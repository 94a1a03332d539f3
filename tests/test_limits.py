"""Limits that are refused with an error instead of being launched: a launch holds fewer than 2^32 threads; the static
device array of unet_verify_static serves one call at a time."""
import threading

import numpy as np
import pytest

import unet_amd
from unet_amd import _ffi

pytestmark = pytest.mark.gpu

SEED = 0x756E6574


def test_a_batch_that_would_need_2_pow_32_threads_is_refused_not_launched(offload):
    """One record per wavefront (here: a wrong length hint on 2^26 sixteen-byte records) times 2^26 records is 2^32
    threads: UNET_E_RANGE with the number of records that does fit, and nothing is written."""
    import torch
    n = 1 << 26
    ifc = unet_amd.Iface()
    d_arena = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
    d_len = torch.full((n,), 16, dtype=torch.int32, device="cuda")
    d_out = torch.full((n,), 0x55, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    offload.hint_mean_len(20000)  # -> one record per wave
    try:
        with pytest.raises(_ffi.UnetError) as e:
            offload.verify_dev(d_arena.data_ptr(), n * 16, 0, 1, d_len.data_ptr(), n, ifc, d_out.data_ptr())
        assert e.value.code == -5 and "split the batch" in str(e.value)
        # a few records fewer fit
        m = n - 1024
        offload.verify_dev(d_arena.data_ptr(), m * 16, 0, 1, d_len.data_ptr(), m, ifc, d_out.data_ptr())
        torch.cuda.synchronize()
    finally:
        offload.hint_mean_len(0)
    got = d_out.cpu().numpy()
    assert (got[m:] == 0x55).all()
    # sixteen zero bytes are no IPv4 packet: every record dropped at the version check, none left unjudged
    v = (got[:m].view(np.uint32) >> 16) & 0xFF
    assert (v == v[0]).all() and v[0] != 0


def test_trace_generator_slices_batches_of_2_pow_26_records(offload):
    """The generator runs one wave per record: 2^26 records go out as launches of 2^25, with the same lengths as the
    host twin at both ends and across the seam."""
    import torch
    n = (1 << 26) + 5
    d_len = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    offload.synth_len_dev(unet_amd.SYNTH_IMIX, SEED, 1000, None, n, d_len.data_ptr())
    torch.cuda.synchronize()
    for at in (0, (1 << 25) - 3, (1 << 26) - 3):
        cnt = min(8, n - at)
        _, lens, _ = _ffi.synth_layout(unet_amd.SYNTH_IMIX, SEED, 1000 + at, cnt)
        assert np.array_equal(d_len[at:at + cnt].cpu().numpy().view(np.uint32), lens), at


def test_static_array_serves_one_call_at_a_time():
    """Two contexts on one GPU share the static device array of unet_verify_static.  Calls from two threads either
    return the right verdicts or UNET_E_BUSY -- never verdicts of the other thread's batch."""
    ifc = unet_amd.Iface()
    batches = []
    for seed in (1, 2):
        arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_UDP1500, SEED + seed, 0, 4000, ifc)
        batches.append((arena, off16, lens))
    with unet_amd.Offload(0) as c0, unet_amd.Offload(0) as c1:
        want = [c0.verify(*batches[0], ifc), c0.verify(*batches[1], ifc)]
        assert not np.array_equal(want[0], want[1])
        stats = [{"ok": 0, "busy": 0, "bad": 0, "other": []} for _ in range(2)]

        def worker(k, ctx):
            for _ in range(150):
                try:
                    got = ctx.verify_static(*batches[k], ifc)
                except _ffi.UnetError as e:
                    if e.code == -6:
                        stats[k]["busy"] += 1
                    else:
                        stats[k]["other"].append(str(e))
                    continue
                stats[k]["ok" if np.array_equal(got, want[k]) else "bad"] += 1

        ts = [threading.Thread(target=worker, args=(k, c)) for k, c in enumerate((c0, c1))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for s in stats:
            assert s["bad"] == 0 and not s["other"], s
            assert s["ok"] > 0, s
        print("\nunet_verify_static from two threads:", stats)
