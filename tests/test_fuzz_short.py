"""A few rounds of the randomised parity soak (tests/fuzz_parity.py): random length mixes, layouts and modes through
unet_verify, unet_csum_batch, unet_generate and unet_tx_build, against the CPU oracle."""
import pytest

import unet_amd
from fuzz_parity import TX_CONTEXT_KNOBS, dev_round, one_round, open_with, tx_dev_round, tx_round

pytestmark = pytest.mark.gpu


def test_fuzz_rounds(offload):
    for seed in range(1000, 1040):
        one_round(offload, seed)
        tx_round(offload, seed)
        dev_round(offload, seed)
        tx_dev_round(offload, seed)
    # Against a -DUNET_DEBUG build of the library (UNET_OFFLOAD_LIB=.../variants/debug.so, tools/variants.sh debug) every
    # LDS store of these rounds was bounds-checked: none may have fallen outside its wave's area.  A release build
    # has no such counter.
    bad = offload.debug_lds_violations()
    assert bad in (None, 0), f"{bad} LDS stores outside their wave's area"


@pytest.mark.parametrize("knobs", TX_CONTEXT_KNOBS, ids=lambda k: ",".join(f"{a[5:]}={b}" for a, b in k.items()) or "default")
def test_fuzz_transmit_rounds_per_kernel_configuration(knobs):
    """The transmit rounds through a context opened with each set of kernel knobs: every route builds random batches
    of every length class, layout, hint and flag, not only the batches the library would choose it for."""
    from conftest import close_checked
    ctx = open_with(knobs)
    try:
        for seed in range(2000, 2012):
            tx_round(ctx, seed)
            tx_dev_round(ctx, seed)
    finally:
        close_checked(ctx)


def test_fuzz_rounds_from_two_threads_with_a_context_each():
    """Contexts are thread-compatible: two threads, each with a context of its own on the same GPU, run transmit and
    device-pointer rounds at the same time (the binding releases the interpreter lock inside every call)."""
    import threading
    from conftest import close_checked
    ctxs = [unet_amd.Offload(0), unet_amd.Offload(0)]
    errors = []

    def worker(k):
        try:
            for seed in range(3000 + 100 * k, 3000 + 100 * k + 25):
                tx_round(ctxs[k], seed)
                dev_round(ctxs[k], seed)
                tx_dev_round(ctxs[k], seed)
        except BaseException as e:  # noqa: BLE001 -- reported below, in the main thread
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    try:
        assert not errors, errors
    finally:
        for c in ctxs:
            close_checked(c)
