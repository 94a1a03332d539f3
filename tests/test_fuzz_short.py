"""A few rounds of the randomised parity soak (tests/fuzz_parity.py): random length mixes, layouts and modes through
unet_verify, unet_csum_batch, unet_generate and unet_tx_build, against the CPU oracle."""
import pytest

import unet_amd
from fuzz_parity import dev_round, one_round, tx_dev_round, tx_round

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
