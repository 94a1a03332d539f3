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
