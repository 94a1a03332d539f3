"""A few rounds of the randomised parity soak (tests/fuzz_parity.py): random length mixes, layouts and modes through
unet_verify, unet_csum_batch and unet_generate, against the CPU oracle."""
import pytest

import unet_amd
from fuzz_parity import one_round

pytestmark = pytest.mark.gpu


def test_fuzz_rounds(offload):
    for seed in range(1000, 1040):
        one_round(offload, seed)
