"""MI355X offload of the ikanago/unet receive/transmit integrity path.

The compute lives in ``unet_amd/csrc`` (HIP, gfx950) behind the C ABI declared in
``include/unet_offload.h``; this package is the thin ctypes binding used by the tests and
the benchmark.  Importing it never falls back to a CPU implementation: if the shared
library is missing, ``load_library()`` raises.
"""
from ._ffi import (  # noqa: F401
    RESULT_DTYPE,
    V_COUNT,
    Iface,
    Offload,
    UnetError,
    Verdict,
    flow_hash5,
    load_library,
    pack_arena,
    synth_flow_hash,
    synth_host,
    synth_layout,
    records_per_wave,
    synth_len,
    tx_packets_per_wave,
)

MODE_L3 = 0
MODE_L2 = 1
SYNTH_IPV4_HDR20 = 2
SYNTH_UDP1500 = 3
SYNTH_IMIX = 4
SYNTH_FLOWS = 5
SYNTH_ODD = 6
