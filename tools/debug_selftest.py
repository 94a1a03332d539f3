#!/usr/bin/env python3
"""Self-test of the -DUNET_DEBUG build: against a library that also has -DUNET_DEBUG_SPAN_WHOLE_ROWS (the fault of
storing whole rows into the 112-chunk area of 16-record tiles, brought back on purpose) the counter of out-of-area LDS
stores must move on the batch shapes that used to show the fault; against a plain -DUNET_DEBUG build it must stay 0.
usage: UNET_OFFLOAD_LIB=<variant.so> python tools/debug_selftest.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import unet_amd
from test_short_records import test_short_record_tiles_inside_a_batch_of_long_records as batch_shapes

ctx = unet_amd.Offload(0)
assert ctx.debug_lds_violations(reset=True) is not None, "not a -DUNET_DEBUG build"
batch_shapes(ctx)
print(os.environ.get("UNET_OFFLOAD_LIB", "default library"), "out-of-area LDS stores counted and dropped:", ctx.debug_lds_violations())
ctx.close()
