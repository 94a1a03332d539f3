#!/usr/bin/env python3
"""Microseconds per unet_verify call on a few long records: one wavefront per 4 KiB segment (what unet_verify
does for such a batch) against the ordinary tiling (forced with UNET_TILE, which switches the split off), and the
device-pointer forms of both on records already in HBM.  Prints one JSON object per line."""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def timed(fn, min_s=0.25, min_reps=5):
    fn()
    reps, t0 = 0, time.perf_counter()
    while True:
        fn()
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= min_s and reps >= min_reps:
            return dt / reps * 1e6


def main():
    import numpy as np
    import torch

    import unet_amd
    from unet_amd import _ffi
    from batch_sweep import make_batch

    ifc = unet_amd.Iface()
    ctx = unet_amd.Offload(0)
    mode = "ordinary tiling (UNET_TILE=%s)" % os.environ["UNET_TILE"] if os.environ.get("UNET_TILE") else "library's choice"
    for n, size in [(1, 65535), (4, 65535), (16, 65535), (64, 65535), (16, 20000), (64, 20000), (16, 9000)]:
        flat, slots, lens = make_batch(n, size, ifc)
        row = {"records": n, "bytes": size, "mode": mode, "unet_verify_us": round(timed(lambda: ctx.verify(flat, None, lens, ifc, stride16=slots)), 1)}
        d_arena = torch.from_numpy(flat).cuda()
        d_len = torch.from_numpy(lens.view(np.int32)).cuda()
        d_out = torch.zeros(n, dtype=torch.int32, device="cuda")
        need = _ffi.verify_split_workspace(n, size)
        d_ws = torch.zeros(need // 4 + 1, dtype=torch.int32, device="cuda")

        def dev():
            ctx.verify_dev(d_arena.data_ptr(), flat.nbytes, None, slots, d_len.data_ptr(), n, ifc, d_out.data_ptr())
            ctx.sync()

        def split():
            ctx.verify_split_dev(d_arena.data_ptr(), flat.nbytes, None, slots, d_len.data_ptr(), n, size, ifc, d_out.data_ptr(), d_ws.data_ptr(), need)
            ctx.sync()

        row["unet_verify_dev_us"] = round(timed(dev), 1)
        row["unet_verify_split_dev_us"] = round(timed(split), 1)
        print(json.dumps(row), flush=True)
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "both":
        # second pass in a fresh process with the split switched off
        main()
        env = dict(os.environ, UNET_TILE="4")
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)], env=env))
    main()
