"""Every device-pointer entry point that takes a batch of records refuses bad arguments with UNET_E_INVAL before anything
is launched (outputs keep their fill), and treats n = 0 as a no-op."""
import ctypes as C

import numpy as np
import pytest

import unet_amd
from unet_amd import _ffi

pytestmark = pytest.mark.gpu

E_INVAL = -1
N = 64


def _descs():
    d = np.zeros(N, dtype=_ffi.TX_DESC_DTYPE)
    d["proto"], d["l4a"], d["l4b"], d["dst"] = 17, 7, 9, 0x0A000001
    return d


def _setup():
    import torch
    ifc = unet_amd.Iface()
    arena, off16, lens = unet_amd.synth_host(unet_amd.SYNTH_IMIX, 5, 0, N, ifc)
    t = {
        "arena": torch.from_numpy(arena).cuda(), "off": torch.from_numpy(off16.view(np.int32)).cuda(),
        "len": torch.from_numpy(lens.view(np.int32)).cuda(),
        "out": torch.full((N * 4 + 64,), 0x5A, dtype=torch.uint8, device="cuda"),    # results / sums / fields / hashes
        "out2": torch.full((N + 64,), 0x5A, dtype=torch.uint8, device="cuda"),       # status bytes / segment states
        "seeds": torch.zeros(N, dtype=torch.int16, device="cuda"),
        "desc": torch.from_numpy(_descs().view(np.uint8).reshape(-1).copy()).cuda(),
        "ooff": torch.arange(N, dtype=torch.int32, device="cuda") * 600,
        "olen": torch.full((N,), 0x5A5A5A5A, dtype=torch.int32, device="cuda"),
        "oarena": torch.full((N * 9600,), 0x5A, dtype=torch.uint8, device="cuda"),
        "table": torch.zeros(1024 * 16, dtype=torch.uint8, device="cuda"),
        "ws": torch.zeros(1 << 20, dtype=torch.uint8, device="cuda"),
    }
    torch.cuda.synchronize()
    return ifc, arena.nbytes, t


def _calls(lib, ifc_c, nbytes, t):
    """name -> (function, argument list, index of: arena, off16, stride16, len, n, iface or None, outputs to watch)."""
    p = {k: v.data_ptr() for k, v in t.items()}
    head = [None, p["arena"], nbytes, p["off"], 0, p["len"]]  # stream, arena, bytes, off16, stride16, len  (ctx goes in front)
    I = C.byref(ifc_c)
    return {
        "unet_csum_batch_dev": (head + [p["seeds"], N, p["out"]], None, ["out"]),
        "unet_verify_dev": (head + [N, I, p["out"]], 7, ["out"]),
        "unet_generate_dev": (head + [N, p["out2"]], None, ["out2"]),
        "unet_checksum_fields_dev": (head + [N, p["out"], p["out2"]], None, ["out", "out2"]),
        "unet_flow_hash_dev": (head + [N, I, p["out"]], 7, ["out"]),
        "unet_tx_build_dev": (head + [p["desc"], N, I, p["oarena"], t["oarena"].numel(), p["ooff"], p["olen"], p["out2"]], 8, ["oarena", "olen", "out2"]),
        "unet_verify_split_dev": (head + [N, 9000, I, p["out"], p["ws"], t["ws"].numel()], 8, ["out"]),
    }


def _untouched(t, names):
    import torch
    torch.cuda.synchronize()
    for k in names:
        v = t[k]
        if v.dtype == torch.int32:
            assert bool((v == 0x5A5A5A5A).all()), k
        else:
            assert bool((v == 0x5A).all()), k


@pytest.mark.parametrize("name", ["unet_csum_batch_dev", "unet_verify_dev", "unet_generate_dev", "unet_checksum_fields_dev", "unet_flow_hash_dev",
                                  "unet_tx_build_dev", "unet_verify_split_dev"])
def test_bad_arguments_are_refused_before_anything_runs(offload, name):
    lib = _ffi.load_library()
    ifc, nbytes, t = _setup()
    ic = ifc.c()
    args, iface_at, outs = _calls(lib, ic, nbytes, t)[name]
    fn = getattr(lib, name)
    n_at = {"unet_csum_batch_dev": 7, "unet_tx_build_dev": 7}.get(name, 6)  # where n sits in `args`

    def call(ctx, a):
        return fn(ctx, *a)

    # null context
    assert call(None, args) == E_INVAL
    _untouched(t, outs)
    # null arena, arena off its 16-byte boundary, no offsets and no stride, null lengths
    for at, bad in ((1, None), (1, args[1] + 4), (5, None)):
        a = list(args)
        a[at] = bad
        assert call(offload.h, a) == E_INVAL, (name, at)
        assert _ffi.load_library().unet_last_error()
        _untouched(t, outs)
    a = list(args)
    a[3], a[4] = None, 0
    assert call(offload.h, a) == E_INVAL
    _untouched(t, outs)
    # an interface mode that does not exist
    if iface_at is not None:
        bad_ifc = unet_amd.Iface().c()
        bad_ifc.mode = 7
        a = list(args)
        a[iface_at] = C.byref(bad_ifc)
        assert call(offload.h, a) == E_INVAL, name
        _untouched(t, outs)
    # n = 0: nothing to do, nothing touched (whatever the other arguments are)
    a = list(args)
    a[n_at] = 0
    assert call(offload.h, a) == 0, name
    _untouched(t, outs)
    # and the call as it should be made works
    assert call(offload.h, args) == 0, (name, _ffi.load_library().unet_last_error())
    offload.sync()
    import torch
    torch.cuda.synchronize()
    assert not bool((t[outs[0]] == (0x5A5A5A5A if t[outs[0]].dtype == torch.int32 else 0x5A)).all())
