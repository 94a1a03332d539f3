"""The bandwidth yardsticks bench.py reports beside every kernel read each byte of their buffer exactly once: every
shape leaves the sum of the buffer's 16-bit words in its sink, whatever the buffer's size; the copy yardstick copies."""
import numpy as np
import pytest

import unet_amd

pytestmark = pytest.mark.gpu


def _sum16(a: np.ndarray) -> int:
    return int(a.view("<u2").astype(np.uint64).sum() & 0xFFFFFFFF)


@pytest.mark.parametrize("nbytes", [16, 1008, 1024 * 24, 1024 * 24 + 16, 999_984, 50_000_016])
def test_every_read_shape_sums_the_whole_buffer(offload, nbytes):
    import torch
    rng = np.random.default_rng(nbytes)
    host = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
    d_buf = torch.from_numpy(host).cuda()
    want = _sum16(host)
    shapes = {
        "slabs": lambda sink: offload.stream_read_dev(d_buf.data_ptr(), nbytes, sink),
        "pieces": lambda sink: offload.stream_read_pieces_dev(d_buf.data_ptr(), nbytes, sink),
        "pieces_full_occupancy": lambda sink: offload.stream_read_pieces_full_dev(d_buf.data_ptr(), nbytes, sink),
    }
    for name, fn in shapes.items():
        d_sink = torch.zeros(4096, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        fn(d_sink.data_ptr())
        offload.sync()
        torch.cuda.synchronize()
        got = int(d_sink.cpu().numpy().view(np.uint32).astype(np.uint64).sum() & 0xFFFFFFFF)
        assert got == want, (name, nbytes)
    # records of a given size: as many whole records as the buffer holds (twice, the second call reuses the length table)
    for rec in (16, 100, 1500, 1504, 9000, 65520):
        slots = (rec + 15) // 16
        whole = nbytes // 16 // slots * slots * 16
        for _ in range(2):
            d_sink = torch.zeros(4096, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            offload.stream_read_records_dev(d_buf.data_ptr(), nbytes, rec, d_sink.data_ptr())
            offload.sync()
            torch.cuda.synchronize()
            got = int(d_sink.cpu().numpy().view(np.uint32).astype(np.uint64).sum() & 0xFFFFFFFF)
            assert got == _sum16(host[:whole]), (rec, nbytes)


def test_record_shape_refuses_sizes_it_cannot_read(offload):
    import torch
    from unet_amd import _ffi
    d_buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    d_sink = torch.zeros(4096, dtype=torch.int32, device="cuda")
    for rec in (0, 65521, 65535, 70000):
        with pytest.raises(_ffi.UnetError) as e:
            offload.stream_read_records_dev(d_buf.data_ptr(), 1 << 20, rec, d_sink.data_ptr())
        assert e.value.code == -5


@pytest.mark.parametrize("nbytes", [16, 4096 + 48, 30_000_000 // 16 * 16])
def test_copy_yardstick_copies(offload, nbytes):
    import torch
    rng = np.random.default_rng(nbytes)
    host = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
    d_src = torch.from_numpy(host).cuda()
    d_dst = torch.full((nbytes + 64,), 0xAB, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    offload.stream_copy_dev(d_src.data_ptr(), d_dst.data_ptr(), nbytes)
    offload.sync()
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    assert np.array_equal(got[:nbytes], host) and (got[nbytes:] == 0xAB).all()
