"""examples/drain_loop.c: the C ABI from a plain C program (C11, linked against the library with gcc) -- a drained queue
judged by one unet_verify call, the accepted datagrams answered by one unet_tx_build call, the answers judged again."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "drain_loop")


@pytest.mark.gpu
@pytest.mark.parametrize("frames", [1, 300, 20000])
def test_plain_c_caller(frames):
    assert os.path.exists(BIN), "build it with `python __graft_entry__.py build`"
    p = subprocess.run([BIN, str(frames)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "drain_loop ok" in p.stdout
