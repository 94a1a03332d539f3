"""The C++ host layer (unet_amd/host/unet_host.hpp): arena packer, UDP PCB table, and -- on the GPU --
the batched drain loop with its host-side follow-ups, checked packet by packet against the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "host_test")


def run(mode):
    assert os.path.exists(BIN), "build it with `python __graft_entry__.py build`"
    p = subprocess.run([BIN, mode], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert f"host_test {mode} ok" in p.stdout


def test_host_logic_cpu():
    run("--cpu")


@pytest.mark.gpu
def test_batched_drain_loop_gpu():
    run("--gpu")
