"""The host-only entry points that read bytes they did not make -- unet_pcap_scan / _pack / _write on random and damaged
capture files, unet_pack_length_prefixed on damaged streams, the arena packers and transmit layouts -- into heap buffers
of exactly the size the library asked for, under the host-side address and undefined-behaviour sanitizers
(tests/cpp/host_fuzz.cpp, built by __graft_entry__.build()).  CPU only: no GPU call is made."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "host_fuzz_san")


def test_host_parsers_and_packers_under_sanitizers():
    assert os.path.exists(BIN), "build it with `python __graft_entry__.py build`"
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([BIN, "1500"], capture_output=True, text=True, timeout=300, env=env)
    if p.returncode != 0 and "runtime does not come first" in p.stderr:
        import pytest
        pytest.skip("another preloaded library keeps the sanitizer runtime from starting here")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "host_fuzz ok: 1500 rounds" in p.stdout
