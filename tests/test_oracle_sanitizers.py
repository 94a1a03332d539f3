"""The oracle is what every GPU result is compared with: it must not read past a record itself.  oracle/oracle_san_check.c
feeds it random and header-shaped records in heap buffers of exactly their length, built with the address and
undefined-behaviour sanitizers, and checks the checksum properties the reference's own tests assert (src/utils.rs:43-60)
and that every packet the transmit side builds verifies on the receive side -- up to the two cases where the reference
itself builds what it would drop (a datagram sent from 0.0.0.0, a padded short datagram)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "oracle_san_check")


def test_oracle_reads_nothing_past_a_record():
    assert os.path.exists(BIN), "build it with `make -C oracle`"
    env = dict(os.environ, UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([BIN, "40000"], capture_output=True, text=True, timeout=300, env=env)
    if p.returncode != 0 and "runtime does not come first" in p.stderr:
        import pytest
        pytest.skip("another preloaded library keeps the sanitizer runtime from starting here")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "oracle_san_check ok: 40000 records" in p.stdout
