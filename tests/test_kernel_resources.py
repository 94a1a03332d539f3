"""Registers, scratch and LDS of the tile kernels, from the compiler's own resource remarks (no GPU needed): eight
waves per SIMD (at most 64 VGPRs), no scratch memory, and an LDS footprint that lets eight workgroups share a CU.
A change that looks harmless in the source can cost 16 registers and with them a quarter of the resident waves."""
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")


def _resources():
    out = subprocess.run(["python", "tools/kres.py"], capture_output=True, text=True, timeout=600).stdout
    rows = {}
    for line in out.splitlines():
        m = re.match(r"(\S.*?)\s+VGPR\s+(\d+)\s+SGPR\s+(\d+)\s+scratch\s+(\d+)\s+occ\s+(\d+)\s+LDS\s+(\d+)\s+spill\s+(\d+)", line)
        if m:
            rows[m.group(1)] = dict(vgpr=int(m.group(2)), scratch=int(m.group(4)), occ=int(m.group(5)), lds=int(m.group(6)), spill=int(m.group(7)))
    return rows


def test_tile_kernels_keep_eight_waves_per_simd():
    rows = _resources()
    tile = {k: v for k, v in rows.items() if k.startswith(("k_verify<", "k_csum<", "k_generate<"))}
    assert len(tile) == 15, sorted(rows)
    for name, r in tile.items():
        assert r["vgpr"] <= 64 and r["occ"] == 8, (name, r)
        assert r["scratch"] == 0 and r["spill"] == 0, (name, r)
        assert 8 * r["lds"] <= 160 * 1024, (name, r)
    # (the synthetic trace generators build packets in per-lane arrays; they are not on the measured path)
    for name, r in rows.items():
        if not name.startswith("k_synth"):
            assert r["scratch"] == 0 and r["spill"] == 0, (name, r)
