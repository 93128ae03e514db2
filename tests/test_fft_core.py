"""CPU: the index algebra of the LDS Stockham convolution (csrc/fft_core.h, fft_passes.h) is emulated on the host
thread by thread and checked against a direct double-precision causal convolution for every transform size."""
from __future__ import annotations

import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "chimeralm_amd" / "csrc"


def test_stockham_convolution_emulation(tmp_path):
    exe = tmp_path / "fft_core_test"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), str(CSRC / "fft_core_test.cpp")], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout
    assert out.stdout.count("rel_err") == 35  # 7 sizes x 5 lengths incl. the aliased L = N/2 + 1 case
