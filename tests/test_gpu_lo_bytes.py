"""GPU: the activations' lo bytes of fp16c in isolation (round 4; HISTORY.md section 4.10) -- tools/micro/mfma_lo2.cpp built against the
product's own headers (clm_common.h lo8_pack4 / lo8_unpack4, gemm_common.h frag_to_e5m2t / mfma_lo8 / mfma_lo2) and run on the card:
64-deep products of fp32 operands must come out an order of magnitude closer with the third MFMA term than with the weights' lo
alone, and fp16(x) + unpack(pack(x)) must be x to ~15 bits.  (This probe caught a real bug: a vector-element bit_cast in lo8_pack4
that wrote half 0 into every odd slot -- 'worst relative error 1.24e+04'.)"""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def test_activation_lo_term_and_byte_round_trip(tmp_path):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    exe = tmp_path / "mfma_lo2"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{REPO / 'chimeralm_amd' / 'csrc'}", f"-I{REPO / 'include'}",
                    "-o", str(exe), str(REPO / "tools" / "micro" / "mfma_lo2.cpp")], check=True, timeout=300)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    print(out.stdout)
    m = re.search(r"rms error: fp16 x fp16 ([0-9.e+-]+), \+ weights' lo ([0-9.e+-]+), \+ activations' lo ([0-9.e+-]+)", out.stdout)
    e_plain, e_wlo, e_alo = (float(m.group(i)) for i in (1, 2, 3))
    assert e_alo < e_wlo / 5 < e_plain / 5, out.stdout          # measured: 1.73e-3 -> 1.23e-3 -> 1.16e-4
    rt = float(re.search(r"worst relative error ([0-9.e+-]+)", out.stdout).group(1))
    assert rt < 6e-5, out.stdout                                 # 2^-15 .. 2^-14: measured 4.0e-5 (fp16 alone: 2.4e-4)


def test_fp16x3_product_is_fp32_class(tmp_path):
    """tools/micro/mfma_x3.cpp: 64-deep products of fp32 operands as one fp16 MFMA, as fp16x3's three (hi + lo halfs, weights scaled by
    2^10) and on the fp32 MFMA, against the double product: fp16x3 must sit three orders of magnitude under plain fp16 and within a
    few times the fp32 MFMA's own rounding."""
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    exe = tmp_path / "mfma_x3"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", str(exe), str(REPO / "tools" / "micro" / "mfma_x3.cpp")],
                   check=True, timeout=300)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    print(out.stdout)
    m = re.search(r"rms error: fp16 x fp16 ([0-9.e+-]+), fp16x3 ([0-9.e+-]+), fp32 MFMA ([0-9.e+-]+)", out.stdout)
    e16, ex3, e32 = (float(m.group(i)) for i in (1, 2, 3))
    assert ex3 < e16 / 300 and ex3 < 20 * e32, out.stdout
