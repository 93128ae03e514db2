"""CPU: the host-side C++ of the engine library (BAM feeder with its inflate workers and slot ring, BGZF reader / writer threads,
filter with the SAM parser, spilling coordinate sort + BAI) under AddressSanitizer + UndefinedBehaviorSanitizer and under
ThreadSanitizer.  The sources are compiled with g++ straight from csrc/ (the HIP kernels are not involved; the feeder's pinned
ring is off), linked with a C++ driver that exercises them through the C ABI (tests/sanitize/host_driver.cpp)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "chimeralm_amd" / "csrc"


def _build(tmp: Path, flags: list[str], name: str) -> Path:
    exe = tmp / name
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", *flags, f"-I{REPO / 'include'}", f"-I{CSRC}", "-I/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__", str(CSRC / "bam_feeder.cpp"), str(CSRC / "bam_filter.cpp"), str(REPO / "tests/sanitize/host_driver.cpp"),
           "-o", str(exe), "-L/opt/rocm/lib", "-lamdhip64", "-lz", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "cannot find -lamdhip64" in r.stderr:
        pytest.skip("HIP runtime library not present for linking the host sources")
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.parametrize("kind,flags,env", [
    ("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
     {"ASAN_OPTIONS": "halt_on_error=1:detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}),
    ("tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1:second_deadlock_stack=1"}),
])
def test_host_cpp_under_sanitizers(tmp_path, golden_dir, kind, flags, env):
    from test_host import _bam_to_sam

    exe = _build(tmp_path, flags, f"driver_{kind}")
    bam = tmp_path / "reads.bam"
    shutil.copyfile(golden_dir / "test_chimric_reads.bam", bam)
    sam = tmp_path / "reads.sam"
    sam.write_text(_bam_to_sam(bam))
    r = subprocess.run([str(exe), str(bam), str(sam), str(tmp_path)], capture_output=True, text=True, timeout=900,
                       env={**os.environ, **env, "CLM_BAM_THREADS": "3"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "sanitizer driver OK" in r.stdout
    for marker in ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "ERROR: LeakSanitizer"):
        assert marker not in r.stderr, r.stderr[-4000:]
