"""Build the C-ABI shared library `csrc/libchimeralm_hip.so` for gfx950 with hipcc (in-tree, no JIT cache).

    python -m chimeralm_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
from __future__ import annotations

import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
INCLUDE = Path(__file__).resolve().parent.parent / "include"
LIB = CSRC / "libchimeralm_hip.so"
# hipcc's per-kernel resource remarks of the last build (registers, scratch, LDS): tests/test_kernel_resources.py holds the hot
# kernels to "no scratch" -- spills there are vector-memory traffic inside loops that were tuned to hide it (HISTORY.md section 4.7)
RESOURCES = CSRC / "kernel_resources.txt"
SOURCES = ["clm_api.hip", "gemm.hip", "gemm16.hip", "tail32.hip", "hyena_conv.hip", "head.hip", "pad_prefix.hip", "lone_token.hip", "attention.hip", "tf_model.hip", "tf_fp32.hip", "bam_feeder.cpp", "bam_filter.cpp"]
HEADERS = ["clm_common.h", "clm_lab.h", "gemm_common.h", "gemm16_common.h", "fft_core.h", "fft_passes.h", "bgzf.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function",
         "-Wno-pass-failed"]


def _stale() -> bool:
    if not LIB.exists() or not RESOURCES.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES + HEADERS] + [INCLUDE / "chimeralm_hip.h", INCLUDE / "chimeralm_feed.h"]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        obj = CSRC / (Path(s).stem + ".o")
        flags = FLAGS if s.endswith(".hip") else [f for f in FLAGS if not f.startswith("--offload-arch")]   # host-only C++
        cmd = ["hipcc", *flags, f"-I{INCLUDE}", f"-I{CSRC}", "-c", str(CSRC / s), "-o", str(obj)]
        if s.endswith(".hip"):
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(str(obj))
    failed = False
    remarks = []
    for s, p in procs:
        out, _ = p.communicate()
        kept = [ln for ln in out.splitlines() if "[-Rpass-analysis=kernel-resource-usage]" not in ln]
        remarks += [ln.split("remark: ", 1)[1].split(" [-Rpass-analysis", 1)[0].rstrip()
                    for ln in out.splitlines() if "[-Rpass-analysis=kernel-resource-usage]" in ln and "remark: " in ln]
        if p.returncode != 0 or verbose:
            print(f"--- hipcc {s} (rc={p.returncode})\n" + ("\n".join(kept) if not verbose else out), file=sys.stderr)
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed building the gfx950 engine")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *objs, "-lz", "-lpthread"], check=True)
    RESOURCES.write_text("\n".join(remarks) + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
