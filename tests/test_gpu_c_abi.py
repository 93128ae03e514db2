"""GPU (MI355X): the engine driven from a plain-C program through include/chimeralm_hip.h -- no Python, no torch in that
process -- gives the oracle's logits.  This is the drop-in boundary exactly as a foreign-language binding sees it."""
from __future__ import annotations

import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import hyena_oracle as ho

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def build_client(out_dir: Path, lib: Path) -> Path:
    exe = out_dir / "abi_client"
    subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-O1", f"-I{REPO / 'include'}", "-I/opt/rocm/include",
                    str(REPO / "tests" / "c_abi" / "abi_client.c"), "-o", str(exe), f"-L{lib.parent}", "-lchimeralm_hip",
                    "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def write_weights(path: Path, sd: dict) -> None:
    with path.open("wb") as f:
        for k, v in sd.items():
            a = np.ascontiguousarray(v.detach().cpu().numpy().astype(np.float32))
            kb = k.encode()
            f.write(struct.pack("<I", len(kb)) + kb + struct.pack("<I", a.ndim) + struct.pack(f"<{a.ndim}q", *a.shape))
            f.write(a.tobytes())


@pytest.mark.parametrize("prec,code,tol,bases", [("fp32", 0, 1e-3, 700), ("fp16", 2, 5e-3, 700), ("fp16c", 3, 1e-3, 2600),
                                                 ("fp16x3", 4, 1e-4, 700)])
def test_c_client_matches_oracle(tmp_path, built_lib, prec, code, tol, bases):
    """fp16c (code 3, the CLI / bench default) at a length its fp16 kernels run (>= 2,048 tokens); the client also calls
    clm_selfcheck and prints what it measured."""
    sd = ho.make_state_dict(0, head_scale=3.0)
    ids, _ = ho.synthetic_batch(7, 5, bases)
    ids[1, :9] = 4                                           # a left-padded read
    B, L = ids.shape
    write_weights(tmp_path / "weights.bin", sd)
    ids.astype(np.uint8).tofile(tmp_path / "ids.bin")
    exe = build_client(tmp_path, Path(built_lib))
    r = subprocess.run([str(exe), str(tmp_path / "weights.bin"), str(tmp_path / "ids.bin"), str(B), str(L), str(code)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.array([[float(x) for x in line.split()] for line in r.stdout.strip().splitlines()], dtype=np.float32)
    ref = ho.forward(torch.from_numpy(ids.astype(np.int64)), sd).numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= tol
    assert (got.argmax(1) == ref.argmax(1)).all()
    sc = [ln.split() for ln in r.stderr.splitlines() if ln.startswith("selfcheck ")]
    assert len(sc) == 1 and int(sc[0][2]) == 0
    if prec == "fp32":
        assert float(sc[0][1]) == 0.0
    else:
        assert 0.0 < float(sc[0][1]) <= tol
