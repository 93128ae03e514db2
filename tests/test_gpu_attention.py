"""Self-attention kernel of the SequenceCNNTransformer encoder (csrc/attention.hip) against the oracle's attention
(oracle/transformer_oracle.py::attention, pinned to the reference module through the whole-model goldens)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import transformer_oracle as to

pytestmark = pytest.mark.gpu


def _run(qkv: torch.Tensor, prec: int) -> torch.Tensor:
    from chimeralm_amd import _native as N

    lib = N.load()
    B, L, _ = qkv.shape
    out = torch.empty((B, L, 256), dtype=qkv.dtype, device=qkv.device)
    rc = lib.clm_attention_fwd(C.c_void_p(qkv.data_ptr()), C.c_void_p(out.data_ptr()), B, L, prec,
                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    return out


def _reference(qkv: torch.Tensor) -> torch.Tensor:
    B, L, _ = qkv.shape
    q, k, v = (t.reshape(B, L, 8, 32).transpose(1, 2) for t in qkv.float().split(256, dim=-1))
    return to.attention(q, k, v).transpose(1, 2).reshape(B, L, 256)


@pytest.mark.parametrize("B,L", [(1, 1), (2, 31), (1, 64), (3, 65), (2, 128), (1, 129), (2, 1000), (1, 1024), (1, 4096)])
def test_attention_matches_oracle_fp16(built_lib, B, L):
    from chimeralm_amd import _native as N

    rng = np.random.default_rng(L * 7 + B)
    x = rng.standard_normal((B, L, 768)).astype(np.float32)
    x[..., :256] *= 1.5                                                       # sharper softmax than unit-variance scores
    x[0, L // 2, 256:512] *= 4.0                                              # one dominant key: exercises the running maximum
    qkv = torch.from_numpy(x).half()
    ref = _reference(qkv)
    got = _run(qkv.cuda(), N.PREC_F16).cpu().float()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    assert err < 3e-3, f"max |attention - oracle| = {err:.2e}"
    again = _run(qkv.cuda(), N.PREC_F16).cpu().float()
    assert torch.equal(got, again)                                            # deterministic: fixed reduction order


def test_attention_bf16_and_arguments(built_lib):
    from chimeralm_amd import _native as N

    rng = np.random.default_rng(3)
    qkv = torch.from_numpy(rng.standard_normal((2, 300, 768)).astype(np.float32)).bfloat16()
    got = _run(qkv.cuda(), N.PREC_BF16).cpu().float()
    assert (got - _reference(qkv)).abs().max().item() < 2.5e-2
    lib = N.load()
    assert lib.clm_attention_fwd(None, None, 1, 1, N.PREC_F16, None) == N.E_INVALID
    assert lib.clm_attention_fwd(C.c_void_p(1), C.c_void_p(1), 1, 1, N.PREC_F32, None) == N.E_INVALID   # 16-bit only
