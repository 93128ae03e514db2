"""SequenceCNNTransformer on MI355X (csrc/tf_model.hip + attention.hip through the clm_tf_* C ABI) against the oracle
(oracle/transformer_oracle.py, pinned to the reference module by tests/golden/transformer_golden.npz)."""
import numpy as np
import pytest
import torch

from oracle import transformer_oracle as to

pytestmark = pytest.mark.gpu

# fp32 = the reference's arithmetic (exact-fp32 MFMA, csrc/tf_fp32.hip): held to north_star's gate, 1e-3 on the logits.
# fp16 / bf16 are the throughput modes of this net, REDUCED precision outside the gate: logits of the seeded classifier
# (scale 3) are 3..9 in magnitude, their bounds are ~0.5 % / 4 % of that; the encoder output (O(1) LayerNorm values) is
# checked separately.
GATE = 1e-3
TOL = {"fp32": GATE, "fp16": 4e-2, "bf16": 4e-1}
TOL_HIDDEN = {"fp32": 2e-4, "fp16": 1e-2, "bf16": 1e-1}


def _model(sd, prec, layers=12):
    from chimeralm_amd.transformer import SequenceCNNTransformer

    net = SequenceCNNTransformer(vocab_size=12, max_len=32768, num_encoder_layers=layers, precision=prec)
    assert set(net.state_dict()) == set(sd)                     # reference keys, `pos_encoder.pe` buffer included
    net.load_state_dict(sd, strict=True)
    return net


@pytest.mark.parametrize("prec,seed,B,L,pads", [("fp32", 0, 2, 1000, 0), ("fp32", 1, 3, 777, 40), ("fp32", 2, 1, 8, 0),
                                                ("fp32", 3, 2, 2055, 0), ("fp32", 4, 3, 4101, 7),
                                                ("fp16", 0, 2, 1000, 0), ("fp16", 1, 3, 777, 40), ("fp16", 2, 1, 8, 0),
                                                ("fp16", 3, 2, 2055, 0), ("bf16", 0, 2, 1000, 0)])
def test_forward_matches_oracle(built_lib, golden_dir, prec, seed, B, L, pads):
    sd = to.make_state_dict(seed, to.PRODUCTION, scale=3.0)
    ids = to.synthetic_ids(100 + seed, B, L, pads)
    trace = {}
    ref = to.forward(torch.from_numpy(ids), sd, trace=trace).numpy()
    if (seed, B, L, pads) in ((0, 2, 1000, 0), (1, 3, 777, 40)):   # the very cases the reference module was run on
        g = np.load(golden_dir / "transformer_golden.npz")
        name = "prod" if seed == 0 else "prod_pad"
        assert np.abs(ref - g[f"{name}_logits"]).max() < 5e-5
    net = _model(sd, prec)
    got = net(torch.from_numpy(ids).cuda()).cpu().numpy()
    assert np.isfinite(got).all()
    L3 = L // 8
    hid = net.debug_fetch("hidden", (B, L3, 256))
    err_h = np.abs(hid - trace["layer11"].numpy()).max()
    err = np.abs(got - ref).max()
    print(f"{prec} B={B} L={L}: |logits - oracle| = {err:.2e}, |hidden - oracle| = {err_h:.2e}")
    assert err < TOL[prec] and err_h < TOL_HIDDEN[prec], f"{prec}: |logits - oracle| = {err:.2e} (hidden {err_h:.2e})"
    decided = np.abs(ref[:, 0] - ref[:, 1]) > (2 if prec == "fp32" else 4) * TOL[prec]
    assert (got.argmax(1)[decided] == ref.argmax(1)[decided]).all()
    pooled = net.debug_fetch("pooled", (B, 256))
    assert np.abs(pooled - trace["pooled"].numpy()).max() < TOL_HIDDEN[prec]
    # ids as uint8 with a row stride give the same bits
    wide = torch.zeros((B, L + 5), dtype=torch.uint8).cuda()
    wide[:, :L] = torch.from_numpy(ids.astype(np.uint8)).cuda()
    assert torch.equal(net(wide[:, :L]).cpu(), torch.from_numpy(got))
    net.close()


def test_arguments_and_errors(built_lib):
    from chimeralm_amd.transformer import SequenceCNNTransformer, TransformerEngineError

    with pytest.raises(NotImplementedError):
        SequenceCNNTransformer(vocab_size=12, max_len=64, d_model=128)
    net = SequenceCNNTransformer(vocab_size=12, max_len=16, num_encoder_layers=1)
    with pytest.raises(RuntimeError, match="MI355X only"):
        net(torch.zeros((1, 64), dtype=torch.int64))
    with pytest.raises(TransformerEngineError, match="Sequence too long"):     # transformer.py:21 asserts the same
        net(torch.full((1, 8 * 17), 7, dtype=torch.int64).cuda())
    with pytest.raises(TransformerEngineError, match="L >= 8"):
        net(torch.full((1, 7), 7, dtype=torch.int64).cuda())
    out = net(torch.full((2, 128), 7, dtype=torch.int64).cuda())
    assert out.shape == (2, 2) and torch.isfinite(out).all()
    net.close()
