"""Developer probe (GPU; test infrastructure: it imports the oracle; not collected by pytest): raw error of the transformer's fp16c / fp16 modes against the fp64 oracle, by weight scale and length."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import transformer_oracle as to  # noqa: E402
from chimeralm_amd.transformer import SequenceCNNTransformer  # noqa: E402

for scale in (1.0, 3.0):
    for seed, B, L in ((0, 2, 1000), (1, 2, 2055), (2, 2, 4101), (3, 2, 8193), (4, 2, 8193)):
        sd = to.make_state_dict(seed, to.PRODUCTION, scale=scale)
        ids = to.synthetic_ids(100 + seed, B, L)
        ref = to.forward(torch.from_numpy(ids), {k: v.double() for k, v in sd.items()}, dtype=torch.float64).numpy()
        row = f"scale {scale} seed {seed} L {L} |logit| {np.abs(ref).max():.2f}:"
        for prec in ("fp32", "fp16c", "fp16"):
            net = SequenceCNNTransformer(vocab_size=12, max_len=32768, num_encoder_layers=12, precision=prec, selfcheck=False)
            net.load_state_dict(sd, strict=True)
            got = net(torch.from_numpy(ids).cuda()).cpu().numpy().astype(np.float64)
            row += f"  {prec} {np.abs(got - ref).max():.2e}"
            net.close()
        print(row, flush=True)
