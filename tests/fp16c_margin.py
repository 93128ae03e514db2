"""Margin of the `fp16c` mode under the 1e-3 gate over more weight draws and lengths than the test suite runs (developer tool;
test infrastructure: it imports the oracle).     python tests/fp16c_margin.py [n_draws]

For every seeded state dict (3x head scale, as in the parity tests) and length it prints the largest |logit error| of a batch of
4 reads against the fp32 oracle, the largest |logit| and whether the labels agree; the last line is the worst case."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from oracle import hyena_oracle as ho
from chimeralm_amd.engine import Engine


def main():
    n_draws = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    worst = (0.0, None)
    for wseed in range(n_draws):
        sd = ho.make_state_dict(wseed, head_scale=3.0)
        e = Engine("cuda:0", precision="fp16c", chunk_reads=4)
        e.load_state_dict(sd)
        for L in (2048, 3000, 4097, 8193):
            rng = np.random.default_rng(1000 * wseed + L)
            ids = rng.integers(7, 11, size=(4, L)).astype(np.uint8)
            ids[:, -1] = 1
            ids[0, : L // 3] = 4                                     # one read left-padded by a third
            ref = ho.forward(torch.from_numpy(ids.astype(np.int64)), sd).numpy()
            got = e.forward(torch.from_numpy(ids).cuda()).cpu().numpy()
            err = float(np.abs(got - ref).max())
            same = bool((got.argmax(1) == ref.argmax(1)).all())
            print(f"weights {wseed}  L {L:5d}: max |err| {err:.2e}  max |logit| {np.abs(ref).max():.2f}  labels {'identical' if same else 'DIFFER'}",
                  flush=True)
            if err > worst[0]:
                worst = (err, (wseed, L))
        e.close()
    print(f"worst of {n_draws} draws x 4 lengths: {worst[0]:.2e} at (weights, L) = {worst[1]}  (gate 1e-3)")


if __name__ == "__main__":
    main()
