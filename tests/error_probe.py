"""Accuracy report of the engine precisions against the CPU oracle (fp32 = reference precision, fp64 = referee), at unit and at
3x classifier scale (the tests use 3x so that label checks have margins).  Developer tool (imports the oracle: lives under
tests/).  GPU box:  python tests/error_probe.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from chimeralm_amd.engine import Engine
from oracle import hyena_oracle as ho

precs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fp32", "fp16c", "fp16", "bf16"]
for wseed, hs in ((0, 1.0), (0, 3.0), (3, 3.0)):
    sd = ho.make_state_dict(wseed, head_scale=hs)
    for (B, L) in ((6, 100), (6, 1000), (3, 8193)):
        ids, _ = ho.synthetic_batch(5, B, L - 1, seed=99)
        t = torch.from_numpy(ids.astype(np.int64))
        ref, ref64 = ho.forward(t, sd).numpy(), ho.forward(t, sd, dt=torch.float64).numpy()
        for prec in precs:
            e = Engine("cuda:0", precision=prec, chunk_reads=4)
            e.load_state_dict(sd)
            got = e.forward(torch.from_numpy(ids).cuda()).cpu().numpy()
            e.close()
            print(f"weights {wseed} head x{hs:g}  {B} x {L}  {prec:5s}: max |err| vs fp32 oracle {np.abs(got - ref).max():.2e}, vs fp64 {np.abs(got - ref64).max():.2e}"
                  f"  (fp32 oracle vs fp64 {np.abs(ref - ref64).max():.2e});  max |logit| {np.abs(ref).max():.2f}, smallest margin "
                  f"{np.abs(ref[:, 0] - ref[:, 1]).min():.3f}", flush=True)
