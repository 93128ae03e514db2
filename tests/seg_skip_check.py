"""Developer check (GPU): segment skipping of the long-read convolution on shapes the test suite does not hold -- odd chunks, L just past the
one-shot kernel, a length whose dot-product table does not apply, prefixes at every segment boundary.  python tests/seg_skip_check.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import hyena_oracle as ho
from chimeralm_amd.engine import Engine
sd = ho.make_state_dict(0, head_scale=3.0)
def batch(L, prefixes, seed):
    rng = np.random.default_rng(seed)
    ids = rng.integers(7, 11, size=(len(prefixes), L)).astype(np.uint8); ids[:, -1] = 1
    for b, p in enumerate(prefixes): ids[b, :p] = 4
    return ids
def engines(prec, chunk):
    os.environ.pop("CLM_DEBUG", None)
    a = Engine("cuda:0", precision=prec, chunk_reads=chunk); a.load_state_dict(sd)
    os.environ["CLM_DEBUG"] = "no_pad_skip"
    b = Engine("cuda:0", precision=prec, chunk_reads=chunk); b.load_state_dict(sd)
    os.environ.pop("CLM_DEBUG", None)
    if prec == "fp16c": a.set_f16c_min_len(1); b.set_f16c_min_len(1)
    return a, b
cases = [("fp16c", 3, 16385, (16000, 15000, 9000, 8500, 0, 12000, 16384)),      # LONE at S = 2, chunks of 3 (odd chunks)
         ("fp16c", 8, 8300, (8200, 8192, 8191, 129, 8064)),                     # just past the one-shot kernel
         ("bf16", 4, 24577, (24000, 20000, 17000, 16500, 100, 24576)),          # LONE, L != table length: segments not skipped
         ("fp16c", 8, 32769, (32768, 32768, 32640, 32000, 16384, 16383, 8192, 8193, 1))]
for prec, chunk, L, pf in cases:
    a, b = engines(prec, chunk)
    ids = batch(L, pf, 7 + L); t = torch.from_numpy(ids).cuda()
    x, y = a.forward(t).cpu().numpy(), b.forward(t).cpu().numpy()
    x2 = a.forward(t).cpu().numpy()
    print(prec, chunk, L, "max |skip - full|", float(np.abs(x - y).max()), "finite", bool(np.isfinite(x).all()), "deterministic", bool(np.array_equal(x, x2)), flush=True)
    a.close(); b.close()
