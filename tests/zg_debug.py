"""Developer check of the gated z hand-over (GPU): z / y of the LAST block as the two hand-over forms leave them.
    python tests/zg_debug.py [B L precision]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from oracle import hyena_oracle as ho
from chimeralm_amd.engine import Engine

B, L, prec = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]) if len(sys.argv) > 3 else (2, 700, "fp16")
sd = ho.make_state_dict(0, head_scale=3.0)
ids, _ = ho.synthetic_batch(5, B, L - 1, seed=99)
t = torch.from_numpy(ids).cuda()
Lp = (L + 63) // 64 * 64
dt = np.float16
out = {}
for mode in ("gated", "raw"):
    if mode == "raw":
        os.environ["CLM_DEBUG"] = "raw_z"
    e = Engine("cuda:0", precision=prec, chunk_reads=8)
    os.environ.pop("CLM_DEBUG", None)
    e.load_state_dict(sd)
    lg = e.forward(t).cpu().numpy()
    torch.cuda.synchronize()
    z = e.debug_fetch("z", (B, 768, Lp), dtype=np.uint16).view(dt).astype(np.float32)
    y = e.debug_fetch("y", (B, 256, Lp), dtype=np.uint16).view(dt).astype(np.float32)
    out[mode] = (lg, z, y)
    e.close()
print("logits gated", out["gated"][0].ravel(), "\nlogits raw  ", out["raw"][0].ravel())
zr = out["raw"][1]
p = f"{ho.BB}layers.3.mixer.short_filter."
sw = sd[p + "weight"].numpy().reshape(768, 3); sb = sd[p + "bias"].numpy()
zp = np.concatenate([np.zeros((B, 768, 2), np.float32), zr[:, :, :L]], axis=2)
f = sb[None, :, None] + sw[None, :, 0, None] * zp[:, :, :-2] + sw[None, :, 1, None] * zp[:, :, 1:-1] + sw[None, :, 2, None] * zp[:, :, 2:]
x0f, g = f[:, :256], f[:, 256:512] * f[:, 512:]
zg = out["gated"][1]
for name, ref, got in (("x0f", x0f, zg[:, :256, :L]), ("g", g, zg[:, 256:512, :L])):
    err = np.abs(got - ref)
    scale = np.abs(ref).max()
    print(f"{name}: max |gated - filtered raw| {err.max():.3e} (scale {scale:.2f}); mean {err.mean():.2e}")
    pt = err.max(axis=(0, 1))                      # per token
    bad = np.where(pt > 0.02 * scale)[0]
    print(f"   tokens with error > 2% of scale: {len(bad)} of {L}; first {bad[:24]}; (mod 128) {sorted(set(bad % 128))[:20]}")
    pc = err.max(axis=(0, 2)); badc = np.where(pc > 0.02 * scale)[0]
    print(f"   channels with such errors: {len(badc)}; first {badc[:16]}")
np.set_printoptions(precision=4, suppress=True, linewidth=200)
for ch in (0, 5, 40):
    print(f"channel {ch}: x0f expected", x0f[0, ch, :16], "\n            x0f got     ", zg[0, ch, :16])
    print(f"            raw x0 (+bias, fp16)", zr[0, ch, :16])
    print(f"            g expected  ", g[0, ch, :16], "\n            g got       ", zg[0, 256 + ch, :16])
    print(f"            x0f exp 120..136", x0f[0, ch, 120:136], "\n            x0f got 120..136", zg[0, ch, 120:136])
ey = np.abs(out["gated"][2][:, :, :L] - out["raw"][2][:, :, :L])
print(f"y: max |gated - raw| {ey.max():.3e} (scale {np.abs(out['raw'][2]).max():.2f}); tokens > 2%: {np.where(ey.max(axis=(0,1)) > 0.02 * np.abs(out['raw'][2]).max())[0][:24]}")
