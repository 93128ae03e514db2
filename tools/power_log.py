"""Socket power, shader clock and temperature of the GPU while a workload runs, sampled from a side thread (hwmon / sysfs when
readable, else `amd-smi metric` / `rocm-smi`), to test the "power-bound" reading of the tail kernel with telemetry instead of
clock arithmetic.   GPU box:

    python tools/power_log.py tail   [--precision fp16c] [--seconds 6]     # the engine's forward in a loop
    python tools/power_log.py conv / idle / probe                          # conv only (debug stop) / nothing / build/mfma_probe

Prints per-phase mean / max of every quantity it can read plus the raw samples (CSV) for profiles/.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def _read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def leased_pci_slot() -> str | None:
    """PCI slot ("0000:c5:00.0") of the GPU this process computes on (cuda:0 of the lease), for picking ITS hwmon directory:
    the box's sysfs shows every GPU of a shared 8-GPU host, and "the busiest card" can be another tenant's."""
    try:
        import torch

        p = torch.cuda.get_device_properties(0)
        return f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:  # noqa: BLE001
        return None


class Sampler:
    def __init__(self, period=0.05, pci_slot: str | None = None):
        self.period, self.rows, self._stop = period, [], threading.Event()
        self.hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        self.mine = None                         # hwmon directory of the leased card, by PCI slot
        if pci_slot:
            for h in self.hw:
                ue = _read(os.path.join(os.path.dirname(os.path.dirname(h)), "uevent")) or ""
                if f"PCI_SLOT_NAME={pci_slot.lower()}" in ue.lower() or pci_slot.lower() in os.path.realpath(h).lower():
                    self.mine = h
        self.src = "sysfs" if any(_read(f"{h}/power1_average") or _read(f"{h}/power1_input") for h in self.hw) else None
        if self.src is None:
            for tool in ("amd-smi", "rocm-smi"):
                if subprocess.run(["which", tool], capture_output=True).returncode == 0:
                    self.src = tool
                    break
        self.t = threading.Thread(target=self._run, daemon=True)

    def sample(self) -> dict:
        row = {"t": time.perf_counter()}
        if self.src == "sysfs":
            # the box's sysfs shows every GPU of the host, the process sees one of them: sample all; clock / temperature / the
            # headline power are those of the LEASED card (matched by PCI slot), the busiest one only if that match failed
            best = None
            for h in self.hw:
                p = _read(f"{h}/power1_average") or _read(f"{h}/power1_input")
                if not p:
                    continue
                card = h.split("/")[4]
                row[f"power_w_{card}"] = float(p) / 1e6
                if (self.mine == h) or (self.mine is None and (best is None or float(p) > best[0])):
                    best = (float(p), h)
            if best:
                h = best[1]
                row["card"] = h.split("/")[4]
                row["power_w"] = best[0] / 1e6
                f = _read(f"{h}/freq1_input")
                row["sclk_mhz"] = float(f) / 1e6 if f else None
                t = _read(f"{h}/temp2_input") or _read(f"{h}/temp1_input")
                row["temp_c"] = float(t) / 1e3 if t else None
                cap = _read(f"{h}/power1_cap")
                row["cap_w"] = float(cap) / 1e6 if cap else None
        elif self.src == "amd-smi":
            r = subprocess.run(["amd-smi", "metric", "-g", "0", "--power", "--clock", "--temperature", "--json"],
                               capture_output=True, text=True)
            try:
                d = json.loads(r.stdout)
                d = d[0] if isinstance(d, list) else d
                d = d.get("gpu_data", [d])[0] if isinstance(d, dict) and "gpu_data" in d else d
                pw = d.get("power", {})
                row["power_w"] = float((pw.get("socket_power") or pw.get("current_socket_power") or {}).get("value"))
                clk = d.get("clock", {})
                g = next((v for k, v in clk.items() if k.startswith("gfx")), {})
                row["sclk_mhz"] = float((g.get("clk") or {}).get("value")) if g else None
                tmp = d.get("temperature", {})
                row["temp_c"] = float((tmp.get("hotspot") or tmp.get("edge") or {}).get("value"))
            except Exception as e:  # noqa: BLE001
                row["err"] = f"{type(e).__name__}: {r.stdout[:120]!r} {r.stderr[:120]!r}"
        elif self.src == "rocm-smi":
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True)
            try:
                d = json.loads(r.stdout)
                c = next(iter(d.values()))
                for k, v in c.items():
                    kl = k.lower()
                    if "power" in kl and "w" in kl and "power_w" not in row:
                        row["power_w"] = float(v)
                    if "sclk" in kl and "sclk_mhz" not in row:
                        m = re.search(r"(\d+)\s*mhz", str(v).lower())
                        row["sclk_mhz"] = float(m.group(1)) if m else None
                    if "temperature" in kl and "junction" in kl:
                        row["temp_c"] = float(v)
            except Exception as e:  # noqa: BLE001
                row["err"] = f"{type(e).__name__}: {r.stdout[:120]!r}"
        return row

    def _run(self):
        while not self._stop.is_set():
            self.rows.append(self.sample())
            time.sleep(self.period)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self.t.join()


def summarize(name, rows):
    out = [f"== {name}: {len(rows)} samples"]
    for k in ("power_w", "sclk_mhz", "temp_c", "cap_w"):
        v = [r[k] for r in rows if r.get(k) is not None]
        if v:
            out.append(f"   {k:9s} mean {sum(v) / len(v):8.1f}   min {min(v):8.1f}   max {max(v):8.1f}")
    cards = sorted({k for r in rows for k in r if k.startswith("power_w_")})
    if cards:
        out.append("   mean W per card: " + "  ".join(f"{c[8:]} {sum(r.get(c, 0) for r in rows) / len(rows):.0f}" for c in cards)
                   + f"   (card sampled for power / clock / temperature: {rows[len(rows) // 2].get('card')})")
    errs = [r["err"] for r in rows if "err" in r]
    if errs:
        out.append(f"   {len(errs)} failed samples, first: {errs[0]}")
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["tail", "conv", "idle", "probe", "all"])
    ap.add_argument("--precision", default="fp16c")
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--csv", default=None)
    a = ap.parse_args()
    import numpy as np
    import torch

    slot = leased_pci_slot()
    s0 = Sampler(pci_slot=slot)
    print(f"telemetry source: {s0.src}; leased GPU at PCI {slot} -> hwmon {s0.mine or 'NOT MATCHED (busiest card reported)'}; "
          f"hwmon dirs: {s0.hw}", flush=True)
    if s0.src is None:
        print("no readable power telemetry on this box (no hwmon power file, no amd-smi / rocm-smi)")
    allrows = []

    def run(name, fn):
        with Sampler(pci_slot=slot) as s:
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < a.seconds:
                fn()
                n += 1
        print(summarize(f"{name} ({n} iterations in {a.seconds:.0f} s)", s.rows), flush=True)
        allrows.extend({**r, "phase": name} for r in s.rows)

    phases = ["idle", "tail", "conv", "probe"] if a.what == "all" else [a.what]
    eng = None
    if any(p in ("tail", "conv") for p in phases):
        from bench import synthetic_ids
        from chimeralm_amd import lm
        from chimeralm_amd.engine import Engine

        torch.manual_seed(0)
        eng = Engine("cuda:0", precision=a.precision, chunk_reads=64)
        eng.load_state_dict(lm.ChimeraLM.new(precision=a.precision).state_dict())
        ids = torch.from_numpy(synthetic_ids(0, 64, 8192)).cuda()
        logits = torch.empty((64, 2), dtype=torch.float32, device="cuda")
        eng.forward(ids, out=logits)
        torch.cuda.synchronize()
    for p in phases:
        if p == "idle":
            run("idle", lambda: time.sleep(0.2))
        elif p == "tail":
            def f():
                for _ in range(4):
                    eng.forward(ids, out=logits)
                torch.cuda.synchronize()
            run(f"whole forward ({a.precision}: tail kernel ~2/3 of the time, convolution ~1/3)", f)
        elif p == "conv":
            from chimeralm_amd import _native as N

            eng.debug_stop_after(0, N.STAGES.index("short_long_conv"))

            def f():
                for _ in range(40):
                    eng.forward(ids, out=logits)
                torch.cuda.synchronize()
            run("block-0 convolution only (debug stop after the first long convolution)", f)
            eng.debug_stop_after(-1, -1)
        elif p == "probe":
            exe = Path(__file__).resolve().parent.parent / "build" / "mfma_probe"
            if exe.exists():
                run("mfma_probe (back-to-back MFMA on every SIMD)", lambda: subprocess.run([str(exe)], capture_output=True))
            else:
                print("build/mfma_probe missing: hipcc --offload-arch=gfx950 -O3 -o build/mfma_probe tools/micro/mfma_probe.cpp")
    if a.csv and allrows:
        keys = ["phase", "t", "power_w", "sclk_mhz", "temp_c", "cap_w"]
        with open(a.csv, "w") as f:
            f.write(",".join(keys) + "\n")
            for r in allrows:
                f.write(",".join("" if r.get(k) is None else str(r.get(k)) for k in keys) + "\n")


if __name__ == "__main__":
    main()
