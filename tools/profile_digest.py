"""Condense gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into the small files kept under profiles/:

    profiles/<tag>_bench.json          the default bench line of the round
    profiles/<tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary of the same command
    profiles/<tag>_pmc_summary.txt     per kernel: mean duration and mean counter values per dispatch
    profiles/<tag>_traffic.json        per kernel HBM bytes per dispatch = (2*FETCH_SIZE + WRITE_SIZE) * 1024
                                       (gfx950 correction of MI355X_MICROARCH.md "HBM": FETCH_SIZE tallies 128-B requests
                                        at 64 B), keyed by engine stage for bench.py's roofline.traffic

    python tools/profile_digest.py r01
"""
import csv
import glob
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
STAGE_OF = {"tail16_kernel": "out_proj_ln2_mlp", "tail32_kernel": "out_proj_ln2_mlp", "mlp16_kernel": "ln2_mlp", "in_proj16_kernel": "ln1_in_proj",
            "out_proj16_kernel": "out_proj", "hyena_conv_kernel": "short_long_conv", "hyena_conv_seg_kernel": "short_long_conv",
            "hyena_conv_pers_kernel": "short_long_conv",
            "embed_kernel": "embed", "gemm_kernel": "lnf_pool_score", "softmax_stats_kernel": "softmax_pool", "pool_kernel": "softmax_pool",
            "head_mlp_kernel": "head_mlp", "head_tiles_kernel": "head_mlp", "ids8_kernel": "embed",
            "score_pool16_kernel": "lnf_pool_score", "attention_fwd_kernel": "attention", "enc_ffn16_kernel": "encoder_layer",
            "conv3_relu_pool_kernel": "conv_stack_pe_ln"}


def short(name: str) -> str:
    n = name.replace("void ", "").replace("clm::", "")
    return n.split("(")[0].split("<")[0]


def counters(path_glob):
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    per = defaultdict(lambda: defaultdict(dict))
    for f in glob.glob(path_glob, recursive=True):
        for r in csv.DictReader(open(f)):
            n = short(r["Kernel_Name"])
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[n][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            per[n][r["Counter_Name"]][r["Dispatch_Id"]] = float(r["Counter_Value"])
    return acc, dur, per


def full_size_mean(values_by_dispatch, dur_by_dispatch):
    """Mean over the FULL-SIZE dispatches of a kernel in one counter pass: those that ran at least half as long as its longest one.
    (The guarded product also launches the kernels on 4-read self-check samples; averaged in, they understate the bytes of a
    256-read launch by the share of small launches -- a share that changes with the number of steps of the pass.)"""
    if not values_by_dispatch:
        return None, 0
    top = max(dur_by_dispatch[i] for i in values_by_dispatch)
    ids = [i for i in values_by_dispatch if dur_by_dispatch[i] >= 0.5 * top]
    return sum(values_by_dispatch[i] for i in ids) / len(ids), len(ids)


def main(tag):
    src = ROOT / "gpurun_out" / tag
    dst = ROOT / "profiles"
    dst.mkdir(exist_ok=True)
    bench = [l for l in open(src / "bench_default.json") if l.startswith("{")]
    if bench:
        (dst / f"{tag}_bench.json").write_text(bench[-1])
    stats = glob.glob(str(src / "kt" / "**" / "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], dst / f"{tag}_kernel_stats.csv")
    lines, traffic = [], defaultdict(lambda: {"fetch_kb": None, "write_kb": None})
    busy = {}          # kernel -> (mean SQ_VALU_MFMA_BUSY_CYCLES per dispatch, mean dispatch duration in us of the SAME counter pass)
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_inst", "pmc_icache"):
        acc, dur, per = counters(str(src / sub / "**" / "*counter_collection.csv"))
        if not acc:
            continue
        lines.append(f"==== {sub} (counter run: kernels are serialised, durations are for reference only)")
        for n in sorted(acc, key=lambda k: -sum(dur[k].values())):
            d = list(dur[n].values())
            if sum(d) < 100:
                continue
            lines.append(f"{n:28s} dispatches={len(d):4d} avg_us={sum(d)/len(d):9.1f}")
            for c, v in sorted(acc[n].items()):
                fm, nf = full_size_mean(per[n][c], dur[n])
                fd, _ = full_size_mean(dur[n], dur[n])
                lines.append(f"    {c:28s} {sum(v)/len(v):18.1f}   full-size launches ({nf:3d}, avg_us {fd:9.1f}): {fm:18.1f}")
                if c == "FETCH_SIZE":
                    traffic[n]["fetch_kb"], traffic[n]["fetch_kb_all"] = fm, sum(v) / len(v)
                if c == "WRITE_SIZE":
                    traffic[n]["write_kb"], traffic[n]["write_kb_all"] = fm, sum(v) / len(v)
                if c == "SQ_VALU_MFMA_BUSY_CYCLES" and sum(v) > 0:
                    busy[n] = (fm, fd)
    (dst / f"{tag}_pmc_summary.txt").write_text("\n".join(lines) + "\n")
    out = {}
    for n, t in traffic.items():
        if t["fetch_kb"] is None or t["write_kb"] is None:
            continue
        stage = next((s for k, s in sorted(STAGE_OF.items(), key=lambda kv: -len(kv[0])) if k in n), None)   # longest name first
        e = {"kernel": n, "fetch_size_kb": t["fetch_kb"], "write_size_kb": t["write_kb"],
             "hbm_bytes_per_dispatch": (2.0 * t["fetch_kb"] + t["write_kb"]) * 1024.0,      # full-size launches only (full_size_mean)
             "hbm_bytes_per_dispatch_all_launches": (2.0 * t.get("fetch_kb_all", t["fetch_kb"]) + t.get("write_kb_all", t["write_kb"])) * 1024.0}
        if n in busy:
            # MFMA pipe occupancy: busy cycles are summed over the chip's 1,024 SIMDs; both figures are means over the
            # full-size dispatches of ONE counter pass, so their ratio is the kernel's
            e["mfma_busy_cycles_per_dispatch"], e["pmc_pass_avg_us"] = busy[n]
        if stage and (stage not in out or e["hbm_bytes_per_dispatch"] > out[stage]["hbm_bytes_per_dispatch"]):
            out[stage] = e
    cfg = json.loads(bench[-1])["config"] if bench else {}
    sys.path.insert(0, str(ROOT))
    from bench import kernel_sources_sha        # the digest is only valid for the kernel sources it was measured on

    (dst / f"{tag}_traffic.json").write_text(json.dumps({"config": cfg, "dtype": json.loads(bench[-1])["dtype"] if bench else None,
                                                         "kernel_sources_sha": kernel_sources_sha(), "stages": out}, indent=1) + "\n")
    # the bench line of this round carries the traffic of the previous digest (bench.py reads profiles/*_traffic.json at run
    # time): put the figure of THIS run's counter passes into the committed copy
    if bench:
        b = json.loads(bench[-1])
        st = out.get(b.get("roofline", {}).get("kernel"))
        if st:
            b["roofline"]["traffic"] = st["hbm_bytes_per_dispatch"]
            b["roofline"]["traffic_unit"] = "bytes/launch"
            b["roofline"]["traffic_source"] = f"profiles/{tag}_traffic.json"
            b["roofline"].pop("traffic_stale", None)      # the line was printed before this round's digest existed
            if st.get("mfma_busy_cycles_per_dispatch"):   # as bench.py computes it from the digest at run time
                from bench import SCLK_UNDER_TAIL_HZ
                cyc, us = st["mfma_busy_cycles_per_dispatch"], st["pmc_pass_avg_us"]
                b["roofline"]["mfma_busy_frac"] = cyc / (1024 * us * 1e-6 * SCLK_UNDER_TAIL_HZ)
                b["roofline"]["mfma_busy_note"] = (f"SQ_VALU_MFMA_BUSY_CYCLES {cyc:.3g} per dispatch / (1,024 SIMDs x {us:.0f} us x "
                                                   f"{SCLK_UNDER_TAIL_HZ / 1e9:.2f} GHz), profiles/{tag}_traffic.json")
            (dst / f"{tag}_bench.json").write_text(json.dumps(b) + "\n")
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
