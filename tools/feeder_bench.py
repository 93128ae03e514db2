"""Throughput of the native BAM feeder (include/chimeralm_feed.h) on a synthetic BAM, beside the Python data path it replaces.

    python tools/feeder_bench.py [--reads 4000] [--bases 8192] [--batch 256] [--pinned]

The BAM holds `reads` primary records with an SA tag, random A/C/G/T bases and Phred-like qualities, BGZF-compressed at zlib
level 6 in 64-KiB members (what samtools writes).  Reported: selected reads/s and MB/s of inflated BAM stream through
BGZF inflate -> record decode -> selection -> tokenisation -> left-padded batches in the (pinned) ring, one decoder thread;
then the same file through `chimeralm_amd.bam.parse_bam_file` + tokenizer + collator (the reference's Python path, mirrored)."""
from __future__ import annotations

import argparse
import struct
import sys
import tempfile
import time
import zlib
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def write_bam(path: Path, n_reads: int, n_bases: int, seed: int = 0, min_bases: int | None = None) -> int:
    """`n_reads` records of `n_bases` bases, or of lengths uniform in [min_bases, n_bases] (real long-read files are ragged)."""
    rng = np.random.default_rng(seed)
    max_bases = n_bases
    sa = b"SAZchr1,100,+,50M,60,0;\0"
    header = b"BAM\1" + struct.pack("<i", 4) + b"@HD\n" + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", 1 << 28)
    codes = np.array([1, 2, 4, 8], dtype=np.uint8)                      # A C G T in the 4-bit alphabet
    total, pending = 0, bytearray(header)
    with path.open("wb") as f:
        def flush(final=False):
            nonlocal pending, total
            while len(pending) >= 0xff00 or (final and pending):
                chunk = bytes(pending[:0xff00])
                del pending[:0xff00]
                comp = zlib.compressobj(6, 8, -15)
                data = comp.compress(chunk) + comp.flush()
                f.write(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(data) + 25))
                f.write(data + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
                total += len(chunk)
        for i in range(n_reads):
            n_bases = max_bases if min_bases is None else int(rng.integers(min_bases, max_bases + 1))
            name = f"read_{i:08d}".encode()
            b = codes[rng.integers(0, 4, n_bases)]
            bp = b if n_bases % 2 == 0 else np.append(b, np.uint8(0))
            packed = ((bp[0::2] << 4) | bp[1::2]).astype(np.uint8).tobytes()
            qual = np.clip(rng.normal(25, 6, n_bases), 2, 40).astype(np.uint8).tobytes()
            body = struct.pack("<iiBBHHHiiii", 0, 1000 + i, len(name) + 1, 60, 4681, 1, 0, n_bases, -1, -1, 0) + name + b"\0"
            body += struct.pack("<I", n_bases << 4) + packed + qual + sa
            pending += struct.pack("<i", len(body)) + body
            flush()
        flush(final=True)
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4000)
    ap.add_argument("--bases", type=int, default=8192)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--pinned", action="store_true", help="page-locked ring slots (needs a HIP device)")
    ap.add_argument("--threads", type=int, nargs="*", default=[1, 0], help="inflate_threads values to time (0 = automatic)")
    ap.add_argument("--python-reads", type=int, default=300, help="reads pushed through the Python path for comparison")
    a = ap.parse_args()
    from chimeralm_amd.feeder import BamFeeder

    with tempfile.TemporaryDirectory() as td:
        path = Path(td) / "synthetic.bam"
        raw = write_bam(path, a.reads, a.bases)
        comp = path.stat().st_size
        print(f"synthetic BAM: {a.reads} reads x {a.bases} bases, {raw / 1e6:.1f} MB inflated, {comp / 1e6:.1f} MB on disk")
        for threads in a.threads:
            for rep in range(2):
                t0 = time.perf_counter()
                n = 0
                with BamFeeder(path, batch_size=a.batch, max_tokens=32769, pinned=a.pinned, inflate_threads=threads) as f:
                    while (b := f.next()) is not None:
                        n += b.n_reads
                        f.release(b)
                dt = time.perf_counter() - t0
            print(f"native feeder ({'pinned' if a.pinned else 'pageable'} ring, inflate_threads = {threads or 'auto'}): {n / dt:,.0f} reads/s, "
                  f"{raw / dt / 1e6:,.0f} MB/s inflated, {comp / dt / 1e6:,.0f} MB/s compressed")
        from chimeralm_amd import bam as pybam
        from chimeralm_amd.tokenizer import DataCollator, load_tokenizer_from_hyena_model, tokenize_and_align_labels_and_quals_ids

        tok = load_tokenizer_from_hyena_model("hyenadna-small-32k-seqlen")
        coll = DataCollator(tok)
        t0 = time.perf_counter()
        feats = []
        for i, r in enumerate(pybam.parse_bam_file(path)):
            if i >= a.python_reads:
                break
            feats.append(tokenize_and_align_labels_and_quals_ids(r, tok, tok.max_len_single_sentence))
            if len(feats) == 12:
                coll(feats)
                feats = []
        dt = time.perf_counter() - t0
        print(f"Python data path (parse + tokenise + collate, batches of 12): {min(a.python_reads, a.reads) / dt:,.0f} reads/s")


if __name__ == "__main__":
    main()
