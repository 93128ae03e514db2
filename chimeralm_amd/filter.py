"""`chimeralm filter`: the step after `predict`, mirroring /root/reference/chimeralm/__main__.py:26-69,99-153.

`load_predicts` / `load_predictions_from_folder` read the per-batch `name<TAB>label` files the prediction writer produced;
`filter_bam_by_predcition` (the reference's spelling is kept) drops every record of the reads labelled 1 (chimera artifact),
writes `<bam>.filtered.bam`, then -- with `index=True` -- `<bam>.filtered.sorted.bam` + its `.bai`.  The BAM work (BGZF
inflate/deflate, SAM text parsing, record copy, coordinate sort with spill-to-disk runs, BAI binning index) is native C++
(csrc/bam_filter.cpp) behind `clm_bam_filter_ex` / `clm_bam_sort_index`; the reference does it through pysam / samtools.
As there (`file_mode = "rb" if suffix == ".bam" else "r"`, :127) any other suffix is read as SAM text.  Records of a BAM
that have no reference placement are left out, as the reference's index walk (`bam_file.fetch()`, :131) never yields them.
"""
from __future__ import annotations

import ctypes as C
import logging
from collections import Counter
from pathlib import Path

from . import _native as N

log = logging.getLogger("chimeralm_amd")


def load_predicts(path: Path | str) -> dict[str, int]:
    predicts: dict[str, int] = {}
    try:
        path = Path(path)
        if not path.exists():
            raise FileNotFoundError(f"File not found: {path}")
        with path.open(encoding="utf-8") as f:
            for line_num, line in enumerate(f, 1):
                line = line.strip()
                if not line:
                    continue
                parts = line.split("\t")
                if len(parts) != 2:
                    raise ValueError(f"Invalid line format at line {line_num}: {line}")
                predicts[parts[0]] = int(parts[1])
    except Exception as e:
        raise ValueError(f"Error reading file {path}: {e}") from e
    return predicts


def load_predictions_from_folder(path: Path | str) -> dict[str, int]:
    predictions: dict[str, int] = {}
    for file in Path(path).glob("*.txt"):
        predictions.update(load_predicts(file))
    return predictions


def _check(rc: int):
    if rc != 0:
        raise RuntimeError(N.load().clm_bam_last_error().decode())


def filter_bam_by_predcition(bam_path: Path, prediction_path: Path, *, index: bool = True,
                             output_prediction: bool = False) -> dict | None:
    """Returns {"kept", "dropped", "filtered", "sorted"} (record counts and output paths), None when there are no predictions."""
    bam_path, prediction_path = Path(bam_path), Path(prediction_path)
    predictions = load_predictions_from_folder(prediction_path)
    if not predictions:
        log.warning("No predictions found")
        return None
    if output_prediction:
        log.info(f"Writing all predictions to {prediction_path / 'predictions.txt'}")
        with (prediction_path / "predictions.txt").open("w") as f:
            for name, label in predictions.items():
                f.write(f"{name}\t{label}\n")
    log.info(f"Loaded {len(predictions)} predictions from {prediction_path}")
    counter = Counter(predictions.values())
    log.info(f"Biological: {counter.get(0, 0)} ({counter.get(0, 0) / len(predictions) * 100:.1f}%), "
             f"Chimera artifact: {counter.get(1, 0)} ({counter.get(1, 0) / len(predictions) * 100:.1f}%)")
    flags = 0 if bam_path.suffix == ".bam" else N.BAM_INPUT_SAM          # __main__.py:127
    lib = N.load()
    drop = [n.encode() for n, label in predictions.items() if label == 1]
    arr = (C.c_char_p * max(1, len(drop)))(*drop)
    kept, dropped, unplaced = C.c_int64(), C.c_int64(), C.c_int64()
    output_path = bam_path.with_suffix(".filtered.bam")
    _check(lib.clm_bam_filter_ex(str(bam_path).encode(), str(output_path).encode(), arr, len(drop), flags, C.byref(kept),
                               C.byref(dropped), C.byref(unplaced)))
    if unplaced.value:
        log.info(f"{unplaced.value} records without a reference placement left out (an index walk does not reach them)")
    if kept.value == 0:
        # the reference's `bam_file.fetch()` (:131) RAISES on a BAM without an index; an unaligned BAM gives no record here
        # either -- say so loudly instead of handing back an empty, sorted, indexed file without a word
        why = (f"all {unplaced.value} remaining records lack a reference placement (an unaligned / unindexed BAM: the reference's "
               "index walk would have raised here)" if unplaced.value else "every record belongs to a read labelled 1")
        log.warning(f"{output_path} holds NO records: {why}")
    result = {"kept": kept.value, "dropped": dropped.value, "unplaced": unplaced.value, "filtered": output_path, "sorted": None}
    if index:
        log.info(f"Sorting {output_path}")
        sorted_output_path = output_path.with_suffix(".sorted.bam")
        n = C.c_int64()
        log.info(f"Indexing {sorted_output_path}")
        _check(lib.clm_bam_sort_index(str(output_path).encode(), str(sorted_output_path).encode(), None, C.byref(n)))
        result["sorted"] = sorted_output_path
    return result
