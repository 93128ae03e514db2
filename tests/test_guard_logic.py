"""CPU: the decisions of `HyenaDna`'s self-check (chimeralm_amd/hyena.py) with the engine replaced by a stub
whose `selfcheck` answers are scripted -- fallback, the measured short-read switch, when a later batch is checked again.  (The
measurements themselves -- `clm_selfcheck` against the oracle -- are GPU tests: tests/test_gpu_parity.py.)"""
from __future__ import annotations

import warnings

import pytest
import torch

from chimeralm_amd import hyena, lm


class StubEngine:
    """What `HyenaDna._selfcheck` touches of `chimeralm_amd.engine.Engine`."""

    def __init__(self, err_by_len, batch_err=1e-4, precision_code=3, level2=None):
        self.err_by_len, self.batch_err = err_by_len, batch_err
        self.err_level2, self.mlp_lo = level2, False
        self.device = torch.device("cpu")
        self.cfg = type("Cfg", (), {"precision": precision_code})()
        self.min_len, self.fallback, self.calls = 2048, 0, []

    def selfcheck(self, ids):
        n, L = ids.shape
        self.calls.append((n, L))
        return (self.err_by_len.get(L, self.batch_err), 0)

    def set_f16c_min_len(self, n):
        self.min_len = n

    def set_fallback(self, level):
        self.fallback = int(level)                           # clm_set_fallback: 0 the mode, 1 the next arithmetic inside the gate, 2 exact fp32

    def set_mlp_compensation(self, on):
        self.mlp_lo = bool(on)
        if on and self.err_level2 is not None:               # the second level answers with its own scripted errors
            self.err_by_len, self.batch_err = self.err_level2

    def effective_precision(self, L):
        mode = {0: "fp32", 1: "bf16", 2: "fp16", 3: "fp16c", 4: "fp16x3"}[self.cfg.precision]
        if self.fallback == 2 or mode == "fp32" or (self.fallback and mode == "fp16x3"):
            return "fp32"
        if self.fallback or (mode == "fp16c" and L < self.min_len):
            return "fp16x3"                                  # a 16-bit handle's short reads and first fall-back level (ABI 5)
        return mode


def _net():
    return lm.ChimeraLM.new(precision="fp16c").net


def _ids(B, L):
    return torch.full((B, L), 7, dtype=torch.uint8)


def test_all_samples_pass_lowers_the_switch_to_the_shortest_sample():
    net = _net()
    eng = StubEngine({4097: 2e-4, 2048: 4.5e-4, 1024: 2.4e-4, 512: 2.2e-4, 256: 2.5e-4})
    net.guard(eng, _ids(6, 3000))
    rep = net.selfcheck_report
    assert rep["fallback"] is False and eng.fallback == 0 and rep["f16c_min_len"] == 256 == eng.min_len
    assert [c[1] for c in eng.calls] == [4097, 2048, 1024, 512, 256, 3000]       # descending samples, then four rows of the batch
    assert eng.calls[-1][0] == 4 and abs(rep["max_abs_dlogit"] - 4.5e-4) < 1e-12
    n = len(eng.calls)
    net.guard(eng, _ids(6, 3000))                                                # a length inside the checked range: nothing to do ...
    net.guard(eng, _ids(6, 2200))
    net.guard(eng, _ids(6, 4400))
    assert len(eng.calls) == n
    net.guard(eng, _ids(2, 1400))                                                # ... a batch more than 1.5x SHORTER than any checked
    assert eng.calls[-1] == (2, 1400) and len(eng.calls) == n + 1
    net.guard(eng, _ids(3, 4600))                                                # ... or more than 1.5x LONGER (the error is
    assert eng.calls[-1] == (3, 4600) and len(eng.calls) == n + 2                #     not monotone in the length)
    net.guard(eng, _ids(3, 2000))                                                # [1400, 4600] is covered now
    assert len(eng.calls) == n + 2


def test_periodic_recheck_and_rows_spread_over_the_batch():
    net = lm.ChimeraLM.new(precision="fp16c", selfcheck_every=3).net
    net.selfcheck_min_reads = 0                                                    # (the periodic rule by batches alone: next test for the reads)
    eng = StubEngine({4097: 2e-4, 2048: 3e-4, 1024: 7e-4})
    seen = []
    eng_selfcheck = eng.selfcheck
    eng.selfcheck = lambda ids: (seen.append(ids.clone()), eng_selfcheck(ids))[1]
    ids = torch.arange(40, dtype=torch.uint8)[:, None].repeat(1, 3000)            # row r holds the value r
    net.guard(eng, ids)
    assert seen[-1][:, 0].tolist() == [0, 13, 26, 39]                              # four rows spread over the batch, not rows 0..3
    n = len(eng.calls)
    net.guard(eng, ids), net.guard(eng, ids)
    assert len(eng.calls) == n and net.selfcheck_report["checks"] == 1
    net.guard(eng, ids)                                                            # the third batch since the last check
    assert len(eng.calls) == n + 1 and net.selfcheck_report["checks"] == 2
    eng.batch_err = 9e-4                                                           # the mode drifts above the threshold on LATER data
    net.guard(eng, ids), net.guard(eng, ids)
    assert eng.fallback == 0
    with pytest.warns(RuntimeWarning, match="falling back to fp16x3"):
        net.guard(eng, ids)
    assert eng.fallback == 1 and net.selfcheck_report["fallback"] is True and net.selfcheck_report["fallback_precision"] == "fp16x3"


def test_periodic_recheck_waits_for_reads_as_well_as_batches():
    """Round 5: at the reference's default batch of 12, sixteen batches are 192 reads and a check (two passes over 4 reads, one in exact
    fp32) would cost 10 % of them -- a periodic check is due after `selfcheck_every` batches AND `selfcheck_min_reads` reads."""
    net = lm.ChimeraLM.new(precision="fp16c", selfcheck_every=2).net
    net.selfcheck_min_reads = 30
    eng = StubEngine({4097: 2e-4, 2048: 3e-4, 1024: 7e-4})
    ids = _ids(12, 3000)
    net.guard(eng, ids)
    assert net.selfcheck_report["checks"] == 1
    net.guard(eng, ids), net.guard(eng, ids)                                       # two batches, 24 reads: not yet
    assert net.selfcheck_report["checks"] == 1
    net.guard(eng, ids)                                                            # 36 reads
    assert net.selfcheck_report["checks"] == 2
    net.guard(eng, lambda: ids, n_tokens=3000), net.guard(eng, lambda: ids, n_tokens=3000)   # batch size unknown: batches alone decide
    assert net.selfcheck_report["checks"] == 3


def test_lowering_the_switch_below_its_default_needs_a_margin_of_two():
    """ONE seeded sample places the switch; round 4 lowered it to 512 tokens on a sample at 4.7e-4 and the first
    real batch there measured 94 % of the threshold."""
    net = _net()
    eng = StubEngine({4097: 2e-4, 2048: 4.9e-4, 1024: 2.6e-4, 512: 1e-4})       # 2,048 passes at tol, 1,024 would need tol / 2
    net.guard(eng, _ids(4, 5000))
    assert net.selfcheck_report["f16c_min_len"] == 2048 == eng.min_len and [c[1] for c in eng.calls] == [4097, 2048, 1024, 5000]
    net2, eng2 = _net(), StubEngine({4097: 2e-4, 2048: 4.9e-4, 1024: 2.5e-4, 512: 2.6e-4})
    net2.guard(eng2, _ids(4, 5000))
    assert net2.selfcheck_report["f16c_min_len"] == 1024 == eng2.min_len


def test_a_measurement_within_ten_percent_of_the_threshold_rearms_the_check_for_the_next_batch():
    net = lm.ChimeraLM.new(precision="fp16c", selfcheck_every=16).net
    eng = StubEngine({4097: 2e-4, 2048: 3e-4, 1024: 7e-4}, batch_err=4.8e-4)
    net.guard(eng, _ids(8, 8193))
    n = len(eng.calls)
    net.guard(eng, _ids(8, 8193))                                                  # kept at 96 % of tol: the very next batch is measured
    assert len(eng.calls) == n + 1 and net.selfcheck_report["fallback"] is False
    eng.batch_err = 3e-4
    net.guard(eng, _ids(8, 8193))                                                  # ... and again (4.8e-4 re-armed it), now comfortably inside
    assert len(eng.calls) == n + 2
    for _ in range(14):
        net.guard(eng, _ids(8, 8193))
    assert len(eng.calls) == n + 2                                                 # back to every 16th batch


def test_batches_below_the_switch_are_not_on_trial_and_do_not_widen_the_checked_range():
    """a batch that ran in the short-read kernels was recorded as 'checked'; the first batch that really ran fp16c, within
    1.5x of it, then went unmeasured."""
    net = _net()
    eng = StubEngine({4097: 2e-4, 2048: 6e-4})                                     # switch at 4,097
    net.guard(eng, _ids(4, 3000))                                                  # runs in the fp16x3 kernels: samples only
    assert [c[1] for c in eng.calls] == [4097, 2048] and net._checked_min_len is None
    net.guard(eng, _ids(4, 3500))                                                  # below the switch: nothing on trial, nothing counted
    assert len(eng.calls) == 2
    net.guard(eng, _ids(4, 4400))                                                  # the first batch that runs the mode IS measured
    assert eng.calls[-1] == (4, 4400) and net._checked_min_len == 4400


def test_an_fp16x3_module_asked_to_check_itself_falls_back_to_exact_fp32():
    """the fall-back of an fp16x3 handle did nothing; level 1 on such a handle is exact fp32 now."""
    net = lm.ChimeraLM.new(precision="fp16x3", selfcheck=True).net
    eng = StubEngine({4097: 1e-5}, batch_err=9e-4, precision_code=4)
    with pytest.warns(RuntimeWarning, match="falling back to exact fp32"):
        net.guard(eng, _ids(4, 3000))
    assert eng.fallback == 1 and eng.effective_precision(3000) == "fp32" and net.selfcheck_report["fallback_precision"] == "fp32"


def test_guard_takes_a_callable_and_only_calls_it_when_a_check_is_due():
    net = _net()
    eng = StubEngine({4097: 2e-4, 2048: 3e-4, 1024: 7e-4})
    made = []
    make = lambda: (made.append(1), _ids(5, 3000))[1]
    net.guard(eng, make, n_tokens=3000)
    assert len(made) == 1
    for _ in range(10):
        net.guard(eng, make, n_tokens=3000)
    assert len(made) == 1                                                          # run_predict_native: no copy on the other batches


def test_an_engine_error_inside_the_sample_loop_restores_the_length_switch():
    net = _net()
    eng = StubEngine({4097: 2e-4})
    def boom(ids):
        if ids.shape[1] == 2048:
            raise RuntimeError("out of memory")
        return (2e-4, 0)
    eng.selfcheck = boom
    with pytest.raises(RuntimeError):
        net.guard(eng, _ids(4, 5000))
    assert eng.min_len == 4097                                                     # never left at 1 (every length in 16 bits)


def test_first_failing_sample_sets_the_switch_and_keeps_the_mode_for_longer_reads():
    net = _net()
    eng = StubEngine({4097: 2e-4, 2048: 3e-4, 1024: 7e-4})
    net.guard(eng, _ids(4, 5000))
    rep = net.selfcheck_report
    assert rep["fallback"] is False and rep["f16c_min_len"] == 2048 == eng.min_len
    assert [c[1] for c in eng.calls] == [4097, 2048, 1024, 5000]                  # stops at the first failure
    assert rep["max_abs_dlogit"] == 3e-4                                          # the failing SHORT sample does not count against the mode
    eng2, net2 = StubEngine({4097: 2e-4, 2048: 6e-4}), _net()                     # 2,048 fails: reads below 4,097 take fp32, the mode stays
    net2.guard(eng2, _ids(4, 3000))
    assert net2.selfcheck_report["f16c_min_len"] == 4097 and net2.selfcheck_report["fallback"] is False
    assert [c[1] for c in eng2.calls] == [4097, 2048]                             # the 3,000-token batch itself runs in fp32: nothing to measure


def test_a_failing_longest_sample_moves_the_switch_above_it_and_the_batch_decides():
    net = _net()
    eng = StubEngine({4097: 8e-4})                                                # the longest sample fails, the 8,193-token batch does not
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        net.guard(eng, _ids(4, 8193))
    rep = net.selfcheck_report
    assert rep["fallback"] is False and eng.fallback == 0 and eng.mlp_lo is False
    assert rep["f16c_min_len"] == 4098 == eng.min_len and rep["max_abs_dlogit"] == 1e-4   # reads up to 4,097 tokens: fp32 kernels
    assert [c[1] for c in eng.calls] == [4097, 8193]


def test_batch_above_the_threshold_at_both_levels_falls_back_for_good():
    net = _net()
    eng = StubEngine({4097: 8e-4}, batch_err=8e-4)                                # (both levels answer 8e-4)
    with pytest.warns(RuntimeWarning, match="falling back to fp16x3"):
        net.guard(eng, _ids(4, 8193))
    assert net.selfcheck_report["fallback"] is True and eng.fallback == 1 and eng.min_len == 4098 and eng.mlp_lo is True
    n = len(eng.calls)
    net.guard(eng, _ids(4, 100))
    assert len(eng.calls) == n                                                    # nothing more to check once it has fallen back
    net2, eng2 = _net(), StubEngine({4097: 1e-4, 2048: 1e-4, 1024: 1e-4, 512: 1e-4, 256: 1e-4}, batch_err=9e-4)
    with pytest.warns(RuntimeWarning):
        net2.guard(eng2, _ids(4, 8193))                                           # the samples pass, the real batch does not
    assert net2.selfcheck_report["fallback"] is True and eng2.fallback == 1


def test_second_level_is_heard_before_the_fallback():
    """Round 4: fp16c whose plain-MLP form fails switches fc1 / fc2 to hi + lo weights and is measured again from the start."""
    net = _net()
    eng = StubEngine({4097: 8e-4}, batch_err=8e-4, level2=({4097: 2e-4, 2048: 3e-4, 1024: 9e-4}, 1.5e-4))
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        net.guard(eng, _ids(4, 8193))
    rep = net.selfcheck_report
    assert eng.mlp_lo is True and rep["mlp_compensation"] is True and rep["fallback"] is False and eng.fallback == 0
    assert [c[1] for c in eng.calls] == [4097, 8193, 4097, 2048, 1024, 8193]      # level 1: sample fails (+ the batch); level 2: from the start
    assert rep["f16c_min_len"] == 2048 == eng.min_len and rep["max_abs_dlogit"] == 3e-4
    net2 = _net()
    eng2 = StubEngine({4097: 8e-4}, batch_err=8e-4, level2=({4097: 7e-4}, 7e-4))  # both levels fail on the batch: fp32
    with pytest.warns(RuntimeWarning, match="falling back to fp16x3"):
        net2.guard(eng2, _ids(4, 8193))
    assert eng2.mlp_lo is True and eng2.fallback == 1 and net2.selfcheck_report["fallback"] is True
    net3 = _net()                                                                  # a LATER batch drifts: escalate then, not fall back
    eng3 = StubEngine({4097: 2e-4, 2048: 3e-4, 1024: 7e-4}, level2=({4097: 1e-4, 2048: 1e-4, 1024: 1e-4, 512: 1e-4, 256: 1e-4}, 1e-4))
    net3.selfcheck_every, net3.selfcheck_min_reads = 2, 0
    net3.guard(eng3, _ids(4, 5000))
    assert eng3.mlp_lo is False
    eng3.batch_err = 9e-4
    net3.guard(eng3, _ids(4, 5000)), net3.guard(eng3, _ids(4, 5000))
    assert eng3.mlp_lo is True and eng3.fallback == 0 and net3.selfcheck_report["f16c_min_len"] == 256


def test_guard_is_off_for_fp32_and_optional_for_the_reduced_modes():
    m32 = lm.ChimeraLM.new(precision="fp32").net
    assert m32.selfcheck is False
    eng = StubEngine({}, precision_code=0)
    m32.guard(eng, _ids(2, 5000))
    assert eng.calls == []
    m16 = lm.ChimeraLM.new(precision="fp16").net                                  # reduced precision on request: not guarded unless asked
    assert m16.selfcheck is False
    m16b = lm.ChimeraLM.new(precision="fp16", selfcheck=True).net
    eng16 = StubEngine({4097: 2e-4}, precision_code=2)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        m16b.guard(eng16, _ids(3, 3000))
    assert [c[1] for c in eng16.calls] == [4097, 3000] and eng16.min_len == 2048  # one verdict, no length switch outside fp16c
    off = lm.ChimeraLM.new(precision="fp16c", selfcheck=False).net
    eng_off = StubEngine({})
    off.guard(eng_off, _ids(2, 5000))
    assert eng_off.calls == [] and off.selfcheck_report == {}
