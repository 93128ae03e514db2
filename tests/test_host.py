"""CPU: the C-ABI library loads and exports every symbol of include/chimeralm_hip.h (no compute without a GPU), and the
host-side mirror of the reference interface (tokenizer, collator, BAM module, writer, CLI, model containers)."""
from __future__ import annotations

import ctypes
import json
import re
from pathlib import Path

import numpy as np
import pytest
import torch

REPO = Path(__file__).resolve().parent.parent


def test_abi_exports_every_header_symbol(built_lib):
    header = (REPO / "include" / "chimeralm_hip.h").read_text() + (REPO / "include" / "chimeralm_feed.h").read_text()
    declared = set(re.findall(r"\b(clm_[a-z_]+)\s*\(", header))
    lib = ctypes.CDLL(str(built_lib))
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    from chimeralm_amd import _native

    assert declared == set(_native.SYMBOLS), "python binding and header disagree"
    lib2 = _native.load()
    assert lib2.clm_abi_version() == _native.ABI_VERSION
    cfg = _native.ClmConfig()
    assert lib2.clm_default_config(ctypes.byref(cfg)) == 0
    assert (cfg.d_model, cfg.n_layer, cfg.d_inner, cfg.head_hidden, cfg.n_classes) == (256, 4, 1024, 512, 2)
    assert cfg.struct_size == ctypes.sizeof(_native.ClmConfig)
    assert lib2.clm_profile_stage_name(2) == b"short_long_conv"


def test_logit_deviation_keeps_a_non_finite_difference(built_lib):
    """the self-checks' reduction (`clm_logit_deviation`, shared by clm_selfcheck and clm_tf_selfcheck; pure host
    arithmetic) -- a NaN / inf in ANY read is +inf for good, whatever finite differences follow it."""
    from chimeralm_amd import _native

    lib = _native.load()

    def dev(a, b):
        a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
        d, n = ctypes.c_float(-1.0), ctypes.c_int(-1)
        fp = ctypes.POINTER(ctypes.c_float)
        assert lib.clm_logit_deviation(a.ctypes.data_as(fp), b.ctypes.data_as(fp), a.shape[0], a.shape[1], ctypes.byref(d),
                                       ctypes.byref(n)) == 0
        return d.value, n.value

    ref = np.array([[0.5, -0.5], [1.0, 2.0], [3.0, -1.0]], np.float32)
    got = ref + np.array([[1e-4, 0], [0, -3e-4], [2e-4, 0]], np.float32)
    d, n = dev(got, ref)
    assert abs(d - 3e-4) < 1e-6 and n == 0
    for bad in (np.nan, np.inf, -np.inf):
        for row in range(3):                                  # row 0 is the case the old transformer reduction lost
            g = got.copy()
            g[row, 1] = bad
            assert dev(g, ref)[0] == np.inf, (bad, row)
    flipped = ref.copy()
    flipped[1] = flipped[1, ::-1]
    assert dev(flipped, ref) == (1.0, 1)
    assert lib.clm_logit_deviation(None, None, 1, 2, None, None) == _native.E_INVALID


def test_headers_are_plain_c_and_a_c_client_links(built_lib, tmp_path):
    """The boundary is a C ABI: both headers compile as strict C99 on their own, and the plain-C client of
    tests/c_abi/ (run on the GPU by tests/test_gpu_c_abi.py) compiles and links against the library with gcc."""
    import shutil
    import subprocess

    inc = REPO / "include"
    for h in ("chimeralm_hip.h", "chimeralm_feed.h"):
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                        str(inc / h)], check=True)
    if not (Path("/opt/rocm/include/hip/hip_runtime_api.h").exists() and shutil.which("gcc")):
        pytest.skip("HIP runtime headers not installed")
    lib = Path(built_lib)
    subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-O1", f"-I{inc}", "-I/opt/rocm/include",
                    str(REPO / "tests" / "c_abi" / "abi_client.c"), "-o", str(tmp_path / "abi_client"), f"-L{lib.parent}",
                    "-lchimeralm_hip", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib.parent}",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)


def test_no_cpu_path_and_loud_failure(built_lib):
    from chimeralm_amd import lm
    from chimeralm_amd.engine import Engine, EngineError

    model = lm.ChimeraLM.new()
    with pytest.raises(RuntimeError, match="MI355X"):
        model(torch.zeros(1, 8, dtype=torch.long))
    with pytest.raises(EngineError):
        Engine("cpu")
    if not torch.cuda.is_available():            # clm_create must fail cleanly, not crash, without a device
        with pytest.raises(EngineError):
            Engine("cuda:0")


def test_product_package_never_imports_oracle():
    for f in (REPO / "chimeralm_amd").rglob("*.py"):
        assert "oracle" not in f.read_text(), f"{f} mentions the oracle: the product path must not use it"
    # ... and nothing under tools/ imports it either (developer probes that need it live under tests/)
    for f in (REPO / "tools").rglob("*.py"):
        txt = f.read_text()
        assert "from oracle" not in txt and "import oracle" not in txt, f"{f} imports the oracle: move it under tests/"


def test_state_dict_keys_match_reference_checkpoint_layout():
    from chimeralm_amd import lm
    from oracle import hyena_oracle as ho

    model = lm.ChimeraLM.new()
    sd = ho.make_state_dict(3)
    assert set(model.state_dict()) == set(sd)
    model.load_state_dict(sd, strict=True)
    for k in ("net.backbone.backbone.layers.0.mixer.in_proj.weight", "net.head.classifier.6.layers.3.bias",
              "net.backbone.backbone.layers.3.mixer.filter_fn.implicit_filter.5.freq"):
        assert k in model.state_dict()
    assert model.net.number_of_classes == 2


def test_head_must_be_production_configuration():
    from chimeralm_amd.hyena import BinarySequenceClassifier

    with pytest.raises(NotImplementedError):
        BinarySequenceClassifier(256, 512, 2, 0.1, "mean")


def test_tokenizer_and_collator_match_reference_goldens(golden_dir):
    from chimeralm_amd import tokenizer as T

    assert T.CharTokenizer(add_cls=True).encode("ATCG") == [0, 7, 10, 8, 9, 1]        # tests/test_tokenzier.py:11
    prod = T.load_tokenizer_from_hyena_model("hyenadna-small-32k-seqlen")
    assert prod.encode("ATCG") == [7, 10, 8, 9, 1] and prod.padding_side == "left"
    assert prod.max_len_single_sentence == 32769                                       # <= 32768 bases + [SEP]
    assert len(prod("A" * 40000, truncation=True, max_length=prod.max_len_single_sentence)["input_ids"]) == 32769
    with pytest.raises(ValueError):
        T.load_tokenizer_from_hyena_model("nope")
    gold = json.loads((golden_dir / "collate_golden.json").read_text())
    for case in gold["cases"]:
        tok = T.CharTokenizer(model_max_length=case["model_max_length"], padding_side=case["padding_side"], add_cls=True)
        assert tok.max_len_single_sentence == case["max_length"]
        feats = [T.tokenize_and_align_labels_and_quals_ids({"id": n, "seq": s}, tok, case["max_length"])
                 for n, s in gold["reads"]]
        assert [f["input_ids"] for f in feats] == case["per_read_input_ids"]
        assert [f["id"] for f in feats] == case["per_read_id"]
        batch = T.DataCollator(tok).torch_call(feats)
        assert batch["input_ids"].dtype == torch.int64 and batch["input_ids"].tolist() == case["input_ids"]
        assert batch["id"].dtype == torch.int8 and batch["id"].tolist() == case["id_int8"]
        assert batch["labels"].tolist() == case["labels"]
    # superset behaviour: names of 128..255 characters survive (the reference collator raises on them)
    tok = T.CharTokenizer()
    long_name = "n" * 200
    b = T.DataCollator(tok).torch_call([T.tokenize_and_align_labels_and_quals_ids({"id": long_name, "seq": "ACGT"}, tok, 100)])
    from chimeralm_amd.callbacks import resume_read_name

    assert resume_read_name(b["id"][0]) == long_name


def test_resume_read_name_matches_reference_where_it_accepts(golden_dir):
    from chimeralm_amd.callbacks import resume_read_name

    for row in json.loads((golden_dir / "readname_golden.json").read_text()):
        if not isinstance(row["resumed"], dict):
            assert resume_read_name(torch.tensor(row["row_int8"], dtype=torch.int8)) == row["resumed"]
    with pytest.raises(ValueError):
        resume_read_name(torch.zeros(256, dtype=torch.int8))
    assert resume_read_name(torch.zeros(0)) == ""


def test_bam_module_batches_like_the_reference(golden_dir):
    from chimeralm_amd import bam, tokenizer as T
    from oracle import data_oracle as do

    path = golden_dir / "test_chimric_reads.bam"
    assert list(bam.parse_bam_file(path)) == list(do.chimeric_reads(path))
    tok = T.load_tokenizer_from_hyena_model("hyenadna-small-32k-seqlen")
    dm = bam.BamDataModule(tokenizer=tok, train_data_path="dummy.bam", predict_data_path=path, batch_size=12)
    dm.setup("predict")
    batches = list(dm.predict_dataloader())
    assert len(batches) == 9 and [len(b["labels"]) for b in batches] == [12] * 8 + [4]
    assert all((b["labels"] == -1).all() for b in batches)
    reads = list(do.chimeric_reads(path))
    b0 = batches[0]
    want = do.collate([{"input_ids": do.tokenize(r["seq"], 32769), "id": do.pack_read_name(r["id"]), "labels": -1}
                       for r in reads[:12]], padding_side="left")
    assert (b0["input_ids"].numpy() == want["input_ids"]).all() and (b0["id"].numpy() == want["id"]).all()
    assert max(b["input_ids"].shape[1] for b in batches) == 32769            # truncation is exercised
    with pytest.raises(RuntimeError, match="not divisible"):
        dm.setup("predict", world_size=5)
    dm.setup("predict", world_size=2, rank=1)                                # rank r sees reads r, r+G, ...
    first = next(iter(dm.predict_dataloader()))
    assert first["input_ids"].shape[0] == 6


def _feeder_batches(path, **kw):
    from chimeralm_amd.feeder import BamFeeder

    with BamFeeder(path, pinned=False, **kw) as f:          # plain host memory: the decoder itself needs no GPU
        out = list(f)
        return out, f.stats()


def test_native_feeder_matches_the_reference_data_path(golden_dir):
    """C++ feeder (BGZF inflate, record decode, SA/primary selection, tokenisation from the 4-bit codes, [SEP],
    truncation, id rows, left-padded collation, per-rank batches) == oracle restatement of the reference's Python path."""
    from oracle import data_oracle as do

    path = golden_dir / "test_chimric_reads.bam"
    reads = list(do.chimeric_reads(path))
    assert len(reads) == 100                                                 # tests/test_data_module.py: all selected
    feats = [{"input_ids": do.tokenize(r["seq"], 32769), "id": do.pack_read_name(r["id"]), "labels": -1} for r in reads]
    for world, rank, bs in ((1, 0, 12), (2, 0, 6), (2, 1, 6), (4, 3, 3), (1, 0, 7)):
        mine = feats[rank::world]
        got, st = _feeder_batches(path, batch_size=bs, world=world, rank=rank, max_tokens=32769)
        assert len(got) == -(-len(mine) // bs)
        for i, (ids, names) in enumerate(got):
            want = do.collate(mine[i * bs: (i + 1) * bs], padding_side="left")
            assert ids.dtype == np.uint8 and np.array_equal(ids.astype(np.int64), want["input_ids"])
            assert names.dtype == np.int8 and np.array_equal(names, want["id"])
        assert st["records"] == 100 and st["selected"] == 100 and st["delivered"] == len(mine)
    # truncation length, right padding and max_reads are honoured
    got, st = _feeder_batches(path, batch_size=5, max_tokens=1000, pad_left=False, max_reads=10)
    assert len(got) == 2 and st["selected"] == 10 and st["truncated_bases"] > 0
    want = do.collate([{"input_ids": do.tokenize(r["seq"], 1000), "id": do.pack_read_name(r["id"]), "labels": -1}
                       for r in reads[:5]], padding_side="right")
    assert np.array_equal(got[0][0].astype(np.int64), want["input_ids"]) and got[0][0].shape[1] <= 1000


def test_native_feeder_selection_and_errors(tmp_path, golden_dir):
    """Unmapped / secondary / supplementary / SA-less records are skipped like bam.py:21-23; broken files fail loudly."""
    import gzip
    import struct

    from chimeralm_amd.feeder import BamFeeder, FeederError

    def record(name, seq, flag, aux):
        codes = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
        packed = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            packed[i // 2] |= codes[ch] << (4 if i % 2 == 0 else 0)
        body = struct.pack("<iiBBHHHiiii", 0, 0, len(name) + 1, 60, 0, 1, flag, len(seq), -1, -1, 0) + name.encode() + b"\0"
        body += struct.pack("<I", len(seq) << 4) + bytes(packed) + b"\xff" * len(seq) + aux
        return struct.pack("<i", len(body)) + body

    def bgzf(payload):
        out = b""
        for i in range(0, len(payload), 40000):                                   # several members, records straddle them
            comp = __import__("zlib").compressobj(6, 8, -15)
            chunk = payload[i: i + 40000]
            data = comp.compress(chunk) + comp.flush()
            out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(data) + 25)
            out += data + struct.pack("<II", __import__("zlib").crc32(chunk), len(chunk))
        return out + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")

    sa = b"SAZchr1,1,+,10M,60,0;\0"
    recs = [("keep1", "ACGTNACGTRY", 0, b"NMi\x01\0\0\0" + sa), ("unmapped", "ACGT", 4, sa), ("secondary", "ACGT", 256, sa),
            ("supp", "ACGT", 2048, sa), ("no_sa", "ACGT", 0, b"NMi\x01\0\0\0"), ("keep2", "T" * 70001, 16, b"XBBc\x02\0\0\0\x01\x02" + sa),
            ("n" * 200, "GATTACA", 0, sa)]
    header = b"BAM\1" + struct.pack("<i", 4) + b"@HD\n" + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", 1000)
    path = tmp_path / "t.bam"
    path.write_bytes(bgzf(header + b"".join(record(*r) for r in recs)))
    from chimeralm_amd import bam as pybam
    assert [r["id"] for r in pybam.parse_bam_file(path)] == ["keep1", "keep2", "n" * 200]       # the Python mirror agrees
    (ids, names), = _feeder_batches(path, batch_size=8, max_tokens=32769)[0]
    assert ids.shape == (3, 32769)
    assert ids[0, -12:].tolist() == [7, 8, 9, 10, 11, 7, 8, 9, 10, 6, 6, 1]          # R, Y -> [UNK]; trailing [SEP]
    assert (ids[0, :-12] == 4).all() and (ids[1, :-1] == 10).all() and ids[1, -1] == 1   # left pad; truncated to 32768 bases
    assert names[0, 0] == 5 and bytes(names[0, 1:6].view(np.uint8)) == b"keep1" and names[2, 0] == np.int8(200 - 256)
    # not a BAM / truncated file / bad arguments
    (tmp_path / "x.bam").write_bytes(gzip.compress(b"not a bam"))
    with pytest.raises(FeederError, match="BGZF|BAM"):
        BamFeeder(tmp_path / "x.bam", pinned=False)
    with pytest.raises(FeederError, match="cannot open"):
        BamFeeder(tmp_path / "missing.bam", pinned=False)
    data = path.read_bytes()
    (tmp_path / "cut.bam").write_bytes(data[: len(data) // 2])
    with pytest.raises(FeederError, match="truncated|corrupt|ends inside"):
        with BamFeeder(tmp_path / "cut.bam", pinned=False, batch_size=8) as f:
            list(f)
    with pytest.raises(FeederError, match="bad batch_size"):
        BamFeeder(path, pinned=False, batch_size=0)


def test_native_feeder_parallel_inflate_is_order_preserving(tmp_path):
    """BGZF members are inflated by worker threads: the stream the decoder sees, hence every batch, is the same for any number
    of them (file order), and a member corrupted in the middle of the file fails the run whichever worker meets it."""
    import sys

    sys.path.insert(0, str(REPO / "tools"))
    from feeder_bench import write_bam

    from chimeralm_amd.feeder import BamFeeder, FeederError

    path = tmp_path / "many_blocks.bam"
    write_bam(path, 400, 3000, seed=3)                        # ~30 members, records straddle them
    runs = {}
    for threads in (1, 2, 5):
        with BamFeeder(path, batch_size=7, pinned=False, inflate_threads=threads, slots=2) as f:
            runs[threads] = [(ids.copy(), names.copy()) for ids, names in f]
            assert f.stats()["selected"] == 400
    assert len(runs[1]) == -(-400 // 7)
    for threads in (2, 5):
        assert len(runs[threads]) == len(runs[1])
        for (a_ids, a_names), (b_ids, b_names) in zip(runs[1], runs[threads]):
            assert np.array_equal(a_ids, b_ids) and np.array_equal(a_names, b_names)
    raw = bytearray(path.read_bytes())
    raw[len(raw) // 2] ^= 0x5A                                 # flip bits inside some member's deflate data
    (tmp_path / "bad.bam").write_bytes(bytes(raw))
    for threads in (1, 4):
        with pytest.raises(FeederError, match="corrupt|BGZF|truncated"):
            with BamFeeder(tmp_path / "bad.bam", batch_size=7, pinned=False, inflate_threads=threads) as f:
                list(f)
    with pytest.raises(FeederError, match="inflate_threads"):
        BamFeeder(path, pinned=False, inflate_threads=-1)


def test_native_feeder_ring_backpressure(golden_dir):
    """A consumer that holds slots stalls the decoder instead of being overwritten; releasing resumes it."""
    from chimeralm_amd.feeder import BamFeeder

    with BamFeeder(golden_dir / "test_chimric_reads.bam", batch_size=4, slots=2, pinned=False, max_tokens=500) as f:
        a, b = f.next(), f.next()
        a_ids, b_ids = a.ids.copy(), b.ids.copy()
        import time
        time.sleep(0.2)                                       # decoder would have lapped the ring by now if it could
        assert np.array_equal(a.ids, a_ids) and np.array_equal(b.ids, b_ids) and a.first_index == 0 and b.first_index == 4
        f.release(a), f.release(b)
        n = 8
        while (c := f.next()) is not None:
            assert c.first_index == n
            n += c.n_reads
            f.release(c)
        assert n == 100


def test_prediction_writer_file_format(tmp_path):
    from types import SimpleNamespace

    from chimeralm_amd.callbacks import PredictionWriter
    from chimeralm_amd.tokenizer import pack_read_name

    ids = torch.tensor([pack_read_name("read/1"), pack_read_name("read;2"), [0] * 256], dtype=torch.int64).to(torch.int8)
    logits = torch.tensor([[1.0, -1.0], [-2.0, 3.0], [0.5, 0.4]])
    w = PredictionWriter(tmp_path / "out", "batch")
    w.write_on_batch_end(SimpleNamespace(global_rank=3), None, (logits, torch.full((3,), -1)), None, {"id": ids}, 7, 0)
    assert (tmp_path / "out" / "3_7.txt").read_text() == "read/1\t0\nread;2\t1\nerror_read_2\t0\n"


def test_cli_surface():
    from typer.testing import CliRunner

    from chimeralm_amd.__main__ import app

    r = CliRunner().invoke(app, ["predict", "--help"])
    assert r.exit_code == 0
    for opt in ("--gpus", "-g", "--output", "-o", "--batch-size", "-b", "--workers", "-w", "--random", "-r",
                "--verbose", "-v"):
        assert opt in r.output
    r = CliRunner().invoke(app, ["predict", "x.bam", "--gpus", "0"])
    assert r.exit_code != 0 and "CPU" in " ".join(r.output.split())


def test_hydra_style_configs_instantiate():
    """configs/model/hyena.yaml keeps the reference's shape; resolve `_target_`s by hand (hydra is not installed)."""
    import importlib

    import yaml

    def build(node):
        if isinstance(node, dict) and "_target_" in node:
            mod, _, name = node["_target_"].rpartition(".")
            fn = getattr(importlib.import_module(mod), name)
            kwargs = {k: build(v) for k, v in node.items() if k not in ("_target_", "_partial_")}
            if node.get("_partial_"):
                from functools import partial

                return partial(fn, **kwargs)
            return fn(**kwargs)
        return node

    cfg = yaml.safe_load((REPO / "configs" / "model" / "hyena.yaml").read_text())
    model = build(cfg)
    assert type(model).__name__ == "ClassificationLit" and model.net.number_of_classes == 2
    assert model.net.precision == "fp16c"
    tcfg = yaml.safe_load((REPO / "configs" / "model" / "transformer.yaml").read_text())   # reference transformer.yaml:3-12
    tmodel = build(tcfg)
    assert type(tmodel.net).__name__ == "SequenceCNNTransformer" and tmodel.net.number_of_classes == 2
    assert "transformer_encoder.layers.11.self_attn.in_proj_weight" in tmodel.net.state_dict()
    assert tuple(tmodel.net.state_dict()["pos_encoder.pe"].shape) == (1, 32768, 256)
    data = yaml.safe_load((REPO / "configs" / "data" / "bam.yaml").read_text())
    dm = build({**data, "predict_data_path": str(REPO / "tests/golden/test_chimric_reads.bam"), "batch_size": 4})
    assert dm.tokenizer.padding_side == "left"


def test_eval_yaml_composes_like_hydra(tmp_path):
    """The Hydra entry route (reference eval.py:87-101 + configs/eval.yaml): defaults list, group override, value / add
    overrides, interpolations, mandatory ckpt_path -- composed by chimeralm_amd.config (hydra is not in the image)."""
    from chimeralm_amd.config import ConfigError, compose, instantiate, instantiate_callbacks

    with pytest.raises(ConfigError, match="ckpt_path"):
        compose(REPO / "configs", "eval.yaml", [])
    bam = str(REPO / "tests/golden/test_chimric_reads.bam")
    cfg = compose(REPO / "configs", "eval.yaml", ["ckpt_path=/x/y.ckpt", f"+data.predict_data_path={bam}", "data.batch_size=12",
                                                    "trainer=ddp", "model.net.precision=fp32"], output_dir=tmp_path)
    assert cfg.ckpt_path == "/x/y.ckpt" and cfg.task_name == "eval"
    assert cfg.trainer._target_ == "chimeralm_amd.trainer.Trainer" and cfg.trainer.devices == 4 and cfg.trainer.strategy == "ddp"
    assert cfg.trainer.default_root_dir == str(tmp_path)                       # ${paths.output_dir} -> ${hydra:runtime.output_dir}
    assert cfg.callbacks.write.output_dir == f"{tmp_path}/predicts"
    assert cfg.model.net.precision == "fp32" and cfg.data.batch_size == 12
    with pytest.raises(ConfigError, match=r"use \+"):
        compose(REPO / "configs", "eval.yaml", ["ckpt_path=/x", "data.predict_data_path=/z"])   # new key needs '+', like hydra
    dm = instantiate(cfg.data)
    assert type(dm).__name__ == "BamDataModule" and dm.batch_size == 12 and dm.tokenizer.padding_side == "left"
    cbs = instantiate_callbacks(cfg.get("callbacks"))
    assert len(cbs) == 1 and type(cbs[0]).__name__ == "PredictionWriter"
    model = instantiate(cfg.model)
    assert type(model).__name__ == "ClassificationLit" and model.net.precision == "fp32"
    gpu = compose(REPO / "configs", "eval.yaml", ["ckpt_path=/x"], output_dir=tmp_path)          # default trainer: gpu.yaml
    assert gpu.trainer.devices == -1 and gpu.trainer.accelerator == "gpu"


def test_untrusted_checkpoint_pickles_are_refused(tmp_path, monkeypatch):
    """A .ckpt that pickles arbitrary objects is not unpickled unless CLM_TRUST_CHECKPOINT=1 (tensor-only ones load)."""
    import torch

    from chimeralm_amd import lm

    class Evil:
        def __reduce__(self):
            return (print, ("code from the checkpoint ran",))

    model = lm.ChimeraLM.new()
    good, bad = tmp_path / "good.ckpt", tmp_path / "bad.ckpt"
    torch.save({"state_dict": model.state_dict()}, good)
    torch.save({"state_dict": model.state_dict(), "extra": Evil()}, bad)
    lm.ChimeraLM.new().load_reference_checkpoint(good)
    monkeypatch.delenv("CLM_TRUST_CHECKPOINT", raising=False)
    with pytest.raises(RuntimeError, match="CLM_TRUST_CHECKPOINT"):
        lm.ChimeraLM.new().load_reference_checkpoint(bad)


def _bam_records(path):
    """(voffset, refID, pos, flag, name) of every record + header text, walking the BGZF members by hand."""
    import struct
    import zlib

    raw = Path(path).read_bytes()
    blocks, off = [], 0
    while off < len(raw):
        assert raw[off: off + 4] == b"\x1f\x8b\x08\x04"
        xlen = struct.unpack_from("<H", raw, off + 10)[0]
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        data = zlib.decompress(raw[off + 12 + xlen: off + bsize - 8], -15) if bsize - xlen - 20 > 0 else b""
        assert zlib.crc32(data) == struct.unpack_from("<I", raw, off + bsize - 8)[0]
        blocks.append((off, data))
        off += bsize
    assert blocks[-1][1] == b"", "BGZF EOF marker missing"
    stream = b"".join(d for _, d in blocks)
    starts, acc = [], 0
    for o, d in blocks:
        starts.append((acc, o))
        acc += len(d)

    def voff(p):
        i = max(j for j, (s, _) in enumerate(starts) if s <= p and (blocks[j][1] or s < p or True))
        while i + 1 < len(starts) and starts[i + 1][0] <= p and blocks[i][1] == b"":
            i += 1
        # a position at a block boundary belongs to the later, non-empty block
        while i + 1 < len(starts) and starts[i + 1][0] == p:
            i += 1
        return (starts[i][1] << 16) | (p - starts[i][0])

    assert stream[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", stream, 4)[0]
    text = stream[8: 8 + l_text].rstrip(b"\0").decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", stream, p)[0]
    p += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", stream, p)[0]
        p += 8 + ln
    recs = []
    while p < len(stream):
        bs = struct.unpack_from("<i", stream, p)[0]
        ref, pos, l_name = struct.unpack_from("<iiB", stream, p + 4)
        flag = struct.unpack_from("<H", stream, p + 4 + 14)[0]
        name = stream[p + 36: p + 36 + l_name - 1].decode()
        recs.append((voff(p), ref, pos, flag, name, stream[p: p + 4 + bs]))
        p += 4 + bs
    return text, n_ref, recs


def test_filter_outputs_do_not_depend_on_the_thread_count(tmp_path, monkeypatch):
    """Filter / sort / index inflate and deflate their BGZF members on worker threads: filtered BAM, sorted BAM and BAI are
    byte-identical for 1 and 5 workers (blocks are cut at the same payload offsets and written in order; the index's virtual
    offsets are resolved after the compressed sizes are known)."""
    import hashlib
    import sys

    sys.path.insert(0, str(REPO / "tools"))
    from feeder_bench import write_bam

    from chimeralm_amd import _native as N

    lib = N.load()
    src = tmp_path / "src.bam"
    write_bam(src, 300, 2500, seed=9)
    drop = [f"read_{i:08d}".encode() for i in range(0, 300, 4)]
    arr = (ctypes.c_char_p * len(drop))(*drop)
    digests = {}
    for threads in ("1", "5"):
        monkeypatch.setenv("CLM_BAM_THREADS", threads)
        kept, dropped, n = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        f, fs = tmp_path / f"f{threads}.bam", tmp_path / f"fs{threads}.bam"
        assert lib.clm_bam_filter(str(src).encode(), str(f).encode(), arr, len(drop), ctypes.byref(kept), ctypes.byref(dropped)) == 0
        assert (kept.value, dropped.value) == (225, 75)
        assert lib.clm_bam_sort_index(str(f).encode(), str(fs).encode(), None, ctypes.byref(n)) == 0 and n.value == 225
        digests[threads] = [hashlib.md5(p.read_bytes()).hexdigest() for p in (f, fs, Path(str(fs) + ".bai"))]
    assert digests["1"] == digests["5"]


def test_filter_drops_artifacts_sorts_and_indexes(tmp_path, golden_dir):
    """`filter` (reference __main__.py:99-153): reads labelled 1 disappear, the rest is copied bit for bit, then coordinate
    sort + BAI whose chunks / linear index / counts are consistent with the records' real virtual offsets."""
    import shutil
    import struct

    from chimeralm_amd import filter as flt

    bam = tmp_path / "reads.bam"
    shutil.copyfile(golden_dir / "test_chimric_reads.bam", bam)
    _, _, recs_in = _bam_records(bam)
    names = [r[4] for r in recs_in]
    pred = tmp_path / "reads.predictions"
    pred.mkdir()
    labels = {n: (i % 3 == 0) * 1 for i, n in enumerate(dict.fromkeys(names))}
    items = list(labels.items())
    (pred / "0_0.txt").write_text("".join(f"{n}\t{v}\n" for n, v in items[:60]))
    (pred / "0_1.txt").write_text("".join(f"{n}\t{v}\n" for n, v in items[60:]) + "\n")
    assert flt.load_predictions_from_folder(pred) == labels
    (pred / "bad.txt").write_text("only_one_column\n")
    with pytest.raises(ValueError, match="Invalid line format"):
        flt.load_predictions_from_folder(pred)
    (pred / "bad.txt").unlink()

    res = flt.filter_bam_by_predcition(bam, pred, index=True, output_prediction=True)
    assert (pred / "predictions.txt").read_text().count("\n") == len(labels)
    text_f, n_ref, recs_f = _bam_records(res["filtered"])
    want = [r for r in recs_in if labels[r[4]] == 0]
    assert res["kept"] == len(want) and res["dropped"] == len(recs_in) - len(want) and res["dropped"] > 0
    assert [r[5] for r in recs_f] == [r[5] for r in want]                     # same records, same order, same bytes
    text_s, _, recs_s = _bam_records(res["sorted"])
    assert text_s.startswith("@HD") and "SO:coordinate" in text_s.split("\n")[0]
    key = lambda r: (r[1] if r[1] >= 0 else 1 << 31, r[2], (r[3] >> 4) & 1)   # noqa: E731
    assert [key(r) for r in recs_s] == sorted(key(r) for r in recs_s)
    assert sorted(r[5] for r in recs_s) == sorted(r[5] for r in want)
    # ---- BAI
    bai = Path(str(res["sorted"]) + ".bai").read_bytes()
    assert bai[:4] == b"BAI\x01" and struct.unpack_from("<i", bai, 4)[0] == n_ref
    p, index = 8, []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", bai, p)[0]
        p += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", bai, p)
            p += 8
            bins[b] = [struct.unpack_from("<QQ", bai, p + 16 * i) for i in range(n_chunk)]
            p += 16 * n_chunk
        n_intv = struct.unpack_from("<i", bai, p)[0]
        p += 4
        lin = list(struct.unpack_from(f"<{n_intv}Q", bai, p))
        p += 8 * n_intv
        index.append((bins, lin))
    n_no_coor = struct.unpack_from("<Q", bai, p)[0]
    assert p + 8 == len(bai) and n_no_coor == sum(1 for r in recs_s if r[1] < 0)

    def reg2bin(beg, end):
        end -= 1
        for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
            if beg >> shift == end >> shift:
                return base + (beg >> shift)
        return 0

    def span(rec):
        l_name, n_cigar = rec[12], struct.unpack_from("<H", rec, 16)[0]
        ops = struct.unpack_from(f"<{n_cigar}I", rec, 36 + l_name)
        return max(1, sum(v >> 4 for v in ops if (v & 15) in (0, 2, 3, 7, 8)))

    for ref in range(n_ref):
        mine = [r for r in recs_s if r[1] == ref]
        bins, lin = index[ref]
        if not mine:
            assert not bins
            continue
        meta = bins.pop(37450)
        assert meta[0][0] == mine[0][0] and meta[1] == (sum(1 for r in mine if not r[3] & 4), sum(1 for r in mine if r[3] & 4))
        for v, _, pos, flag, _, rec in mine:
            b = reg2bin(pos, pos + (1 if flag & 4 else span(rec)))
            assert any(c0 <= v < c1 for c0, c1 in bins[b]), "record start not covered by a chunk of its bin"
            assert lin[pos >> 14] <= v and lin[pos >> 14] != 0
        assert all(c0 < c1 for cs in bins.values() for c0, c1 in cs) and lin == sorted(lin)
    # no predictions -> nothing happens; non-BAM input is refused
    empty = tmp_path / "none"
    empty.mkdir()
    assert flt.filter_bam_by_predcition(bam, empty) is None
    with pytest.raises(RuntimeError, match="cannot open"):
        flt.filter_bam_by_predcition(tmp_path / "missing.bam", pred)


# ---------------------------------------------------------------------------------------------- filter: SAM, spill, unplaced
def _bam_body(name, tid, pos, flag, n_bases, rng, tags=b"SAZchr1,100,+,50M,60,0;\0"):
    import struct

    b = rng.integers(0, 4, n_bases)
    codes = np.array([1, 2, 4, 8], dtype=np.uint8)[b]
    bp = codes if n_bases % 2 == 0 else np.append(codes, np.uint8(0))
    packed = ((bp[0::2] << 4) | bp[1::2]).astype(np.uint8).tobytes()
    qual = rng.integers(2, 41, n_bases).astype(np.uint8).tobytes()
    nm = name.encode() + b"\0"
    cigar = struct.pack("<I", n_bases << 4) if not flag & 4 else b""
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(nm), 60, 4680, 1 if cigar else 0, flag, n_bases, -1, -1, 0) + nm
    return body + cigar + packed + qual + tags


def _write_bam(path, text, refs, bodies):
    import struct
    import zlib

    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for n, ln in refs:
        hdr += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", ln)
    stream = hdr + b"".join(struct.pack("<i", len(b)) + b for b in bodies)
    with open(path, "wb") as f:
        for i in range(0, len(stream), 0xff00):
            chunk = stream[i: i + 0xff00]
            c = zlib.compressobj(6, 8, -15)
            data = c.compress(chunk) + c.flush()
            f.write(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(data) + 25))
            f.write(data + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


def test_sort_spills_runs_and_merges_to_the_same_bytes(tmp_path, monkeypatch):
    """`pysam.sort` spills to disk beyond its memory budget (reference __main__.py:148-151): so does clm_bam_sort_index.  A BAM
    of ~9 MB of records, many equal keys, three references and unplaced reads, sorted with a 1 MiB budget (about ten runs on
    disk, k-way merged) must give the byte-identical sorted BAM and BAI as the in-memory sort, leave no temporary file, and
    keep input order among equal keys (stability across runs)."""
    import hashlib

    from chimeralm_amd import _native as N

    lib = N.load()
    rng = np.random.default_rng(3)
    refs = [("chr1", 1 << 27), ("chr2", 1 << 26), ("chrM", 16569)]
    bodies = []
    for i in range(3000):
        unplaced = i % 97 == 0
        tid = -1 if unplaced else int(rng.integers(0, 3))
        pos = -1 if unplaced else int(rng.integers(0, 40)) * 1000          # 40 distinct positions: thousands of ties
        flag = 4 if unplaced else (16 if rng.random() < 0.5 else 0)
        bodies.append(_bam_body(f"r{i:05d}", tid, pos, flag, 2000 + int(rng.integers(0, 1500)), rng))
    src = tmp_path / "src.bam"
    _write_bam(src, "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:134217728\n", refs, bodies)
    out = {}
    for tag, mb in (("mem", "4096"), ("spill", "1")):
        monkeypatch.setenv("CLM_SORT_MEM_MB", mb)
        dst = tmp_path / f"{tag}.sorted.bam"
        n = ctypes.c_int64()
        assert lib.clm_bam_sort_index(str(src).encode(), str(dst).encode(), None, ctypes.byref(n)) == 0, lib.clm_bam_last_error()
        assert n.value == 3000
        out[tag] = [hashlib.md5(p.read_bytes()).hexdigest() for p in (dst, Path(str(dst) + ".bai"))]
        assert not list(tmp_path.glob("*.run")), "temporary sort runs left behind"
    assert out["mem"] == out["spill"]
    text, _, recs = _bam_records(tmp_path / "spill.sorted.bam")
    assert "SO:coordinate" in text.split("\n")[0]
    key = lambda r: (r[1] if r[1] >= 0 else 1 << 31, r[2], (r[3] >> 4) & 1)   # noqa: E731
    keys = [key(r) for r in recs]
    assert keys == sorted(keys) and len(recs) == 3000
    for a, b in zip(recs, recs[1:]):                                          # stable: equal keys keep the input order
        if key(a) == key(b):
            assert a[4] < b[4], (a[4], b[4])


def _bam_to_sam(path) -> str:
    """Independent BAM -> SAM text decoder (SAM specification 4.2), used to feed the SAM input of `filter`."""
    import struct

    text, n_ref, recs = _bam_records(path)
    raw_refs = []
    # reference names from the header blob
    import zlib

    raw = Path(path).read_bytes()
    stream, off = b"", 0
    while off < len(raw):
        xlen = struct.unpack_from("<H", raw, off + 10)[0]
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        if bsize - xlen - 20 > 0:
            stream += zlib.decompress(raw[off + 12 + xlen: off + bsize - 8], -15)
        off += bsize
    l_text = struct.unpack_from("<i", stream, 4)[0]
    p = 8 + l_text + 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", stream, p)[0]
        raw_refs.append(stream[p + 4: p + 4 + ln - 1].decode())
        p += 8 + ln
    lines = [text.rstrip("\n")] if text.strip() else []
    for _, tid, pos, flag, name, rec in recs:
        b = rec[4:]
        l_name, mapq, _bin, n_cig, _flag, l_seq, ntid, npos, tlen = struct.unpack_from("<BBHHHiiii", b, 8)
        q = 32 + l_name
        cig = "".join(f"{v >> 4}{'MIDNSHP=X'[v & 15]}" for v in struct.unpack_from(f"<{n_cig}I", b, q)) or "*"
        q += 4 * n_cig
        sq = b[q: q + (l_seq + 1) // 2]
        seq = "".join("=ACMGRSVTWYHKDBN"[(sq[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq)) or "*"
        q += (l_seq + 1) // 2
        ql = b[q: q + l_seq]
        qual = "*" if (l_seq == 0 or ql[0] == 0xFF) else "".join(chr(c + 33) for c in ql)
        q += l_seq
        tags = []
        while q < len(b):
            tag, typ = b[q: q + 2].decode(), chr(b[q + 2])
            q += 3
            if typ == "A":
                tags.append(f"{tag}:A:{chr(b[q])}"); q += 1
            elif typ in "cCsSiI":
                fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[typ]
                tags.append(f"{tag}:i:{struct.unpack_from(fmt, b, q)[0]}"); q += struct.calcsize(fmt)
            elif typ == "f":
                tags.append(f"{tag}:f:{struct.unpack_from('<f', b, q)[0]:.9g}"); q += 4
            elif typ in "ZH":
                e = b.index(b"\0", q)
                tags.append(f"{tag}:{typ}:{b[q:e].decode()}"); q = e + 1
            elif typ == "B":
                sub, cnt = chr(b[q]), struct.unpack_from("<i", b, q + 1)[0]
                fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
                vals = struct.unpack_from(f"<{cnt}{fmt}", b, q + 5)
                tags.append(f"{tag}:B:{sub}," + ",".join(f"{v:.9g}" if sub == "f" else str(v) for v in vals))
                q += 5 + cnt * struct.calcsize(fmt)
            else:
                raise ValueError(typ)
        rn = raw_refs[tid] if tid >= 0 else "*"
        rnext = "*" if ntid < 0 else ("=" if ntid == tid else raw_refs[ntid])
        lines.append("\t".join([name, str(flag), rn, str(pos + 1), str(mapq), cig, rnext, str(npos + 1), str(tlen), seq, qual] + tags))
    return "\n".join(lines) + "\n"


def test_filter_reads_sam_text_like_the_reference(tmp_path, golden_dir):
    """The reference opens any path whose suffix is not .bam as SAM text (__main__.py:127).  The reference's test BAM, written
    out as SAM by an independent decoder, must filter to the same records as the BAM itself -- byte for byte where the
    encoding is canonical (bin, smallest integer tag types: what htslib writes), and semantically everywhere."""
    import shutil

    from chimeralm_amd import filter as flt

    bam = tmp_path / "reads.bam"
    shutil.copyfile(golden_dir / "test_chimric_reads.bam", bam)
    sam = tmp_path / "reads.sam"
    sam.write_text(_bam_to_sam(bam))
    _, _, recs_in = _bam_records(bam)
    names = list(dict.fromkeys(r[4] for r in recs_in))
    pred = tmp_path / "pred"
    pred.mkdir()
    (pred / "0_0.txt").write_text("".join(f"{n}\t{i % 2}\n" for i, n in enumerate(names)))
    res_s = flt.filter_bam_by_predcition(sam, pred, index=True)
    assert res_s["filtered"].name == "reads.filtered.bam" and res_s["sorted"].exists()   # same output names for both inputs:
    ts, ns, rs = _bam_records(res_s["filtered"])                                          # read one back before the other runs
    res_b = flt.filter_bam_by_predcition(bam, pred, index=False)
    assert (res_s["kept"], res_s["dropped"]) == (res_b["kept"], res_b["dropped"]) and res_s["dropped"] > 0
    tb, nb, rb = _bam_records(res_b["filtered"])
    assert ns == nb and ts.split("\n")[1:] == tb.split("\n")[1:]            # same references and header lines
    assert [(r[1], r[2], r[3], r[4]) for r in rs] == [(r[1], r[2], r[3], r[4]) for r in rb]
    same = sum(a[5] == b[5] for a, b in zip(rs, rb))
    assert same == len(rb), f"only {same} of {len(rb)} records re-encode to the original bytes"
    # every optional-field type and the corner encodings
    text = ("@HD\tVN:1.6\n@SQ\tSN:c1\tLN:5000\n@SQ\tSN:c2\tLN:900\n"
            "q1\t0\tc1\t101\t60\t3S10M2D5M1I4M\t=\t301\t200\tACGTNACGTNACGTNACGTNACG\t" + "I" * 23 +
            "\tXA:A:k\tXi:i:-70000\tXj:i:-5\tXk:i:300\tXl:i:70000\tXf:f:1.5\tXz:Z:hello world\tXh:H:1AE301\tXb:B:s,-3,4,5\tXc:B:f,0.5,2\n"
            "q2\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\n"
            "q3\t16\tc2\t1\t0\t4M\tc1\t7\t-9\tACGT\t*\n")
    sam2 = tmp_path / "t.sam"
    sam2.write_text(text)
    (pred / "0_0.txt").write_text("q3\t1\nq1\t0\n")
    res = flt.filter_bam_by_predcition(sam2, pred, index=True)
    assert (res["kept"], res["dropped"], res["unplaced"]) == (2, 1, 0)       # SAM text: the unplaced q2 is iterated and kept
    assert _bam_to_sam(res["filtered"]) == "\n".join(text.split("\n")[:5]) + "\n"
    sam2.write_text(text.replace("3S10M", "3S10Q"))
    with pytest.raises(RuntimeError, match="line 4: bad CIGAR"):
        flt.filter_bam_by_predcition(sam2, pred)
    assert not (tmp_path / "t.filtered.bam").exists()                         # no partial output left behind (:140-144)


def test_filter_leaves_out_unplaced_records_of_a_bam(tmp_path):
    """`bam_file.fetch()` on a BAM walks the index reference by reference (reference __main__.py:131): records without a
    reference placement never reach the output.  CLM_BAM_KEEP_UNPLACED keeps them."""
    from chimeralm_amd import _native as N

    lib = N.load()
    rng = np.random.default_rng(5)
    bodies = [_bam_body(f"r{i}", -1 if i % 4 == 3 else 0, -1 if i % 4 == 3 else 100 * i, 4 if i % 4 == 3 else 0, 50, rng) for i in range(20)]
    src = tmp_path / "u.bam"
    _write_bam(src, "@HD\tVN:1.6\tSO:coordinate\n", [("chr1", 100000)], bodies)
    drop = (ctypes.c_char_p * 1)(b"r0")
    k, d, u = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    assert lib.clm_bam_filter_ex(str(src).encode(), str(tmp_path / "a.bam").encode(), drop, 1, 0, ctypes.byref(k), ctypes.byref(d), ctypes.byref(u)) == 0
    assert (k.value, d.value, u.value) == (14, 1, 5)
    assert all(r[1] >= 0 for r in _bam_records(tmp_path / "a.bam")[2])
    assert lib.clm_bam_filter_ex(str(src).encode(), str(tmp_path / "b.bam").encode(), drop, 1, N.BAM_KEEP_UNPLACED, ctypes.byref(k),
                               ctypes.byref(d), ctypes.byref(u)) == 0
    assert (k.value, d.value, u.value) == (19, 1, 0)
