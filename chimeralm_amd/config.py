"""The slice of Hydra that `python eval.py ckpt_path=... +data.predict_data_path=...` needs
(/root/reference/eval.py:87-101, configs/eval.yaml:1-22): compose a root YAML with its `defaults` list from config
groups, apply `key=value` / `+key=value` / `~key` command-line overrides, resolve `${a.b}` interpolations and build
objects from `_target_` nodes (`_partial_` supported).  When the real `hydra` / `omegaconf` are importable `eval.py`
uses them instead; this module exists because neither is in the MI355X image, and the predict route must not depend on
packages the box lacks.
"""
from __future__ import annotations

import importlib
import re
import time
from functools import partial
from pathlib import Path
from typing import Any

import yaml


class ConfigError(ValueError):
    pass


class Node(dict):
    """dict with attribute access, like the DictConfig the reference code reads (`cfg.data._target_`, `cfg.get("x")`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Node({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def to_container(x):
    """Plain dict / list copy of a composed config (for printing / yaml.safe_dump)."""
    if isinstance(x, dict):
        return {k: to_container(v) for k, v in x.items()}
    if isinstance(x, list):
        return [to_container(v) for v in x]
    return x


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _load(path: Path) -> dict:
    if not path.exists():
        raise ConfigError(f"config file not found: {path}")
    return yaml.safe_load(path.read_text()) or {}


def _compose_file(config_dir: Path, rel: str, group_overrides: dict[str, str]) -> dict:
    """One YAML with its own `defaults` list resolved (entries: `_self_`, `name` (same group), `group: option|null`)."""
    path = config_dir / (rel if rel.endswith(".yaml") else rel + ".yaml")
    raw = _load(path)
    defaults = raw.pop("defaults", ["_self_"])
    if "_self_" not in defaults:
        defaults = list(defaults) + ["_self_"]
    out: dict = {}
    group_dir = str(Path(rel).parent)
    for d in defaults:
        if d == "_self_":
            _merge(out, raw)
        elif isinstance(d, str):                                  # sibling in the same group, merged at this level
            _merge(out, _compose_file(config_dir, str(Path(group_dir) / d) if group_dir != "." else d, group_overrides))
        elif isinstance(d, dict):
            (group, option), = d.items()
            group = group.replace("override ", "").strip()
            option = group_overrides.get(group, option)
            if option in (None, "null"):
                continue
            sub = _compose_file(config_dir, f"{group}/{option}", group_overrides)
            _merge(out.setdefault(group.split("/")[-1], {}), sub)
        else:
            raise ConfigError(f"{path}: bad defaults entry {d!r}")
    return out


def _set_path(cfg: dict, dotted: str, value, *, must_exist: bool | None):
    keys = dotted.split(".")
    cur = cfg
    for k in keys[:-1]:
        if k not in cur or not isinstance(cur[k], dict):
            if must_exist:
                raise ConfigError(f"override {dotted}: no such node (use +{dotted}=... to add)")
            cur[k] = {}
        cur = cur[k]
    if must_exist is True and keys[-1] not in cur:
        raise ConfigError(f"override {dotted}: key is not in the config (use +{dotted}=... to add)")
    if must_exist is False and keys[-1] in cur:
        raise ConfigError(f"override +{dotted}: key already exists (drop the '+')")
    cur[keys[-1]] = value


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _resolve(cfg: dict, runtime: dict):
    def lookup(expr: str):
        if expr.startswith("hydra:"):
            cur: Any = runtime
            for k in expr[6:].split("."):
                cur = cur[k]
            return cur
        if expr.startswith("now:"):
            return time.strftime(expr[4:])
        if expr.startswith("oc.env:"):
            import os

            name, _, default = expr[7:].partition(",")
            return os.environ.get(name, default)
        cur = cfg
        for k in expr.split("."):
            cur = cur[k]
        return cur

    def walk(x, depth=0):
        if depth > 16:
            raise ConfigError("interpolation cycle")
        if isinstance(x, dict):
            return {k: walk(v, depth) for k, v in x.items()}
        if isinstance(x, list):
            return [walk(v, depth) for v in x]
        if isinstance(x, str) and "${" in x:
            m = _INTERP.fullmatch(x)
            if m:                                                # whole-string interpolation keeps the value's type
                return walk(lookup(m.group(1)), depth + 1)
            return walk(_INTERP.sub(lambda mm: str(lookup(mm.group(1))), x), depth + 1)
        return x

    return walk(cfg)


def compose(config_dir: str | Path, config_name: str, overrides: list[str] | None = None, *, output_dir: str | Path | None = None,
            cwd: str | Path | None = None) -> Node:
    """Hydra-style composition.  Missing mandatory values (`???`) raise when still unset after the overrides."""
    config_dir = Path(config_dir)
    overrides = list(overrides or [])
    group_over, value_over = {}, []
    groups = {p.name for p in config_dir.iterdir() if p.is_dir()}
    for ov in overrides:
        if ov.startswith("~"):
            value_over.append(("del", ov[1:], None))
            continue
        if "=" not in ov:
            raise ConfigError(f"override {ov!r}: expected key=value")
        k, v = ov.split("=", 1)
        add = k.startswith("+")
        k = k.lstrip("+")
        if not add and k in groups:
            group_over[k] = v
        else:
            value_over.append(("add" if add else "set", k, yaml.safe_load(v)))
    cfg = _compose_file(config_dir, config_name, group_over)
    for kind, k, v in value_over:
        if kind == "del":
            cur = cfg
            ks = k.split(".")
            for kk in ks[:-1]:
                cur = cur.get(kk, {})
            cur.pop(ks[-1], None)
        else:
            _set_path(cfg, k, v, must_exist=(kind == "set"))
    cwd = Path(cwd or Path.cwd())
    task = cfg.get("task_name", "run")
    out = Path(output_dir) if output_dir else cwd / "logs" / str(task) / "runs" / time.strftime("%Y-%m-%d_%H-%M-%S")
    cfg = _resolve(cfg, {"runtime": {"output_dir": str(out), "cwd": str(cwd)}})

    def missing(x, path=""):
        if isinstance(x, dict):
            for k, v in x.items():
                missing(v, f"{path}.{k}" if path else k)
        elif x == "???":
            raise ConfigError(f"Missing mandatory value: {path}")

    missing(cfg)
    return _wrap(cfg)


def instantiate(node, **extra):
    """`hydra.utils.instantiate`: build `_target_(**kwargs)` depth-first; `_partial_: true` gives functools.partial."""
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    if not isinstance(node, dict):
        return node
    if "_target_" not in node:
        return Node({k: instantiate(v) for k, v in node.items()})
    mod, _, name = node["_target_"].rpartition(".")
    fn = getattr(importlib.import_module(mod), name)
    kwargs = {k: instantiate(v) for k, v in node.items() if k not in ("_target_", "_partial_")}
    kwargs.update(extra)
    return partial(fn, **kwargs) if node.get("_partial_") else fn(**kwargs)


def instantiate_callbacks(callbacks_cfg) -> list:
    """reference chimeralm/utils/instantiators.py: every child node with a `_target_` becomes one callback."""
    out = []
    for _, cb in (callbacks_cfg or {}).items():
        if isinstance(cb, dict) and "_target_" in cb:
            out.append(instantiate(cb))
    return out
