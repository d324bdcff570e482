"""Minimal Hydra/OmegaConf stand-in for the features the reference's CLI uses (hydra-core / omegaconf are not installed
in the target image; when they are importable the real ``hydra.utils.instantiate`` semantics are what this mirrors).

Covers exactly what `main_v2.py` + `configs/**` of the reference rely on (SURVEY.md section 5 "Config / flags"):
  * YAML composition with a `defaults:` list (`- /group: name`, `- group: name`, `- _self_`, `override /group: name`),
    `# @package _global_` group files, and command-line overrides `a.b.c=value` / `group=name`;
  * `${a.b}` interpolation, `${eval:"..."}`, `${uuid4:x}`, `${now:%fmt}`, `${hydra:runtime.choices.<group>}`,
    `${hydra:runtime.cwd}`, `${hydra:run.dir}`;
  * `instantiate(node, *args, **kw)` for `_target_` / `_args_` / `_partial_`-free nodes, recursively.
"""
from __future__ import annotations

import copy
import importlib
import os
import re
import uuid
from datetime import datetime
from typing import Any, Dict, List, Optional

import yaml


class DictConfig(dict):
    """dict with attribute access (the slice of omegaconf.DictConfig the reference code touches)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def __setitem__(self, k, v):
        super().__setitem__(k, _wrap(v))

    def get(self, k, default=None):
        return self[k] if k in self else default

    def __deepcopy__(self, memo):
        return DictConfig({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(v):
    if isinstance(v, DictConfig):
        return v
    if isinstance(v, dict):
        d = DictConfig()
        for k, x in v.items():
            dict.__setitem__(d, k, _wrap(x))
        return d
    if isinstance(v, (list, tuple)):
        return [_wrap(x) for x in v]
    return v


def to_container(cfg) -> Any:
    if isinstance(cfg, dict):
        return {k: to_container(v) for k, v in cfg.items()}
    if isinstance(cfg, list):
        return [to_container(v) for v in cfg]
    return cfg


def _set_path(cfg: dict, path: str, value):
    keys = path.split(".")
    cur = cfg
    for k in keys[:-1]:
        if k not in cur or not isinstance(cur[k], dict):
            cur[k] = DictConfig()
        cur = cur[k]
    cur[keys[-1]] = value


def _get_path(cfg: dict, path: str):
    cur = cfg
    for k in path.split("."):
        if isinstance(cur, list):
            cur = cur[int(k)]
        else:
            cur = cur[k]
    return cur


def _merge(dst: dict, src: dict):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _parse_value(text: str):
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


_INTERP = re.compile(r"\$\{([^${}]*)\}")


class _Resolver:
    def __init__(self, root: dict, choices: Dict[str, str], cwd: str, run_dir_template: Optional[str]):
        self.root, self.choices, self.cwd = root, choices, cwd
        self.now = datetime.now()
        self.run_dir_template = run_dir_template
        self._uuid: Dict[str, str] = {}

    def lookup(self, expr: str):
        expr = expr.strip()
        if ":" in expr and not expr.startswith("."):
            kind, _, arg = expr.partition(":")
            kind = kind.strip()
            if kind == "eval":
                arg = arg.strip()
                if len(arg) >= 2 and arg[0] == arg[-1] and arg[0] in "\"'":
                    arg = arg[1:-1]
                return eval(arg, {"__builtins__": {"int": int, "float": float, "min": min, "max": max, "round": round,
                                                   "len": len, "abs": abs}})
            if kind == "uuid4":
                return self._uuid.setdefault(arg, str(uuid.uuid4())[-4:])
            if kind == "now":
                return self.now.strftime(arg)
            if kind == "hydra":
                if arg.startswith("runtime.choices."):
                    return self.choices.get(arg[len("runtime.choices."):], "")
                if arg == "runtime.cwd":
                    return self.cwd
                if arg in ("run.dir", "runtime.output_dir"):
                    if self.run_dir_template is None:
                        return self.cwd
                    return self.resolve_value(self.run_dir_template)
                raise KeyError(f"unsupported hydra interpolation '{arg}'")
            if kind == "oc.env":
                name, _, default = arg.partition(",")
                return os.environ.get(name.strip(), default.strip() or None)
            raise KeyError(f"unknown resolver '{kind}'")
        return self.resolve_value(_get_path(self.root, expr))

    def resolve_value(self, v):
        if isinstance(v, str):
            for _ in range(32):
                m = _INTERP.search(v)
                if not m:
                    break
                val = self.lookup(m.group(1))
                if m.start() == 0 and m.end() == len(v):
                    return self.resolve_value(val) if isinstance(val, str) else self.resolve_tree(val)
                v = v[:m.start()] + str(val) + v[m.end():]
            return v
        return self.resolve_tree(v)

    def resolve_tree(self, node):
        if isinstance(node, dict):
            for k in list(node.keys()):
                node[k] = self.resolve_value(node[k])
            return node
        if isinstance(node, list):
            return [self.resolve_value(x) for x in node]
        return node


def _load_yaml(path: str):
    with open(path) as f:
        text = f.read()
    is_global = bool(re.search(r"^#\s*@package\s+_global_", text, flags=re.M))
    return (yaml.safe_load(text) or {}), is_global


def compose(config_dir: str, config_name: str, overrides: Optional[List[str]] = None, cwd: Optional[str] = None) -> DictConfig:
    """Hydra-style composition of `<config_dir>/<config_name>` with `key=value` overrides."""
    overrides = list(overrides or [])
    cwd = cwd or os.getcwd()
    if not config_name.endswith((".yaml", ".yml")):
        config_name += ".yaml"
    primary, _ = _load_yaml(os.path.join(config_dir, config_name))
    defaults = primary.pop("defaults", [])
    hydra_node = primary.pop("hydra", {}) or {}

    groups = {d for d in os.listdir(config_dir) if os.path.isdir(os.path.join(config_dir, d))}
    group_over, value_over = {}, []
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not of the form key=value")
        k, _, v = ov.partition("=")
        k = k.lstrip("+~")
        if k in groups and "." not in k:
            group_over[k] = v
        else:
            value_over.append((k, _parse_value(v)))

    cfg = DictConfig()
    choices: Dict[str, str] = {}
    order = []
    saw_self = False
    for d in defaults:
        if d == "_self_":
            order.append(("_self_", None))
            saw_self = True
        elif isinstance(d, dict):
            (g, name), = d.items()
            g = g.replace("override ", "").strip().lstrip("/")
            order.append((g, name))
        else:
            raise ValueError(f"unsupported defaults entry {d!r}")
    if not saw_self:
        order.append(("_self_", None))
    for g in group_over:
        if g not in [o[0] for o in order]:
            order.append((g, None))

    def apply_group(g, name):
        name = group_over.get(g, name)
        if name in (None, "null", "???"):
            return
        choices[g] = str(name)
        body, is_global = _load_yaml(os.path.join(config_dir, g, str(name) + ".yaml"))
        sub_defaults = body.pop("defaults", [])
        for sd in sub_defaults:  # nested `override /group: name` entries inside a group file
            if isinstance(sd, dict):
                (sg, sname), = sd.items()
                apply_group(sg.replace("override ", "").strip().lstrip("/"), sname)
        if is_global:
            _merge(cfg, _wrap(body))
        else:
            if g not in cfg or not isinstance(cfg[g], dict):
                cfg[g] = DictConfig()
            _merge(cfg[g], _wrap(body))

    for g, name in order:
        if g == "_self_":
            _merge(cfg, _wrap(primary))
        else:
            apply_group(g, name)
    for k, v in value_over:
        _set_path(cfg, k, v)
    run_dir = ((hydra_node.get("run") or {}).get("dir")) if isinstance(hydra_node, dict) else None
    res = _Resolver(cfg, choices, cwd, run_dir)
    res.resolve_tree(cfg)
    return cfg


def load(path: str) -> DictConfig:
    """OmegaConf.load for an already-resolved config.yaml (e.g. <logdir>/config.yaml)."""
    with open(path) as f:
        return _wrap(yaml.safe_load(f) or {})


def locate(target: str):
    mod, _, attr = target.rpartition(".")
    if not mod:
        raise ImportError(f"'{target}' is not a dotted path")
    try:
        return getattr(importlib.import_module(mod), attr)
    except ModuleNotFoundError:
        # torchvision is absent in the target image: its handful of eval transforms are restated in utils.transforms
        if mod == "torchvision.transforms":
            return getattr(importlib.import_module("utils.transforms"), attr)
        raise


def instantiate(node, *args, **kwargs):
    """hydra.utils.instantiate for `_target_` nodes (recursive); non-target containers are returned converted."""
    if isinstance(node, dict):
        if "_target_" in node:
            fn = locate(node["_target_"])
            pos = [instantiate(a) for a in node.get("_args_", [])]
            kw = {k: instantiate(v) for k, v in node.items() if k not in ("_target_", "_args_", "_recursive_", "_convert_")}
            kw.update(kwargs)
            return fn(*pos, *args, **kw)
        return DictConfig({k: instantiate(v) for k, v in node.items()})
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    return node
