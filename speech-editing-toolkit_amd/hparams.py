"""Config surface: one global mutable dict `hparams` filled from yaml + `-hp` overrides.

Mirrors the behaviour of the reference's utils/commons/hparams.py:25-131 (CLI flags
--config --exp_name -hp/--hparams --infer --validate --reset --remove --debug; recursive
`base_config` inheritance; checkpoints/<exp>/config.yaml merge unless --reset; typed
"a=1,b.c=2,d=[1 1]" overrides) so the reference's yaml files and command lines work unchanged.
"""
import argparse
import os
import shutil

import yaml

hparams = {}
_printed = False


def _merge(old, new):
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(old.get(k), dict):
            _merge(old[k], v)
        else:
            old[k] = v


def _load_yaml_chain(fn, seen, chain):
    if not os.path.exists(fn):
        return {}
    with open(fn) as f:
        cfg = yaml.safe_load(f) or {}
    seen.add(fn)
    out = {}
    bases = cfg.get("base_config")
    if bases is not None:
        for c in bases if isinstance(bases, list) else [bases]:
            if c.startswith("."):
                c = os.path.normpath(os.path.join(os.path.dirname(fn), c))
            if c not in seen:
                _merge(out, _load_yaml_chain(c, seen, chain))
    _merge(out, cfg)
    chain.append(fn)
    return out


def _apply_overrides(cfg, spec):
    """-hp "a=1,b.c=2,d=[1 1]": the value is typed by the existing entry (hparams.py:93-106)."""
    for item in spec.split(","):
        k, v = item.split("=")
        v = v.strip("'\" ")
        node = cfg
        parts = k.split(".")
        for kk in parts[:-1]:
            node = node[kk]
        k = parts[-1]
        cur = node[k]
        if v in ("True", "False") or type(cur) in (bool, list, dict):
            if type(cur) == list:
                v = v.replace(" ", ",")
            node[k] = eval(v)  # noqa: S307 -- same contract as the reference CLI
        else:
            node[k] = type(cur)(v)


def set_hparams(config="", exp_name="", hparams_str="", print_hparams=True, global_hparams=True):
    global _printed
    if config == "" and exp_name == "":
        ap = argparse.ArgumentParser(description="")
        ap.add_argument("--config", type=str, default="")
        ap.add_argument("--exp_name", type=str, default="")
        ap.add_argument("-hp", "--hparams", type=str, default="")
        for flag in ("infer", "validate", "reset", "remove", "debug"):
            ap.add_argument("--" + flag, action="store_true")
        args, unknown = ap.parse_known_args()
        print("| Unknow hparams: ", unknown)
    else:
        args = argparse.Namespace(config=config, exp_name=exp_name, hparams=hparams_str, infer=False,
                                  validate=False, reset=False, remove=False, debug=False)
    assert args.config != "" or args.exp_name != ""
    if args.config != "":
        assert os.path.exists(args.config), args.config
    chain = []
    saved = {}
    work_dir = ""
    ckpt_cfg = ""
    if args.exp_name != "":
        work_dir = "checkpoints/%s" % args.exp_name
        ckpt_cfg = "%s/config.yaml" % work_dir
        if os.path.exists(ckpt_cfg):
            with open(ckpt_cfg) as f:
                saved.update(yaml.safe_load(f) or {})
    cfg = {}
    if args.config != "":
        cfg.update(_load_yaml_chain(args.config, set(), chain))
    if not args.reset:
        cfg.update(saved)
    cfg["work_dir"] = work_dir
    if args.hparams != "":
        _apply_overrides(cfg, args.hparams)
    if work_dir != "" and args.remove:
        if input("REMOVE old checkpoint? Y/N [Default: N]: ").lower() == "y":
            shutil.rmtree(work_dir, ignore_errors=True)
    if work_dir != "" and (not os.path.exists(ckpt_cfg) or args.reset) and not args.infer:
        os.makedirs(work_dir, exist_ok=True)
        with open(ckpt_cfg, "w") as f:
            yaml.safe_dump(cfg, f)
    cfg["infer"], cfg["debug"], cfg["validate"], cfg["exp_name"] = args.infer, args.debug, args.validate, args.exp_name
    if global_hparams:
        hparams.clear()
        hparams.update(cfg)
    if print_hparams and global_hparams and not _printed:
        print("| Hparams chains: ", chain)
        print("| Hparams: " + ", ".join("%s: %s" % kv for kv in sorted(cfg.items())))
        _printed = True
    return cfg
