"""Checkpoint loading with the reference's file layout (utils/commons/ckpt_utils.py:7-66):
`<dir>/model_ckpt_steps_<N>.ckpt` = {'state_dict': {<model_name>: {...}} | flat 'model_name.key' dict}."""
import glob
import os
import re

import torch


def get_all_ckpts(work_dir, steps=None):
    pat = "%s/model_ckpt_steps_%s.ckpt" % (work_dir, "*" if steps is None else steps)
    return sorted(glob.glob(pat), key=lambda x: -int(re.findall(r".*steps\_(\d+)\.ckpt", x)[0]))


def get_last_checkpoint(work_dir, steps=None):
    paths = get_all_ckpts(work_dir, steps)
    if not paths:
        return None, None
    return torch.load(paths[0], map_location="cpu", weights_only=False), paths[0]


def load_ckpt(cur_model, ckpt_base_dir, model_name="model", force=True, strict=True):
    if os.path.isfile(ckpt_base_dir):
        base_dir, ckpt_path = os.path.dirname(ckpt_base_dir), ckpt_base_dir
        checkpoint = torch.load(ckpt_base_dir, map_location="cpu", weights_only=False)
    else:
        base_dir = ckpt_base_dir
        checkpoint, ckpt_path = get_last_checkpoint(ckpt_base_dir)
    if checkpoint is None:
        msg = "| ckpt not found in %s." % base_dir
        assert not force, msg
        print(msg)
        return
    sd = checkpoint["state_dict"]
    if any("." in k for k in sd.keys()):
        sd = {k[len(model_name) + 1:]: v for k, v in sd.items() if k.startswith(model_name + ".")}
    elif "." not in model_name:
        sd = sd[model_name]
    else:
        base, rest = model_name.split(".", 1)
        sd = {k[len(rest) + 1:]: v for k, v in sd[base].items() if k.startswith(rest + ".")}
    if not strict:
        cur = cur_model.state_dict()
        for k in [k for k, v in sd.items() if k in cur and cur[k].shape != v.shape]:
            print("| Unmatched keys: ", k, cur[k].shape, sd[k].shape)
            del sd[k]
    cur_model.load_state_dict(sd, strict=strict)
    print("| load '%s' from '%s'." % (model_name, ckpt_path))
