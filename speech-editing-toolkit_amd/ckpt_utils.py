"""Checkpoints in the reference's file layout.

Loading (utils/commons/ckpt_utils.py:7-66): `<dir>/model_ckpt_steps_<N>.ckpt` = {'state_dict': {<model_name>: {...}} |
flat 'model_name.key' dict}.  Saving / resuming (utils/commons/trainer.py:384-471): {'epoch', 'global_step',
'checkpoint_callback_best', 'optimizer_states': [AdamW state_dict], 'state_dict': {'model': ...}}, written atomically,
oldest files beyond `num_ckpt_keep` removed."""
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


def save_ckpt(work_dir, model, optimizer=None, global_step=0, epoch=0, best=None, num_ckpt_keep=3, model_name="model"):
    """trainer.py:430-471 (`save_checkpoint` / `_atomic_save` / `dump_checkpoint`) for one model + one optimizer."""
    os.makedirs(work_dir, exist_ok=True)
    # `global_step_unit`: this trainer counts optimizer updates, the reference counts batches (they differ only under
    # accumulate_grad_batches > 1; trainer.Trainer checks the key before resuming such a run)
    ckpt = {"epoch": int(epoch), "global_step": int(global_step), "checkpoint_callback_best": best,
            "global_step_unit": "updates",
            "optimizer_states": [optimizer.state_dict()] if optimizer is not None else [],
            "state_dict": {model_name: {k: v.detach().cpu() for k, v in model.state_dict().items()}}}
    path = "%s/model_ckpt_steps_%d.ckpt" % (work_dir, int(global_step))
    tmp = path + ".part"
    torch.save(ckpt, tmp)
    os.replace(tmp, path)
    for old in get_all_ckpts(work_dir)[num_ckpt_keep:]:
        os.remove(old)
    return path


def restore_ckpt(work_dir, model, optimizer=None, model_name="model", strict=True, return_meta=False):
    """trainer.py:384-428 (`restore_weights` + `restore_opt_state`).  Returns (global_step, epoch), (0, 0) if the
    directory holds no checkpoint (the reference then trains from its random init, trainer.py:153-157); with
    `return_meta` a third value: the checkpoint's scalar entries (so that callers need not read the file again)."""
    checkpoint, path = get_last_checkpoint(work_dir)
    if checkpoint is None:
        return (0, 0, {}) if return_meta else (0, 0)
    model.load_state_dict(checkpoint["state_dict"][model_name], strict=strict)
    if optimizer is not None and checkpoint.get("optimizer_states"):
        try:
            optimizer.load_state_dict(checkpoint["optimizer_states"][0])
        except ValueError:
            print("| WARMING: optimizer parameters not match !!!")  # the reference's message, trainer.py:420
    from . import ops
    ops.bump_weights_epoch()  # packed weight images are stale now
    if return_meta:
        meta = {k: v for k, v in checkpoint.items() if k not in ("state_dict", "optimizer_states")}
        return int(checkpoint["global_step"]), int(checkpoint["epoch"]), meta
    return int(checkpoint["global_step"]), int(checkpoint["epoch"])
