"""Small helpers the harness needs (mirror of the importable subset of util.py:103-152)."""
import os

import torch
import yaml

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "grafp.yaml")


def load_config(config_path=DEFAULT_CONFIG):
    with open(config_path, "r") as fp:
        return yaml.safe_load(fp)


def override(config_val, arg):
    return arg if arg is not None else config_val


def query_len_from_seconds(seconds, overlap, dur):
    hop = dur * (1 - overlap)
    return int((seconds - dur) / hop + 1)


def seconds_from_query_len(query_len, overlap, dur):
    hop = dur * (1 - overlap)
    return int((query_len - 1) * hop + dur)


def save_ckp(state, model_name, model_folder, text):
    os.makedirs(model_folder, exist_ok=True)
    torch.save(state, "{}/model_{}_{}.pth".format(model_folder, model_name, text))


def strip_module_prefix(state_dict):
    """DataParallel/DDP checkpoints carry a 'module.' prefix (generate.py:93-94, test_fp.py:295-296)."""
    if state_dict and all(k.startswith("module.") for k in state_dict):
        return {k[len("module."):]: v for k, v in state_dict.items()}
    return state_dict


def load_ckp(checkpoint_fpath, model, optimizer=None, scheduler=None, map_location=None):
    """Checkpoint dict layout of train.py:212-220: epoch, loss, valid_acc, hit_rate, state_dict, optimizer,
    scheduler."""
    ckp = torch.load(checkpoint_fpath, map_location=map_location, weights_only=False)
    model.load_state_dict(strip_module_prefix(ckp["state_dict"]))
    if optimizer is not None:
        optimizer.load_state_dict(ckp["optimizer"])
    if scheduler is not None:
        scheduler.load_state_dict(ckp["scheduler"])
    return model, optimizer, scheduler, ckp["epoch"], ckp["loss"], ckp["valid_acc"]


def create_fp_dir(resume=None, ckp=None, epoch=1, train=True, large=False, parent_dir="logs/emb"):
    if resume is not None:
        ckp = os.path.splitext(os.path.basename(resume))[0]
    sub = "valid" if train else ("test_large" if large else "test")
    out = os.path.join(parent_dir, sub, f"{ckp}_{epoch}")
    os.makedirs(out, exist_ok=True)
    return out
