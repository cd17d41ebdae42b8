"""Checkpoint compatibility with the reference (NFModel.load_checkpoint, pl_module.py:112-129).

The reference accepts either a raw ``state_dict`` (``.pth``) or a PyTorch-Lightning checkpoint whose
``"state_dict"`` holds the whole LightningModule; in the latter case only keys starting with ``student.`` are
kept and the prefix is stripped.  Keys and shapes of this package's modules equal the reference's, so both kinds
load unchanged.  Files are read with ``weights_only=True`` (nothing from the file is executed).
"""
import torch


def extract_state_dict(obj, prefix="student."):
    """Raw state_dict -> itself; Lightning checkpoint -> the `prefix`-ed sub-dict with the prefix stripped."""
    if isinstance(obj, dict) and "state_dict" in obj:
        return {".".join(k.split(".")[1:]): v for k, v in obj["state_dict"].items() if k.startswith(prefix)}
    return obj


def load_checkpoint(model, checkpoint_path, map_location=None, prefix="student."):
    if map_location is None:
        map_location = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    state = torch.load(checkpoint_path, map_location=map_location, weights_only=True)
    model.load_state_dict(extract_state_dict(state, prefix))
    eng = getattr(getattr(model, "flow", None), "_engine", None)
    if eng is not None:
        eng.mark_dirty()   # cached parameter products of a frozen model must be rebuilt
    return model
