"""Checkpoint compatibility (pl_module.py:112-129): raw state_dicts with the reference's keys and Lightning-style
checkpoints with `student.`-prefixed keys both load into the package's modules (CPU-only: no kernels run)."""
import torch

import nf_distillation_amd as nfa
from golden_util import Golden
from nf_distillation_amd.checkpoint import load_checkpoint


def test_raw_and_lightning_checkpoints_load(tmp_path):
    g = Golden("glow8")
    sd = g.group("sd")                       # keys and shapes produced by the REFERENCE's state_dict()
    model = nfa.create_glow_model(dict(g.cfg))
    assert list(model.state_dict().keys()) == list(sd.keys())
    assert all(tuple(model.state_dict()[k].shape) == tuple(v.shape) for k, v in sd.items())
    raw = tmp_path / "teacher.pth"
    torch.save(sd, raw)
    load_checkpoint(model, str(raw), map_location="cpu")
    assert all(torch.equal(model.state_dict()[k], v) for k, v in sd.items())

    pl_ckpt = {"epoch": 3, "state_dict": {**{"student." + k: v + 1.0 for k, v in sd.items()},
                                           **{"teacher." + k: v for k, v in sd.items()}}}
    path = tmp_path / "epoch=3.ckpt"
    torch.save(pl_ckpt, path)
    load_checkpoint(model, str(path), map_location="cpu")
    assert all(torch.equal(model.state_dict()[k], v + 1.0) for k, v in sd.items())


def test_state_dict_keys_match_reference_for_every_fixture():
    for name in ("glow32", "glow8_additive", "glow8_nolu", "glow8_shuffle", "glow8_learntop", "tab6", "tab63",
                 "tab5_cond"):
        g = Golden(name)
        model = nfa.create_glow_model(dict(g.cfg))
        ref = g.group("sd")
        mine = model.state_dict()
        assert list(mine.keys()) == list(ref.keys()), name
        assert all(tuple(mine[k].shape) == tuple(ref[k].shape) for k in ref), name
