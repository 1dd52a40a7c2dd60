#!/usr/bin/env python3
"""Golden record of the reference's training ENGINE, made by RUNNING /root/reference/laed/main.py (train, validate,
LossManager; lines 17-49 and 81-181) unmodified on a scripted stand-in model (tests/engine_stub.py).

Same rules as make_goldens.py (whose import shim is reused): build container only; only the record is committed.
`train` / `validate` run under torch 2.x when every loss term is a 1-element 1-D tensor (`val.data[0]` then works), the
preview generator is a recording no-op, and torch.save is replaced by a recorder for the duration of the run.

Per scenario (engine_stub.scenarios()):
  <s>/events      every call the engine made on the model / optimizer / feeds / gen / torch.save, in order
  <s>/log_train   the windowed "Train" report lines (print_step), as logged
  <s>/log_valid   the per-validation-pass report lines
  <s>/valid_loss  what validate() returned per pass (parsed from "Total valid loss ...")
  <s>/patience    every "Update patience to ..." value
  <s>/best        the final "Best validation loss ..."
  <s>/save_paths  where each checkpoint went (relative to session_dir)
plus validate() called on its own, and LossManager.pprint strings for a small fixed input.

usage:  python tests/golden/make_goldens_engine.py
"""
import logging
import os
import sys
import tempfile

import numpy as np

import make_goldens as mg          # noqa: F401  (sets up the reference import + nltk stand-in)
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from engine_stub import ScriptedFeed, ScriptedModel, engine_config, scenarios  # noqa: E402
from laed import main as ref_engine  # noqa: E402


class _Capture(logging.Handler):
    def __init__(self):
        logging.Handler.__init__(self)
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


def run(spec, session_dir):
    model = ScriptedModel(spec["valid"], spec["train"], spec["flush_at"], shape=(1,))
    cfg = engine_config(session_dir, **spec["cfg"])
    saves = []
    real_save = torch.save

    def fake_save(obj, path):
        model.events.append("save")
        saves.append(os.path.relpath(path, session_dir))
    cap = _Capture()
    root = logging.getLogger()
    root.addHandler(cap)
    old_level = root.level
    root.setLevel(logging.INFO)
    torch.save = fake_save
    try:
        ref_engine.train(model, ScriptedFeed("Train", spec["train_batches"], model),
                         ScriptedFeed("Valid", spec["valid_batches"], model), ScriptedFeed("Test", 1, model), cfg, "EVAL",
                         gen=lambda m, feed, config, evaluator, num_batch=1: m.events.append(
                             "gen:%s:%s:%s" % (feed.name, evaluator, num_batch)))
    finally:
        torch.save = real_save
        root.removeHandler(cap)
        root.setLevel(old_level)
    lines = cap.lines
    return {
        "events": np.array(model.events),
        "log_train": np.array([l for l in lines if ": Train " in l] or [""]),
        "log_valid": np.array([l for l in lines if l.startswith("Valid ")]),
        "valid_loss": np.array([float(l.split()[-1]) for l in lines if l.startswith("Total valid loss")]),
        "patience": np.array([float(l.split()[-1]) for l in lines if l.startswith("Update patience to")]),
        "best": np.array(float([l for l in lines if l.startswith("Best validation loss")][-1].split()[-1])),
        "save_paths": np.array(saves or [""]),
        "optimizers": np.array(sum(1 for e in model.events if e == "get_optimizer")),
    }


def main():
    G = {"meta/torch_version": np.array(torch.__version__)}
    with tempfile.TemporaryDirectory() as tmp:
        for name, spec in scenarios().items():
            for k, v in run(spec, tmp).items():
                G["%s/%s" % (name, k)] = v
            print(name, "events", len(G[name + "/events"]), "best", float(G[name + "/best"]),
                  "patience", G[name + "/patience"].tolist(), "saves", int((G[name + "/events"] == "save").sum()))
        # validate() on its own: three batches with losses v, v + 0.5, v + 1.0
        model = ScriptedModel([2.0], shape=(1,))
        cap = _Capture()
        logging.getLogger().addHandler(cap)
        logging.getLogger().setLevel(logging.INFO)
        G["validate/return"] = np.array(float(ref_engine.validate(model, ScriptedFeed("Valid", 3, model),
                                                                  engine_config(tmp), batch_cnt=7)))
        logging.getLogger().removeHandler(cap)
        G["validate/events"] = np.array(model.events)
        G["validate/log"] = np.array([l for l in cap.lines if l.startswith("Valid ")])
    # LossManager on a fixed input
    lm = ref_engine.LossManager()
    for a, b in ((2.0, 0.5), (4.0, 0.25), (3.0, 0.125)):
        lm.add_loss({"nll": torch.tensor([a]), "flag": True, "absent": None, "reg_kl": torch.tensor([b])})
        lm.add_backward_loss(torch.tensor([a + b]))
    G["lossmanager/pprint_all"] = np.array(lm.pprint("Train"))
    G["lossmanager/pprint_window2_prefix"] = np.array(lm.pprint("Train", window=2, prefix="1/2-(0.000)"))
    G["lossmanager/avg_loss"] = np.array(float(lm.avg_loss()))
    lm.clear()
    G["lossmanager/pprint_cleared"] = np.array(lm.pprint("Train"))
    path = os.path.join(HERE, "engine.npz")
    np.savez_compressed(path, **G)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
