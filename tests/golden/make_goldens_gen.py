#!/usr/bin/env python3
"""Goldens for the GEN-side latent analysis entry points of DiVAE (sent_models.py:167-274): sweep, enumerate,
exp_enumerate -- produced by RUNNING THE REFERENCE on CPU in the build container (same harness as make_goldens.py:
nltk stub in sys.modules, Gumbel noise injected by overriding cat_connector.sample_gumbel on the instance).
Only inputs and expected outputs are stored.    usage: python tests/golden/make_goldens_gen.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as M  # noqa: E402  (imports the reference)
import torch  # noqa: E402
from laed.models import sent_models  # noqa: E402


def seqs(dec_ctx):
    return np.stack([t.view(-1).numpy() for t in dec_ctx['sequence']], axis=1).astype(np.int64)     # [B, max_dec_len]


def main():
    V, B, T = 60, 5, 9
    rs = np.random.RandomState(77)
    torch.manual_seed(77)
    config = M.make_config(max_utt_len=T, y_size=3, k=4, max_dec_len=7, enc_cell_size=32, dec_cell_size=32)
    model = sent_models.DiVAE(M.FakeCorpus(V), config)
    model.eval()
    out = {"config/" + k: np.array(v) for k, v in vars(config).items()}
    out["config/vocab_size"] = np.array(V)
    for k, v in model.state_dict().items():
        out["sd/" + k] = M.f32(v)
    batch, lens = M.make_batch(rs, B, T, V)
    u = rs.uniform(size=(B * config.y_size, config.k)).astype(np.float32)
    out["in/outputs"], out["in/output_lens"], out["in/u"] = batch, np.array(lens), u
    with torch.no_grad():
        M.inject_gumbel(model, u)
        ctx, labels, all_y = model.sweep({"outputs": batch, "output_lens": lens}, gen_type='greedy')
        out["sweep/sequence"], out["sweep/labels"], out["sweep/all_y_ids"] = seqs(ctx), labels.numpy(), all_y.numpy()
        for rep in (1, 2):
            out["enumerate%d/sequence" % rep] = seqs(model.enumerate(repeat=rep, gen_type='greedy'))
        ctx, ids = model.exp_enumerate(repeat=1, gen_type='greedy')
        out["exp_enumerate/sequence"], out["exp_enumerate/all_y_ids"] = seqs(ctx), np.array(ids, dtype=np.int64)
    out["meta/torch_version"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(HERE, "divae_gen_tiny.npz"), **out)
    print({k: v.shape for k, v in out.items() if not k.startswith(("sd/", "config/"))})


if __name__ == "__main__":
    main()
