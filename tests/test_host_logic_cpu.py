"""CPU-only tests of the host-side mirror of the reference API (no kernels are launched)."""
import os

import numpy as np
import pytest
import torch

from helpers import golden_config, load_golden


class Corpus(object):
    def __init__(self, V):
        self.vocab = ['<pad>', '<unk>', '<s>', '</s>'] + ['w%d' % i for i in range(4, V)]
        self.rev_vocab = {w: i for i, w in enumerate(self.vocab)}


def test_state_dict_keys_match_reference_checkpoints():
    from laed_amd.models.sent_models import DiVAE, DiVST
    for name, cls in (("divae_tiny", DiVAE), ("divst_tiny", DiVST)):
        G = load_golden(name)
        model = cls(Corpus(int(G["config/vocab_size"])), golden_config(G))
        assert list(model.state_dict().keys()) == [str(k) for k in G["meta/state_dict_keys"]]
        for k, v in model.state_dict().items():
            assert tuple(v.shape) == G["sd/" + k].shape, k
        # the decoder embedding is the shared table (sent_models.py:70) and PAD row is zero
        dec = model.decoder if name == "divae_tiny" else model.prev_decoder
        assert dec.embedding.weight is model.embedding.weight
        assert float(model.embedding.weight[0].abs().sum()) == 0.0
        model.load_state_dict({k: torch.from_numpy(np.array(G["sd/" + k])) for k in model.state_dict()})


def test_summary_counts_parameters_like_the_reference():
    """model_bases.summary: per-child parameter counts; modules that share a table (the decoder's embedding) are counted
    under every child that holds them, as the reference's does."""
    from laed_amd.models.model_bases import summary
    from laed_amd.models.sent_models import DiVAE
    G = load_golden("divae_tiny")
    model = DiVAE(Corpus(int(G["config/vocab_size"])), golden_config(G))
    text = summary(model, show_weights=False)
    per_child = sum(sum(p.numel() for p in m.parameters()) for m in model._modules.values())
    assert text.startswith("DiVAE (") and text.endswith("Total Parameters={}".format(per_child))
    assert "(x_encoder):" in text and "(decoder):" in text and "weights=" not in text
    assert "weights=" in summary(model)


def test_session_dirs_and_forward_only_reload(tmp_path):
    """prepare_dirs_loggers / process_config / load_config (utils.py:58-117): a training run gets a time-stamped session
    directory with session.log and params.json; a forward_only run reloads them and keeps its decoding-time flags."""
    import json
    import logging
    import os
    from argparse import Namespace
    from laed_amd.utils import load_config, prepare_dirs_loggers, process_config
    root = logging.getLogger()
    before = list(root.handlers)
    try:
        cfg = Namespace(log_dir=str(tmp_path / "logs"), forward_only=False, y_size=20, k=10, gen_type="greedy", beam_size=10,
                        backward_size=5, batch_size=30, load_sess="")
        prepare_dirs_loggers(cfg, "ptb-utt.py")
        assert os.path.isdir(cfg.session_dir) and cfg.session_dir.endswith("-ptb-utt.py")
        logging.getLogger().info("hello session")
        for h in root.handlers:
            h.flush()
        assert "hello session" in open(os.path.join(cfg.session_dir, "session.log")).read()
        saved = json.load(open(os.path.join(cfg.session_dir, "params.json")))
        assert saved["y_size"] == 20 and saved["session_dir"] == cfg.session_dir
        again = Namespace(log_dir=cfg.log_dir, forward_only=True, load_sess=os.path.basename(cfg.session_dir),
                          backward_size=9, beam_size=3, gen_type="sample", batch_size=7, y_size=-1)
        out = process_config(again)
        assert (out.y_size, out.k, out.backward_size, out.beam_size, out.gen_type, out.batch_size) == (20, 10, 9, 3, "sample", 50)
        assert out.forward_only and out.load_sess == again.load_sess
        assert vars(load_config(os.path.join(cfg.session_dir, "params.json")))["k"] == 10
        assert process_config(cfg) is cfg
    finally:
        for h in list(root.handlers):
            if h not in before:
                root.removeHandler(h)
                h.close()


def test_same_seed_same_init_as_torch_modules():
    """GRU / Linear / Embedding keep torch's initialisation order: same manual_seed -> same weights as nn.GRU etc."""
    from laed_amd.enc2dec.base_modules import GRU
    torch.manual_seed(3)
    a = GRU(6, 10, bidirectional=True)
    torch.manual_seed(3)
    b = torch.nn.GRU(6, 10, 1, batch_first=True, bidirectional=True)
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert na == nb and torch.equal(pa, pb)


def test_unsupported_options_fail_loudly():
    from laed_amd.enc2dec.decoders import DecoderRNN
    from laed_amd.enc2dec.encoders import EncoderRNN
    from laed_amd.nn_lib import LinearConnector
    with pytest.raises(NotImplementedError):
        EncoderRNN(4, 8, rnn_cell='lstm')
    with pytest.raises(NotImplementedError):
        EncoderRNN(4, 8, variable_lengths=True)
    with pytest.raises(NotImplementedError):
        DecoderRNN(10, 5, 4, 8, 2, 3, rnn_cell='gru', use_attention=True)
    with pytest.raises(NotImplementedError):
        LinearConnector(4, 8, True)
    with pytest.raises(ValueError):
        EncoderRNN(4, 8, rnn_cell='rnn')


def test_forward_needs_the_gpu_no_cpu_fallback():
    from laed_amd._lib import LaedError
    from laed_amd.enc2dec.decoders import TEACH_FORCE
    from laed_amd.models.sent_models import DiVAE
    G = load_golden("divae_tiny")
    cfg = golden_config(G)
    model = DiVAE(Corpus(int(G["config/vocab_size"])), cfg)      # use_gpu False in the golden config
    with pytest.raises(LaedError):
        model({"outputs": G["in/outputs"], "output_lens": G["in/output_lens"].tolist()}, mode=TEACH_FORCE)


def test_criterions_have_no_cpu_path():
    """NLLEntropy (dense input), CatKLLoss and Entropy run on the HIP kernels only: CPU tensors are refused (their
    values against the reference goldens are checked on the GPU, tests/test_kernels_gpu.py)."""
    from argparse import Namespace
    from laed_amd import criterions
    from laed_amd._lib import LaedError
    G = load_golden("modules")
    logp, labels = torch.from_numpy(G["nll/logp"]), torch.from_numpy(G["nll/labels"])
    with pytest.raises(LaedError):
        criterions.NLLEntropy(0, Namespace(avg_type="word", use_gpu=False))(logp, labels)
    with pytest.raises(ValueError):
        criterions.NLLEntropy(0, Namespace(avg_type="bogus", use_gpu=False))(logp, labels)
    lq, lp = torch.from_numpy(G["catkl/log_qy"]), torch.from_numpy(G["catkl/log_py"])
    with pytest.raises(LaedError):
        criterions.CatKLLoss()(lq, lp, 4, unit_average=True)
    with pytest.raises(LaedError):
        criterions.Entropy()(lq, unit_average=True)


def test_pack_and_cast_type():
    from laed_amd.utils import FLOAT, INT, LONG, Pack, cast_type, str2bool
    p = Pack(a=1)
    p.add(b=[1, 2])
    q = p.copy()
    q.b.append(3)
    assert p.a == 1 and p.b == [1, 2] and q.b == [1, 2, 3]
    with pytest.raises(AttributeError):
        p.missing
    t = torch.tensor([1.7, 2.2])
    assert cast_type(t, LONG, False).dtype == torch.int64 and cast_type(t, INT, False).dtype == torch.int32
    assert cast_type(t.long(), FLOAT, False).dtype == torch.float32
    with pytest.raises(ValueError):
        cast_type(t, 7, False)
    assert str2bool("True") and str2bool("1") and not str2bool("no")


def test_proj_chunk_rows_fill_whole_cu_waves():
    from laed_amd.ops import proj_chunk_rows
    assert proj_chunk_rows(159744, 50000, 1024) == 40960        # 5 x (64 x 8 = 512 block tiles = 2 per CU)
    assert proj_chunk_rows(19968, 50000, 1024) == 9984          # one GPU's share at 8 GPUs: two balanced chunks, never the whole tensor
    assert proj_chunk_rows(1170, 10002, 512) == 1170            # small problem: one chunk
    r = proj_chunk_rows(10 ** 6, 50000, 512)
    assert r % 128 == 0 and r * 50016 * 4 <= (1 << 33)


def test_greedy_connector():
    from laed_amd.nn_lib import GreedyConnector
    logits = torch.tensor([[0.1, 2.0, 2.0], [3.0, -1.0, 0.0]])
    y, ids = GreedyConnector()(logits, False, return_max_id=True)
    assert ids.view(-1).tolist() == [1, 0] and y.tolist() == [[0, 1, 0], [1, 0, 0]]


def test_dialog_models_state_dict_and_freeze_switch():
    """AE-ED / ST-ED: reference state_dict keys/shapes, shared embedding tables, and the freeze_step phase switch of
    valid_loss (dialog_models.py:278-307, :549-585) -- host logic only, no kernel launches."""
    from laed_amd.models.dialog_models import AeED, StED
    from laed_amd.utils import Pack
    for name, cls in (("aeed_tiny", AeED), ("sted_tiny", StED)):
        G = load_golden(name)
        cfg = golden_config(G)
        model = cls(Corpus(int(G["config/vocab_size"])), cfg)
        assert list(model.state_dict().keys()) == [str(k) for k in G["meta/state_dict_keys"]]
        for k, v in model.state_dict().items():
            assert tuple(v.shape) == G["sd/" + k].shape, k
        assert model.decoder.embedding.weight is model.embedding.weight
        assert model.utt_encoder.embedding.weight is model.embedding.weight
        assert model.x_embedding.padding_idx is None and float(model.x_embedding.weight[0].abs().sum()) > 0.0
        one = torch.ones(())
        loss = Pack(nll=one * 2, pi_kl=one * 3, pi_nll=one * 5, attribute_nll=one * 10, vae_nll=one * 7,
                    vst_prev_nll=one * 7, vst_next_nll=one * 11, reg_kl=one * 13, mi=one, pi_err=one * 0.25,
                    adv_err=one * 0.5)
        phase_a = 7 + 13 if cls is AeED else 7 + 11 + 13
        phase_b = 2 + 5 + (1.0 if cls is AeED else 10.0)
        assert float(model.valid_loss(loss, 0)) == phase_a and model.kl_w == 0.0 and not model.flush_valid
        assert float(model.model_sel_loss(loss, 0)) == phase_a
        assert float(model.valid_loss(loss, cfg.freeze_step + 1)) == phase_b
        assert model.kl_w == 1.0 and model.flush_valid
        assert [int(p.requires_grad) for _, p in model.named_parameters()] == [int(r) for r in G["flip/requires_grad"]]
        assert float(model.model_sel_loss(loss, cfg.freeze_step + 1)) == 0.75
        assert float(loss.nll) == 2.0          # valid_loss is out of place here (the reference mutates loss.nll)


def test_corpus_bleu_known_answers_and_tokenizers():
    """corpus_bleu restates nltk's (absent here): the worked examples nltk's own docstrings quote for the BLEU paper's
    sentences (sentence_bleu 0.5045..., corpus_bleu 0.5920...), smoothing 1 on an empty 4-gram match, and the edge
    rules (no unigram match -> 0, brevity penalty from the closest reference length)."""
    import math
    from laed_amd.evaluators import corpus_bleu
    from laed_amd.utils import get_chat_tokenize, get_dekenize, get_tokenize
    hyp1 = "It is a guide to action which ensures that the military always obeys the commands of the party".split()
    ref1a = "It is a guide to action that ensures that the military will forever heed Party commands".split()
    ref1b = ("It is the guiding principle which guarantees the military forces always being under the command of the "
             "Party").split()
    ref1c = "It is the practical guide for the army always to heed the directions of the party".split()
    hyp2 = "he read the book because he was interested in world history".split()
    ref2a = "he was interested in world history because he read the book".split()
    assert abs(corpus_bleu([[ref1a, ref1b, ref1c]], [hyp1]) - 0.5045666840058485) < 1e-12
    assert abs(corpus_bleu([[ref1a, ref1b, ref1c], [ref2a]], [hyp1, hyp2]) - 0.5920778868801042) < 1e-12
    assert corpus_bleu([[ref2a]], [ref2a]) == 1.0
    assert corpus_bleu([[["a", "b"]]], [["c", "d"]]) == 0.0
    # 1-gram 3/4, 2-gram 1/3, 3- and 4-gram numerators 0 -> 0.1/2 and 0.1/1; hyp (4) shorter than ref (5): bp = e^(1-5/4)
    got = corpus_bleu([[["a", "b", "c", "d", "e"]]], [["a", "b", "d", "x"]])
    want = math.exp(1 - 5.0 / 4) * math.exp(0.25 * (math.log(3 / 4) + math.log(1 / 3) + math.log(0.1 / 2) + math.log(0.1 / 1)))
    assert abs(got - want) < 1e-12
    assert corpus_bleu([[["a", "b", "c", "d", "e"]]], [["a", "b", "d", "x"]], smooth=False) == 0.0
    assert get_tokenize()("hello <s> #tag %x, isn't it?") == ['hello', '<s>', '#tag', '%x', ',', 'isn', "'", 't', 'it', '?']
    assert get_chat_tokenize()("ok <sil> fine!") == ['ok', '<sil>', 'fine', '!']
    assert get_dekenize()(['i', 'do', "n't", 'know', ',', 'it', "'s", '(', 'ok', ')', '.']) == "i don't know, it's (ok)."


# ---- engine control flow (laed/main.py:81-181) on a stub model: no GPU, scripted validation losses --------------------
class _ScriptedModel(object):
    """Counts what the engine calls, in order; `validate` sees the scripted losses one per pseudo-epoch."""
    vocab = ['<pad>']
    kl_w = 0.0

    def __init__(self, valid_losses, flush_at=None):
        self.valid_losses = list(valid_losses)
        self.flush_at = flush_at
        self.flush_valid = False
        self.calls = []
        self.training = True
        self.optimizers = 0
        self.saved = []
        self._modules = {}

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def get_optimizer(self, config):
        self.optimizers += 1
        model = self

        class Opt(object):
            def zero_grad(self):
                model.calls.append("zero_grad")

            def step(self):
                model.calls.append("step")
        return Opt()

    def __call__(self, batch, mode=None):
        import torch
        if self.training:
            self.calls.append("forward")
            if self.flush_at is not None and self.calls.count("forward") == self.flush_at:
                self.flush_valid = True
            return {"nll": torch.tensor(1.0)}
        return {"nll": torch.tensor(float(self.valid_losses[0]))}

    def backward(self, batch_cnt, loss):
        self.calls.append("backward")

    def model_sel_loss(self, loss, batch_cnt):
        return loss["nll"]

    def state_dict(self):
        return {"epoch_marker": len(self.saved)}


class _OneBatchFeed(object):
    def __init__(self, name, n, model=None):
        self.name, self.n, self.model, self.ptr, self.num_batch = name, n, model, 0, n

    def epoch_init(self, config, shuffle=True, verbose=True):
        self.ptr = 0

    def next_batch(self):
        if self.ptr >= self.n:
            if self.model is not None and self.name == "Valid":
                self.model.valid_losses.pop(0)          # one scripted loss per validation pass
            return None
        self.ptr += 1
        return {"outputs": None}


def _engine_cfg(tmp_path, **kw):
    from argparse import Namespace
    cfg = dict(print_step=1000, ckpt_step=2, max_epoch=50, early_stop=True, improve_threshold=0.996,
               patient_increase=2.0, save_model=True, session_dir=str(tmp_path), preview_batch_num=1)
    cfg.update(kw)
    return Namespace(**cfg)


def test_engine_step_order_patience_and_checkpoints(tmp_path, monkeypatch):
    """The reference's control flow (laed/main.py:81-162) without a GPU: per step zero_grad -> forward -> backward ->
    step; a checkpoint at every new best validation loss; patience 10 grows to max(patience, epochs x patient_increase)
    only on an improvement beyond improve_threshold; training stops once patience <= epochs done."""
    import torch
    from laed_amd import main as engine
    saves = []
    monkeypatch.setattr(torch, "save", lambda sd, path: saves.append(path))
    monkeypatch.setattr(engine, "summary", lambda *a, **k: "stub")
    # epochs 1-6 improve (patience -> 12 at epoch 6), then nothing improves: stop when 12 <= epochs done
    script = [5.0, 4.0, 3.0, 2.0, 1.5, 1.0] + [1.2] * 40
    model = _ScriptedModel(script)
    best = engine.train(model, _OneBatchFeed("Train", 2), _OneBatchFeed("Valid", 1, model), None, _engine_cfg(tmp_path))
    assert best == 1.0
    assert len(saves) == 6 and all(p == os.path.join(str(tmp_path), "model") for p in saves)
    steps = model.calls
    assert steps[:4] == ["zero_grad", "forward", "backward", "step"] and len(steps) % 4 == 0
    assert all(steps[i:i + 4] == steps[:4] for i in range(0, len(steps), 4))
    assert len(steps) // 4 == 12 * 2                                   # 12 pseudo-epochs of ckpt_step = 2 batches
    # an improvement that is NOT significant (> 0.996 x the last significant best) is saved but earns no patience
    es = engine.EarlyStopping(max_epoch=100, early_stop=True, improve_threshold=0.996, patient_increase=2.0)
    assert es.observe(10.0) == (True, True) and es.patience == 10
    for _ in range(5):
        es.observe(11.0)
    assert es.observe(9.99) == (True, False) and es.patience == 10 and es.best == 9.99 and es.significant_best == 10.0
    assert es.observe(9.0) == (True, True) and es.patience == 16 and es.significant_best == 9.0      # epoch 8 x 2.0
    assert not es.finished()
    # max_epoch ends training whatever the patience; early_stop=False ignores patience
    model = _ScriptedModel([3.0, 2.0, 1.0, 0.5])
    best = engine.train(model, _OneBatchFeed("Train", 2), _OneBatchFeed("Valid", 1, model), None,
                        _engine_cfg(tmp_path, max_epoch=3))
    assert best == 1.0 and model.calls.count("step") == 6
    es = engine.EarlyStopping(max_epoch=30, early_stop=False, improve_threshold=0.996, patient_increase=2.0)
    for _ in range(20):
        es.observe(1.0)
    assert not es.finished()


def test_engine_flush_valid_rebuilds_the_optimizer_and_forgets_the_best(tmp_path, monkeypatch):
    """laed/main.py:106-110: when a forward raises model.flush_valid (AE-ED / ST-ED at freeze_step) the engine forgets
    the best validation loss -- a worse loss after the switch is a new best and is saved -- clears the flag and asks
    the model for a new optimizer BEFORE that step's backward."""
    import torch
    from laed_amd import main as engine
    saves = []
    monkeypatch.setattr(torch, "save", lambda sd, path: saves.append(path))
    monkeypatch.setattr(engine, "summary", lambda *a, **k: "stub")
    model = _ScriptedModel([1.0, 5.0, 4.0], flush_at=3)                # third training forward = first of pseudo-epoch 2
    best = engine.train(model, _OneBatchFeed("Train", 2), _OneBatchFeed("Valid", 1, model), None,
                        _engine_cfg(tmp_path, max_epoch=3))
    assert model.optimizers == 2 and model.flush_valid is False
    assert best == 4.0 and len(saves) == 3                             # 1.0, then (after the flush) 5.0 and 4.0
    assert model.calls[8:12] == ["zero_grad", "forward", "backward", "step"]


def test_engine_preview_generation_defaults_to_generate_with_an_evaluator(tmp_path, monkeypatch):
    """laed/main.py:82-83,132: `gen` defaults to `generate` and runs at every pseudo-epoch; here the default applies when
    an evaluator is passed (without one there is nothing to hand to it and the preview is skipped)."""
    import torch
    from laed_amd import main as engine
    monkeypatch.setattr(torch, "save", lambda sd, path: None)
    monkeypatch.setattr(engine, "summary", lambda *a, **k: "stub")
    seen = []
    monkeypatch.setattr(engine, "generate", lambda model, feed, config, evaluator, num_batch=1: seen.append((evaluator, num_batch)))
    model = _ScriptedModel([2.0, 1.0])
    engine.train(model, _OneBatchFeed("Train", 2), _OneBatchFeed("Valid", 1, model), "test-feed",
                 _engine_cfg(tmp_path, max_epoch=2), evaluator="EV")
    assert seen == [("EV", 1), ("EV", 1)]
    seen[:] = []
    model = _ScriptedModel([2.0, 1.0])
    engine.train(model, _OneBatchFeed("Train", 2), _OneBatchFeed("Valid", 1, model), None, _engine_cfg(tmp_path, max_epoch=2))
    assert seen == []
    lm = engine.LossManager()
    lm.add_loss({"nll": torch.tensor(2.0), "flag": True, "none": None, "rows": torch.zeros(3)})
    lm.add_loss({"nll": torch.tensor(4.0)})
    assert lm.pprint("x") == "x nll 3.000 PPL %.3f" % np.exp(3.0)
    assert lm.pprint("x", window=1, prefix="p") == "p: x nll 4.000 PPL %.3f" % np.exp(4.0)
    assert list(lm.losses["nll"]) == [2.0, 4.0] and len(lm.losses["nll"]) == 2


def test_projection_chunks_never_hold_a_whole_big_logits_tensor():
    """North-star item 2 at every config: once the [N, V] logits tensor would exceed 512 MiB, a training step's workspace
    tile holds at most half of it (the 8-GPU point's per-rank N = 19968 at V = 50000 -> two 9984-row chunks), while
    small problems keep one chunk."""
    from laed_amd import ops
    for N, V, H in ((19968, 50000, 1024), (159744, 50000, 1024), (40000, 50000, 1024), (79 * 512, 50000, 512)):
        rows = ops.proj_chunk_rows(N, V, H)
        assert rows <= (N + 1) // 2 + 127, (N, rows)
        assert rows * ops.proj_ldl(V) * 4 <= (8 << 30) + (1 << 20)
    assert ops.proj_chunk_rows(19968, 50000, 1024) == 9984
    assert ops.proj_chunk_rows(1170, 10002, 512) == 1170 and ops.proj_chunk_rows(2496, 50000, 1024) == 2496


def test_persistent_recurrence_budget_streams_divide_processes_forbid(monkeypatch):
    """The CU budget handed to laed_gru_{fwd,bwd}_cus: the whole device for one stream of one process, a share of it for
    recurrences that run side by side on several streams of this process, and ONE CU -- i.e. never the persistent
    kernels -- once another process shares the GPU (time-sliced spin grids: DESIGN.md section 5)."""
    from types import SimpleNamespace
    from laed_amd import ops
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: SimpleNamespace(multi_processor_count=256))
    monkeypatch.setitem(ops._CU_SHARE, "streams", 1)
    monkeypatch.setitem(ops._CU_SHARE, "processes", 1)
    assert ops._cu_budget(0) == 0
    with ops.concurrent_streams(2):
        assert ops._cu_budget(0) == 128
        with ops.concurrent_streams(4):                 # nested regions multiply: 2 x 4 concurrent recurrences
            assert ops._cu_budget(0) == 32
        assert ops._cu_budget(0) == 128
    assert ops._cu_budget(0) == 0
    ops.set_device_sharers(2)
    assert ops._cu_budget(0) == 1
    with ops.concurrent_streams(2):
        assert ops._cu_budget(0) == 1
    ops.set_device_sharers(1)
    assert ops._cu_budget(0) == 0
    # data parallelism: a quarter of the CUs stays free for the gradient reducer's RCCL kernels
    from laed_amd import parallel
    monkeypatch.setattr(parallel, "_ENABLED", True)
    monkeypatch.setattr(parallel, "_GROUP", object())
    assert ops._cu_budget(0) == 192
    with ops.concurrent_streams(2):
        assert ops._cu_budget(0) == 96


# ---- the same engine, PINNED against the reference's own train() / validate() / LossManager ---------------------------
def _drive_engine(spec, tmp_path, shape):
    """Run laed_amd.main.train on the scripted stand-in of tests/engine_stub.py; -> the same record
    tests/golden/make_goldens_engine.py took from the reference's train()."""
    import logging
    import torch
    from engine_stub import ScriptedFeed, ScriptedModel, engine_config
    from laed_amd import main as engine
    model = ScriptedModel(spec["valid"], spec["train"], spec["flush_at"], shape=shape)
    cfg = engine_config(tmp_path, **spec["cfg"])
    saves, patience, lines = [], [], []

    class Cap(logging.Handler):
        def emit(self, record):
            lines.append(record.getMessage())
    real_save, real_observe = torch.save, engine.EarlyStopping.observe

    def fake_save(obj, path):
        model.events.append("save")
        saves.append(os.path.relpath(path, str(tmp_path)))

    def observe(self, valid_loss):
        out = real_observe(self, valid_loss)
        if out[1]:
            patience.append(float(self.patience))
        return out
    cap, root = Cap(), logging.getLogger()
    old_level = root.level
    root.addHandler(cap)
    root.setLevel(logging.INFO)
    torch.save, engine.EarlyStopping.observe = fake_save, observe
    try:
        best = engine.train(model, ScriptedFeed("Train", spec["train_batches"], model),
                            ScriptedFeed("Valid", spec["valid_batches"], model), ScriptedFeed("Test", 1, model), cfg, "EVAL",
                            gen=lambda m, feed, config, evaluator, num_batch=1: m.events.append(
                                "gen:%s:%s:%s" % (feed.name, evaluator, num_batch)))
    finally:
        torch.save, engine.EarlyStopping.observe = real_save, real_observe
        root.removeHandler(cap)
        root.setLevel(old_level)
    return model, best, saves, patience, lines


@pytest.mark.parametrize("shape", [(1,), ()])
def test_engine_reproduces_the_reference_train_loop(tmp_path, shape):
    """`laed_amd.main.train` against tests/golden/engine.npz, the record of /root/reference/laed/main.py:81-181 driven
    UNMODIFIED over the same scripted scenarios (tests/golden/make_goldens_engine.py): every call on the model, the
    optimizer, the feeds (epoch_init with its shuffle flag), the preview generator and torch.save in the same order with
    the same batch counters; the same checkpoints; the same patience updates; the same best validation loss and stopping
    point; the same windowed "Train" report lines and validation report lines, character for character."""
    from engine_stub import scenarios
    from helpers import load_golden
    G = load_golden("engine")
    for name, spec in scenarios().items():
        model, best, saves, patience, lines = _drive_engine(spec, tmp_path, shape)
        assert model.events == [str(e) for e in G[name + "/events"]], name
        assert best == float(G[name + "/best"]), name
        assert patience == G[name + "/patience"].tolist(), name
        assert (saves or [""]) == [str(p) for p in G[name + "/save_paths"]], name
        assert sum(1 for e in model.events if e == "get_optimizer") == int(G[name + "/optimizers"]), name
        want_train = [str(l) for l in G[name + "/log_train"] if str(l)]
        assert [l for l in lines if ": Train " in l] == want_train, name
        got_valid = [l for l in lines if l.startswith("Valid ")]
        want_valid = [str(l) for l in G[name + "/log_valid"]]
        assert len(got_valid) == len(want_valid), name
        for g, w, v in zip(got_valid, want_valid, G[name + "/valid_loss"]):
            assert g.startswith(w + ";") and abs(float(g.split()[-1]) - float(v)) < 1e-6, (name, g, w)


def test_validate_and_lossmanager_reproduce_the_reference():
    """validate() on its own (call order, eval mode, the mean of model_sel_loss over the batches) and LossManager's
    report strings / avg_loss against the reference's (tests/golden/engine.npz)."""
    import torch
    from engine_stub import ScriptedFeed, ScriptedModel, engine_config
    from helpers import load_golden
    from laed_amd import main as engine
    G = load_golden("engine")
    model = ScriptedModel([2.0], shape=())
    got = engine.validate(model, ScriptedFeed("Valid", 3, model), engine_config("."), batch_cnt=7)
    assert abs(got - float(G["validate/return"])) < 1e-6 and model.training is False
    assert model.events == [str(e) for e in G["validate/events"]]
    lm = engine.LossManager()
    for a, b in ((2.0, 0.5), (4.0, 0.25), (3.0, 0.125)):
        lm.add_loss({"nll": torch.tensor(a), "flag": True, "absent": None, "reg_kl": torch.tensor([b])})
        lm.add_backward_loss(torch.tensor(a + b))
    assert lm.pprint("Train") == str(G["lossmanager/pprint_all"])
    assert lm.pprint("Train", window=2, prefix="1/2-(0.000)") == str(G["lossmanager/pprint_window2_prefix"])
    assert abs(lm.avg_loss() - float(G["lossmanager/avg_loss"])) < 1e-6
    lm.clear()
    assert lm.pprint("Train") == str(G["lossmanager/pprint_cleared"])
