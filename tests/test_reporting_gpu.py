"""The reporting utilities either side of the hot path (laed_amd.utt_utils / utt_skip_utils / main.generate-side
helpers / evaluators = laed/utt_utils.py, utt_skip_utils.py, main.py:52-78, evaluators.py) on the HIP models:
what they print must be exactly what the model's GEN / TEACH_FORCE passes return, decoded through the vocabulary."""
import io
import pickle
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

V, T = 40, 9


class Corpus(object):
    vocab = ['<pad>', '<unk>', '<s>', '</s>'] + ['w%d' % i for i in range(4, V)]
    rev_vocab = {w: i for i, w in enumerate(vocab)}


def _cfg(**kw):
    c = dict(embed_size=16, max_utt_len=T, num_layer=1, dropout=0.0, enc_cell_size=32, dec_cell_size=32, rnn_cell='gru',
             max_dec_len=T, use_attn=False, beam_size=4, utt_type='rnn', bi_enc_cell=False, attn_type='cat',
             fix_batch=False, y_size=2, k=3, use_gpu=True, avg_type='word', use_mutual=True, use_reg_kl=True,
             train_prior=False, init_lr=0.01, op='adam', batch_size=8, gen_type='greedy', backward_size=3)
    c.update(kw)
    return Namespace(**c)


def _ptb_feed(cfg, n=48, seed=0):
    """PTB-style loader over arithmetic token runs; every row carries a meta dict (its start token class)."""
    from laed_amd.dataset import data_loaders
    from laed_amd.utils import Pack
    rs = np.random.RandomState(seed)

    class Loader(data_loaders.PTBDataLoader):
        def _prepare_batch(self, selected_index):
            batch = super(Loader, self)._prepare_batch(selected_index)
            batch['metas'] = [self.data[i].meta for i in selected_index]
            return batch

    rows = []
    for _ in range(n):
        start, L = int(rs.randint(4, V)), int(rs.randint(3, T - 1))
        body = [(start + j - 4) % (V - 4) + 4 for j in range(L)]
        rows.append(Pack(utt=[2] + body + [3], meta={"start_class": start % 3, "tags": [start % 2, L % 2]}))
    return Loader("Test", rows, cfg)


def _decode(ids):
    words = []
    for i in ids:
        w = Corpus.vocab[int(i)]
        if w in ('</s>', '<pad>'):
            break
        words.append(w)
    return " ".join(words)


def test_get_sent_and_bleu_report():
    from laed_amd import main as engine
    from laed_amd.evaluators import BleuEvaluator
    from laed_amd.utils import get_dekenize

    class M(object):
        vocab = ['<pad>', '<unk>', '<s>', '</s>', 'it', "'s", 'fine', ',', 'thanks', '.', '</t>']
    data = np.array([[2, 4, 5, 6, 7, 8, 9, 3, 4, 4], [2, 6, 10, 4, 0, 0, 0, 0, 0, 0], [2, 6, 0, 6, 3, 0, 0, 0, 0, 0]])
    de = get_dekenize()
    assert engine.get_sent(M, de, data, 0) == ("<s> it's fine, thanks.", "")
    assert engine.get_sent(M, de, data, 1)[0] == "<s> fine </t>"               # </t> ends the turn and is kept
    assert engine.get_sent(M, de, data, 2)[0] == "<s> fine"                    # PAD stops the row
    assert engine.get_sent(M, de, data, 2, stop_pad=False)[0] == "<s> fine fine"
    ev = BleuEvaluator("unit")
    for _ in range(3):
        ev.add_example("<s> a b it's fine , thanks . </s>", "<s> a b it's fine , thanks . </s>")
    assert abs(ev.bleu() - 1.0) < 1e-12 and "Domain: default BLEU 1.000000" in ev.get_report()
    ev.add_example("<s> x y completely different words here </s>", "<s> x y nothing in common at all </s>", domain="other")
    assert 0.0 <= ev.bleu("other") < 0.2 and "Domain: other" in ev.get_report()


def test_divae_reporting_utilities():
    from laed_amd import main as engine
    from laed_amd import utt_utils
    from laed_amd.enc2dec.decoders import GEN, TEACH_FORCE
    from laed_amd.evaluators import BleuEvaluator
    from laed_amd.models.sent_models import DiVAE
    cfg = _cfg()
    torch.manual_seed(0)
    model = DiVAE(Corpus(), cfg).cuda()
    feed = _ptb_feed(cfg)
    # fixed noise, so that the direct model calls below see what the utilities saw
    model.cat_connector.sample_uniform = lambda logits: torch.full(logits.shape, 0.5, device=logits.device)

    # generate: every printed pair = the model's own greedy ids, decoded
    out = io.StringIO()
    ev = BleuEvaluator("unit")
    utt_utils.generate(model, feed, cfg, ev, num_batch=None, dest_f=out)
    text = out.getvalue()
    assert text.count("Target: ") == 48 and text.count("Predict: ") == 48 and "BLEU" in text
    assert cfg.batch_size == 8
    feed.epoch_init(cfg, shuffle=False, verbose=False)
    batch = feed.next_batch()
    model.eval()
    with torch.no_grad():
        outputs, labels = model(batch, mode=GEN, gen_type='greedy')
    pred = engine.sequences_of(outputs)
    assert pred.shape == (8, T)
    lines = text.split("\n")
    assert lines[0] == "Target: " + _decode(labels[0].cpu().numpy())
    assert lines[1] == "Predict: " + _decode(pred[0])
    # preview mode: 3 rows per batch, one batch
    out = io.StringIO()
    utt_utils.generate(model, feed, cfg, ev, num_batch=1, dest_f=out)
    assert out.getvalue().count("Target: ") == 3 and cfg.batch_size == 8

    # dump_latent: the pickle holds the TEACH_FORCE latents of every batch
    buf = io.BytesIO()
    utt_utils.dump_latent(model, feed, cfg, buf, num_batch=None)
    dump = pickle.loads(buf.getvalue())                      # a file this test wrote itself
    assert len(dump['z']) == 6 and len(dump['labels']) == 48 and len(dump['metas']) == 48
    with torch.no_grad():
        res = model(batch, mode=TEACH_FORCE, return_latent=True)
    lq, di, yi = dump['z'][0]
    assert lq.shape == (8 * cfg.y_size, cfg.k) and di.shape == (8, 32) and yi.shape == (8, cfg.y_size)
    assert np.array_equal(yi, res.y_ids.cpu().numpy()) and np.allclose(lq, res.log_qy.cpu().numpy(), atol=1e-6)
    assert dump['labels'][0] == _decode(batch.outputs[0]) and dump['metas'][0] == batch.metas[0]

    # sweep: Start / Predict (code) / End blocks, codes as the model reports them
    out = io.StringIO()
    utt_utils.sweep(model, feed, cfg, num_batch=2, dest_f=out)
    text = out.getvalue()
    assert text.count("Start: ") == 2 and text.count("End: ") == 2 and "Predict (" in text

    # latent_cluster: 3^2 = 9 states; the clusters partition the utterances they keep
    clusters = utt_utils.latent_cluster(model, feed, cfg, num_batch=None, rng=np.random.RandomState(0))
    assert clusters is not None
    for c in clusters:
        assert len(c['code'].split('-')) == cfg.y_size and 1 <= len(c['examples']) <= 5
    assert utt_utils.latent_cluster(model, feed, _cfg(y_size=20, k=10), num_batch=None) is None

    # find_mi: homogeneity of each meta attribute given the code
    scores = utt_utils.find_mi(model, feed, cfg)
    assert set(k for k, _ in scores) == {"start_class", "tags"}
    assert all(0.0 <= v <= 1.0 + 1e-9 for v in scores.values()) and len(scores) == 2 * (1 + cfg.y_size)


def test_divst_generate():
    from laed_amd import utt_skip_utils
    from laed_amd.dataset import data_loaders
    from laed_amd.evaluators import BleuEvaluator
    from laed_amd.models.sent_models import DiVST
    from test_engine_gpu import _toy_dialogs
    cfg = _cfg(max_utt_len=10, max_dec_len=10)
    torch.manual_seed(1)
    model = DiVST(Corpus(), cfg).cuda()
    feed = data_loaders.DailyDialogSkipLoader("Test", _toy_dialogs(np.random.RandomState(3), 6, V, 10), cfg)
    out = io.StringIO()
    utt_skip_utils.generate(model, feed, cfg, BleuEvaluator("unit"), num_batch=2, dest_f=out)
    text = out.getvalue()
    assert text.count("Response: ") == 10 and text.count("Prev Predict: ") == 10 and text.count("Next Target: ") == 10
    assert "BLEU" in text and cfg.batch_size == 8


def test_dialog_reporting_utilities():
    """laed_amd.dialog_utils (= laed/dialog_utils.py) on a small AE-ED: the printed responses are the model's own GEN
    output, the policy accuracy is recomputed from a direct pass, the latent dump holds the TEACH_FORCE latents."""
    from laed_amd import dialog_utils
    from laed_amd import main as engine
    from laed_amd.dataset import data_loaders
    from laed_amd.enc2dec.decoders import GEN, TEACH_FORCE, DecoderRNN
    from laed_amd.evaluators import BleuEvaluator
    from laed_amd.models.dialog_models import AeED
    from laed_amd.utils import Pack

    class DCorpus(object):
        vocab = ['<pad>', '<unk>', '<s>', '</s>', '<sys>', '<usr>'] + ['w%d' % i for i in range(6, V)]
        rev_vocab = {w: i for i, w in enumerate(vocab)}

    Td = 10
    rs = np.random.RandomState(5)
    dialogs = []
    for _ in range(8):
        turns = [Pack(utt=[2, 4, 6, 3], speaker=0, meta=None)]
        start = int(rs.randint(6, V))
        for t in range(int(rs.randint(4, 7))):
            L = int(rs.randint(3, Td - 3))
            body = [(start + t + j - 6) % (V - 6) + 6 for j in range(L)]
            turns.append(Pack(utt=[2, 4 + t % 2] + body + [3], speaker=t % 2, meta=None))
        dialogs.append(turns)
    cfg = _cfg(max_utt_len=Td, max_dec_len=Td, utt_type='attn_rnn', utt_cell_size=8, ctx_cell_size=32, use_attribute=True,
               greedy_q=True, freeze_step=1000, batch_size=4)
    torch.manual_seed(2)
    model = AeED(DCorpus(), cfg).cuda()
    model.cat_connector.sample_uniform = lambda logits: torch.full(logits.shape, 0.5, device=logits.device)
    feed = data_loaders.SMDDataLoader("Test", dialogs, cfg)
    feed.epoch_init(cfg, shuffle=False, verbose=False)
    n_batches, n_rows = feed.num_batch, feed.num_batch * cfg.batch_size

    def first_batch():
        feed.epoch_init(cfg, shuffle=False, verbose=False)
        return feed.next_batch()

    # generate: Source / Target / Predict per response, Predict = the model's greedy ids decoded
    out = io.StringIO()
    ev = BleuEvaluator("unit")
    dialog_utils.generate(model, feed, cfg, ev, num_batch=None, dest_f=out)
    text = out.getvalue()
    assert text.count("Source: ") == n_rows and text.count("Predict: ") == n_rows and "BLEU" in text
    assert len(ev.domain_hyps['default']) == n_rows and cfg.batch_size == 4
    batch = first_batch()
    model.eval()
    with torch.no_grad():
        outputs, labels = model(batch, mode=GEN, gen_type='greedy')
    lines = text.split("\n")
    assert lines[1] == "Target: " + _decode_d(DCorpus, labels[0].cpu().numpy())
    assert lines[2] == "Predict: " + _decode_d(DCorpus, engine.sequences_of(outputs)[0])
    assert lines[0] == "Source: <sys> w6"                         # the first response follows the opening turn only
    assert any(l.startswith("Source: ") and "<t>" in l for l in lines)          # later ones carry several turns

    # generate_with_adv: policy accuracy over the whole feed, recomputed from direct passes
    hits = rows = 0
    feed.epoch_init(cfg, shuffle=False, verbose=False)
    while True:
        b = feed.next_batch()
        if b is None:
            break
        with torch.no_grad():
            o, _ = model(b, mode=GEN, gen_type='greedy')
        hits += int((torch.max(o[DecoderRNN.KEY_POLICY], dim=1)[1] == o[DecoderRNN.KEY_RECOG_LATENT].view(-1)).sum())
        rows += o[DecoderRNN.KEY_POLICY].shape[0]
    pi_acc = dialog_utils.generate_with_adv(model, feed, cfg, None, num_batch=None)
    assert abs(pi_acc - hits / float(rows)) < 1e-12 and rows == n_rows * cfg.y_size

    # dump_latent
    buf = io.BytesIO()
    dialog_utils.dump_latent(model, feed, cfg, buf, num_batch=None)
    dump = pickle.loads(buf.getvalue())                          # a file this test wrote itself
    assert len(dump['z']) == n_batches and len(dump['labels']) == n_rows and len(dump['contexts']) == n_rows
    with torch.no_grad():
        res = model(batch, mode=TEACH_FORCE, return_latent=True)
    lq, lp, yi, di = dump['z'][0]
    assert lq.shape == (4 * cfg.y_size, cfg.k) and lp.shape == lq.shape and di.shape == (4, 32)
    assert np.array_equal(yi, res.y_ids.cpu().numpy()) and np.allclose(lp, res.log_py.cpu().numpy(), atol=1e-6)

    # gen_with_cond: 5 policy samples per context, decoded greedily (Z) and by sampling words (W)
    out = io.StringIO()
    dialog_utils.gen_with_cond(model, feed, cfg, num_batch=2, dest_f=out)
    text = out.getvalue()
    assert text.count("Source: ") == 6 and text.count("Sample Z (") == 30 and text.count("Sample W (") == 30
    assert cfg.batch_size == 4

    # selective_generate: exactly the responses whose code is selected
    codes = ['-'.join(map(str, r)) for r in outputs[DecoderRNN.KEY_LATENT].cpu().numpy().reshape(-1, cfg.y_size)]
    data = dialog_utils.selective_generate(model, feed, cfg, [{'code': codes[0]}])
    assert len(data) >= 1 and all(d['code'] == codes[0] for d in data)
    assert data[0]['predict'] == _decode_d(DCorpus, engine.sequences_of(outputs)[codes.index(codes[0])])


def _decode_d(corpus, ids):
    words = []
    for i in ids:
        w = corpus.vocab[int(i)]
        if w in ('</s>', '<pad>'):
            break
        words.append(w)
    return " ".join(words)
