"""Dialog-model (AE-ED / ST-ED) reporting utilities with the reference's entry points (laed/dialog_utils.py):
`generate`, `generate_with_adv`, `dump_latent`, `gen_with_cond`, `selective_generate` -- the callers stanford-ae.py /
stanford-skip.py hang behind `engine.train(..., gen=dialog_utils.generate_with_adv)` and run after training
(stanford-ae.py:131-165).

Host composition only: every forward pass is the model's own GEN / TEACH_FORCE pass on the HIP path; generated ids
leave the device in one copy per batch (`engine.sequences_of`).
"""
import logging
import pickle

import numpy as np
import torch

from . import main as engine
from . import ops
from .dataset.corpora import EOS, USR
from .enc2dec.decoders import GEN, TEACH_FORCE, DecoderRNN
from .utils import get_dekenize
from .utt_utils import _batches, _writer

logger = logging.getLogger()


def _context_text(model, de_tknize, ctx, b_id, skip_first, keep_empty):
    """The dialog context of row b_id as '<t>'-joined turns (ctx [B,C,T]; skip_first drops each turn's <s>)."""
    turns = []
    for i in range(ctx.shape[1]):
        turn = ctx[:, i, 1:] if skip_first else ctx[:, i, :]
        text, _ = engine.get_sent(model, de_tknize, turn, b_id)
        if text or keep_empty:
            turns.append(text)
    return '<t>'.join(turns)


def generate(model, data_feed, config, evaluator, num_batch=1, dest_f=None):
    """Decode every response from its context and score it with the evaluator (dialog_utils.py:19-76)."""
    model.eval()
    de_tknize = get_dekenize()
    evaluator.initialize()
    write = _writer(dest_f)
    for batch in _batches(data_feed, config, num_batch, batch_size=5):
        with torch.no_grad():
            outputs, labels = model(batch, mode=GEN, gen_type=config.gen_type)
        pred_labels = engine.sequences_of(outputs)
        true_labels = labels.cpu().numpy()
        ctx = batch.get('contexts')
        for b_id in range(pred_labels.shape[0]):
            pred_str, _ = engine.get_sent(model, de_tknize, pred_labels, b_id)
            true_str, _ = engine.get_sent(model, de_tknize, true_labels, b_id)
            evaluator.add_example(true_str, pred_str)
            write("Source: {}".format(_context_text(model, de_tknize, ctx, b_id, True, False)))
            write("Target: {}".format(true_str))
            write("Predict: {}\n".format(pred_str))
    report = evaluator.get_report(include_error=dest_f is not None)
    if dest_f is None:
        logger.info(report)
    else:
        dest_f.write(report)
    logger.info("Generation Done")


def generate_with_adv(model, data_feed, config, evaluator, num_batch=1, dest_f=None):
    """Policy accuracy and the 'adversarial' check of dialog_utils.py:79-168: the response generated from a latent
    action is cut at its first </s>, re-encoded by the recognition network q(z|x), and scored by how well q recovers
    the action that produced it.  Returns the policy accuracy (the engine's model-selection hook ignores it)."""
    eos_id = model.rev_vocab[EOS]
    usr_id = model.rev_vocab.get(USR)
    model.eval()
    logger.info("Generation with Adversarial: {} batches".format(data_feed.num_batch if num_batch is None else num_batch))
    adv_nll = corr_cnt = adv_cnt = corr_pi_cnt = 0.0
    sys_pi_cnt, usr_pi_cnt = 1e-18, 1e-18
    sys_corr_pi_cnt = usr_corr_pi_cnt = 0.0
    if num_batch is not None:
        gen_with_cond(model, data_feed, config, num_batch)

    data_feed.epoch_init(config, shuffle=False, verbose=False)
    while True:
        batch = data_feed.next_batch()
        if batch is None:
            break
        with torch.no_grad():
            outputs, labels = model(batch, mode=GEN, gen_type=config.gen_type)
            if DecoderRNN.KEY_LATENT not in outputs or DecoderRNN.KEY_RECOG_LATENT not in outputs:
                logger.info("No latent. Skip")
                return None
            y_ids = outputs[DecoderRNN.KEY_LATENT]
            qy_ids = outputs[DecoderRNN.KEY_RECOG_LATENT]
            log_py = outputs[DecoderRNN.KEY_POLICY]
            corr_flag = (torch.max(log_py, dim=1)[1] == qy_ids.view(-1)).cpu().numpy()
            corr_pi_cnt += float(np.sum(corr_flag))
            outs_np = np.asarray(batch['outputs'])
            n_rows = outs_np.shape[0]
            for b_id in range(n_rows):
                is_usr = usr_id in outs_np[b_id]
                hits = float(np.sum(corr_flag[b_id * config.y_size:(b_id + 1) * config.y_size]))
                if is_usr:
                    usr_pi_cnt += config.y_size
                    usr_corr_pi_cnt += hits
                else:
                    sys_pi_cnt += config.y_size
                    sys_corr_pi_cnt += hits

            # blank everything from the first </s> on (ids -> PAD = 0), then re-encode with q(z|x)
            seq = torch.cat([t.reshape(-1, 1) for t in outputs[DecoderRNN.KEY_SEQUENCE]], dim=1)
            ended = torch.cumsum((seq == eos_id).long(), dim=1) > 0
            norm_out_utts = seq * (~ended).long()
            qzx_results = model.qzx_forward(norm_out_utts)
            log_qy = ops.LogSoftmaxKFunction.apply(qzx_results.qy_logits)
            nll = -1.0 * log_qy.gather(1, y_ids.view(-1, 1))
            max_qy = torch.max(log_qy, dim=1)[1]
            corr_cnt += float((max_qy == y_ids.view(-1)).sum())
            adv_nll += float(nll.sum())
            adv_cnt += log_qy.size(0)

    avg_adv_nll = adv_nll / adv_cnt
    acc = corr_cnt / adv_cnt
    pi_acc = corr_pi_cnt / adv_cnt
    logger.info("Adversarial NLL {}, PPL {} Acc {} PI Acc {} Sys Acc {} Usr Acc {}"
                .format(avg_adv_nll, np.exp(avg_adv_nll), acc, pi_acc, sys_corr_pi_cnt / sys_pi_cnt,
                        usr_corr_pi_cnt / usr_pi_cnt))
    logger.info("Generation Done")
    return pi_acc


def collect_latent(model, data_feed, config, num_batch=1):
    """{'z': [(log_qy, log_py, y_ids, dec_init) per batch], 'labels': [...], 'contexts': [...]} (dialog_utils.py:171-210)."""
    model.eval()
    de_tknize = get_dekenize()
    all_zs, all_labels, all_ctxs = [], [], []
    for batch in _batches(data_feed, config, num_batch):
        with torch.no_grad():
            results = model(batch, mode=TEACH_FORCE, return_latent=True)
        labels, ctx = np.asarray(batch['outputs']), np.asarray(batch['contexts'])
        for b_id in range(labels.shape[0]):
            all_labels.append(engine.get_sent(model, de_tknize, labels, b_id)[0])
            all_ctxs.append(_context_text(model, de_tknize, ctx, b_id, False, True))
        all_zs.append((results.log_qy.detach().cpu().squeeze(0).numpy(), results.log_py.detach().cpu().squeeze(0).numpy(),
                       results.y_ids.cpu().numpy(), results.dec_init_state.detach().cpu().squeeze().numpy()))
    return {'z': all_zs, 'labels': all_labels, 'contexts': all_ctxs}


def dump_latent(model, data_feed, config, dest_f, num_batch=1):
    logger.info("Dumping")
    pickle.dump(collect_latent(model, data_feed, config, num_batch), dest_f)
    logger.info("Dumping Done")


def gen_with_cond(model, data_feed, config, num_batch=1, dest_f=None):
    """For every context: sample_n latent actions from the policy, each decoded greedily ('Sample Z') and by sampling
    words ('Sample W'), printed with their codes (dialog_utils.py:213-281).  Row b + batch_size * n of the model's
    sample_n-fold output is sample n of context b."""
    model.eval()
    de_tknize = get_dekenize()
    write = _writer(dest_f)
    sample_n = 5
    for batch in _batches(data_feed, config, num_batch, batch_size=3):
        ctx = batch.get('contexts')
        with torch.no_grad():
            sample_outputs, _ = model(batch, mode=GEN, gen_type="sample", sample_n=sample_n)
            greedy_outputs, labels = model(batch, mode=GEN, gen_type="greedy", sample_n=sample_n)
        sample_labels, greedy_labels = engine.sequences_of(sample_outputs), engine.sequences_of(greedy_outputs)
        greedy_y_ids = greedy_outputs[DecoderRNN.KEY_LATENT].cpu().numpy().reshape(greedy_labels.shape[0], -1)
        sample_y_ids = sample_outputs[DecoderRNN.KEY_LATENT].cpu().numpy().reshape(sample_labels.shape[0], -1)
        true_labels = labels.cpu().numpy()
        rows = true_labels.shape[0]
        for b_id in range(rows):
            write("Source: {}".format(_context_text(model, de_tknize, ctx, b_id, False, False)))
            write("Target: {}".format(engine.get_sent(model, de_tknize, true_labels, b_id)[0]))
            for tag, ids, codes in (("Sample Z", greedy_labels, greedy_y_ids), ("Sample W", sample_labels, sample_y_ids)):
                for n_id in range(sample_n):
                    r = b_id + rows * n_id
                    write("{} ({}): {}".format(tag, " ".join(map(str, codes[r])), engine.get_sent(model, de_tknize, ids, r)[0]))
            write('\n')
    logger.info("Generation Done\n")


def selective_generate(model, data_feed, config, selected_clusters):
    """Responses whose latent code ('-'-joined y ids) is one of the selected clusters' codes (dialog_utils.py:284-331)."""
    model.eval()
    de_tknize = get_dekenize()
    codes = set(d['code'] for d in selected_clusters)
    data = []
    total_cnt = in_cnt = 0.0
    for batch in _batches(data_feed, config, None):
        with torch.no_grad():
            outputs, labels = model(batch, mode=GEN, gen_type=config.gen_type)
        pred_labels = engine.sequences_of(outputs)
        true_labels = labels.cpu().numpy()
        y_ids = outputs[DecoderRNN.KEY_LATENT].cpu().numpy().reshape(-1, config.y_size)
        ctx = batch.get('contexts')
        for b_id in range(pred_labels.shape[0]):
            code = '-'.join(map(str, y_ids[b_id]))
            total_cnt += 1
            if code in codes:
                in_cnt += 1
                data.append({'context': _context_text(model, de_tknize, ctx, b_id, True, True),
                             'target': engine.get_sent(model, de_tknize, true_labels, b_id)[0],
                             'predict': engine.get_sent(model, de_tknize, pred_labels, b_id)[0], 'code': code})
    logger.info("In rate {}".format(in_cnt / max(total_cnt, 1.0)))
    return data
