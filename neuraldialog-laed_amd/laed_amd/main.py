"""Train / validate loop with the reference's control flow (mirror of laed/main.py:17-181).

Same step order (zero_grad -> forward(TEACH_FORCE) -> [flush_valid] -> backward -> step), ckpt_step pseudo-epochs,
patience rule, best-valid checkpoint (`torch.save(model.state_dict())`, reference-compatible keys) and LossManager
windows/PPL.  Differences: loss scalars stay on the device until a print / checkpoint step reads them (the reference's
per-step `val.data[0]` is a device synchronisation per loss term and no longer runs on 0-dim tensors anyway),
validation runs under torch.no_grad().  `get_sent` / `generate` (laed/main.py:52-78,184-243) turn greedy / sampled id
matrices back into text for the `gen` callback and the BLEU evaluator.
"""
import logging
import os
from collections import defaultdict

import numpy as np
import torch

from .dataset.corpora import EOS, EOT, PAD
from .enc2dec.decoders import GEN, TEACH_FORCE, DecoderRNN
from .models.model_bases import summary
from .utils import get_dekenize

logger = logging.getLogger()


class _LazyScalars(list):
    """List of loss scalars that keeps device tensors as they are appended and copies them to the host in ONE transfer
    when somebody reads the list.  The reference reads every loss term with `val.data[0]` per step (main.py:25), i.e.
    several device synchronisations per step; here the train loop never waits for the GPU between print steps."""

    def __init__(self):
        super(_LazyScalars, self).__init__()
        self._pending = []

    def append(self, val):
        if torch.is_tensor(val):
            self._pending.append(val.detach().reshape(1))
        else:
            self._flush()
            super(_LazyScalars, self).append(float(val))

    def _flush(self):
        if self._pending:
            vals, self._pending = self._pending, []
            super(_LazyScalars, self).extend(torch.cat(vals).float().cpu().tolist())

    def __len__(self):
        self._flush()
        return super(_LazyScalars, self).__len__()

    def __iter__(self):
        self._flush()
        return super(_LazyScalars, self).__iter__()

    def __getitem__(self, i):
        self._flush()
        return super(_LazyScalars, self).__getitem__(i)


class LossManager(object):
    def __init__(self):
        self.losses = defaultdict(_LazyScalars)
        self.backward_losses = _LazyScalars()

    def add_loss(self, loss):
        for key, val in loss.items():
            if val is not None and type(val) is not bool and torch.is_tensor(val) and val.dim() == 0:
                self.losses[key].append(val)

    def add_backward_loss(self, loss):
        self.backward_losses.append(loss)

    def clear(self):
        self.losses = defaultdict(_LazyScalars)
        self.backward_losses = _LazyScalars()

    def pprint(self, name, window=None, prefix=None):
        str_losses = []
        for key, loss in self.losses.items():
            if loss is None:
                continue
            loss = list(loss)
            avg_loss = np.average(loss) if window is None else np.average(loss[-window:])
            str_losses.append("{} {:.3f}".format(key, avg_loss))
            if 'nll' in key:
                str_losses.append("PPL {:.3f}".format(np.exp(avg_loss)))
        if prefix:
            return "{}: {} {}".format(prefix, name, " ".join(str_losses))
        return "{} {}".format(name, " ".join(str_losses))

    def avg_loss(self):
        return np.mean(list(self.backward_losses))


def train(model, train_feed, valid_feed, test_feed, config, evaluator=None, gen=None):
    """laed/main.py:81-162.  Returns the best validation loss."""
    patience = 10
    valid_loss_threshold = np.inf
    best_valid_loss = np.inf
    batch_cnt = 0
    optimizer = model.get_optimizer(config)
    done_epoch = 0
    train_loss = LossManager()
    model.train()
    logger.info(summary(model, show_weights=False))
    logger.info("**** Training Begins ****")
    logger.info("**** Epoch 0/{} ****".format(config.max_epoch))
    while True:
        train_feed.epoch_init(config, verbose=done_epoch == 0, shuffle=True)
        while True:
            batch = train_feed.next_batch()
            if batch is None:
                break
            optimizer.zero_grad()
            loss = model(batch, mode=TEACH_FORCE)
            if model.flush_valid:
                logger.info("Flush previous valid loss")
                best_valid_loss = np.inf
                model.flush_valid = False
                optimizer = model.get_optimizer(config)
            model.backward(batch_cnt, loss)
            optimizer.step()
            batch_cnt += 1
            train_loss.add_loss(loss)
            if batch_cnt % config.print_step == 0:
                logger.info(train_loss.pprint("Train", window=config.print_step,
                                              prefix="{}/{}-({:.3f})".format(batch_cnt % config.ckpt_step,
                                                                             config.ckpt_step, model.kl_w)))
            if batch_cnt % config.ckpt_step == 0:
                logger.info("\n=== Evaluating Model ===")
                logger.info(train_loss.pprint("Train"))
                done_epoch += 1
                valid_loss = validate(model, valid_feed, config, batch_cnt)
                if gen is not None:
                    gen(model, test_feed, config, evaluator, num_batch=getattr(config, 'preview_batch_num', 1))
                if valid_loss < best_valid_loss:
                    if valid_loss <= valid_loss_threshold * config.improve_threshold:
                        patience = max(patience, done_epoch * config.patient_increase)
                        valid_loss_threshold = valid_loss
                        logger.info("Update patience to {}".format(patience))
                    if config.save_model:
                        logger.info("Model Saved.")
                        torch.save(model.state_dict(), os.path.join(config.session_dir, "model"))
                    best_valid_loss = valid_loss
                if done_epoch >= config.max_epoch or (config.early_stop and patience <= done_epoch):
                    if done_epoch < config.max_epoch:
                        logger.info("!!Early stop due to run out of patience!!")
                    logger.info("Best validation loss %f" % best_valid_loss)
                    return best_valid_loss
                model.train()
                train_loss.clear()
                logger.info("\n**** Epcoch {}/{} ****".format(done_epoch, config.max_epoch))


def validate(model, valid_feed, config, batch_cnt=None):
    """laed/main.py:165-181 (noise is still drawn in eval mode, as in the reference)."""
    model.eval()
    valid_feed.epoch_init(config, shuffle=False, verbose=True)
    losses = LossManager()
    with torch.no_grad():
        while True:
            batch = valid_feed.next_batch()
            if batch is None:
                break
            loss = model(batch, mode=TEACH_FORCE)
            losses.add_loss(loss)
            losses.add_backward_loss(model.model_sel_loss(loss, batch_cnt))
    valid_loss = losses.avg_loss()
    logger.info(losses.pprint(valid_feed.name))
    logger.info("Total valid loss {}".format(valid_loss))
    return valid_loss


def get_sent(model, de_tknize, data, b_id, attn=None, attn_ctx=None, stop_eos=True, stop_pad=True):
    """Row b_id of an id matrix as text (laed/main.py:52-78): stops at </s> / </t> (keeping </t>) and at PAD; with
    attention scores, also lists the context words that got more than 0.1 of the mass."""
    words, attn_words = [], []
    has_attn = attn is not None and attn_ctx is not None
    for t_id in range(data.shape[1]):
        w = model.vocab[int(data[b_id, t_id])]
        if has_attn:
            a_val = np.max(attn[b_id, t_id])
            if a_val > 0.1:
                attn_words.append("{}({})".format(model.vocab[int(attn_ctx[b_id, np.argmax(attn[b_id, t_id])])], a_val))
        if (stop_eos and w in (EOS, EOT)) or (stop_pad and w == PAD):
            if w == EOT:
                words.append(w)
            break
        if w != PAD:
            words.append(w)
    att = "Attention: {}".format(" ".join(attn_words)) if attn_words else ""
    try:
        return de_tknize(words), att if has_attn else ""
    except Exception:                       # a detokeniser that rejects the tokens: plain join, as the reference
        return " ".join(words), ""


def sequences_of(ret_dict):
    """[B, max_len] int matrix of generated ids from a decoder's ret_dict (list of [B,1] id tensors per step)."""
    seq = ret_dict[DecoderRNN.KEY_SEQUENCE]
    if len(seq) == 0:
        return np.zeros((0, 0), dtype=int)
    return torch.cat([t.reshape(-1, 1) for t in seq], dim=1).cpu().numpy().astype(int)      # one device->host copy


def generate(model, data_feed, config, evaluator, num_batch=1, dest_f=None):
    """Response generation over a dialog feed with per-domain BLEU (laed/main.py:184-243)."""
    model.eval()
    de_tknize = get_dekenize()

    def write(msg):
        if msg is None or msg == '':
            return
        if dest_f is None:
            logger.info(msg)
        else:
            dest_f.write(msg + '\n')

    data_feed.epoch_init(config, shuffle=num_batch is not None, verbose=False)
    evaluator.initialize()
    logger.info("Generation: {} batches".format(data_feed.num_batch if num_batch is None else num_batch))
    while True:
        batch = data_feed.next_batch()
        if batch is None or (num_batch is not None and data_feed.ptr > num_batch):
            break
        with torch.no_grad():
            outputs, labels = model(batch, mode=GEN, gen_type=config.gen_type)
        pred_labels = sequences_of(outputs)
        true_labels = labels.cpu().numpy()
        pred_attns = None
        if (getattr(config, 'use_attn', False) or getattr(config, 'use_ptr', False)) and DecoderRNN.KEY_ATTN_SCORE in outputs:
            pred_attns = np.array([t.cpu().numpy() for t in outputs[DecoderRNN.KEY_ATTN_SCORE]], dtype=float).squeeze(2).swapaxes(0, 1)
        ctx, ctx_len = batch.get('contexts'), batch.get('context_lens')
        domains = batch.get('domains')
        for b_id in range(pred_labels.shape[0]):
            pred_str, attn = get_sent(model, de_tknize, pred_labels, b_id, attn=pred_attns)
            true_str, _ = get_sent(model, de_tknize, true_labels, b_id)
            prev_ctx = ""
            if ctx is not None:
                ctx_str, _ = get_sent(model, de_tknize, ctx[:, ctx_len[b_id] - 1, :], b_id)
                prev_ctx = "Source: {}".format(ctx_str)
            domain = domains[b_id] if domains is not None else 'default'
            evaluator.add_example(true_str, pred_str, domain)
            if num_batch is None or num_batch <= 2:
                write(prev_ctx)
                write("{}:: True: {} ||| Pred: {}".format(domain, true_str, pred_str))
                if attn:
                    write("[[{}]]".format(attn))
    write(evaluator.get_report(include_error=dest_f is not None))
    logger.info("Generation Done")
