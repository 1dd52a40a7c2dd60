"""Train / validate loop with the reference's control flow (mirror of laed/main.py:17-181).

Same step order (zero_grad -> forward(TEACH_FORCE) -> [flush_valid] -> backward -> step), ckpt_step pseudo-epochs,
patience rule, best-valid checkpoint (`torch.save(model.state_dict())`, reference-compatible keys) and LossManager
windows/PPL.  Differences: loss scalars stay on the device until a print / checkpoint step reads them (the reference's
per-step `val.data[0]` is a device synchronisation per loss term and no longer runs on 0-dim tensors anyway),
validation runs under torch.no_grad(), and text generation previews are left to the caller
(`gen` callback) since tokenisers / BLEU are outside the hot-path scope.
"""
import logging
import os
from collections import defaultdict

import numpy as np
import torch

from .enc2dec.decoders import TEACH_FORCE

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
