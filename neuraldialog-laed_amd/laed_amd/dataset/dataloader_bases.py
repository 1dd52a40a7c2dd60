"""Batch iteration shared by every loader (laed/dataset/dataloader_bases.py:6-64, Python 3).

Same public surface as the reference's DataLoader: epoch_init(config, shuffle, verbose) cuts `indexes` into
`data_size // batch_size` full batches (the remainder is dropped), next_batch() returns a Pack or None, pad_to()
truncates keeping the last token.  Shuffling works here (the reference shuffles a `range`, which Python 3 rejects):
plain loaders shuffle examples, fix_batch loaders keep their length-sorted batches together and shuffle batch order.
"""
import logging

import numpy as np


class DataLoader(object):
    logger = logging.getLogger()

    def __init__(self, name, fix_batch=True):
        self.name = name
        self.fix_batch = fix_batch
        self.batch_size = 0
        self.ptr = 0
        self.num_batch = None
        self.indexes = None
        self.data_size = None
        self.batch_indexes = None
        self.max_utt_size = None

    def _prepare_batch(self, selected_index):
        raise NotImplementedError("Have to override prepare batch")

    def epoch_init(self, config, shuffle=True, verbose=True):
        self.ptr = 0
        self.batch_size = config.batch_size
        self.num_batch = self.data_size // config.batch_size
        if verbose:
            self.logger.info("Number of left over sample %d" % (self.data_size - config.batch_size * self.num_batch))
        order = list(self.indexes)
        if shuffle and not self.fix_batch:
            np.random.shuffle(order)
            self.indexes = order
        bs = self.batch_size
        self.batch_indexes = [order[i * bs:(i + 1) * bs] for i in range(self.num_batch)]
        if shuffle and self.fix_batch:
            np.random.shuffle(self.batch_indexes)
        if verbose:
            self.logger.info("%s begins with %d batches" % (self.name, self.num_batch))

    def next_batch(self):
        if self.ptr >= self.num_batch:
            return None
        selected = self.batch_indexes[self.ptr]
        self.ptr += 1
        return self._prepare_batch(selected_index=selected)

    def pad_to(self, max_len, tokens, do_pad=True):
        """Clip to max_len keeping the final token (normally </s>); optionally right-pad with 0."""
        tokens = list(tokens)
        if len(tokens) >= max_len:
            return tokens[:max_len - 1] + tokens[-1:]
        if do_pad:
            return tokens + [0] * (max_len - len(tokens))
        return tokens
