"""BaseModel: np2var, backward, valid_loss, model_sel_loss, get_optimizer (mirror of
laed/models/model_bases.py:45-133; the Adam branch runs the fused HIP Adam kernel)."""
import logging

import numpy as np
import torch
import torch.nn as nn

from .. import optim, parallel
from ..utils import cast_type


logger = logging.getLogger()


def summary(model, show_weights=True, show_parameters=True):
    """One line per child module with its weight shapes and parameter count, then the total -- the model description
    the training loop logs before it starts (model_bases.py:12-42, main.py:93)."""
    lines, total = [model.__class__.__name__ + ' ('], 0
    for key, module in model._modules.items():
        text = summary(module) if isinstance(module, nn.Sequential) else repr(module)
        text = text.replace('\n', '\n  ')
        shapes = tuple(tuple(p.size()) for p in module.parameters())
        count = int(sum(p.numel() for p in module.parameters()))
        total += count
        line = '  (' + key + '): ' + text
        if show_weights:
            line += ', weights={}'.format(shapes)
        if show_parameters:
            line += ', parameters={}'.format(count)
        lines.append(line)
    return '\n'.join(lines) + '\n) Total Parameters={}'.format(total)


_ONES = {}


def _one_like(t, weight=1.0):
    """A cached 0-dim constant `weight` (1 by default) on t's device: the seed of a backward pass; never written."""
    key = (t.device, t.dtype, float(weight))
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.full((), float(weight), device=t.device, dtype=t.dtype)
    return one


class BaseModel(nn.Module):
    # Under data parallelism these loss terms are batch STATISTICS, formed from all-reduced sums and therefore already the
    # global-batch value on every rank; every other scalar term of a model's loss Pack is the rank's share of one
    # (parallel.global_terms, used by the engine's reports and model selection).
    DP_REPLICATED = ("reg_kl", "mi", "bpr", "b_pr")

    def __init__(self, config):
        super(BaseModel, self).__init__()
        self.use_gpu = config.use_gpu
        self.flush_valid = False
        self.config = config
        self.kl_w = 0.0

    def np2var(self, inputs, dtype):
        """numpy (or tensor) -> device tensor of the requested tag (model_bases.py:53-57)."""
        if inputs is None:
            return None
        if not torch.is_tensor(inputs):
            inputs = torch.from_numpy(np.ascontiguousarray(inputs))
        return cast_type(inputs, dtype, self.use_gpu)

    def forward(self, *input):
        raise NotImplementedError

    def backward(self, batch_cnt, loss):
        """model_bases.py:88-90: backward of valid_loss.  Where valid_loss is a plain sum (loss_terms says so) the terms are
        the roots themselves, each seeded with a cached constant 1: the same gradients, without the add kernel(s) that form
        the sum and the fill that seeds it -- three to four launches of a 1.5 ms step."""
        terms = self.loss_terms(loss, batch_cnt)
        if terms:
            pairs = [t if isinstance(t, tuple) else (t, 1.0) for t in terms]
            if all(torch.is_tensor(t) and t.is_cuda and t.dim() == 0 and t.requires_grad for t, _ in pairs):
                torch.autograd.backward([t for t, _ in pairs], [_one_like(t, w) for t, w in pairs])
                return
        total_loss = self.valid_loss(loss, batch_cnt)
        total_loss.backward()

    def loss_terms(self, loss, batch_cnt=None):
        """The terms whose (weighted) sum is valid_loss(loss, batch_cnt) -- tensors, or (tensor, weight) pairs -- or None
        where valid_loss does more than that on this step (a phase switch): backward() then goes through valid_loss."""
        return None

    def valid_loss(self, loss, batch_cnt=None):
        total_loss = 0.0
        for key, l in loss.items():
            if l is not None:
                total_loss += l
        return total_loss

    def model_sel_loss(self, loss, batch_cnt):
        return self.valid_loss(loss, batch_cnt)

    def get_optimizer(self, config):
        """model_bases.py:121-133 on the fused HIP optimizers: Adam over the parameters that require grad (the
        scripts' default, op='adam'), SGD / RMSprop over all parameters with config.momentum; under data parallelism
        the optimizer first joins the overlapped gradient all-reduce."""
        if config.op not in ('adam', 'sgd', 'rmsprop'):
            return None                                             # the reference falls through and returns None
        trainable = [p for p in self.parameters() if p.requires_grad]
        old = getattr(self, "_grad_reducer", None)
        if old is not None:                      # e.g. the optimizer rebuilt after flush_valid (main.py:106-110): the
            old.close()                          # previous reducer's hooks must not reduce every gradient a second time
        reducer = parallel.GradReducer(trainable) if parallel.group() is not None else None
        object.__setattr__(self, "_grad_reducer", reducer)
        if config.op == 'adam':
            logger.info("Use adam")
            return optim.FusedAdam(trainable, lr=config.init_lr, reducer=reducer)
        if config.op == 'sgd':
            logger.info("Use SGD")
            return optim.FusedSGD(self.parameters(), lr=config.init_lr, momentum=config.momentum, reducer=reducer)
        logger.info("RMSProp")
        return optim.FusedRMSprop(self.parameters(), lr=config.init_lr, momentum=config.momentum, reducer=reducer)
