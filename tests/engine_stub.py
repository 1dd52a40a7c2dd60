"""A scripted stand-in for a model, its optimizer and its data feeds, for driving a training ENGINE without a GPU.

Used twice with the same scenarios: by tests/golden/make_goldens_engine.py, which drives the REFERENCE's own
`laed.main.train` / `validate` (/root/reference/laed/main.py:81-181, unmodified) and records what it does, and by
tests/test_host_logic_cpu.py, which drives `laed_amd.main.train` / `validate` and must reproduce that record.

Every call the engine makes on the model, the optimizer, the feeds, the preview generator and torch.save is appended to
`model.events` as a short string.  Loss terms are 1-element tensors of shape `shape`: (1,) for the reference
(`LossManager.add_loss` reads `val.data[0]`, which needs one dimension under torch 2.x), () or (1,) for this repo.
"""
import torch


class ScriptedModel(object):
    vocab = ['<pad>']

    def __init__(self, valid_losses, train_losses=(1.0,), flush_at=None, shape=(1,), kl_w=0.0):
        self.valid_losses = list(valid_losses)      # one per validation pass
        self.train_losses = list(train_losses)      # cycled over training steps
        self.flush_at = flush_at                    # 1-based training forward that raises flush_valid
        self.shape = tuple(shape)
        self.kl_w = kl_w
        self.flush_valid = False
        self.training = True
        self.events = []
        self.train_forwards = 0
        self.valid_passes = 0
        self._modules = {}                          # what the reference's summary() walks

    def _t(self, v):
        return torch.full(self.shape, float(v))

    def train(self):
        self.training = True
        self.events.append("train()")

    def eval(self):
        self.training = False
        self.valid_passes += 1
        self.events.append("eval()")

    def get_optimizer(self, config):
        self.events.append("get_optimizer")
        model = self

        class Opt(object):
            def zero_grad(self):
                model.events.append("zero_grad")

            def step(self):
                model.events.append("step")
        return Opt()

    def __call__(self, batch, mode=None):
        if self.training:
            self.train_forwards += 1
            self.events.append("forward")
            if self.flush_at is not None and self.train_forwards == self.flush_at:
                self.flush_valid = True
            v = self.train_losses[(self.train_forwards - 1) % len(self.train_losses)]
            return {"nll": self._t(v), "reg_kl": self._t(0.25 * v), "flag": True, "absent": None}
        self.events.append("valid_forward")
        v = self.valid_losses[min(self.valid_passes, len(self.valid_losses)) - 1] + 0.5 * batch["index"]
        return {"nll": self._t(v)}

    def backward(self, batch_cnt, loss):
        self.events.append("backward:%d" % batch_cnt)

    def model_sel_loss(self, loss, batch_cnt):
        self.events.append("sel_loss:%s" % batch_cnt)
        return loss["nll"]

    def state_dict(self):
        return {"marker": torch.zeros(1)}


class ScriptedFeed(object):
    def __init__(self, name, n, model):
        self.name, self.n, self.model, self.ptr, self.num_batch = name, n, model, 0, n

    def epoch_init(self, config, shuffle=True, verbose=True):
        self.ptr = 0
        self.model.events.append("epoch_init:%s:shuffle=%d" % (self.name, int(bool(shuffle))))

    def next_batch(self):
        if self.ptr >= self.n:
            return None
        self.ptr += 1
        return {"outputs": None, "index": self.ptr - 1}


def scenarios():
    """name -> dict(valid, train, flush_at, train_batches, valid_batches, cfg overrides)."""
    base = dict(train=(1.0,), flush_at=None, train_batches=2, valid_batches=1, cfg={})
    S = {}
    # six improving pseudo-epochs (patience 10 -> 12 at the sixth), then nothing improves: stop when 12 <= epochs done
    S["improve_then_stall"] = dict(base, valid=[5.0, 4.0, 3.0, 2.0, 1.5, 1.0] + [1.2] * 40)
    # max_epoch ends training whatever the patience
    S["max_epoch"] = dict(base, valid=[3.0, 2.0, 1.0, 0.5], cfg=dict(max_epoch=3))
    # flush_valid raised by the third training forward (first of pseudo-epoch 2): best loss forgotten, new optimizer
    S["flush_valid"] = dict(base, valid=[1.0, 5.0, 4.0], flush_at=3, cfg=dict(max_epoch=3))
    # an improvement that is not significant (> improve_threshold x the last significant best) is saved, earns no patience
    S["insignificant"] = dict(base, valid=[10.0, 11.0, 11.0, 11.0, 11.0, 11.0, 9.99, 9.0] + [9.5] * 40)
    # early_stop off: patience is ignored, max_epoch ends it
    S["no_early_stop"] = dict(base, valid=[2.0] + [3.0] * 40, cfg=dict(early_stop=False, max_epoch=14))
    # pseudo-epochs that straddle the feed's epochs (3 batches per epoch, ckpt_step 2), three validation batches whose
    # losses differ (the selection loss is their mean), windowed Train reports every step, no checkpoints
    S["straddle"] = dict(base, valid=[4.0, 3.5, 3.6, 3.0], train=(2.0, 1.5, 1.0, 0.5, 0.25), train_batches=3, valid_batches=3,
                         cfg=dict(max_epoch=4, print_step=1, save_model=False))
    # a loss that comes back equal to the best is not an improvement; patient_increase 3
    S["ties"] = dict(base, valid=[2.0, 2.0, 1.0, 1.0] + [1.0] * 40, cfg=dict(patient_increase=3.0, max_epoch=40))
    return S


def engine_config(session_dir, **kw):
    from argparse import Namespace
    cfg = dict(print_step=1000, ckpt_step=2, max_epoch=50, early_stop=True, improve_threshold=0.996,
               patient_increase=2.0, save_model=True, session_dir=str(session_dir), preview_batch_num=1)
    cfg.update(kw)
    return Namespace(**cfg)
