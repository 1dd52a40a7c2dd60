"""Training engine: the behaviour of laed/main.py:17-244 (the reference's train / validate / generate and LossManager),
written against that behaviour rather than its text.

What is kept, because callers and checkpoints depend on it:
  * the step order  zero_grad -> forward(TEACH_FORCE) -> [flush_valid: forget the best loss, rebuild the optimizer]
    -> backward -> optimizer.step  (laed/main.py:104-113);
  * a "pseudo-epoch" every `config.ckpt_step` batches: validate, preview generation, then the patience rule
    (laed/main.py:123-158): a new best validation loss is saved; one that also beats `improve_threshold` x the last
    significant best stretches patience to max(patience, epochs_done x patient_increase); training ends at `max_epoch`
    pseudo-epochs or, with `early_stop`, once patience <= epochs_done;
  * the checkpoint is `torch.save(model.state_dict(), <session_dir>/model)` with the reference's key names;
  * LossManager's interface (add_loss / add_backward_loss / clear / pprint / avg_loss) and PPL = exp(mean) for every
    key that contains "nll" (laed/main.py:17-49).
What differs: loss scalars stay on the device until a report reads them (the reference synchronises per loss term per
step with `val.data[0]`, which no longer runs on 0-dim tensors), validation runs under `torch.no_grad()`, and the
patience bookkeeping lives in `EarlyStopping`, which `tests/test_host_logic_cpu.py` drives through scripted
validation losses without a GPU.
New (the reference has no multi-GPU code): under data parallelism (`parallel.init()` called: one process per GPU) every rank
runs this same loop over the same global batches, keeps its own rows of each (`parallel.shard_batch`), and sees the same
GLOBAL loss values wherever a decision or a report depends on them -- validation reduces each batch's terms
(`parallel.global_terms`), the training reports reduce lazily when they are printed -- so patience, phase changes and
the stopping epoch are identical on every rank; rank 0 alone writes checkpoints (the preview generations run everywhere,
which keeps the ranks' random-number counters in step; only rank 0's logger is expected to print them).  The order of
each training pass is drawn from a per-pass seed so that nothing a single rank does in between can make the ranks disagree
about it.  `tests/test_y_dp_gpu.py::test_engine_train_data_parallel_equals_single_process`.
"""
import logging
import os

import numpy as np
import torch

from . import parallel
from .dataset.corpora import EOS, EOT, PAD
from .enc2dec.decoders import GEN, TEACH_FORCE, DecoderRNN
from .models.model_bases import summary
from .utils import get_dekenize

logger = logging.getLogger()


# ----------------------------------------------------------------------------------------------------------------
# loss bookkeeping
# ----------------------------------------------------------------------------------------------------------------
class _Series(object):
    """Scalars in arrival order.  Device tensors are parked as they come and fetched in ONE device->host copy when the
    series is read, so the step loop never waits for the GPU between reports."""

    def __init__(self):
        self._host = []
        self._parked = []

    def push(self, value):
        if torch.is_tensor(value):
            self._parked.append(value.detach().reshape(1))
        else:
            self.values()                       # keep arrival order
            self._host.append(float(value))

    def values(self):
        if self._parked:
            parked, self._parked = self._parked, []
            self._host.extend(torch.cat(parked).float().cpu().tolist())
        return self._host

    def mean(self, last=None):
        """fp32 mean, as the reference's np.average / np.mean over its list of fp32 loss scalars (laed/main.py:39,49):
        the early-stopping comparisons and the printed digits then agree with it exactly."""
        v = self.values()
        return float(np.asarray(v if last is None else v[-last:], dtype=np.float32).mean())

    # list protocol, for callers that index / iterate `manager.losses[key]` as the reference's plain lists
    def __len__(self):
        return len(self.values())

    def __iter__(self):
        return iter(self.values())

    def __getitem__(self, i):
        return self.values()[i]


class LossManager(object):
    """Per-key running record of the 0-dim loss terms of a Pack plus the scalar that drives model selection.

    `replicated` (data-parallel training only): the keys that are already global-batch values on every rank; every other
    key holds this rank's SHARES, and whenever the record is read all shares parked since the last read are summed over
    the ranks in one all-reduce (every rank reads at the same steps, so the collectives line up).  The step loop itself
    never communicates for the sake of a report."""

    def __init__(self, replicated=None):
        self._replicated = replicated
        self.clear()

    def _sync(self):
        if self._replicated is None or parallel.group() is None:
            return
        todo = [s for k, s in self.losses.items() if k not in self._replicated and s._parked]
        if not todo:
            return
        chunks = [torch.cat(s._parked).float() for s in todo]
        vec = parallel.all_reduce_(torch.cat(chunks), kind="report").cpu().tolist()
        at = 0
        for s, c in zip(todo, chunks):
            s._host.extend(vec[at:at + c.numel()])
            s._parked = []
            at += c.numel()

    def clear(self):
        self.losses = {}
        self.backward_losses = _Series()

    def add_loss(self, loss):
        for key, val in loss.items():
            if torch.is_tensor(val) and val.numel() == 1:   # 0-dim or 1-element terms; skips None, flags, per-row tensors
                self.losses.setdefault(key, _Series()).push(val)

    def add_backward_loss(self, loss):
        self.backward_losses.push(loss)

    def avg_loss(self):
        return self.backward_losses.mean()

    def pprint(self, name, window=None, prefix=None):
        self._sync()
        cells = []
        for key, series in self.losses.items():
            m = series.mean(window)
            cells.append("%s %.3f" % (key, m))
            if 'nll' in key:
                cells.append("PPL %.3f" % np.exp(np.float32(m)))      # fp32 exp, as the reference prints it
        head = name if not prefix else "%s: %s" % (prefix, name)
        return head + " " + " ".join(cells)         # an empty record prints "name " (trailing blank), as the reference does


class EarlyStopping(object):
    """The patience rule of laed/main.py:86-88,133-158 as a state object.

    observe(valid_loss) is called once per pseudo-epoch and returns (is_new_best, patience_was_extended); finished()
    then says whether training ends.  flush() is the `flush_valid` reset (laed/main.py:106-110): only the best loss is
    forgotten -- the significance threshold and the patience already earned stay."""

    def __init__(self, max_epoch, early_stop, improve_threshold, patient_increase, initial_patience=10):
        self.max_epoch = max_epoch
        self.early_stop = early_stop
        self.improve_threshold = improve_threshold
        self.patient_increase = patient_increase
        self.patience = initial_patience
        self.epochs_done = 0
        self.best = np.inf
        self.significant_best = np.inf

    def flush(self):
        self.best = np.inf

    def observe(self, valid_loss):
        self.epochs_done += 1
        if not valid_loss < self.best:
            return False, False
        extended = False
        if valid_loss <= self.significant_best * self.improve_threshold:
            self.patience = max(self.patience, self.epochs_done * self.patient_increase)
            self.significant_best = valid_loss
            extended = True
        self.best = valid_loss
        return True, extended

    def out_of_patience(self):
        return bool(self.early_stop) and self.patience <= self.epochs_done

    def finished(self):
        return self.epochs_done >= self.max_epoch or self.out_of_patience()


# ----------------------------------------------------------------------------------------------------------------
# train / validate
# ----------------------------------------------------------------------------------------------------------------
def _batches(feed):
    """The feed's batches; under data parallelism every rank walks the same global batches and keeps its own rows."""
    while True:
        batch = feed.next_batch()
        if batch is None:
            return
        yield parallel.shard_batch(batch)


def _save_checkpoint(model, config):
    path = os.path.join(config.session_dir, "model")
    torch.save(model.state_dict(), path)
    return path


def check_persistent_recurrences(model, where):
    """Raise if any bounded in-kernel wait of the persistent GRU kernels has timed out on this device.

    A wait that gives up poisons its outputs with NaN so that the grid drains (gru_small.hip); in a training run that
    reads as a loss that diverged.  The counter is sticky and costs one 4-byte device read, so the engine looks at it
    where the host synchronises anyway (each pseudo-epoch) and names the real cause: something the CU budget does not
    know about (another process on the GPU, a whole-device grid on a side stream) kept a persistent grid from becoming
    co-resident.  Models that do not live on a GPU (the scripted stand-ins of the tests) are skipped."""
    params = getattr(model, "parameters", None)
    first = next(iter(params()), None) if callable(params) else None
    if first is None or not first.is_cuda:
        return 0
    from . import ops
    with torch.cuda.device(first.device):
        n = ops.gru_persist_timeouts()
    if n:
        raise RuntimeError("%s: %d inter-workgroup wait(s) of the persistent GRU kernels timed out on %s -- their outputs "
                           "were poisoned with NaN. The persistent grids could not become co-resident: is another "
                           "process using this GPU (ops.set_device_sharers), or a whole-device kernel running on a side "
                           "stream? LAED_GRU_PERSIST=0 selects per-step launches." % (where, n, first.device))
    return n


def train(model, train_feed, valid_feed, test_feed, config, evaluator=None, gen=None):
    """Train until `max_epoch` pseudo-epochs or the patience rule stops it; returns the best validation loss.

    `gen(model, test_feed, config, evaluator, num_batch=...)` previews generations at every pseudo-epoch.  As in the
    reference (laed/main.py:82-83) it defaults to `generate` -- when there is an evaluator to hand to it; with neither,
    the preview is skipped."""
    if gen is None and evaluator is not None:
        gen = generate
    dp = parallel.group() is not None
    lead = parallel.rank() == 0
    stopper = EarlyStopping(config.max_epoch, config.early_stop, config.improve_threshold, config.patient_increase)
    optimizer = model.get_optimizer(config)
    seen = LossManager(replicated=getattr(model, "DP_REPLICATED", ()) if dp else None)
    step_no = 0
    passes = 0
    model.train()
    logger.info(summary(model, show_weights=False))
    logger.info("Training starts: up to %d pseudo-epochs of %d batches", config.max_epoch, config.ckpt_step)
    while True:
        if dp:          # every rank must draw the same order for this pass, whatever else has used numpy's generator since
            np.random.seed((int(getattr(config, "seed", 0)) + 7919 * passes) & 0x7FFFFFFF)
        passes += 1
        train_feed.epoch_init(config, verbose=stopper.epochs_done == 0, shuffle=True)
        for batch in _batches(train_feed):
            optimizer.zero_grad()
            loss = model(batch, mode=TEACH_FORCE)
            if model.flush_valid:               # the model switched training phase inside this forward
                logger.info("Training phase changed: forgetting the best validation loss, new optimizer")
                stopper.flush()
                model.flush_valid = False
                optimizer = model.get_optimizer(config)
            model.backward(step_no, loss)
            optimizer.step()
            step_no += 1
            seen.add_loss(loss)

            if step_no % config.print_step == 0:
                where = "%d/%d-(%.3f)" % (step_no % config.ckpt_step, config.ckpt_step, model.kl_w)
                logger.info(seen.pprint("Train", window=config.print_step, prefix=where))
            if step_no % config.ckpt_step != 0:
                continue

            # ---- pseudo-epoch boundary
            logger.info("Pseudo-epoch %d done; %s", stopper.epochs_done + 1, seen.pprint("Train"))
            valid_loss = validate(model, valid_feed, config, step_no)
            check_persistent_recurrences(model, "pseudo-epoch %d" % (stopper.epochs_done + 1))
            # data parallel: the preview runs on EVERY rank -- a GEN-mode forward exchanges nothing, the ranks' noise counters
            # stay in step, and the other ranks would only wait for rank 0 at the next step's exchange anyway
            if gen is not None:
                gen(model, test_feed, config, evaluator, num_batch=getattr(config, 'preview_batch_num', 1))
            is_best, extended = stopper.observe(valid_loss)
            if extended:
                logger.info("Significant improvement: patience is now %s pseudo-epochs", stopper.patience)
            if is_best and config.save_model and lead:
                logger.info("New best validation loss %f, saved to %s", valid_loss, _save_checkpoint(model, config))
            if stopper.finished():
                if stopper.epochs_done < config.max_epoch:
                    logger.info("Stopping early: patience %s used up after %d pseudo-epochs", stopper.patience,
                                stopper.epochs_done)
                logger.info("Best validation loss %f", stopper.best)
                return stopper.best
            model.train()
            seen.clear()
            logger.info("Pseudo-epoch %d/%d starts", stopper.epochs_done, config.max_epoch)


def validate(model, valid_feed, config, batch_cnt=None):
    """Mean model-selection loss over one pass of `valid_feed` in eval mode (laed/main.py:165-181).  The Gumbel noise
    is still drawn in eval mode, as in the reference; no tape is built."""
    model.eval()
    valid_feed.epoch_init(config, shuffle=False, verbose=True)
    seen = LossManager()
    replicated = getattr(model, "DP_REPLICATED", ())
    with torch.no_grad():
        for batch in _batches(valid_feed):
            # data parallel: the global-batch value of every term on every rank, so all ranks select and stop alike
            loss = parallel.global_terms(model(batch, mode=TEACH_FORCE), replicated)
            seen.add_loss(loss)
            seen.add_backward_loss(model.model_sel_loss(loss, batch_cnt))
    valid_loss = seen.avg_loss()
    logger.info("%s; selection loss %f", seen.pprint(valid_feed.name), valid_loss)
    return valid_loss


# ----------------------------------------------------------------------------------------------------------------
# ids -> text, preview generation
# ----------------------------------------------------------------------------------------------------------------
def get_sent(model, de_tknize, data, b_id, attn=None, attn_ctx=None, stop_eos=True, stop_pad=True):
    """(text, attention note) for row `b_id` of an id matrix (laed/main.py:52-78).  Reading stops at the first </s> or
    </t> (a </t> is kept in the text) when `stop_eos`, and at the first PAD when `stop_pad`; PAD is never printed.  With
    attention scores [B,T,C] and the attended ids, context words holding more than 0.1 of a step's mass are listed."""
    row = data[b_id]
    with_attn = attn is not None and attn_ctx is not None
    words, noted = [], []
    for t_id in range(len(row)):
        word = model.vocab[int(row[t_id])]
        if with_attn:
            weight = float(np.max(attn[b_id, t_id]))
            if weight > 0.1:
                src = model.vocab[int(attn_ctx[b_id, int(np.argmax(attn[b_id, t_id]))])]
                noted.append("%s(%s)" % (src, weight))
        ends_turn = stop_eos and word in (EOS, EOT)
        if ends_turn or (stop_pad and word == PAD):
            if word == EOT:
                words.append(word)
            break
        if word != PAD:
            words.append(word)
    note = "Attention: " + " ".join(noted) if (with_attn and noted) else ""
    try:
        return de_tknize(words), note
    except Exception:                           # a detokeniser that rejects the tokens: plain join, no note
        return " ".join(words), ""


def sequences_of(ret_dict):
    """[B, max_len] int matrix of generated ids from a decoder's ret_dict (a list of [B,1] id tensors per step): one
    device->host copy per batch instead of one per step."""
    steps = ret_dict[DecoderRNN.KEY_SEQUENCE]
    if not steps:
        return np.zeros((0, 0), dtype=int)
    return torch.cat([s.reshape(-1, 1) for s in steps], dim=1).cpu().numpy().astype(int)


def _attention_of(ret_dict, config):
    """[B,T,C] attention scores if the decoder produced any (never with the scripts' defaults: use_attn=False)."""
    wants = getattr(config, 'use_attn', False) or getattr(config, 'use_ptr', False)
    if not wants or DecoderRNN.KEY_ATTN_SCORE not in ret_dict:
        return None
    per_step = [a.cpu().numpy() for a in ret_dict[DecoderRNN.KEY_ATTN_SCORE]]
    return np.array(per_step, dtype=float).squeeze(2).swapaxes(0, 1)


def generate(model, data_feed, config, evaluator, num_batch=1, dest_f=None):
    """Free-running generation over `data_feed` (all of it when `num_batch` is None), every (reference, hypothesis)
    pair handed to `evaluator` under its dialog domain, the evaluator's report at the end (laed/main.py:184-243).
    Lines go to `dest_f` if given, else to the log; single examples are only printed for short previews."""
    model.eval()
    de_tknize = get_dekenize()
    emit = (lambda line: dest_f.write(line + '\n')) if dest_f is not None else logger.info
    verbose = num_batch is None or num_batch <= 2
    data_feed.epoch_init(config, shuffle=num_batch is not None, verbose=False)
    evaluator.initialize()
    logger.info("Generating over %s batches", data_feed.num_batch if num_batch is None else num_batch)
    for batch in _batches(data_feed):
        if num_batch is not None and data_feed.ptr > num_batch:
            break
        with torch.no_grad():
            outputs, labels = model(batch, mode=GEN, gen_type=config.gen_type)
        hyp_ids, ref_ids = sequences_of(outputs), labels.cpu().numpy()
        scores = _attention_of(outputs, config)
        contexts, context_lens, domains = batch.get('contexts'), batch.get('context_lens'), batch.get('domains')
        for b_id in range(hyp_ids.shape[0]):
            hyp, note = get_sent(model, de_tknize, hyp_ids, b_id, attn=scores)
            ref, _ = get_sent(model, de_tknize, ref_ids, b_id)
            domain = 'default' if domains is None else domains[b_id]
            evaluator.add_example(ref, hyp, domain)
            if not verbose:
                continue
            if contexts is not None:            # the turn right before the response
                last_turn, _ = get_sent(model, de_tknize, contexts[:, context_lens[b_id] - 1, :], b_id)
                emit("Source: " + last_turn)
            emit("%s:: True: %s ||| Pred: %s" % (domain, ref, hyp))
            if note:
                emit("[[%s]]" % note)
    report = evaluator.get_report(include_error=dest_f is not None)
    if report:
        emit(report)
    logger.info("Generation done")
