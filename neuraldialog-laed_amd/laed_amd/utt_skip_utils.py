"""Skip-thought (DI-VST) reporting utilities with the reference's entry points (laed/utt_skip_utils.py): `generate`
decodes the previous and the next utterance from the latent code of the current one and scores both with the
evaluator; `dump_latent` is the sentence-model dump (same record layout)."""
import logging

import torch

from . import main as engine
from .enc2dec.decoders import GEN
from .utils import get_dekenize
from .utt_utils import _batches, _writer, collect_latent, dump_latent      # noqa: F401  (dump_latent re-exported)

logger = logging.getLogger()


def generate(model, data_feed, config, evaluator, num_batch=1, dest_f=None):
    """utt_skip_utils.py:19-83."""
    model.eval()
    de_tknize = get_dekenize()
    evaluator.initialize()
    write = _writer(dest_f)
    for batch in _batches(data_feed, config, num_batch, batch_size=5):
        with torch.no_grad():
            (prev_out, next_out), (prev_labels, next_labels) = model(batch, mode=GEN, gen_type=config.gen_type)
        cur = batch.get('outputs')
        prev_true, next_true = prev_labels.cpu().numpy(), next_labels.cpu().numpy()
        prev_pred, next_pred = engine.sequences_of(prev_out), engine.sequences_of(next_out)
        for b_id in range(cur.shape[0]):
            text = lambda m: engine.get_sent(model, de_tknize, m, b_id)[0]
            evaluator.add_example(text(prev_true), text(prev_pred))
            evaluator.add_example(text(next_true), text(next_pred))
            write("Response: {}".format(text(cur)))
            write("Prev Target: {}".format(text(prev_true)))
            write("Prev Predict: {}".format(text(prev_pred)))
            write("Next Target: {}".format(text(next_true)))
            write("Next Predict: {}\n".format(text(next_pred)))
    report = evaluator.get_report(include_error=dest_f is not None)
    if dest_f is None:
        logger.info(report)
    else:
        dest_f.write(report)
    logger.info("Generation Done")
