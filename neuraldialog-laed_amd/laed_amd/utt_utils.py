"""Sentence-model (DI-VAE) reporting utilities with the reference's entry points (laed/utt_utils.py):
`generate`, `dump_latent`, `sweep`, `latent_cluster`, `find_mi` -- the callers the training scripts hang behind
`engine.train(..., gen=utt_utils.generate)` and run after training (ptb-utt.py:131-149, stanford-ae.py:141).

Host composition only: every forward pass goes through the same HIP path as training (`model(batch, mode=...)`).
Id matrices leave the device in one copy per batch (`engine.sequences_of`), not one per decoding step.
"""
import logging
import pickle
from collections import defaultdict

import numpy as np
import torch

from . import main as engine
from .enc2dec.decoders import GEN, TEACH_FORCE
from .utils import get_dekenize

logger = logging.getLogger()


def _batches(data_feed, config, num_batch, batch_size=None):
    """Unshuffled pass over the feed; with a preview count (num_batch) the feed is re-cut into small batches and the
    walk stops after num_batch of them (utt_utils.py:18-35)."""
    old = config.batch_size
    if num_batch is not None and batch_size is not None:
        config.batch_size = batch_size
    try:
        data_feed.epoch_init(config, shuffle=False, verbose=False)
    finally:
        config.batch_size = old
    logger.info("{} batches".format(data_feed.num_batch if num_batch is None else num_batch))
    while True:
        batch = data_feed.next_batch()
        if batch is None or (num_batch is not None and data_feed.ptr > num_batch):
            return
        yield batch


def _writer(dest_f):
    def write(msg):
        if dest_f is None:
            logger.info(msg)
        else:
            dest_f.write(msg + '\n')
    return write


def generate(model, data_feed, config, evaluator, num_batch=1, dest_f=None):
    """Reconstruct every utterance from its latent code and score it against itself (utt_utils.py:16-61)."""
    model.eval()
    de_tknize = get_dekenize()
    evaluator.initialize()
    write = _writer(dest_f)
    for batch in _batches(data_feed, config, num_batch, batch_size=3):
        with torch.no_grad():
            outputs, labels = model(batch, mode=GEN, gen_type=config.gen_type)
        pred_labels = engine.sequences_of(outputs)
        true_labels = labels.cpu().numpy()
        for b_id in range(pred_labels.shape[0]):
            pred_str, _ = engine.get_sent(model, de_tknize, pred_labels, b_id)
            true_str, _ = engine.get_sent(model, de_tknize, true_labels, b_id)
            evaluator.add_example(true_str, pred_str)
            write("Target: {}".format(true_str))
            write("Predict: {}\n".format(pred_str))
    report = evaluator.get_report(include_error=dest_f is not None)
    if dest_f is None:
        logger.info(report)
    else:
        dest_f.write(report)
    logger.info("Generation Done")


def collect_latent(model, data_feed, config, num_batch=1):
    """{'z': [(log_qy, dec_init, y_ids) per batch], 'labels': [...], 'metas': [...]} (utt_utils.py:64-94)."""
    model.eval()
    de_tknize = get_dekenize()
    all_zs, all_labels, all_metas = [], [], []
    for batch in _batches(data_feed, config, num_batch):
        with torch.no_grad():
            results = model(batch, mode=TEACH_FORCE, return_latent=True)
        labels = np.asarray(batch.outputs)
        for b_id in range(labels.shape[0]):
            all_labels.append(engine.get_sent(model, de_tknize, labels, b_id)[0])
            all_metas.append(batch.metas[b_id])
        all_zs.append((results.log_qy.detach().cpu().squeeze(0).numpy(),
                       results.dec_init_state.detach().cpu().squeeze().numpy(),
                       results.y_ids.detach().cpu().numpy()))
    return {'z': all_zs, 'labels': all_labels, 'metas': all_metas}


def dump_latent(model, data_feed, config, dest_f, num_batch=1):
    """Pickle of `collect_latent` into the open binary file dest_f (same layout as the reference's dump)."""
    pickle.dump(collect_latent(model, data_feed, config, num_batch), dest_f)
    logger.info("Dumping Done")


def sweep(model, data_feed, config, num_batch=1, dest_f=None):
    """Walk the latent code from the first to the last utterance of each batch and print the decoded sentence every
    time the code changes (utt_utils.py:98-148; model.sweep = sent_models.py:167-213)."""
    model.eval()
    de_tknize = get_dekenize()
    write = _writer(dest_f)
    for batch in _batches(data_feed, config, num_batch, batch_size=10):
        with torch.no_grad():
            outputs, labels, all_y_ids = model.sweep(batch, gen_type=config.gen_type)
        true_labels = labels.cpu().numpy()
        all_y_ids = all_y_ids.cpu().numpy()
        pred_labels = engine.sequences_of(outputs)
        write("Start: {}".format(engine.get_sent(model, de_tknize, true_labels, 0)[0]))
        prev_code = None
        for b_id in range(pred_labels.shape[0]):
            code = '-'.join(map(str, all_y_ids[b_id]))
            if code != prev_code:
                write("Predict ({}): {}".format(code, engine.get_sent(model, de_tknize, pred_labels, b_id)[0]))
                prev_code = code
        write("End: {}\n".format(engine.get_sent(model, de_tknize, true_labels, true_labels.shape[0] - 1)[0]))
    logger.info("Generation Done")


def _codes_of(results, config):
    """First-maximum class of every latent variable, per row: [B, y_size] ints (utt_utils.py:185-191)."""
    qy = torch.exp(results.log_qy.detach().view(-1, config.y_size, config.k)).cpu().numpy()
    return np.argmax(qy, axis=2)                 # numpy's argmax is the first maximum, like the reference's scan


def latent_cluster(model, data_feed, config, num_batch=1, max_samples=5, rng=None):
    """Group utterances by their latent code and list up to max_samples examples per code (utt_utils.py:151-219).
    Returns [{'code', 'meaning', 'examples'}]; None when k ** y_size > 1000, as the reference."""
    if np.power(float(config.k), config.y_size) > 1000:
        logger.info("Skip latent cluster too many states")
        return None
    rng = rng if rng is not None else np.random
    model.eval()
    de_tknize = get_dekenize()
    clusters = defaultdict(list)
    for batch in _batches(data_feed, config, num_batch):
        with torch.no_grad():
            results = model(batch, mode=TEACH_FORCE, return_latent=True)
        labels = np.asarray(batch.outputs)
        codes = _codes_of(results, config)
        for b_id in range(labels.shape[0]):
            code = '-'.join(str(int(c)) for c in codes[b_id])
            clusters[code].append((engine.get_sent(model, de_tknize, labels, b_id)[0], batch.metas[b_id]))
    logger.info("Find {} clusters".format(len(clusters)))
    selected = []
    for symbol in sorted(clusters):
        sents = clusters[symbol]
        if len(sents) < 5:
            logger.info("Skip tiny cluster with {} utts - {}".format(len(sents), symbol))
            continue
        logger.info("Symbol {}".format(symbol))
        ids = rng.permutation(len(sents)) if len(sents) < max_samples else rng.choice(len(sents), max_samples, replace=False)
        for s_id in ids[:5]:
            logger.info(sents[s_id][0])
        logger.info("")
        selected.append({'code': symbol, 'meaning': '', 'examples': [sents[i][0] for i in ids]})
    return selected


def find_mi(model, data_feed, config):
    """Homogeneity (normalised mutual information) between every meta attribute of the utterances and their latent
    code, as a whole and per latent variable (utt_utils.py:222-292).  Returns {(key, 'ID' | dim): score}."""
    from sklearn import metrics
    model.eval()
    all_codes, all_metas, meta_keys = [], [], []
    for batch in _batches(data_feed, config, None):
        with torch.no_grad():
            results = model(batch, mode=TEACH_FORCE, return_latent=True)
        for key in batch.metas[0].keys():
            if key not in meta_keys:
                meta_keys.append(key)
        y_ids = results.y_ids.cpu().numpy()
        all_codes.extend(y_ids[b] for b in range(y_ids.shape[0]))
        all_metas.extend(batch.metas[:y_ids.shape[0]])
    vec_codes = np.array(all_codes)
    weights = np.power(config.k, np.arange(config.y_size)).astype(np.int64)
    _, vec_idxes = np.unique(vec_codes.astype(np.int64) @ weights, return_inverse=True)
    scores = {}
    for key in meta_keys:
        vals = []
        for m in all_metas:
            if isinstance(m[key], list):
                vals.append(" ".join(map(str, m[key])))
            elif isinstance(m[key], dict):
                break
            else:
                vals.append(m[key])
        if len(vals) != len(all_metas) or not vals:
            continue
        _, vals = np.unique(np.array([str(v) for v in vals]), return_inverse=True)
        scores[(key, 'ID')] = metrics.homogeneity_score(vals, vec_idxes)
        logger.info("{} mi with ID is {}".format(key, scores[(key, 'ID')]))
        for y_id in range(config.y_size):
            scores[(key, y_id)] = metrics.homogeneity_score(vals, vec_codes[:, y_id])
            logger.info("{} mi with dim {} is {}".format(key, y_id, scores[(key, y_id)]))
    return scores
