"""Generation-side evaluator with the reference's interface (laed/evaluators.py:17-90): `BleuEvaluator(data_name)` with
`initialize / add_example(ref, hyp, domain) / get_report`, one corpus-BLEU line per domain.

The reference calls `nltk.translate.bleu_score.corpus_bleu(refs, hyps, smoothing_function=SmoothingFunction().method1)`
(evaluators.py:84).  nltk is not installable here, so `corpus_bleu` below restates the published algorithm
(Papineni et al. 2002; smoothing 1 of Chen & Cherry 2014 as nltk 3.x applies it): uniform 4-gram weights, modified
n-gram precisions pooled over the corpus, closest-reference-length brevity penalty, 0 when no unigram matches, and
a zero numerator replaced by epsilon = 0.1.  PARITY UNPINNED against nltk itself; pinned by the worked examples of the
BLEU paper as nltk's own docstrings quote them (tests/test_host_logic_cpu.py).
"""
import logging
import math
from collections import Counter, defaultdict

from .dataset.corpora import BOS, EOS
from .utils import get_tokenize


def _ngrams(tokens, n):
    return Counter(tuple(tokens[i:i + n]) for i in range(len(tokens) - n + 1))


def _closest_ref_length(refs, hyp_len):
    return min((len(r) for r in refs), key=lambda rl: (abs(rl - hyp_len), rl))


def corpus_bleu(list_of_references, hypotheses, weights=(0.25, 0.25, 0.25, 0.25), epsilon=0.1, smooth=True):
    """Corpus-level BLEU: list_of_references[i] = list of token lists for hypotheses[i] (a token list)."""
    assert len(list_of_references) == len(hypotheses)
    num, den = Counter(), Counter()
    hyp_len = ref_len = 0
    for refs, hyp in zip(list_of_references, hypotheses):
        for n in range(1, len(weights) + 1):
            counts = _ngrams(hyp, n)
            best = Counter()
            for r in refs:
                for g, c in _ngrams(r, n).items():
                    if c > best[g]:
                        best[g] = c
            num[n] += sum(min(c, best[g]) for g, c in counts.items())      # clipped matches
            den[n] += max(1, sum(counts.values()))
        hyp_len += len(hyp)
        ref_len += _closest_ref_length(refs, len(hyp))
    if num[1] == 0:
        return 0.0
    if hyp_len == 0:
        return 0.0
    bp = 1.0 if hyp_len > ref_len else math.exp(1.0 - float(ref_len) / hyp_len)
    log_sum = 0.0
    for n, w in enumerate(weights, start=1):
        p = float(num[n]) / den[n]
        if num[n] == 0:
            if not smooth:
                return 0.0
            p = epsilon / den[n]
        log_sum += w * math.log(p)
    return bp * math.exp(log_sum)


class EvaluatorBase(object):
    def initialize(self):
        raise NotImplementedError

    def add_example(self, ref, hyp, domain='default'):
        raise NotImplementedError

    def get_report(self, include_error=False):
        raise NotImplementedError

    @staticmethod
    def _get_prec_recall(tp, fp, fn):
        precision = tp / (tp + fp + 10e-20)
        recall = tp / (tp + fn + 10e-20)
        return precision, recall, 2 * precision * recall / (precision + recall + 1e-20)

    @staticmethod
    def _get_tp_fp_fn(label_list, pred_list):
        tp = sum(1 for t in pred_list if t in label_list)
        return tp, max(0, len(pred_list) - tp), max(0, len(label_list) - tp)


class BleuEvaluator(EvaluatorBase):
    """Per-domain corpus BLEU of (reference string, hypothesis string) pairs (evaluators.py:43-90)."""
    logger = logging.getLogger(__name__)

    def __init__(self, data_name):
        self.data_name = data_name
        self.initialize()

    def initialize(self):
        self.domain_labels = defaultdict(list)
        self.domain_hyps = defaultdict(list)

    def add_example(self, ref, hyp, domain='default'):
        self.domain_labels[domain].append(ref)
        self.domain_hyps[domain].append(hyp)

    def bleu(self, domain='default'):
        tokenize = get_tokenize()
        strip = lambda s: tokenize(s.replace(EOS, '').replace(BOS, ''))[2:]      # evaluators.py:77-78 drops 2 tokens
        refs = [[strip(r)] for r in self.domain_labels[domain]]
        hyps = [strip(h) for h in self.domain_hyps[domain]]
        return corpus_bleu(refs, hyps)

    def get_report(self, include_error=False):
        reports = []
        for domain in self.domain_labels:
            self.logger.info("Generate report for {} for {} samples".format(domain, len(self.domain_hyps[domain])))
            reports.append("\nDomain: %s BLEU %f\n" % (domain, self.bleu(domain)))
        return "\n==== REPORT===={report}".format(report="========".join(reports))
