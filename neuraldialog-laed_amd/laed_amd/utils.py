"""Host plumbing shared with the reference's `laed.utils` (Pack, cast_type, dtype tags, str2bool).

Reference: laed/utils.py:17-55,119-157.  The tokenisers are the reference's regular expressions (`nltk.RegexpTokenizer`
= `re.findall` with UNICODE | MULTILINE | DOTALL); the detokeniser stands in for nltk's MosesDetokenizer (absent here:
PARITY UNPINNED), with the English punctuation / contraction rules that matter for the corpora's lower-cased text.
Session-dir helpers are out of scope.
"""
import re
from collections import defaultdict

import torch

INT = 0
LONG = 1
FLOAT = 2


class Pack(dict):
    """Attribute-style dict (laed/utils.py:22-37)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def add(self, **kwargs):
        for k, v in kwargs.items():
            self[k] = v

    def copy(self):
        pack = Pack()
        for k, v in self.items():
            pack[k] = list(v) if type(v) is list else v
        return pack


def str2bool(v):
    return v.lower() in ('true', '1')


_DTYPES = {INT: torch.int32, LONG: torch.int64, FLOAT: torch.float32}


def cast_type(var, dtype, use_gpu):
    """laed/utils.py:123-142: cast to Int/Long/Float on the GPU (use_gpu) or the CPU."""
    if dtype not in _DTYPES:
        raise ValueError("Unknown dtype")
    var = var.to(_DTYPES[dtype])
    return var.cuda(non_blocking=True) if use_gpu else var.cpu()


_RE_FLAGS = re.UNICODE | re.MULTILINE | re.DOTALL
_TOKEN = re.compile(r'\w+|#\w+|<\w+>|%\w+|[^\w\s]+', _RE_FLAGS)          # laed/utils.py:150
_CHAT_TOKEN = re.compile(r'\w+|<sil>|[^\w\s]+', _RE_FLAGS)                # laed/utils.py:154
_NO_SPACE_BEFORE = re.compile(r" ([.,!?;:%)\]}]|'s|'m|'d|'re|'ve|'ll|n't)(?= |$)")
_NO_SPACE_AFTER = re.compile(r"(^| )([(\[{$#]) ")


def get_tokenize():
    return _TOKEN.findall


def get_chat_tokenize():
    return _CHAT_TOKEN.findall


def get_dekenize():
    """tokens -> sentence: joined by blanks, closing punctuation and English clitics attached to the word before
    them, opening brackets / currency signs to the word after."""
    def detokenize(tokens):
        s = " ".join(tokens)
        s = _NO_SPACE_BEFORE.sub(r"\1", s)
        return _NO_SPACE_AFTER.sub(r"\1\2", s)
    return detokenize


class missingdict(defaultdict):
    """defaultdict whose missing keys are not stored (laed/utils.py:157-159)."""

    def __missing__(self, key):
        return self.default_factory()
