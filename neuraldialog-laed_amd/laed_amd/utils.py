"""Host plumbing shared with the reference's `laed.utils` (Pack, cast_type, dtype tags, str2bool).

Reference: laed/utils.py:17-55,119-157.  The tokenisers are the reference's regular expressions (`nltk.RegexpTokenizer`
= `re.findall` with UNICODE | MULTILINE | DOTALL); the detokeniser stands in for nltk's MosesDetokenizer (absent here:
PARITY UNPINNED), with the English punctuation / contraction rules that matter for the corpora's lower-cased text.
The session helpers at the end (run directory, session.log, params.json, forward_only reload) follow utils.py:58-117.
"""
import json
import logging
import os
import re
import sys
from argparse import Namespace
from collections import defaultdict
from datetime import datetime

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


# ---- session plumbing of the driver scripts (utils.py:58-117): a run's directory, its log file and params.json ----
def get_time():
    return datetime.now().strftime("%Y-%m-%dT%H-%M-%S")


def load_config(load_path):
    """params.json of a session -> Namespace (utils.py:108-112)."""
    with open(load_path, "r") as fp:
        return Namespace(**json.load(fp))


def process_config(config):
    """forward_only runs take every setting from the loaded session's params.json except the decoding-time ones
    (utils.py:58-73)."""
    if getattr(config, "forward_only", False):
        keep = {k: getattr(config, k) for k in ("load_sess", "backward_size", "beam_size", "gen_type") if hasattr(config, k)}
        config = load_config(os.path.join(config.log_dir, config.load_sess, "params.json"))
        config.forward_only = True
        for k, v in keep.items():
            setattr(config, k, v)
        config.batch_size = 50
    return config


def prepare_dirs_loggers(config, script=""):
    """Root logger to stdout; for training runs also <log_dir>/<time>[-script]/session.log, with the configuration saved
    next to it as params.json (utils.py:76-105)."""
    fmt = logging.Formatter("%(message)s")
    root = logging.getLogger()
    root.setLevel(logging.DEBUG)
    console = logging.StreamHandler(sys.stdout)
    console.setLevel(logging.DEBUG)
    console.setFormatter(fmt)
    root.addHandler(console)
    if getattr(config, "forward_only", False):
        return
    os.makedirs(config.log_dir, exist_ok=True)
    config.session_dir = os.path.join(config.log_dir, "{}-{}".format(get_time(), script) if script else get_time())
    os.mkdir(config.session_dir)
    log_file = logging.FileHandler(os.path.join(config.session_dir, "session.log"))
    log_file.setLevel(logging.DEBUG)
    log_file.setFormatter(fmt)
    root.addHandler(log_file)
    with open(os.path.join(config.session_dir, "params.json"), "w") as fp:
        json.dump(vars(config), fp, indent=4, sort_keys=True)
