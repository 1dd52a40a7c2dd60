#!/usr/bin/env python3
"""Golden record of the reference's four driver scripts' FLAG TABLES (the benchmark configurations):
every flag name with the type it is parsed as and its default, read off the argparse parser each script builds at import
(ptb-utt.py:18-90, dailydialog-utt-skip.py:18-90, stanford-ae.py:18-91, stanford-skip.py:18-93), plus what `get_config()` of
each script returns for one non-default command line.

Same rules as make_goldens.py (whose import shim is reused): build container only, the scripts are IMPORTED from
/root/reference (their `__main__` bodies do not run), only the record is committed: tests/golden/script_flags.json.
`nltk.translate` is absent like the rest of nltk; laed/evaluators.py imports it at module level, so an empty stand-in module
is placed in sys.modules for the import (harness side; nothing of it is called here).

usage:  python tests/golden/make_goldens_flags.py
"""
import importlib.util
import json
import os
import sys
import types

import make_goldens as mg          # noqa: F401  (sets up the reference import + nltk stand-in)

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SCRIPTS = ["ptb-utt.py", "dailydialog-utt-skip.py", "stanford-ae.py", "stanford-skip.py"]
ARGV = ["--batch_size", "16", "--use_gpu", "false", "--data_dir", "a", "b", "--init_lr", "0.01", "--gen_type", "beam",
        "--forward_only", "True", "--not_a_flag", "7"]


def _stub_translate():
    tr = types.ModuleType("nltk.translate")
    bs = types.ModuleType("nltk.translate.bleu_score")
    bs.SmoothingFunction = object
    tr.bleu_score = bs
    sys.modules["nltk"].translate = tr
    sys.modules.update({"nltk.translate": tr, "nltk.translate.bleu_score": bs})


def kind(action):
    t = action.type
    if t is None:
        return "str"
    name = getattr(t, "__name__", str(t))
    return {"str2bool": "bool"}.get(name, name)


def main():
    _stub_translate()
    out = {}
    for script in SCRIPTS:
        spec = importlib.util.spec_from_file_location("ref_script_" + script.replace("-", "_").replace(".", "_"),
                                                      os.path.join(REF, script))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        flags = {}
        for a in mod.parser._actions:
            if not a.option_strings or a.dest == "help":
                continue
            flags[a.dest] = {"type": kind(a), "default": a.default, "nargs": a.nargs}
        old = sys.argv
        sys.argv = [script] + ARGV
        try:
            cfg, unparsed = mod.get_config()
        finally:
            sys.argv = old
        out[script] = {"flags": flags, "parsed": {"argv": ARGV, "config": vars(cfg), "unparsed": unparsed}}
        print("%-28s %d flags" % (script, len(flags)))
    with open(os.path.join(HERE, "script_flags.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
