"""The four driver scripts' flag tables (laed_amd/cli.py) against the parsers the reference scripts build.

tests/golden/script_flags.json is made by tests/golden/make_goldens_flags.py, which imports the reference's ptb-utt.py,
dailydialog-utt-skip.py, stanford-ae.py and stanford-skip.py and reads every flag's name, parse type, nargs and default off
their argparse parsers, then records what each script's get_config() returns for one non-default command line."""
import json
import os
import subprocess
import sys

import pytest

from laed_amd import cli

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "script_flags.json")))
PKG_DIR = os.path.join(HERE, "..", "neuraldialog-laed_amd")
SCRIPTS = sorted(GOLD)


def _kind(t):
    return "bool" if t is cli.utils.str2bool else t.__name__


def test_the_four_scripts_are_the_reference_s_four():
    assert SCRIPTS == sorted(cli.FLAGS) == ["dailydialog-utt-skip.py", "ptb-utt.py", "stanford-ae.py", "stanford-skip.py"]
    for s in SCRIPTS:
        assert os.path.exists(os.path.join(PKG_DIR, s)), s


@pytest.mark.parametrize("script", SCRIPTS)
def test_flag_table_equals_the_reference_parser(script):
    ref = GOLD[script]["flags"]
    assert sorted(cli.FLAGS[script]) == sorted(ref)
    parser = cli.build_parser(script)
    mine = {a.dest: a for a in parser._actions if a.option_strings and a.dest != "help"}
    for name, want in ref.items():
        a = mine[name]
        assert _kind(a.type) == want["type"], (script, name)
        assert a.default == want["default"] and type(a.default) is type(want["default"]), (script, name, a.default, want["default"])
        assert a.nargs == want["nargs"], (script, name)
    for name in cli.IGNORED_FLAGS:                   # documented no-ops are still parsed wherever the reference parses them
        assert (name in cli.FLAGS[script]) == (name in ref)


@pytest.mark.parametrize("script", SCRIPTS)
def test_get_config_parses_a_command_line_as_the_reference_does(script):
    rec = GOLD[script]["parsed"]
    config, unparsed = cli.get_config(script, rec["argv"])
    assert vars(config) == rec["config"]
    assert unparsed == rec["unparsed"]


def test_no_cpu_path_behind_use_gpu_false():
    config, _ = cli.get_config("ptb-utt.py", ["--use_gpu", "false"])
    with pytest.raises(SystemExit) as e:
        cli.run("ptb-utt.py", config)
    assert "no CPU path" in str(e.value)


def test_script_files_start_and_list_their_flags():
    out = subprocess.run([sys.executable, os.path.join(PKG_DIR, "stanford-skip.py"), "--help"], stdout=subprocess.PIPE,
                         universal_newlines=True, timeout=120, check=True).stdout
    for flag in ("--freeze_step", "--include_eod", "--greedy_q", "--backward_size"):
        assert flag in out
