"""The four driver scripts end to end on the GPU (laed_amd/cli.py; reference ptb-utt.py / dailydialog-utt-skip.py /
stanford-ae.py / stanford-skip.py `main(config)`), at small sizes on real text:

  corpus files -> reader -> loaders -> model -> engine.train (session dir, params.json, session.log, best checkpoint) -> reload
  the checkpoint -> the script's reports (test loss, sweep / latent dump / generations with BLEU; for the dialog scripts also
  the attribute accuracy, the latent-action clusters, the selected and the conditioned generations)

and then the same script again with `--forward_only true --load_sess <that session>`: every model setting comes back from the
session's params.json (utils.process_config) and only the reports run.

The text is what tests/golden/corpora.npz carries of the reference's data files (DailyDialog validation / test dialogues, reduced
Stanford dev / test JSON); the PTB-format files are the DailyDialog utterances written one per line (the reference ships no PTB
text).  Nothing here is compared with reference numbers -- the reference's scripts need its absent training sets, Python 2 and
nltk; what the scripts are MADE of is pinned elsewhere (flag tables: test_cli_cpu.py; readers, loaders, engine, models: their
own golden tests)."""
import glob
import json
import logging
import os
import pickle

import numpy as np
import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu

SMALL = ["--embed_size", "32", "--dec_cell_size", "64", "--max_vocab_cnt", "800", "--max_utt_len", "20", "--max_dec_len", "20",
         "--early_stop", "false", "--init_lr", "0.005", "--preview_batch_num", "1"]
SENT = SMALL + ["--enc_cell_size", "64", "--y_size", "4", "--k", "5", "--batch_size", "30", "--max_epoch", "2",
                "--ckpt_step", "30", "--print_step", "15"]
DIALOG = SMALL + ["--utt_cell_size", "32", "--ctx_cell_size", "64", "--batch_size", "8", "--max_epoch", "3", "--ckpt_step", "12",
                  "--print_step", "6", "--freeze_step", "18"]


@pytest.fixture(scope="module")
def data_dirs(tmp_path_factory):
    root = tmp_path_factory.mktemp("corpora")
    G = load_golden("corpora")
    dd = root / "daily_dialog"
    for split, s in (("train", "train"), ("validation", "valid"), ("test", "test")):
        (dd / split).mkdir(parents=True)
        for fname, k in (("dialogues.txt", "txt"), ("dialogues_act.txt", "act"), ("dialogues_emotion.txt", "emo")):
            (dd / split / fname).write_bytes(G["daily_dialog/inputs/%s/%s" % (s, k)].tobytes())
    st = root / "stanford"
    st.mkdir()
    inputs = json.loads(str(G["stanford/inputs_json"]))
    for fname, s in (("kvret_train_public.json", "train"), ("kvret_dev_public.json", "valid"), ("kvret_test_public.json", "test")):
        (st / fname).write_text(json.dumps(inputs[s]), encoding="utf-8")
    ptb = root / "ptb"
    ptb.mkdir()
    for s in ("train", "valid", "test"):                     # PTB format: one lower-cased sentence per line, blank-led
        text = G["daily_dialog/inputs/%s/txt" % s].tobytes().decode("utf-8", errors="ignore").lower()
        utts = [u.strip() for line in text.splitlines() for u in line.split("__eou__") if u.strip()]
        (ptb / ("ptb.%s.txt" % s)).write_text("".join(" %s \n" % u for u in utts), encoding="utf-8")
    return {"ptb-utt.py": str(ptb), "dailydialog-utt-skip.py": str(dd), "stanford-ae.py": str(st), "stanford-skip.py": str(st)}


@pytest.fixture()
def quiet_root_logger():
    root = logging.getLogger()
    before, level = list(root.handlers), root.level
    yield
    for h in list(root.handlers):
        if h not in before:
            root.removeHandler(h)
            h.close()
    root.setLevel(level)


def _run(script, argv, data_dirs, log_dir):
    from laed_amd import cli
    torch.manual_seed(0)
    np.random.seed(0)
    return cli.main(script, ["--data_dir", data_dirs[script], "--log_dir", str(log_dir)] + argv)


def _check_session(paths, script):
    sess = os.path.dirname(paths["model"])
    assert os.path.basename(sess).endswith("-" + script)
    params = json.load(open(os.path.join(sess, "params.json")))
    assert params["session_dir"] == sess and params["dec_cell_size"] == 64
    log = open(os.path.join(sess, "session.log")).read()
    assert "Best validation loss" in log and "Pseudo-epoch 1 done" in log
    sd = torch.load(paths["model"], weights_only=True)          # the reference's checkpoint format: a plain state_dict
    assert all(torch.is_tensor(v) for v in sd.values()) and any(k.endswith("decoder.project.weight") for k in sd)
    return sess, params, log


@pytest.mark.parametrize("script", ["ptb-utt.py", "dailydialog-utt-skip.py"])
def test_sentence_model_scripts_end_to_end(script, data_dirs, tmp_path, quiet_root_logger):
    from laed_amd import ops
    paths = _run(script, SENT, data_dirs, tmp_path / "logs")
    sess, params, log = _check_session(paths, script)
    assert params["y_size"] == 4 and params["batch_size"] == 30           # as parsed; the report phase's 50 is not written back
    if script == "ptb-utt.py":
        assert "Start:" in log and "End:" in log                         # the latent sweep goes to the log
    z = pickle.load(open(paths["dump"], "rb"))
    assert len(z["labels"]) == len(z["metas"]) > 100 and z["z"][0][2].shape[1] == 4
    text = open(paths["test"]).read()
    assert "BLEU" in text and text.count("Target:" if script == "ptb-utt.py" else "Prev Target:") > 100
    assert ops.gru_persist_timeouts() == 0
    # reports only, from the saved session
    again = _run(script, ["--forward_only", "true", "--load_sess", os.path.basename(sess), "--gen_type", "greedy"], data_dirs,
                 tmp_path / "logs")
    assert os.path.dirname(again["model"]) == sess                       # written next to the session it read
    assert len(glob.glob(os.path.join(sess, "*-z.pkl"))) >= 1 and "BLEU" in open(again["test"]).read()
    z2 = pickle.load(open(again["dump"], "rb"))
    assert z2["labels"] == z["labels"]                                   # same model (params.json + checkpoint), same test set


@pytest.mark.parametrize("script", ["stanford-ae.py", "stanford-skip.py"])
def test_dialog_model_scripts_end_to_end(script, data_dirs, tmp_path, quiet_root_logger):
    from laed_amd import ops
    paths = _run(script, DIALOG, data_dirs, tmp_path / "logs")
    sess, params, log = _check_session(paths, script)
    assert "Training phase changed" in log                               # freeze_step 18 lies inside the run
    clusters = json.load(open(paths["dump"] + ".json"))
    selected = json.load(open(paths["dump"] + ".out.json"))
    assert isinstance(clusters, (dict, list)) and isinstance(selected, list)
    z = pickle.load(open(paths["dump"], "rb"))
    assert len(z["z"]) > 0
    assert os.path.getsize(paths["test"]) > 0                            # generations conditioned on every code
    assert "BLEU" in open(paths["test"] + ".txt").read()
    assert ops.gru_persist_timeouts() == 0


@pytest.mark.parametrize("script", ["ptb-utt.py", "stanford-skip.py"])
def test_script_under_a_launcher_trains_data_parallel(script, data_dirs, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 <script> ...`: two ranks (gloo rehearsal transport, both on the
    box's GPU: LAED_DP_BACKEND=gloo; production is one rank per GPU over RCCL) train through the engine on halves of every
    batch; rank 0 alone makes the session directory, the log, the checkpoint and -- after the ranks have parted -- the
    reports.  ST-ED goes through its phase change (freeze, new optimizer, new gradient reducer) on both ranks at once."""
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "neuraldialog-laed_amd")
    # a rank still running after 100 s dumps its threads' stacks and exits: a failure with a place, not a silent hang
    env = dict(os.environ, LAED_DP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", LAED_WATCHDOG="100")
    import socket
    probe = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    probe.bind(("127.0.0.1", 0))
    port = probe.getsockname()[1]
    probe.close()
    sent = script == "ptb-utt.py"
    flags = [a if a != "30" else "32" for a in SENT] if sent else DIALOG
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(pkg, script), "--data_dir", data_dirs[script], "--log_dir",
           str(tmp_path / "logs")] + flags
    with open(str(tmp_path / "run.log"), "w") as out:
        rc = subprocess.run(cmd, env=env, stdin=subprocess.DEVNULL, stdout=out, stderr=subprocess.STDOUT, timeout=150).returncode
    assert rc == 0, open(str(tmp_path / "run.log")).read()[-6000:]
    sessions = os.listdir(str(tmp_path / "logs"))
    assert len(sessions) == 1 and sessions[0].endswith("-" + script)                      # one session: rank 0's
    sess = str(tmp_path / "logs" / sessions[0])
    log = open(os.path.join(sess, "session.log")).read()
    assert log.count("Pseudo-epoch 1 done") == 1 and "Best validation loss" in log and "Generation Done" in log
    assert json.load(open(os.path.join(sess, "params.json")))["batch_size"] == (32 if sent else 8)
    assert os.path.exists(os.path.join(sess, "model")) and len(glob.glob(os.path.join(sess, "*-z.pkl"))) == 1
    if sent:
        assert "BLEU" in open(glob.glob(os.path.join(sess, "*-test-greedy.txt"))[0]).read()
    else:
        assert log.count("Training phase changed") == 1
        assert "BLEU" in open(glob.glob(os.path.join(sess, "*-test-greedy.txt.txt"))[0]).read()
