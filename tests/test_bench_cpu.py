"""bench.py's host-side pieces (no GPU): the algorithmic-FLOP formula reproduces SURVEY.md 8(d)'s figures, the synthetic
batches have the documented shape, and the workload table matches the reference scripts' defaults."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_flops_formula_matches_survey():
    V, E, H, y, k, T, B = bench.SHAPES["S-fixed"]
    assert (V, E, H, y, k, T, B) == (50000, 200, 1024, 20, 10, 40, 4096)
    f = bench.flops_per_step(V, E, H, y, k, T, B)
    assert abs(f - 56.4e12) / 56.4e12 < 5e-3                     # "train 56.4 TFLOP/step"
    assert abs(f / (B * (T - 1)) - 353e6) / 353e6 < 5e-3         # "353 MFLOP per target token"
    V, E, H, y, k, T, B = bench.SHAPES["P-fixed"]
    assert abs(bench.flops_per_step(V, E, H, y, k, T, B) - 51.5e9) / 51.5e9 < 5e-3
    # DI-VST: two decoders
    V, E, H, y, k, T, B = bench.SHAPES["VST-fixed"]
    two = bench.flops_per_step(V, E, H, y, k, T, B, dec_positions=2 * (T - 1))
    one = bench.flops_per_step(V, E, H, y, k, T, B)
    assert two > 1.8 * one - bench.flops_per_step(V, E, H, y, k, T, B, dec_positions=0)


def test_synthetic_batches():
    out = bench.synth_batch(1000, 12, 7)
    assert out.shape == (7, 12) and out.dtype == np.int32
    assert (out[:, 0] == 2).all() and (out[:, -1] == 3).all() and out[:, 1:-1].min() >= 4 and out.max() < 1000
    assert np.array_equal(out, bench.synth_batch(1000, 12, 7))   # seeded
    full, lens = bench.synth_varlen(1000, 40, 30)
    assert full.shape == (30, int(lens.max())) and lens.min() >= 3 and lens.max() <= 40
    for b, L in enumerate(lens):
        assert full[b, 0] == 2 and full[b, L - 1] == 3 and (full[b, L:] == 0).all() and (full[b, 1:L - 1] >= 4).all()


def test_script_default_configs():
    c = bench.make_config(200, 512, 20, 10, 40, 30)
    assert (c.dropout, c.avg_type, c.use_mutual, c.train_prior, c.init_lr, c.op) == (0.3, 'word', True, False, 0.001, 'adam')
    ae, st = bench.make_dialog_config(200, 512, 3, 5, 40, 30, False), bench.make_dialog_config(200, 512, 3, 5, 40, 30, True)
    assert (ae.utt_type, ae.freeze_step, ae.backward_size, ae.utt_cell_size) == ('attn_rnn', 4000, 30, 256)
    assert (st.utt_type, st.freeze_step, st.backward_size) == ('rnn', 5000, 10)


def test_gpus_n_self_launches_its_ranks(monkeypatch, capsys):
    """`python bench.py --gpus N` with no launcher around it (no RANK in the environment) starts N fresh ranks through
    torch.distributed.run on 127.0.0.1 and relays rank 0's single JSON line; other stdout chatter goes to stderr."""
    import io
    import subprocess
    seen = {}

    class FakeProc(object):
        def __init__(self, cmd, **kw):
            seen["cmd"], seen["env"] = cmd, kw["env"]
            self.stdout = io.StringIO('gloo chatter\n{"metric": "train tokens/sec (DI-VAE, whole node)", "n_gpus": 2}\n')

        def wait(self):
            return 0
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.delenv("RANK", raising=False)
    rc = bench.self_launch(2, ["--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo"])
    cmd = seen["cmd"]
    assert rc == 0 and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-9] == os.path.abspath(bench.__file__) and cmd[-8:-6] == ["--gpus", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["LAED_BENCH_SELF_LAUNCHED"] == "1"
    cap = capsys.readouterr()
    assert cap.out.strip().splitlines() == ['{"metric": "train tokens/sec (DI-VAE, whole node)", "n_gpus": 2}']
    assert "gloo chatter" in cap.err


def test_gpus_n_spawn_reaches_real_ranks():
    """The real thing, on this GPU-less container: two ranks are started (each sees WORLD_SIZE=2 and a RANK, so none of
    them launches again), each refuses to run without an MI355X, and the parent reports the failure instead of a line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, bench.__file__, "--gpus", "2", "--backend", "gloo", "--global-batch", "128",
                        "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert r.stderr.count("bench.py needs an MI355X") >= 1
    assert "without a result line" in r.stderr
