"""The reference's four driver scripts as ONE table-driven module.

    ptb-utt.py                DI-VAE on PTB                       (reference ptb-utt.py:18-155)
    dailydialog-utt-skip.py   DI-VST on DailyDialog               (dailydialog-utt-skip.py:18-151)
    stanford-ae.py            AE-ED on Stanford multi-domain      (stanford-ae.py:18-169)
    stanford-skip.py          ST-ED on Stanford multi-domain      (stanford-skip.py:18-173)

The scripts of the same names next to this package are three lines each: `cli.main("<script>")`.  What a script IS in the
reference is (1) a flag table -- the hyper-parameters of the benchmark configurations -- and (2) one wiring: corpus reader ->
three loaders -> model -> engine.train -> reload the best checkpoint -> the script's own set of reports.  Both are data here:
FLAGS (shared defaults + per-script differences; pinned flag by flag, type by type and default by default against the parsers
the reference scripts build, tests/golden/script_flags.json) and RECIPES.

Flags the reference parses and never reads are accepted and ignored, as there: `grad_clip`, `init_w`,
`lr_decay`, `lr_hold`, `step_size`, `loss_type`; `momentum` matters for `--op sgd / rmsprop` only; `backward_size` is unused by
the two sentence-model scripts.  Flags that select paths this build leaves out raise in the model constructors as before
(`--rnn_cell lstm`, `--use_attn true`: NotImplementedError).  There is no CPU path: `--use_gpu false` is refused.

Data parallelism (new): started under a launcher --
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 <script> ...` -- every rank (one per GPU, RCCL over
xGMI) builds the same model from the same seed and runs the same engine loop on its own rows of every batch; `--batch_size` is
the GLOBAL batch and must divide by N.  Rank 0 alone creates the session directory, logs, writes checkpoints and, when training
is over, produces the reports on its own.  `--forward_only true` needs one process only.
Environment (diagnostics only): `LAED_WATCHDOG=<seconds>` dumps every thread's stack and exits if the script is still running
then (where is it stuck?); `LAED_DP_BACKEND=gloo` rehearses the multi-rank path with the ranks sharing the GPUs that exist.
"""
import argparse
import json
import logging
import os
import sys

import torch

from laed_amd import utils

logger = logging.getLogger()

_B = utils.str2bool
# flag -> (type, default) shared by every script that has the flag
_SHARED = {
    "log_dir": (str, "logs"), "rnn_cell": (str, "gru"), "embed_size": (int, 200), "dec_cell_size": (int, 512),
    "max_utt_len": (int, 40), "max_dec_len": (int, 40), "max_vocab_cnt": (int, 10000), "num_layer": (int, 1),
    "use_attn": (_B, False), "attn_type": (str, "cat"), "op": (str, "adam"), "step_size": (int, 1),
    "grad_clip": (float, 3.0), "init_lr": (float, 0.001), "lr_hold": (int, 1), "lr_decay": (float, 0.6),
    "dropout": (float, 0.3), "improve_threshold": (float, 0.996), "early_stop": (_B, True), "save_model": (_B, True),
    "use_gpu": (_B, True), "fix_batch": (_B, False), "preview_batch_num": (int, 1), "gen_type": (str, "greedy"),
    "avg_type": (str, "word"), "beam_size": (int, 10), "forward_only": (_B, False),
}
_SENT = {      # the two sentence-model scripts
    "y_size": (int, 20), "k": (int, 10), "use_mutual": (_B, True), "use_reg_kl": (_B, True), "utt_type": (str, "rnn"),
    "enc_cell_size": (int, 512), "bi_enc_cell": (_B, False), "backward_size": (int, 5), "max_epoch": (int, 50),
    "print_step": (int, 500),
}
_DIALOG = {    # the two encoder-decoder scripts
    "data_dir": (str, ["data/stanford"]), "y_size": (int, 3), "k": (int, 5), "use_attribute": (_B, True),
    "utt_cell_size": (int, 256), "ctx_cell_size": (int, 512), "bi_ctx_cell": (_B, False), "greedy_q": (_B, True),
    "momentum": (float, 0.0), "loss_type": (str, "e2e"), "batch_size": (int, 30),
}
FLAGS = {
    "ptb-utt.py": dict(_SHARED, **_SENT, **{
        "data_dir": (str, ["data/ptb"]), "init_w": (float, 0.08), "momentum": (float, 0.0), "patient_increase": (float, 2.0),
        "train_prior": (_B, False), "ckpt_step": (int, 2000), "batch_size": (int, 30),
        "load_sess": (str, "2018-02-04T01-20-45")}),
    "dailydialog-utt-skip.py": dict(_SHARED, **_SENT, **{
        "data_dir": (str, ["data/daily_dialog"]), "init_w": (float, 0.1), "momentum": (float, 0.1),
        "patient_increase": (float, 3.0), "include_eod": (_B, True), "ckpt_step": (int, 1500), "batch_size": (int, 60),
        "load_sess": (str, "2018-05-28T14-08-32-dailydialog-utt-skip.py")}),
    "stanford-ae.py": dict(_SHARED, **_DIALOG, **{
        "utt_type": (str, "attn_rnn"), "backward_size": (int, 30), "init_w": (float, 0.1), "patient_increase": (float, 4.0),
        "max_epoch": (int, 100), "print_step": (int, 100), "ckpt_step": (int, 500), "freeze_step": (int, 4000),
        "load_sess": (str, "2018-02-14T12-34-00-stanford-ae.py")}),
    "stanford-skip.py": dict(_SHARED, **_DIALOG, **{
        "use_reg_kl": (_B, True), "utt_type": (str, "rnn"), "backward_size": (int, 10), "init_w": (float, 0.08),
        "patient_increase": (float, 2.0), "max_epoch": (int, 50), "include_eod": (_B, True), "print_step": (int, 200),
        "ckpt_step": (int, 1000), "freeze_step": (int, 5000), "load_sess": (str, "2018-02-14T12-55-34-stanford-skip.py")}),
}
IGNORED_FLAGS = ("grad_clip", "init_w", "lr_decay", "lr_hold", "step_size", "loss_type")


def build_parser(script):
    p = argparse.ArgumentParser(prog=script, description=__doc__.split("\n\n")[0])
    for name in sorted(FLAGS[script]):
        kind, default = FLAGS[script][name]
        extra = {"nargs": "+"} if name == "data_dir" else {}
        p.add_argument("--" + name, type=kind, default=default,
                       help="accepted and ignored, as in the reference" if name in IGNORED_FLAGS else None, **extra)
    return p


def get_config(script, argv=None):
    """-> (config Namespace, unparsed arguments): what the reference script's `get_config()` returns."""
    return build_parser(script).parse_known_args(sys.argv[1:] if argv is None else argv)


def _recipes():
    from laed_amd import dialog_utils, utt_skip_utils, utt_utils
    from laed_amd.dataset import corpora, data_loaders
    from laed_amd.models import dialog_models, sent_models
    return {
        # corpus reader, loader, model, preview / report generator, evaluator name, reports after training
        "ptb-utt.py": (corpora.PTBCorpus, data_loaders.PTBDataLoader, sent_models.DiVAE, utt_utils.generate, "CornellMovie",
                       _sentence_reports(utt_utils, sweep=True)),
        "dailydialog-utt-skip.py": (corpora.DailyDialogCorpus, data_loaders.DailyDialogSkipLoader, sent_models.DiVST,
                                    utt_skip_utils.generate, "dailydialog-utt-skip.py", _sentence_reports(utt_skip_utils, sweep=False)),
        "stanford-ae.py": (corpora.StanfordCorpus, data_loaders.SMDDataLoader, dialog_models.AeED, dialog_utils.generate_with_adv,
                           "stanford-ae.py", _dialog_reports(None)),
        "stanford-skip.py": (corpora.StanfordCorpus, data_loaders.SMDDialogSkipLoader, dialog_models.StED,
                             dialog_utils.generate_with_adv, "CornellMovie", _dialog_reports(0)),
    }


def _sentence_reports(mod, sweep):
    """ptb-utt.py:137-149 / dailydialog-utt-skip.py:135-145: test-set loss, (DI-VAE) the latent sweep, the latent dump, the
    generations with BLEU -- at batch size 50."""
    def run(model, feeds, config, evaluator, paths):
        from laed_amd import main as engine, utt_utils
        config.batch_size = 50
        engine.validate(model, feeds["test"], config)
        if sweep:
            utt_utils.sweep(model, feeds["test"], config, num_batch=50)
        with open(paths["dump"], "wb") as f:
            logger.info("Dumping test to {}".format(paths["dump"]))
            mod.dump_latent(model, feeds["test"], config, f, num_batch=None)
        with open(paths["test"], "w") as f:
            logger.info("Saving test to {}".format(paths["test"]))
            mod.generate(model, feeds["test"], config, evaluator, num_batch=None, dest_f=f)
    return run


def _dialog_reports(adv_batches):
    """stanford-ae.py:135-163 / stanford-skip.py:137-167: valid and test loss, attribute accuracy of the generations (all test
    batches for AE-ED; the reference passes num_batch=0 for ST-ED, i.e. none), the latent-action clusters of the training set
    and the generations selected by them (two JSON files), the latent dump, generation conditioned on every code, and the
    plain generations with BLEU -- at batch size 10."""
    def run(model, feeds, config, evaluator, paths):
        from laed_amd import dialog_utils, main as engine, utt_utils
        config.batch_size = 10
        engine.validate(model, feeds["valid"], config)
        engine.validate(model, feeds["test"], config)
        dialog_utils.generate_with_adv(model, feeds["test"], config, None, num_batch=adv_batches)
        clusters = utt_utils.latent_cluster(model, feeds["train"], config, num_batch=None)
        selected = dialog_utils.selective_generate(model, feeds["test"], config, clusters)
        logger.info("%d selected generations" % len(selected))
        with open(paths["dump"] + ".json", "w") as f:
            json.dump(clusters, f, indent=2)
        with open(paths["dump"] + ".out.json", "w") as f:
            json.dump(selected, f, indent=2)
        with open(paths["dump"], "wb") as f:
            logger.info("Dumping test to {}".format(paths["dump"]))
            dialog_utils.dump_latent(model, feeds["test"], config, f, num_batch=None)
        with open(paths["test"], "w") as f:
            logger.info("Saving test to {}".format(paths["test"]))
            dialog_utils.gen_with_cond(model, feeds["test"], config, num_batch=None, dest_f=f)
        with open(paths["test"] + ".txt", "w") as f:
            dialog_utils.generate(model, feeds["test"], config, evaluator, num_batch=None, dest_f=f)
    return run


def _init_data_parallel(config):
    """Under a torch.distributed launcher: one rank per GPU over RCCL.  -> (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1")) if "RANK" in os.environ else 1
    if world <= 1:
        return 0, 1
    if config.forward_only:                                  # reports need one process: the others have nothing to do
        return int(os.environ["RANK"]), -world
    import torch.distributed as dist
    from laed_amd import parallel
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    # LAED_DP_BACKEND=gloo: REHEARSAL of the multi-rank path on fewer GPUs than ranks (the ranks share the GPUs that exist;
    # tests/test_cli_gpu.py on the one-GPU box) -- not a way to train
    backend = os.environ.get("LAED_DP_BACKEND", "nccl")
    if backend == "nccl" and local >= ndev:
        raise SystemExit("rank %s has no GPU of its own (%d visible): one rank per GPU" % (os.environ["RANK"], ndev))
    if config.batch_size % world:
        raise SystemExit("--batch_size %d (the global batch) does not divide by %d ranks" % (config.batch_size, world))
    torch.cuda.set_device(local % ndev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    parallel.init()
    if world > ndev:
        from laed_amd import ops
        ops.set_device_sharers((world + ndev - 1) // ndev)
    return dist.get_rank(), dist.get_world_size()


def run(script, config):
    """One driver script, start to finish (the `main(config)` of the reference script).  -> the paths it wrote (rank 0)."""
    from laed_amd import evaluators, main as engine, parallel
    if not config.use_gpu:
        raise SystemExit("%s: --use_gpu false is not available: this build has no CPU path (the kernels are gfx950 HIP)" % script)
    if not torch.cuda.is_available():
        raise SystemExit("%s needs an MI355X" % script)
    rank, world = _init_data_parallel(config)
    if world < 0:                                            # forward_only under a launcher
        if rank != 0:
            return None
        world = 1
    corpus_cls, loader_cls, model_cls, gen, eval_name, reports = _recipes()[script]
    if world > 1:
        import torch.distributed as dist
        torch.manual_seed(int(getattr(config, "seed", 0)))   # identical replicas (the reference's scripts seed nothing)
        box = [None]
        if rank == 0:
            utils.prepare_dirs_loggers(config, script)
            box[0] = getattr(config, "session_dir", None)
        else:
            logging.getLogger().setLevel(logging.WARNING)    # one log: rank 0's
        dist.broadcast_object_list(box, src=0)
        config.session_dir = box[0]
    else:
        utils.prepare_dirs_loggers(config, script)
    corpus_client = corpus_cls(config)
    corpus = corpus_client.get_corpus()
    evaluator = evaluators.BleuEvaluator(eval_name)
    feeds = {"train": loader_cls("Train", corpus["train"], config), "valid": loader_cls("Valid", corpus["valid"], config),
             "test": loader_cls("Test", corpus["test"], config)}
    model = model_cls(corpus_client, config)
    where = os.path.join(config.log_dir, config.load_sess) if config.forward_only else config.session_dir
    stamp = utils.get_time()
    paths = {"test": os.path.join(where, "{}-test-{}.txt".format(stamp, config.gen_type)),
             "dump": os.path.join(where, "{}-z.pkl".format(stamp)), "model": os.path.join(where, "model")}
    model.cuda()
    if not config.forward_only:
        try:
            engine.train(model, feeds["train"], feeds["valid"], feeds["test"], config, evaluator, gen=gen)
        except KeyboardInterrupt:
            print("Training stopped by keyboard.")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()                                       # rank 0's last checkpoint is on disk; every step has been joined
        parallel.shutdown()                                  # from here on rank 0 works alone: no exchange in any forward
        if rank != 0:
            return None
    # the best checkpoint of the session: a plain state_dict (the reference's format; tensors only)
    model.load_state_dict(torch.load(paths["model"], map_location="cuda", weights_only=True))
    reports(model, feeds, config, evaluator, paths)
    return paths


def main(script, argv=None):
    if os.environ.get("LAED_WATCHDOG"):                       # seconds: dump every thread's stack and exit (hang diagnosis)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LAED_WATCHDOG"]), exit=True)
    config, unparsed = get_config(script, argv)
    if unparsed:
        print("%s: ignoring unknown arguments %s" % (script, unparsed), file=sys.stderr)
    return run(script, utils.process_config(config))
