#!/usr/bin/env python3
"""DI-VAE train-step benchmark on MI355X (BASELINE.json: train tokens/s, whole node; fwd+bwd step ms).

    python bench.py --gpus N --steps K --warmup W          # any N: for N > 1 (and no RANK in the environment) this
                                                           # process starts the N ranks itself and relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # the driver's own launcher form: used as given

Workload "S-fixed" (BASELINE.json configs[4]): DI-VAE, V=50000, E=200, H_enc=H_dec=1024, y=20, k=10,
T=40, GLOBAL batch 4096 (strong scaling: 4096/N sequences per GPU), every row [<s>, Zipf ids..., </s>], fp32,
dropout 0.3 on, Gumbel noise + dropout from the library's counter RNG keyed by global row.  A step =
zero_grad + forward(TEACH_FORCE) + backward (+ overlapped RCCL gradient all-reduce) + fused Adam
(laed/main.py:104-113 order).  Token = non-PAD decoder target position (the 'word' NLL denominator).

One JSON line on rank 0.  `roofline` is measured live: HIP events bracket every GEMM launch of the timed steps on
their own stream (laed_prof_*); the dominant GEMM variant's algorithmic FLOP / its summed duration is `achieved`.
`cpu_baseline` (rank 0, N=1 only) times the oracle's nn.Module restatement (the reference's PyTorch CPU path) on a
bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import gc
import os
import sys
import time
from argparse import Namespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "neuraldialog-laed_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
VARIANTS = ["sgemm_128x128_NT", "sgemm_64x64_NT", "sgemm_128x128_NN", "sgemm_64x64_NN", "sgemm_128x128_TN",
            "sgemm_64x64_TN", "xgemm_128x128_NT_bf16x3"]
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA; a bf16x3 product issues 3 bf16 MFMA flops per algorithmic flop

SHAPES = {
    # name: (V, E, H, y, k, T, global batch)
    "S-fixed": (50000, 200, 1024, 20, 10, 40, 4096),      # BASELINE.json configs[4] (headline)
    "P-fixed": (10002, 200, 512, 20, 10, 40, 30),         # ptb-utt.py defaults (configs[0] shape)
    "P-varlen": (10002, 200, 512, 20, 10, 40, 30),        # same, lengths ~ clip(round(lognormal(3.0, 0.45)), 3, 40)
    "VST-fixed": (10004, 200, 512, 20, 10, 40, 60),       # dailydialog-utt-skip.py defaults (configs[1] shape), DI-VST
    # stanford-ae.py / stanford-skip.py defaults (configs[2], configs[3]): y=3, k=5, utt_cell 256, ctx/dec cell 512, B=30;
    # every dialog context = DIALOG_CTX turns of T tokens (the loaders pad contexts to the batch maximum, <= backward_size)
    "AE-fixed": (10004, 200, 512, 3, 5, 40, 30),
    "ST-fixed": (10004, 200, 512, 3, 5, 40, 30),
}
DIALOG_CTX = 10


TRAFFIC_NOTE = ("profiles/r3_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                "(tools/pmc_traffic.py; KiB*1024, FETCH doubled per the gfx950 note), L2<->fabric bytes per launch of the "
                "variant's largest launch shape (the projection product; algorithmic bytes of that launch: dW TN 8.4 GB, "
                "dh NN 8.4 GB, logits NT 8.6 GB with 40960-row chunks), Infinity-Cache hits included; committed file, not "
                "re-measured in this run (PMC passes serialise kernels)")


def pmc_traffic(variant, same_job):
    """Per-launch L2<->fabric bytes of `variant` from the committed PMC passes of this same job (S-fixed, 1 GPU);
    None for any other job (the counters cannot be collected inside a normal run)."""
    path = os.path.join(ROOT, "profiles", "r3_pmc_traffic.json")
    if not same_job or not os.path.exists(path):
        return None
    try:
        v = json.load(open(path))["variants"][variant]
        # the variant's launches are dissimilar (projection product, its split-K tail, GRU products): report the launch
        # shape that moves the most bytes -- the projection product itself -- not an average over all of them
        return v.get("largest_launch", v)["bytes_per_launch"]
    except (KeyError, ValueError):
        return None


class Corpus(object):
    def __init__(self, V):
        self.vocab = ['<pad>', '<unk>', '<s>', '</s>'] + ['w%d' % i for i in range(4, V)]
        self.rev_vocab = {w: i for i, w in enumerate(self.vocab)}


def make_dialog_config(E, H, y, k, T, batch, skip):
    """stanford-ae.py:33-91 / stanford-skip.py:33-93 defaults."""
    return Namespace(embed_size=E, max_utt_len=T, num_layer=1, dropout=0.3, dec_cell_size=H, ctx_cell_size=H,
                     utt_cell_size=256, bi_ctx_cell=False, rnn_cell='gru', max_dec_len=T, use_attn=False, beam_size=10,
                     utt_type='rnn' if skip else 'attn_rnn', attn_type='cat', fix_batch=False, y_size=y, k=k, use_gpu=True,
                     avg_type='word', use_reg_kl=True, use_attribute=True, greedy_q=True, init_lr=0.001, op='adam',
                     momentum=0.0, batch_size=batch, freeze_step=5000 if skip else 4000, backward_size=10 if skip else 30)


def make_config(E, H, y, k, T, batch, use_gpu=True):
    return Namespace(embed_size=E, max_utt_len=T, num_layer=1, dropout=0.3, enc_cell_size=H, dec_cell_size=H,
                     rnn_cell='gru', max_dec_len=T, use_attn=False, beam_size=10, utt_type='rnn', bi_enc_cell=False,
                     attn_type='cat', fix_batch=False, y_size=y, k=k, use_gpu=use_gpu, avg_type='word',
                     use_mutual=True, use_reg_kl=True, train_prior=False, init_lr=0.001, op='adam', batch_size=batch)


IDS = "zipf"      # --ids: interior token distribution (only the embedding-gradient scatter contention depends on it)


def synth_batch(V, T, B, seed=1234):
    """[B,T] int32: <s>=2, Zipf(s=1) (or uniform) interior ids over [4,V), </s>=3 (SURVEY.md 8d 'S-fixed')."""
    rs = np.random.RandomState(seed)
    ranks = np.arange(1, V - 4 + 1, dtype=np.float64)
    p = 1.0 / ranks if IDS == "zipf" else np.ones_like(ranks)
    p /= p.sum()
    out = np.empty((B, T), dtype=np.int32)
    out[:, 0] = 2
    out[:, 1:T - 1] = 4 + rs.choice(V - 4, size=(B, T - 2), p=p)
    out[:, T - 1] = 3
    return out


def synth_varlen(V, T, B, seed=1234):
    """P-varlen: len ~ clip(round(lognormal(3.0, 0.45)), 3, T), zero-padded to the batch maximum."""
    rs = np.random.RandomState(seed)
    lens = np.clip(np.round(rs.lognormal(3.0, 0.45, size=B)), 3, T).astype(np.int64)
    out = np.zeros((B, int(lens.max())), dtype=np.int32)
    for b, L in enumerate(lens):
        out[b, 0] = 2
        out[b, 1:L - 1] = rs.randint(4, V, size=L - 2)
        out[b, L - 1] = 3
    return out, lens


def flops_per_step(V, E, H, y, k, T, B, dec_positions=None):
    """train step = 3 x forward; dec_positions = decoder input positions per sequence (T-1 for DI-VAE,
    (Tp-1)+(Tn-1) for DI-VST)."""
    f_enc = 2.0 * 3 * H * (E + H)
    f_dec = 2.0 * 3 * H * (E + H) + 2.0 * H * V
    f_seq = 2.0 * H * y * k + 2.0 * y * k * H
    dp = (T - 1) if dec_positions is None else dec_positions
    return 3.0 * (B * T * f_enc + B * dp * f_dec + B * f_seq)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(V, E, H, y, k, T, sample_b, steps=2, warmup=1):
    """Oracle nn.Module DiVAE (== the reference's stock-torch CPU path) on `sample_b` sequences of the workload."""
    from oracle.laed_oracle import OracleDiVAE
    torch.manual_seed(0)
    cfg = make_config(E, H, y, k, T, sample_b, use_gpu=False)
    model = OracleDiVAE(V, cfg)
    model.train()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=cfg.init_lr)
    batch = synth_batch(V, T, sample_b)
    rs = np.random.RandomState(7)

    def step():
        u = rs.uniform(size=(sample_b * y, k)).astype(np.float32)
        opt.zero_grad()
        res = model(batch, u)
        res["total"].backward()
        opt.step()
    for _ in range(warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    return sample_b * (T - 1) / dt, dt


def self_launch(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher: start the N ranks as FRESH processes (torch.distributed.run,
    one rank per GPU, rendezvous on 127.0.0.1) before this process has made any GPU call -- it never makes one: a process
    that has initialised the GPU must not be replaced or forked on this pool -- and relay rank 0's single JSON line.
    The ranks' stderr passes through; their other stdout lines (launcher / gloo chatter) go to stderr.  One attempt
    only: a failed GPU run is read, not repeated.  Returns the exit code for this process."""
    import socket
    import subprocess
    probe = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    probe.bind(("127.0.0.1", 0))                             # a free port for rank 0's store
    port = probe.getsockname()[1]
    probe.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this host driver
    env["LAED_BENCH_SELF_LAUNCHED"] = "1"
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
        return rc
    print("bench.py: the %d ranks ended (rc %d) without a result line" % (n, rc), file=sys.stderr)
    return rc or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="S-fixed", choices=sorted(SHAPES))
    ap.add_argument("--global-batch", type=int, default=0, help="override the workload's global batch")
    ap.add_argument("--cpu-sample", type=int, default=48, help="sequences in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket GEMM launches with HIP events")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra split-bf16 measurement")
    ap.add_argument("--gemm-mode", default="fp32", choices=["fp32", "bf16x3"],
                    help="fp32 = exact fp32 MFMA everywhere (default); bf16x3 = opt-in split-bf16 x W^T products")
    ap.add_argument("--ids", default="zipf", choices=["zipf", "uniform"], help="interior token id distribution")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1: nccl = RCCL over xGMI (one rank per GPU); gloo = REHEARSAL of "
                         "the N>1 code path with every rank on the GPUs that exist (ranks share cuda:0 on a 1-GPU box) -- "
                         "not a scaling measurement")
    ap.add_argument("--phase", default="B", choices=["A", "B"],
                    help="AE-fixed / ST-fixed: A = recognition-network pretraining (before freeze_step), "
                         "B = encoder-decoder training with the recognition network frozen (after it)")
    args = ap.parse_args()
    global IDS
    IDS = args.ids

    if os.environ.get("LAED_BENCH_WATCHDOG"):                # seconds: dump every thread's stack and exit (hang diagnosis)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LAED_BENCH_WATCHDOG"]), exit=True)
    if args.gpus > 1 and "RANK" not in os.environ:           # no launcher around us: be the launcher (no GPU call yet)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher's WORLD_SIZE is %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible); RCCL needs one rank per GPU "
                         "(--backend gloo rehearses the N>1 path on fewer GPUs)" % (local_rank, ndev))
    device_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(device_index)

    import torch.distributed as dist
    from laed_amd import parallel
    from laed_amd._lib import lib
    from laed_amd.enc2dec.decoders import TEACH_FORCE
    from laed_amd.models.sent_models import DiVAE, DiVST
    from laed_amd.models.dialog_models import AeED, StED
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index),
                                    timeout=datetime.timedelta(seconds=600))
        else:
            # gloo announces its connections on stdout: keep the one-JSON-line contract by sending fd 1 to stderr meanwhile
            sys.stdout.flush()
            saved_fd = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
            finally:
                os.dup2(saved_fd, 1)
                os.close(saved_fd)
        parallel.init()

    from laed_amd import ops as _ops
    if world > ndev:                                         # gloo rehearsal: several ranks per GPU
        _ops.set_device_sharers((world + ndev - 1) // ndev)
    V, E, H, y, k, T, GB = SHAPES[args.workload]
    if args.global_batch:
        GB = args.global_batch
    if GB % world:
        raise SystemExit("global batch %d not divisible by %d GPUs" % (GB, world))
    per = GB // world
    same_job = args.workload == "S-fixed" and GB == SHAPES["S-fixed"][6] and world == 1 and args.ids == "zipf"   # what the PMC passes ran

    torch.manual_seed(0)                                     # identical replicas on every rank
    is_vst = args.workload == "VST-fixed"
    is_dialog = args.workload in ("AE-fixed", "ST-fixed")
    is_skip = args.workload == "ST-fixed"
    if is_dialog:
        cfg = make_dialog_config(E, H, y, k, T, GB, is_skip)
        model = (StED if is_skip else AeED)(Corpus(V), cfg).cuda()
    else:
        cfg = make_config(E, H, y, k, T, GB)
        model = (DiVST if is_vst else DiVAE)(Corpus(V), cfg).cuda()
    model.train()
    opt = model.get_optimizer(cfg)
    batch_cnt = 0

    def resident(a):                                         # this rank's rows, resident in HBM before timing
        return torch.from_numpy(a[rank * per:(rank + 1) * per].astype(np.int64)).cuda()
    if args.workload == "P-varlen":
        full, lens = synth_varlen(V, T, GB)
        T = full.shape[1]
        tokens_per_step = int((lens - 1).sum())              # non-PAD decoder targets
        enc_tokens_per_step = GB * T                         # the encoder consumes PAD positions too
    else:
        full = synth_batch(V, T, GB)
        tokens_per_step = GB * (T - 1) * (2 if is_vst else 1)
        enc_tokens_per_step = GB * T
    feed = {"outputs": resident(full), "output_lens": [T] * per}
    if is_vst or is_skip:
        feed["prevs"], feed["nexts"] = resident(synth_batch(V, T, GB, seed=1235)), resident(synth_batch(V, T, GB, seed=1236))
    if is_dialog:
        ctx = np.stack([synth_batch(V, T, GB, seed=2000 + c) for c in range(DIALOG_CTX)], axis=1)     # [B, C, T]
        feed["contexts"], feed["context_lens"] = resident(ctx), [DIALOG_CTX] * per
        enc_tokens_per_step = GB * T * (1 + DIALOG_CTX)
    step_flops = flops_per_step(V, E, H, y, k, T, GB, dec_positions=2 * (T - 1) if is_vst else None)

    def step():
        opt.zero_grad()
        loss = model(feed, mode=TEACH_FORCE)
        model.backward(batch_cnt, loss)
        opt.step()
        return loss

    if is_dialog and args.phase == "B":
        # flip to the encoder-decoder phase the way engine.train does (dialog_models.py:278-307, main.py:106-110): the
        # first backward past freeze_step freezes the recognition side, then the optimizer is rebuilt
        batch_cnt = cfg.freeze_step + 1
        opt.zero_grad()
        model.backward(batch_cnt, model(feed, mode=TEACH_FORCE))
        assert model.flush_valid
        model.flush_valid = False
        opt = model.get_optimizer(cfg)

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    def measure(mode):
        """warm-up + timed steps in one GEMM math mode -> dict (rank 0 fills the derived fields)."""
        _ops.set_gemm_mode(mode)
        prof = not args.no_prof
        # The roofline needs every GEMM launch bracketed by a HIP event pair on its stream.  At the headline shape (launches of
        # 1-18 ms) the pairs sit inside the timed region, as the contract asks.  A step of the reference's own script shapes is
        # ~50 launches in 1.5 ms, and there the event pairs themselves cost 8 % (P-fixed 1.47 -> 1.59 ms, VST-fixed 3.04 -> 3.15):
        # those workloads are TIMED without them and a second pass of the same steps, instrumented, fills the roofline object.
        inline = prof and args.workload == "S-fixed"
        prof_out = (ctypes.c_double * 21)()
        for i in range(args.warmup):
            if prof and i == args.warmup - 1:               # count one step's GEMM launches ...
                fence()
                lib.laed_prof_enable(1)
            step()
        gc.collect()
        gc.freeze()     # the model / vocabulary objects are long-lived: keep full collections (25-30 ms of host stall,
        fence()         # a tenth of a 50-step P-fixed run) from walking them inside the timed region
        if prof:                                            # ... and create the instrumented steps' events beforehand
            lib.laed_prof_enable(0)
            lib.laed_prof_collect(prof_out)
            per_step_launches = int(sum(prof_out[v * 3] for v in range(7))) if args.warmup > 0 else 4096
            lib.laed_prof_reserve(2 * per_step_launches * args.steps + 64)
            if inline:
                lib.laed_prof_enable(1)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            loss = step()
            marks[i + 1].record()                           # device-side per-step spread (no host sync inside the loop)
        enqueue = time.perf_counter() - t0                  # what the HOST needed to issue the steps (it never waits inside)
        fence()
        elapsed = time.perf_counter() - t0
        raw_steps = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        if os.environ.get("LAED_BENCH_DEBUG"):
            print("per-step device ms:", " ".join("%.2f" % x for x in raw_steps), file=sys.stderr)
        per_step = sorted(raw_steps)
        if inline:
            lib.laed_prof_enable(0)
            lib.laed_prof_collect(prof_out)
        elif prof:                                          # the same steps again, instrumented (not timed for `value`)
            lib.laed_prof_enable(1)
            for i in range(args.steps):
                step()
            fence()
            lib.laed_prof_enable(0)
            lib.laed_prof_collect(prof_out)
        if is_dialog and args.phase == "A":
            nll = float((loss.vst_prev_nll + loss.vst_next_nll).detach()) if is_skip else float(loss.vae_nll.detach())
        else:
            nll = float((loss.prev_nll + loss.next_nll).detach()) if is_vst else float(loss.nll.detach())
        if dist.is_initialized():
            t = torch.tensor([elapsed, nll], device="cuda", dtype=torch.float64)
            tmax = t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)           # per-rank NLL shares sum to the global NLL
            elapsed, nll = float(tmax[0]), float(t[1])
        # what the host needs to ISSUE a step when nothing holds it back: a burst of 8 steps into an empty queue (over the
        # whole timed loop a host that runs ahead is throttled by the launch queue, so `enqueue` above tends to the device time)
        fence()
        tb = time.perf_counter()
        for _ in range(8):
            step()
        burst = (time.perf_counter() - tb) / 8
        # fwd+bwd (no Adam) step time, measured separately after the timed region
        fence()
        t1 = time.perf_counter()
        nfb = max(2, args.steps // 2)
        for _ in range(nfb):
            opt.zero_grad()
            model.backward(batch_cnt, model(feed, mode=TEACH_FORCE))
            if opt.reducer is not None:
                opt.reducer.wait()
        fence()
        fwd_bwd_ms = (time.perf_counter() - t1) / nfb * 1e3
        ms_per_step = elapsed / args.steps * 1e3
        roofline = None
        if prof:
            rows = [(prof_out[v * 3 + 1], v) for v in range(7) if prof_out[v * 3] > 0]
            tot_ms = sum(r[0] for r in rows)
            ms, v = max(rows)
            cnt, flops = prof_out[v * 3], prof_out[v * 3 + 2]
            achieved = flops / (ms * 1e-3) / 1e12                 # algorithmic TFLOP/s of the dominant GEMM variant
            if v == 6:      # split-bf16: price the MFMA work actually issued (3 bf16 flops per algorithmic flop)
                issued, peak = 3.0 * achieved, PEAK_BF16_MFMA_TFLOPS
            else:
                issued, peak = achieved, PEAK_FP32_MFMA_TFLOPS
            roofline = {"bound": "mfma", "kernel": VARIANTS[v], "achieved": round(issued, 2),
                        "instrumented": ("the timed region" if inline else
                                         "a second pass of the same %d steps (event pairs around ~50 launches cost 8 %% of a "
                                         "1.5 ms step; the timed region runs without them)" % args.steps),
                        "traffic_source": TRAFFIC_NOTE if pmc_traffic(VARIANTS[v], same_job) else None,
                        "peak": peak, "unit": "TFLOP/s", "frac": round(issued / peak, 4),
                        "algorithmic_tflops": round(achieved, 2),
                        "traffic": pmc_traffic(VARIANTS[v], same_job), "launches": int(cnt), "avg_launch_ms": round(ms / cnt, 4),
                        # every GEMM variant of the timed steps (the dominant one above is simply the largest by summed
                        # time; TN and NN are within 1 % of each other at the headline shape, so which one leads varies)
                        "variants": {VARIANTS[i]: {"launches": int(prof_out[i * 3]),
                                                   "ms_per_step": round(prof_out[i * 3 + 1] / args.steps, 3),
                                                   "tflops": round(prof_out[i * 3 + 2] / (prof_out[i * 3 + 1] * 1e-3) / 1e12, 2),
                                                   "frac_of_fp32_mfma_peak": round(prof_out[i * 3 + 2] / (prof_out[i * 3 + 1] * 1e-3) / 1e12
                                                                                   / PEAK_FP32_MFMA_TFLOPS, 4)}
                                     for i in range(6) if prof_out[i * 3] > 0},
                        "gemm_share_of_step": round(tot_ms / args.steps / ms_per_step, 4),
                        "all_gemm_tflops": round(sum(prof_out[i * 3 + 2] for i in range(7)) / (tot_ms * 1e-3) / 1e12, 2)}
        return {"value": round(tokens_per_step * args.steps / elapsed, 1), "ms_per_step": round(ms_per_step, 3),
                "fwd_bwd_ms": round(fwd_bwd_ms, 3),
                "host_enqueue_ms_per_step": round(enqueue / args.steps * 1e3, 3),
                "host_issue_ms_per_step": round(burst * 1e3, 3),
                "step_ms_device": {"median": round(per_step[len(per_step) // 2], 3),
                                   "p10": round(per_step[int(0.1 * (len(per_step) - 1))], 3),
                                   "p90": round(per_step[int(round(0.9 * (len(per_step) - 1)))], 3),
                                   "max": round(per_step[-1], 3)},
                "encoder_tokens_per_s": round(enc_tokens_per_step * args.steps / elapsed, 1),
                "step_tflops": None if is_dialog else round(step_flops / (ms_per_step * 1e-3) / 1e12, 2),
                "final_nll": round(nll, 5), "roofline": roofline}

    main_res = measure(args.gemm_mode)
    alt_res = measure("bf16x3") if (args.gemm_mode == "fp32" and not args.no_alt and not is_dialog) else None
    _ops.set_gemm_mode("fp32")

    if rank == 0:
        out = {
            "metric": "train tokens/sec (%s, whole node)" % (
                "DI-VST" if is_vst else "DI-VAE" if not is_dialog else
                ("ST-ED" if is_skip else "AE-ED") + (", encoder-decoder phase" if args.phase == "B" else ", pretraining phase")),
            "value": main_res["value"], "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
            "fwd_bwd_ms": main_res["fwd_bwd_ms"], "step_ms_device": main_res["step_ms_device"],
            # host time to ISSUE a step (no sync inside the timed loop): well under ms_per_step = the host runs ahead and the GPU
            # never waits for a launch; close to it = the step is host-bound
            "host_enqueue_ms_per_step": main_res["host_enqueue_ms_per_step"],
            # the same for a burst of 8 steps issued into an EMPTY queue: the host's own cost per step, without back-pressure
            "host_issue_ms_per_step": main_res["host_issue_ms_per_step"],
            "encoder_tokens_per_s": main_res["encoder_tokens_per_s"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.gemm_mode == "fp32" else "f32 (x W^T products as split-bf16 bf16x3, fp32 accumulate)",
            "data": "synthetic",
            # what actually carried the exchanges: torch.distributed's group as initialised, not the flag
            "backend": (("rccl" if str(dist.get_backend()).lower() == "nccl" else str(dist.get_backend()).lower())
                        if dist.is_initialized() else None),
            "rccl_ranks": (dist.get_world_size() if (dist.is_initialized() and str(dist.get_backend()).lower() == "nccl")
                           else 0),
            "launched_by": ("bench.py (self-launched torch.distributed.run)" if os.environ.get("LAED_BENCH_SELF_LAUNCHED")
                            else "external launcher" if "RANK" in os.environ else "single process"),
            **({"rehearsal": "backend gloo, %d ranks on %d GPU(s): exercises the N>1 code path (sharding, forward "
                             "exchanges, bucketed gradient all-reduce); NOT a scaling number" % (world, ndev)}
               if (world > 1 and args.backend == "gloo") else {}),
            "config": {"workload": args.workload, "ids": args.ids, "vocab": V, "embed": E, "hidden": H, "y_size": y, "k": k,
                       "seq_len": T, "global_batch": GB, "per_gpu_batch": per, "parallelism": "dp%d" % world,
                       "tokens_per_step": tokens_per_step,
                       **({"context_turns": DIALOG_CTX, "phase": args.phase} if is_dialog else {})},
            "step_tflops": main_res["step_tflops"],
            "final_nll": main_res["final_nll"],
            # sticky count of timed-out in-kernel waits of the persistent GRU kernels (B <= 64 shapes); non-zero = the
            # numbers above come from poisoned outputs and are void
            "gru_persist_timeouts": _ops.gru_persist_timeouts(),
            "roofline": main_res["roofline"],
        }
        if alt_res is not None:
            alt_res["note"] = ("same job with the opt-in split-bf16 GEMM mode (x W^T products as 3 bf16 MFMA products of "
                               "hi/lo-split fp32 operands, fp32 accumulate; same 1e-4 loss / exact greedy-id parity "
                               "tests pass, tests/test_bf16x3_mode_gpu.py). NOT the headline `value`, which is exact fp32.")
            out["split_bf16_mode"] = alt_res
        if world == 1 and args.cpu_sample > 0 and args.workload in ("S-fixed", "P-fixed"):
            # thread sweep: 1 thread on a quarter of the sample, then half and all of this job's cores;
            # the record run is then 2 warm-ups + 5 timed steps at the best thread count
            nmax = int(os.environ.get("OMP_NUM_THREADS", "0")) or torch.get_num_threads()
            sweep = []
            for nt, nseq in ((1, max(4, args.cpu_sample // 4)), (max(1, nmax // 2), args.cpu_sample), (nmax, args.cpu_sample)):
                torch.set_num_threads(nt)
                tps, dt = cpu_baseline(V, E, H, y, k, T, nseq, steps=1, warmup=1)
                sweep.append({"cores": nt, "value": round(tps, 1), "ms_per_step": round(dt * 1e3, 1), "sequences": nseq})
            best = max(sweep, key=lambda r: r["value"])
            torch.set_num_threads(best["cores"])
            nseq = GB if args.workload == "P-fixed" else args.cpu_sample      # P shape: the full 30-sequence step
            tps, dt = cpu_baseline(V, E, H, y, k, T, nseq, steps=5, warmup=2)
            out["cpu_baseline"] = {"value": round(tps, 1), "unit": "tokens/s", "cores": best["cores"],
                                   "kind": "port", "ms_per_step": round(dt * 1e3, 1), "cpu_model": cpu_model_name(),
                                   "sample": "%s shape, %d sequences/step (same step order: zero_grad, forward, backward, "
                                             "Adam), 2 warm-ups + 5 timed steps at the best thread count of the sweep"
                                             % (args.workload, nseq),
                                   "thread_sweep": sweep}
            if args.workload == "S-fixed":
                # SURVEY.md 8d's CPU reference of record: the reference's own script shape (P: ptb-utt.py defaults,
                # B = 30), full step, 2 warm-ups + 5 timed steps
                Vp, Ep, Hp, yp, kp, Tp, Bp = SHAPES["P-fixed"]
                tps_p, dt_p = cpu_baseline(Vp, Ep, Hp, yp, kp, Tp, Bp, steps=5, warmup=2)
                out["cpu_baseline"]["p_shape_full_step"] = {
                    "workload": "P-fixed (V=10002, H=512, B=30, T=40)", "value": round(tps_p, 1), "unit": "tokens/s",
                    "ms_per_step": round(dt_p * 1e3, 1), "cores": best["cores"], "warmup": 2, "steps": 5}
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
