"""Diagnostic: two gloo ranks sharing cuda:0.  Which of (a) plain torch work, (b) the library's GEMMs, (c) its GEMMs with
the event profiler on, in front of a small CUDA-tensor all_reduce, makes the collective wait forever?
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tools/exp/gloo_shared_gpu.py"""
import datetime, faulthandler, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "neuraldialog-laed_amd"))
import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(45, exit=True)
torch.cuda.set_device(0)
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
rank = dist.get_rank()
from laed_amd import ops
from laed_amd._lib import lib


def say(msg):
    print("[rank %d %.2f] %s" % (rank, time.time() % 1000, msg), file=sys.stderr, flush=True)


MODE = os.environ.get("AR_MODE", "cuda")      # cuda: gloo's CUDA-tensor path; sync: synchronize first; host: stage through host


def ar(n):
    t = torch.ones(n, device="cuda")
    if MODE == "sync":
        torch.cuda.current_stream().synchronize()
        dist.all_reduce(t)
    elif MODE == "host":
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)
    return float(t[0])

say("start"); ar(1); say("all_reduce(1 float) ok")
a = torch.randn(8192, 8192, device="cuda")
for _ in range(20):
    b = a @ a
say("torch matmuls queued"); ar(202); say("all_reduce after torch work ok")
h = torch.randn(2496, 1024, device="cuda"); W = torch.randn(50000, 1024, device="cuda")
for _ in range(5):
    o = ops.sgemm(h, W, transB=True)
say("library GEMMs queued"); ar(202); say("all_reduce after library GEMMs ok")
lib.laed_prof_enable(1)
for _ in range(5):
    o = ops.sgemm(h, W, transB=True)
say("library GEMMs (prof on) queued"); ar(202); say("all_reduce after profiled GEMMs ok")
lib.laed_prof_enable(0)
labels = torch.randint(1, 50000, (2496,), device="cuda"); bias = torch.zeros(50000, device="cuda")
scale = torch.full((2496,), 1.0 / 2496, device="cuda")
tok, lse, dh, dW, db = ops.proj_nll_raw(h, W, bias, labels, scale, 0, True)
say("proj_nll queued"); ar(202); say("all_reduce after proj_nll ok")
torch.cuda.synchronize(); say("done")
dist.destroy_process_group()
