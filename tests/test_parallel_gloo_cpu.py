"""world_size-2 data-parallel host path on CPU (gloo): gradient reducer, sharding, and the two forward
exchanges (global non-PAD count for 'word' NLL, batch-mean posterior for BPR) reproduce single-process values."""
import datetime
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, result_dir):
    for p in (ROOT, os.path.join(ROOT, "neuraldialog-laed_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from argparse import Namespace
    from helpers import golden_config, golden_params, load_golden
    from laed_amd import criterions, parallel
    from oracle import laed_oracle as O

    # file:// rendezvous: no port to pick (a probed-then-released TCP port can be taken before rank 0 binds it), and a
    # rank that never arrives fails the group in 60 s instead of the 30-minute default
    dist.init_process_group("gloo", init_method="file://" + os.path.join(result_dir, "rdzv"), rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=60))
    parallel.init()
    assert parallel.world_size() == world and parallel.rank() == rank and parallel.group() is not None
    assert parallel.shard_rows(12) == (rank * 6, rank * 6 + 6)

    # 1. overlapped gradient reducer: SUM over ranks, fired from post-accumulate-grad hooks
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(64, 8)),
              torch.nn.Parameter(torch.randn(3))]
    # big_bytes = 1 KiB: the [64,8] tensor (2 KiB) is reduced on its own, in place; the three small ones are coalesced
    # into one flat buffer (3 + 7 + 15 floats reach the 64-byte bucket size with the last of them)
    red = parallel.GradReducer(params, bucket_bytes=64, big_bytes=1024)
    sum((p * float(rank + 1)).sum() for p in params).backward()
    red.wait()
    for p in params:
        assert torch.allclose(p.grad, torch.full_like(p, 3.0)), p.shape
    assert red.collectives == 2, red.collectives          # 4 gradient tensors, 2 collectives
    # a second optimizer for the same model (engine.train rebuilds it after flush_valid, main.py:106-110) must not leave
    # the first reducer's hooks live: the reduced gradient is world_size x the local one, not world_size squared
    from laed_amd.models.model_bases import BaseModel

    class Tiny(BaseModel):
        def __init__(self):
            super(Tiny, self).__init__(Namespace(use_gpu=False))
            self.w = torch.nn.Parameter(torch.ones(4, 4))

    tiny = Tiny()
    cfg_opt = Namespace(op="adam", init_lr=1e-3)
    opt1 = tiny.get_optimizer(cfg_opt)
    opt2 = tiny.get_optimizer(cfg_opt)
    assert opt1.reducer is not opt2.reducer and not opt1.reducer.enabled and opt2.reducer.enabled
    (tiny.w * float(rank + 1)).sum().backward()
    opt2.reducer.wait()
    assert torch.allclose(tiny.w.grad, torch.full((4, 4), 3.0)), tiny.w.grad
    opt2.reducer.close()

    # 2. sharded loss terms == single-process global-batch values (reference golden)
    G = load_golden("divae_mid")
    cfg = golden_config(G)
    P = golden_params(G)
    B = G["in/outputs"].shape[0]
    b0, b1 = parallel.shard_rows(B)
    y, k = int(cfg.y_size), int(cfg.k)
    u = G["in/u_steps"][0][b0 * y:b1 * y]
    with torch.no_grad():
        res = O.divae_forward(P, cfg, G["in/outputs"][b0:b1], u, training=True)
    # 'word' NLL: the product criterion's row weights (1 / GLOBAL non-PAD count, via its all-reduce) applied to the
    # oracle's per-token NLL of this shard (the criterion's arithmetic itself only exists on the GPU)
    labels = torch.from_numpy(G["in/outputs"][b0:b1, 1:].astype(np.int64))
    crit = criterions.NLLEntropy(0, Namespace(avg_type="word", use_gpu=False))
    f0 = parallel.COLLECTIVES["forward"]
    scale = crit._row_scale(labels, labels.shape[0], labels.shape[1])
    assert parallel.COLLECTIVES["forward"] == f0 + 1        # standalone criterion: its own count exchange
    # the models fold that count into the latent-statistics message instead (one forward collective per step): a count
    # the caller already reduced is used as it is and costs no collective
    cnt = crit.local_count(labels)
    parallel.all_reduce_sum_(cnt)
    f1 = parallel.COLLECTIVES["forward"]
    scale2 = crit._row_scale(labels, labels.shape[0], labels.shape[1], global_count=cnt)
    assert parallel.COLLECTIVES["forward"] == f1 and torch.equal(scale, scale2)
    tok = -res["logp"].gather(2, labels.unsqueeze(2)).squeeze(2) * (labels != 0).float()
    nll_share = (tok * scale).sum().reshape(1)
    dist.all_reduce(nll_share)
    assert abs(float(nll_share) - float(G["out/nll"])) < 1e-5 * abs(float(G["out/nll"]))
    # BPR: all-reduce sum_b q, then the finalize formulas (laed_latent_finalize restated)
    q_sum = torch.exp(res["log_qy"]).view(-1, y, k).sum(0)
    parallel.all_reduce_sum_(q_sum)
    l = torch.log(q_sum / B + 1e-15)
    bpr = (torch.exp(l) * (l - np.log(1.0 / k))).sum(1).mean()
    assert abs(float(bpr) - float(G["out/bpr"])) < 1e-6 + 1e-4 * abs(float(G["out/bpr"]))
    parallel.shutdown()
    assert parallel.group() is None and parallel.world_size() == 1
    dist.destroy_process_group()
    open(os.path.join(result_dir, "ok%d" % rank), "w").write("ok")


def test_data_parallel_host_path_world2(tmp_path):
    mp.spawn(_worker, args=(2, str(tmp_path)), nprocs=2, join=True)
    # the rendezvous file may or may not outlive the group (the store removes it when its last user goes away)
    assert sorted(f for f in os.listdir(str(tmp_path)) if f.startswith("ok")) == ["ok0", "ok1"]
