"""world_size-2 data-parallel host path on CPU (gloo): gradient reducer, sharding, and the two forward
exchanges (global non-PAD count for 'word' NLL, batch-mean posterior for BPR) reproduce single-process values."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, result_dir):
    for p in (ROOT, os.path.join(ROOT, "neuraldialog-laed_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from argparse import Namespace
    from helpers import golden_config, golden_params, load_golden
    from laed_amd import criterions, parallel
    from oracle import laed_oracle as O

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    parallel.init()
    assert parallel.world_size() == world and parallel.rank() == rank and parallel.group() is not None
    assert parallel.shard_rows(12) == (rank * 6, rank * 6 + 6)

    # 1. overlapped gradient reducer: SUM over ranks, fired from post-accumulate-grad hooks
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    red = parallel.GradReducer(params, bucket_bytes=16)
    sum((p * float(rank + 1)).sum() for p in params).backward()
    red.wait()
    for p in params:
        assert torch.allclose(p.grad, torch.full_like(p, 3.0))

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
    scale = crit._row_scale(labels, labels.shape[0], labels.shape[1])
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
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(str(tmp_path))) == ["ok0", "ok1"]
