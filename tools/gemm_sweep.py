"""Sweep (tile, split-K) choices of the fp32 GEMM on the mid-size shapes of the reference's own configs (P: B=30,
VST: B=60; H=512, E=200, V~10004) against the launcher's heuristic.  Each forced choice runs in a child process
(LAED_GEMM_FORCE is read once per process).   usage: gemm_sweep.py            (parent)
                                                      gemm_sweep.py child <M> <N> <K> <lay>"""
import os, subprocess, sys
_PKG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "neuraldialog-laed_amd")
sys.path.insert(0, _PKG)
# LAED_GEMM_FORCE is an experiment hook: only the hooks build reads it (make -C neuraldialog-laed_amd/csrc HOOKS=1)
_HOOKS = os.path.join(_PKG, "laed_amd", "liblaed_hip_hooks.so")
if not os.path.exists(_HOOKS):
    raise SystemExit("gemm_sweep.py needs the hooks build: make -C neuraldialog-laed_amd/csrc HOOKS=1")
os.environ.setdefault("LAED_HIP_LIB", _HOOKS)

SHAPES = [("logits NT  P", 1170, 10004, 512, "NT"), ("logits NT  VST", 2340, 10004, 512, "NT"),
          ("dh NN      P", 1170, 512, 10004, "NN"), ("dh NN      VST", 2340, 512, 10004, "NN"),
          ("dW TN      P", 10004, 512, 1170, "TN"), ("dW TN      VST", 10004, 512, 2340, "TN"),
          ("gi NT      P", 1200, 1536, 200, "NT"), ("gi NT      VST", 2400, 1536, 200, "NT"),
          ("dW_hh TN   P", 1536, 512, 1200, "TN"), ("dW_hh TN   VST", 1536, 512, 2400, "TN"),
          ("dW_ih TN   P", 1536, 200, 1200, "TN"), ("dx NN      VST", 2400, 200, 1536, "NN")]


# the recurrence's per-timestep backward product dh = dgh W_hh at the S shape (H = 1024), per-GPU batches of 8/4/2 ranks
RECUR = [("dh_rec NN B=512", 512, 1024, 3072, "NN"), ("dh_rec NN B=1024", 1024, 1024, 3072, "NN"),
         ("dh_rec NN B=2048", 2048, 1024, 3072, "NN"), ("dh_rec NN B=4096", 4096, 1024, 3072, "NN")]


# the attribute head of AE-ED / ST-ED (softmax(dec_outs) @ x_embedding.weight and its backward; B = 30, V = 10004, E = 200)
ATTR = [("p Wemb NN", 1170, 200, 10004, "NN"), ("dWemb TN", 10004, 200, 1170, "TN"), ("dp NT", 1170, 10004, 200, "NT"),
        ("logits NT", 1170, 10004, 256, "NT"), ("dh NN", 1170, 256, 10004, "NN"), ("dW TN", 10004, 256, 1170, "TN")]


def child(M, N, K, lay):
    import torch
    from laed_amd import ops
    tA, tB = lay[0] == "T", lay[1] == "T"
    A = torch.randn((K, M) if tA else (M, K), device="cuda")
    B = torch.randn((N, K) if tB else (K, N), device="cuda")
    C = torch.empty((M, N), device="cuda")
    fn = lambda: ops.sgemm(A, B, transA=tA, transB=tB, out=C)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(50):
        fn()
    en.record()
    torch.cuda.synchronize()
    print("%.3f" % (st.elapsed_time(en) / 50 * 1e3))        # us


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
        sys.exit(0)
    for name, M, N, K, lay in (RECUR if "--recur" in sys.argv else ATTR if "--attr" in sys.argv else SHAPES):
        res = {}
        forces = [] if "--heuristic-only" in sys.argv else ["%d,%d" % (t, s) for t in (64, 128) for s in (1, 2, 3, 4, 6, 8, 12, 16)]
        for force in [None] + forces:
            env = dict(os.environ)
            env.pop("LAED_GEMM_FORCE", None)
            if force:
                env["LAED_GEMM_FORCE"] = force
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(M), str(N), str(K), lay], env=env,
                                 capture_output=True, text=True, timeout=120)
            try:
                res[force] = float(out.stdout.strip().split()[-1])
            except (ValueError, IndexError):
                res[force] = float("nan")
        if not forces:
            print("%-16s M=%5d N=%5d K=%5d  heuristic %7.1f us (%5.1f TF)" % (name, M, N, K, res[None], 2.0 * M * N * K / 1e6 / res[None]), flush=True)
            continue
        best = min((v, k) for k, v in res.items() if k is not None and v == v)
        gf = 2.0 * M * N * K / 1e6
        print("%-16s M=%5d N=%5d K=%5d  heuristic %7.1f us (%5.1f TF)   best %-7s %7.1f us (%5.1f TF)   all: %s" %
              (name, M, N, K, res[None], gf / res[None], best[1], best[0], gf / best[0],
               " ".join("%s=%.0f" % (k, v) for k, v in res.items() if k)), flush=True)
