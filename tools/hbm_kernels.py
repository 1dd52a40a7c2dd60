#!/usr/bin/env python3
"""Achieved HBM GB/s of the bandwidth-bound kernels of the S-fixed DI-VAE step, from a rocprofv3 kernel-stats CSV.

    python tools/hbm_kernels.py gpurun_out/prof_s6 profiles/r1_hbm_kernels.json

Algorithmic bytes are what each kernel must move per train step at the S-fixed shape
(V=50000, E=200, H=1024, y=20, k=10, T=40, B=4096); achieved = those bytes / the kernel's summed device time per
step in the profile; peak = 8 TB/s (MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md).  The number of steps in
the profile is read off the GRU step launches (79 per step: 40 encoder + 39 decoder).
"""
import csv
import glob
import json
import os
import re
import sys

B, T, E, H, V, Y, K = 4096, 40, 200, 1024, 50000, 20, 10
LDL = (V + 31) // 32 * 32
N_TOK = B * (T - 1)
PARAMS = V * E + 2 * (3 * H * E + 3 * H * H + 6 * H) + H * Y * K + Y * K + Y * K * H + V * H + V + Y * K   # incl. log_py
PEAK = 8000.0   # GB/s

# kernel-name pattern -> (algorithmic bytes per step, what is counted)
KERNELS = [
    (r"proj_rows_reg_kernel|proj_rows_kernel", N_TOK * LDL * 4 * 2,
     "softmax-CE rows: read logits once, write (softmax - onehot) * scale once, [N][ldl] fp32"),
    (r"gru_bwd_gates_kernel", (B * T + B * (T - 1)) * H * 4 * 14,
     "GRU gate gradients: 7 floats read + 7 written per (b, t, unit)"),
    (r"adam_kernel", PARAMS * 28, "Adam: read p, g, m, v; write p, m, v"),
    (r"embed_gather_kernel", (B * T + B * (T - 1)) * E * 4 * 2, "embedding rows gathered and written (enc + dec inputs)"),
    (r"embed_scatter_kernel|embed_scatter_small_kernel|sc_keys_kernel|sc_chunks_kernel|sc_partials_kernel|rocprim.*radix|rocprim.*onesweep|rocprim.*sort",
     (B * T + B * (T - 1)) * E * 4 * 2,
     "embedding gradient rows read + added into the table (sorted scatter-add: keys, radix sort, chunk sums, per-id pass)"),
    (r"gumbel_latent_fwd_kernel", B * Y * K * 4 * 4 + B * Y * 8, "logits, u in; y, log_qy, ids out"),
    (r"gumbel_latent_bwd_kernel", B * Y * K * 4 * 5, "y, log_qy, dy, dlog_qy in; dlogits out"),
    (r"latent_stats_kernel", B * Y * K * 4, "log_qy read once (batch-mean posterior, entropies)"),
    (r"latent_bwd_kernel", B * Y * K * 4 * 2 * 2, "log_qy in, dlog_qy out (two calls per step)"),
    (r"keep_mask_kernel", B * (T - 1) * E, "dropout keep bytes written"),
]


def main():
    d, out = sys.argv[1], sys.argv[2]
    f = max(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    gru = sum(int(r["Calls"]) for r in rows if re.search(r"gru_step_fwd_kernel|gru_small_fwd_kernel", r["Name"]))
    steps = gru / 79.0
    res = []
    for pat, nbytes, what in KERNELS:
        sel = [r for r in rows if re.search(pat, r["Name"])]
        if not sel:
            continue
        ns = sum(int(r["TotalDurationNs"]) for r in sel) / steps
        calls = sum(int(r["Calls"]) for r in sel) / steps
        gbs = nbytes / ns
        res.append({"kernel": pat.split("|")[0], "launches_per_step": round(calls, 1), "device_ms_per_step": round(ns / 1e6, 3),
                    "algorithmic_bytes_per_step": int(nbytes), "achieved_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 3),
                    "counts": what})
    json.dump({"source": os.path.relpath(f), "steps_in_profile": steps, "peak_GBps": PEAK, "kernels": res}, open(out, "w"), indent=1)
    for r in res:
        print("%-28s %6.1f launches/step %8.3f ms/step %8.1f GB/s  (%.2f of peak)" % (
            r["kernel"], r["launches_per_step"], r["device_ms_per_step"], r["achieved_GBps"], r["frac_of_8TBps"]))


if __name__ == "__main__":
    main()
