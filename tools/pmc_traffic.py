#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into
profiles/<round>_pmc_traffic.json: per-kernel, per-launch L2<->fabric bytes.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1_pmc_traffic.json

Corrections, per /opt/skills/guides/MI355X_MICROARCH.md "HBM [CDNA4]": the counters are in KiB (bytes = value*1024);
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) coalesced reads at 64 B, so the read side
is doubled (checked on this run's adam_kernel: 4 arrays read / 3 written, raw FETCH/WRITE = 0.667 -> 1.333 after
doubling); WRITE_SIZE is exact for 16 B/lane stores (checked: the logits GEMM writes 8192*50016*4 = 1.639e9 B per
launch, WRITE_SIZE*1024 = 1.638e9).  Infinity-Cache hits are included in both counters, so `traffic` is an upper
bound on HBM bytes.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

VARIANT_OF = [   # same order as bench.py VARIANTS / laed_prof_collect
    ("sgemm_128x128_NT", r"sgemm_kernel<128, 128, false, true"),
    ("sgemm_64x64_NT", r"sgemm_kernel<64, 64, false, true"),
    ("sgemm_128x128_NN", r"sgemm_kernel<128, 128, false, false"),
    ("sgemm_64x64_NN", r"sgemm_kernel<64, 64, false, false"),
    ("sgemm_128x128_TN", r"sgemm_kernel<128, 128, true, false"),
    ("sgemm_64x64_TN", r"sgemm_kernel<64, 64, true, false"),
    ("xgemm_128x128_NT_bf16x3", r"xgemm_kernel<"),
]


# The projection products of the S-fixed job (V = 50000, H = 1024, 40960-row chunks), by (variant, workgroups of the launch):
# (M, N, K) -> algorithmic bytes (every operand read once, the result written once; dW is read-modify-write) and the bytes
# THIS TILING needs at best: 128x128 tiles, 8 x 8 tiles resident per XCD, so the long operand is read once and the short
# one (N = 1024 wide) once per group of 8 row tiles.  A measured figure well above the second number means the blocks of
# a group have drifted out of phase and re-fetch strips their group already evicted.
KNOWN_SHAPES = {
    ("sgemm_128x128_NT", 125120): ("logits = h W^T + b", 40960, 50000, 1024),
    ("sgemm_128x128_NN", 2560): ("dh = dl W", 40960, 1024, 50000),
    ("sgemm_128x128_TN", 3072): ("dW += dl^T h (whole-rounds part)", 49152, 1024, 40960),
}


def shape_bytes(variant, workgroups):
    k = KNOWN_SHAPES.get((variant, workgroups))
    if k is None:
        return None
    what, M, N, K = k
    if variant.endswith("NT"):      # both operands short (0.37 GB), the [M][ldl] tile written
        algo = 4.0 * (M * K + N * K + M * 50016)
        ideal = algo
    elif variant.endswith("NN"):
        algo = 4.0 * (M * K + K * N + M * N)
        ideal = 4.0 * (M * K + (M // 128 // 8) * K * N + M * N)
    else:
        algo = 4.0 * (K * M + K * N + 2 * M * N)
        ideal = 4.0 * (K * M + (M // 128 // 8) * K * N + 2 * M * N)
    return {"product": what, "M": M, "N": N, "K": K, "algorithmic_bytes": round(algo), "tiling_ideal_bytes": round(ideal)}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].strip()


def load(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under %s" % d)
    newest = max(files, key=os.path.getmtime)        # gpurun merges runs into one tree: take the latest pass
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(newest)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        a = acc[k]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return acc


def main():
    fetch, write, out = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    rows = []
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0, 0])[1] + write.get(k, [0, 0, 0])[1])):
        f, w = fetch.get(k, [0, 0.0, 0.0]), write.get(k, [0, 0.0, 0.0])
        n = max(f[0], w[0])
        rows.append({"kernel": k[0], "workgroups": k[1], "launches": n,
                     "fetch_bytes_per_launch": round(2.0 * 1024.0 * f[1] / max(f[0], 1)),
                     "write_bytes_per_launch": round(1024.0 * w[1] / max(w[0], 1)),
                     "avg_us_under_pmc": round((f[2] + w[2]) / max(f[0] + w[0], 1), 1)})
    variants = {}
    for vname, pat in VARIANT_OF:
        sel = [r for r in rows if re.search(pat, r["kernel"])]
        n = sum(r["launches"] for r in sel)
        if n:
            # a variant covers dissimilar launches (the projection product, its split-K tail, the GRU products): the
            # average over them means little, so the launch shape that moves the most bytes is reported on its own
            top = max(sel, key=lambda r: (r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"]) * r["launches"])
            extra = {}
            sb = shape_bytes(vname, top["workgroups"])
            if sb is not None:
                moved = top["fetch_bytes_per_launch"] + top["write_bytes_per_launch"]
                extra = dict(sb, ratio_to_algorithmic=round(moved / sb["algorithmic_bytes"], 2),
                             ratio_to_tiling_ideal=round(moved / sb["tiling_ideal_bytes"], 2))
            variants[vname] = {
                "launches": n,
                "bytes_per_launch": round(sum((r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"]) * r["launches"]
                                              for r in sel) / n),
                "largest_launch": {"workgroups": top["workgroups"], "launches": top["launches"],
                                   "fetch_bytes_per_launch": top["fetch_bytes_per_launch"],
                                   "write_bytes_per_launch": top["write_bytes_per_launch"],
                                   "bytes_per_launch": top["fetch_bytes_per_launch"] + top["write_bytes_per_launch"],
                                   **extra}}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over "
                         "`python bench.py --steps 2 --warmup 1 --no-alt --cpu-sample 0`",
               "corrections": "bytes = KiB*1024; FETCH_SIZE doubled (gfx950, 16 B/lane reads); includes Infinity-Cache hits",
               "variants": variants, "kernels": rows[:40]}, open(out, "w"), indent=1)
    for v, d in variants.items():
        ll = d["largest_launch"]
        print("%-26s launches=%5d  %.3f GB/launch  largest: %.2f GB%s" % (
            v, d["launches"], d["bytes_per_launch"] / 1e9, ll["bytes_per_launch"] / 1e9,
            "  = %.2fx algorithmic, %.2fx this tiling's best" % (ll["ratio_to_algorithmic"], ll["ratio_to_tiling_ideal"])
            if "ratio_to_algorithmic" in ll else ""))


if __name__ == "__main__":
    main()
