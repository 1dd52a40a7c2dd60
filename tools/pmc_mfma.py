#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel from one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE).

    python tools/pmc_mfma.py gpurun_out/pmc_mfma profiles/r1_pmc_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES sums, over the chip's 1024 SIMDs, the cycles their matrix pipe was busy (it equals
#MFMA x 64 for v_mfma_f32_32x32x2_f32); GRBM_GUI_ACTIVE is reported summed over the 8 XCDs.  So
busy fraction = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024), per kernel over all its launches in the pass.
"""
import collections, csv, glob, json, os, re, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].strip()


def main():
    d, out = sys.argv[1], sys.argv[2]
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[k] += 1
    rows = []
    for k, c in acc.items():
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or c.get("GRBM_GUI_ACTIVE", 0) <= 0:
            continue
        rows.append({"kernel": k, "launches": n[k], "mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"],
                     "gui_active_cycles_sum_8xcd": c["GRBM_GUI_ACTIVE"],
                     "mfma_pipe_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)})
    rows.sort(key=lambda r: -r["mfma_busy_cycles"])
    json.dump({"source": os.path.relpath(f), "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)",
               "kernels": rows}, open(out, "w"), indent=1)
    for r in rows:
        print("%-60s launches=%5d  MFMA pipe busy %.3f" % (r["kernel"][:60], r["launches"], r["mfma_pipe_busy_frac"]))


if __name__ == "__main__":
    main()
