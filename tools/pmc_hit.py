#!/usr/bin/env python3
"""L2 hit / miss per kernel launch shape from one rocprofv3 PMC pass (TCC_HIT_sum TCC_MISS_sum).

    python tools/pmc_hit.py gpurun_out/final/pmc_hit profiles/r3_pmc_l2_hit.json

Requests are 128-byte lines at the L2 (TCC); misses x 128 B is the fabric read traffic FETCH_SIZE reports.
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
        k = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_HIT_sum":
            n[k] += 1
    rows = []
    for k, c in acc.items():
        h, m = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
        if h + m <= 0:
            continue
        rows.append({"kernel": k[0], "workgroups": k[1], "launches": n[k], "requests_per_launch": round((h + m) / n[k]),
                     "misses_per_launch": round(m / n[k]), "miss_bytes_per_launch": round(m / n[k] * 128),
                     "hit_rate": round(h / (h + m), 4)})
    rows.sort(key=lambda r: -r["miss_bytes_per_launch"] * r["launches"])
    json.dump({"source": os.path.relpath(f), "counters": "TCC_HIT_sum, TCC_MISS_sum (128-byte requests, summed over the L2 channels)",
               "kernels": rows[:30]}, open(out, "w"), indent=1)
    for r in rows[:12]:
        print("%-58s wgs %6d  launches %4d  hit rate %.3f  miss %.2f GB/launch" % (
            r["kernel"][:58], r["workgroups"], r["launches"], r["hit_rate"], r["miss_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
