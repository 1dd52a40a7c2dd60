#!/bin/bash
# L2 hit / miss and fetch bytes of the tiled GRU step kernel at H = 1024 (B = 256 .. 4096), one counter block per pass
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_gru
rm -rf $O && mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/hit -- python tools/gru_geom.py > $O/hit.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python tools/gru_geom.py > $O/fetch.log 2>&1 || exit 1
python - <<'PY'
import csv, glob, collections
for name in ("hit", "fetch"):
    f = glob.glob("gpurun_out/pmc_gru/%s/*/*counter_collection.csv" % name)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "gru_step_fwd" not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"])
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
        n[(key, r["Counter_Name"])] += 1
    for key, d in sorted(acc.items()):
        print(name, key, {c: v / n[(key, c)] for c, v in d.items()})
PY
