#!/bin/bash
# L2 <-> fabric bytes and L2 hit / miss of the projection's dW / dh products with and without the phase lock
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3/phase_pmc
rm -rf $O && mkdir -p $O
export LAED_HIP_LIB=$GRAFT_REPO_ROOT/neuraldialog-laed_amd/laed_amd/liblaed_hip_hooks.so
for ph in ${PHASES:-0 64 32}; do
  export LAED_GEMM_PHASE=$ph
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$ph -- python tools/exp/proj_gemms.py > $O/fetch_$ph.log 2>&1 || { tail -3 $O/fetch_$ph.log; exit 1; }
  timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/hit_$ph -- python tools/exp/proj_gemms.py > $O/hit_$ph.log 2>&1 || { tail -3 $O/hit_$ph.log; exit 1; }
done
python - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r3/phase_pmc"
for d in sorted(glob.glob(O + "/*_*/")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "sgemm_kernel" not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"].split("sgemm_kernel")[1][:44], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
        dur[(key, r["Counter_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    for key, c in sorted(acc.items()):
        out = {}
        for cn, v in c.items():
            v /= n[(key, cn)]
            out[cn] = "%.2f GB" % (v * 2 * 1024 / 1e9) if cn == "FETCH_SIZE" else "%.3e" % v
        if "TCC_HIT_sum" in out:
            hsum, msum = c["TCC_HIT_sum"], c["TCC_MISS_sum"]
            out["hit_rate"] = "%.3f" % (hsum / (hsum + msum))
        cn0 = list(c)[0]
        print(os.path.basename(d.rstrip("/")), key, out, "ms %.2f" % (dur[(key, cn0)] / n[(key, cn0)]))
PY
