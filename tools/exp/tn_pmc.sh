#!/bin/bash
# Where do the L2 misses of the A^T B (dW) product come from?  hit / miss per launch for a few variations.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3/tn_pmc
rm -rf $O && mkdir -p $O
export LAED_HIP_LIB=$GRAFT_REPO_ROOT/neuraldialog-laed_amd/laed_amd/liblaed_hip_hooks.so
run() {  # tag, env assignments..., -- args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  for e in "${envs[@]}"; do export "$e"; done
  timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/$tag -- python tools/exp/tn_case.py "$@" > $O/$tag.log 2>&1 || { echo FAILED $tag; tail -3 $O/$tag.log; exit 1; }
  for e in "${envs[@]}"; do unset "${e%%=*}"; done
}
run dW_plain -- 49152 1024 16384 50000 plain
run dW_full -- 49152 1024 16384 50000 full
run dW_dense -- 49152 1024 16384 49152 plain
run dW_g4 LAED_GEMM_GROUP_M=4 -- 49152 1024 16384 50000 plain
run dW_g16 LAED_GEMM_GROUP_M=16 -- 49152 1024 16384 50000 plain
run dW_g64 LAED_GEMM_GROUP_M=64 -- 49152 1024 16384 50000 plain
run one_round -- 8192 1024 16384 50000 plain
run one_round_dense -- 8192 1024 16384 8192 plain
run dW_lock16 LAED_GEMM_PHASE=16 -- 49152 1024 16384 50000 plain
python - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r3/tn_pmc"
for d in sorted(glob.glob(O + "/*/")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "sgemm_kernel" not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"].split("sgemm_kernel")[1][:44], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
        dur[(key, r["Counter_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    for key, c in sorted(acc.items()):
        h, m = c["TCC_HIT_sum"] / n[(key, "TCC_HIT_sum")], c["TCC_MISS_sum"] / n[(key, "TCC_MISS_sum")]
        print("%-16s %s wgs %5d  req %.3e  miss %.3e = %.1f GB  hit rate %.3f  ms %.2f" % (
            os.path.basename(d.rstrip("/")), key[0][:30], key[1], h + m, m, m * 128 / 1e9, h / (h + m),
            dur[(key, "TCC_HIT_sum")] / n[(key, "TCC_HIT_sum")]))
PY
