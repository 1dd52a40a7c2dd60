#!/bin/bash
# A/B of one kernel (AB_KERNEL = substring of its name; default proj_rows_reg_kernel) between the product library and the hooks library (whatever differs between the two builds),
# alternated on one box: kernel average from rocprofv3 --stats over a short S-fixed run, and the step time.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/rows_ab; rm -rf $O; mkdir -p $O
HOOKS=$GRAFT_REPO_ROOT/neuraldialog-laed_amd/laed_amd/liblaed_hip_hooks.so
for i in 1 2; do for lib in product hooks; do
  if [ $lib = hooks ]; then export LAED_HIP_LIB=$HOOKS; else unset LAED_HIP_LIB; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${lib}_$i -- python bench.py --steps 4 --warmup 2 --no-alt --cpu-sample 0 > $O/${lib}_$i.json 2> $O/${lib}_$i.err || exit 1
  rm -f $O/${lib}_$i/*/*kernel_trace.csv
  python - $O/${lib}_$i $O/${lib}_$i.json $lib <<'PY'
import csv, glob, json, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0])))
import os
pat = os.environ.get("AB_KERNEL", "proj_rows_reg_kernel")
r = [x for x in rows if pat in x["Name"]][0]
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(("%-8s " + pat[:60] + " avg %.1f us (%s calls)   step %.2f ms  nll %.5f") % (sys.argv[3], float(r["AverageNs"]) / 1e3, r["Calls"], d["ms_per_step"], d["final_nll"]))
PY
done; done
