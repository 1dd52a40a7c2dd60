#!/usr/bin/env python3
"""Copy / reduce gpurun_out/final (tools/collect_profiles.sh) into profiles/<round>_*.   usage: publish_profiles.py r1"""
import glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r1"


def json_line(path):
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no JSON line in " + path)


for name in ("S-fixed", "S-fixed_b512", "S-fixed_uniform", "P-fixed", "P-varlen", "VST-fixed",
             "AE-fixed_A", "AE-fixed_B", "ST-fixed_A", "ST-fixed_B"):
    d = json_line(os.path.join(SRC, "bench_%s.json" % name))
    json.dump(d, open(os.path.join(DST, "%s_bench_%s_1gpu.json" % (rnd, name)), "w"), indent=1)
    print("%-14s %10.1f tokens/s  %8.3f ms/step" % (name, d["value"], d["ms_per_step"]))
for name in ("S-fixed", "P-fixed", "VST-fixed"):
    f = max(glob.glob(os.path.join(SRC, "prof_" + name, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    shutil.copy(f, os.path.join(DST, "%s_bench_%s_1gpu_kernel_stats.csv" % (rnd, name)))
    json.dump(json_line(os.path.join(SRC, "prof_%s.json" % name)),
              open(os.path.join(DST, "%s_bench_%s_1gpu_under_rocprof.json" % (rnd, name)), "w"), indent=1)
py = sys.executable
subprocess.check_call([py, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(SRC, "pmc_fetch"),
                       os.path.join(SRC, "pmc_write"), os.path.join(DST, rnd + "_pmc_traffic.json")])
subprocess.check_call([py, os.path.join(ROOT, "tools", "pmc_mfma.py"), os.path.join(SRC, "pmc_mfma"),
                       os.path.join(DST, rnd + "_pmc_mfma.json")])
for extra in ("step_P-fixed.txt", "step_VST-fixed.txt"):
    if os.path.exists(os.path.join(SRC, extra)):
        shutil.copy(os.path.join(SRC, extra), os.path.join(DST, "%s_%s" % (rnd, extra)))
if os.path.exists(os.path.join(SRC, "bench_dp2_rehearsal.json")):
    json.dump(json_line(os.path.join(SRC, "bench_dp2_rehearsal.json")), open(os.path.join(DST, rnd + "_bench_dp2_rehearsal.json"), "w"), indent=1)
subprocess.check_call([py, os.path.join(ROOT, "tools", "hbm_kernels.py"), os.path.join(SRC, "prof_S-fixed"),
                       os.path.join(DST, rnd + "_hbm_kernels.json")])
