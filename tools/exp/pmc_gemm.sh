#!/bin/bash
# PMC comparison of GEMM main-loop variants on the dW shape (TN, M=50000 N=1024 K=8192).
#   gpurun --timeout 900 -- 'bash tools/exp/pmc_gemm.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_gemm; rm -rf $O; mkdir -p $O
L=$GRAFT_REPO_ROOT/tools/exp/lib
PA="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
PB="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD"
LAY=${LAY:-TN}; M=${M:-50000}; N=${N:-1024}; K=${K:-8192}
run() {  # name lib glds
  for pass in A B; do
    if [ $pass = A ]; then P="$PA"; else P="$PB"; fi
    LAED_HIP_LIB=$2 LAED_GEMM_GLDS=$3 timeout -s KILL 90 rocprofv3 --pmc $P --output-format csv -d $O/$1_$pass -- python tools/gemm_only.py $M $N $K $LAY > $O/$1_$pass.log 2>&1 || { echo "FAILED $1 $pass"; tail -3 $O/$1_$pass.log; exit 1; }
  done
}
run base $L/liblaed_exp0.so 0
run noload $L/liblaed_exp16.so 0
run noside $L/liblaed_exp2.so 0
run glds $GRAFT_REPO_ROOT/neuraldialog-laed_amd/laed_amd/liblaed_hip_hooks.so 1
python - <<'PY'
import csv, glob, collections
O = "gpurun_out/pmc_gemm"
tab = collections.OrderedDict()
for name in ("base", "noload", "noside", "glds"):
    acc = collections.defaultdict(float); n = collections.Counter()
    for ps in ("A", "B"):
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (O, name, ps), recursive=True):
            for r in csv.DictReader(open(f)):
                if "sgemm" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    tab[name] = {k: v / n[k] for k, v in acc.items()}
keys = sorted({k for t in tab.values() for k in t})
print("%-30s" % "counter (per launch)" + "".join("%14s" % n for n in tab))
for k in keys:
    print("%-30s" % k + "".join("%14.4g" % tab[n].get(k, float("nan")) for n in tab))
PY
