#!/bin/bash
# Runs on the GPU box (via gpurun): every measurement profiles/ is built from, into gpurun_out/final/.
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh'
# then, back in the build container:  python tools/publish_profiles.py r3
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf $O && mkdir -p $O
step() { echo "== $*" >> $O/steps.log; "$@"; rc=$?; echo "rc=$rc" >> $O/steps.log; if [ $rc -ne 0 ]; then echo "FAILED: $*"; exit $rc; fi; }
# 1. bench lines (default run = the driver's command; then the other workloads)
step timeout -k 10 500 python bench.py > $O/bench_S-fixed.json 2> $O/bench_S-fixed.err
step timeout -k 10 200 python bench.py --global-batch 512 --no-alt --cpu-sample 0 > $O/bench_S-fixed_b512.json 2>> $O/bench.err
step timeout -k 10 200 python bench.py --ids uniform --no-alt --cpu-sample 0 > $O/bench_S-fixed_uniform.json 2>> $O/bench.err
for w in P-fixed P-varlen VST-fixed; do
  step timeout -k 10 200 python bench.py --workload $w --steps 50 --warmup 10 --no-alt > $O/bench_$w.json 2>> $O/bench.err
done
for w in AE-fixed ST-fixed; do for ph in A B; do
  step timeout -k 10 200 python bench.py --workload $w --phase $ph --steps 50 --warmup 10 --cpu-sample 0 > $O/bench_${w}_$ph.json 2>> $O/bench.err
done; done
# 2. rocprofv3 kernel stats of the S-fixed step (same command line as the live-roofline line it is compared with)
step timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_S-fixed -- python bench.py --steps 3 --warmup 1 --no-alt --cpu-sample 0 > $O/prof_S-fixed.json 2> $O/prof_S-fixed.err
for w in P-fixed VST-fixed; do
  step timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python bench.py --workload $w --steps 20 --warmup 5 --no-alt --cpu-sample 0 > $O/prof_$w.json 2> $O/prof_$w.err
done
# 3. PMC passes, one counter block per run
step timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 2 --warmup 1 --no-alt --cpu-sample 0 --no-prof > $O/pmc_fetch.log 2>&1
step timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 2 --warmup 1 --no-alt --cpu-sample 0 --no-prof > $O/pmc_write.log 2>&1
step timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python bench.py --steps 2 --warmup 1 --no-alt --cpu-sample 0 --no-prof > $O/pmc_mfma.log 2>&1
step timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_hit -- python bench.py --steps 2 --warmup 1 --no-alt --cpu-sample 0 --no-prof > $O/pmc_hit.log 2>&1
# 4. launch-by-launch view of one step of the reference's own script shapes (tools/exp/trace_step.py)
for w in P-fixed VST-fixed; do
  python tools/exp/trace_step.py $O/prof_$w > $O/step_$w.txt 2>&1 || true
done
# 5. the N > 1 path on this one GPU (rehearsal: gloo, two ranks share the device), started by bench.py itself
step timeout -k 10 300 python bench.py --gpus 2 --backend gloo --global-batch 128 --steps 5 --warmup 2 --no-alt > $O/bench_dp2_rehearsal.json 2> $O/bench_dp2_rehearsal.err
rm -f $O/*/*/*kernel_trace.csv        # large; the stats CSV is what profiles/ keeps
echo done
