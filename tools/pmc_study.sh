#!/bin/bash
# Per-kernel PMC study of the cfg3 bench's LP kernels (one or two counters per pass, --kernel-trace only).
# Writes gpurun_out/pmc_study/<COUNTER>.csv (per-kernel means, tools/pmc_reduce.py).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_study
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
  tag=$(echo $pass | tr ' ' '_')
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/run_$tag -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 0 > $OUT/$tag.log 2>&1
  for c in $pass; do python3 $R/tools/pmc_reduce.py $OUT/run_$tag $c | head -8 > $OUT/$c.csv; done
  rm -rf $OUT/run_$tag
done
ls $OUT
