#!/bin/bash
# Round profile: rocprofv3 kernel stats of the bench command, PMC FETCH_SIZE / WRITE_SIZE passes (separate passes: gpurun
# refuses --pmc together with trace domains other than --kernel-trace), the same for the HBM-regime SpMV microbenchmark and
# the HBM-resident sweep.  Writes summaries under gpurun_out/prof_r04/ (copy what is to be judged into profiles/).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_stats() {  # name, args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 "$@" > $OUT/$name.log 2>&1
  cp $(ls $OUT/$name/*/*kernel_stats.csv | head -1) $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/$name
}
run_pmc() {    # name, counter, args...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${name}_$ctr -- python3 "$@" > $OUT/${name}_$ctr.log 2>&1
  python3 $R/tools/pmc_reduce.py $OUT/${name}_$ctr $ctr > $OUT/${name}_pmc_${ctr}_per_kernel.csv
  rm -rf $OUT/${name}_$ctr
}
run_stats bench $R/bench.py --no-cpu-baseline --no-roofline
run_pmc bench FETCH_SIZE $R/bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 0
run_pmc bench WRITE_SIZE $R/bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 0
run_stats spmv_hbm $R/tools/spmv_bench.py 1000000 30
run_pmc spmv_hbm FETCH_SIZE $R/tools/spmv_bench.py 1000000 8
run_pmc spmv_hbm WRITE_SIZE $R/tools/spmv_bench.py 1000000 8
run_stats sweep_hbm $R/tools/sweep_bench.py cfg3_hbm 20
run_pmc sweep_hbm FETCH_SIZE $R/tools/sweep_bench.py cfg3_hbm 8
run_stats sweep_short $R/tools/sweep_viol.py cfg4 8          # the sweep on 1e6 short rows (bench.py sweep_roofline_short_rows) and the materialising precompute!
cd $R
python3 bench.py > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 tools/spmv_bench.py 1000000 40 2>/dev/null | tail -1 > $OUT/spmv_hbm.json
python3 tools/sweep_bench.py cfg3_hbm 20 2>/dev/null | tail -1 > $OUT/sweep_hbm.json
ls -la $OUT
