cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tools/dev.py shapes > gpurun_out/r3_shapes.log 2>&1
timeout -k 10 300 python tests/tools/fuzz_small.py 0 120 > gpurun_out/r3_fuzz.log 2>&1
timeout -k 10 300 python tests/tools/fuzz_small.py 5 120 >> gpurun_out/r3_fuzz.log 2>&1
tail -n 2 gpurun_out/r3_shapes.log gpurun_out/r3_fuzz.log
