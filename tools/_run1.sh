cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -rf -p no:cacheprovider > gpurun_out/r3_gputests5.log 2>&1; tail -n 6 gpurun_out/r3_gputests5.log
timeout -k 10 300 python tests/tools/fuzz_small.py 0 120 2>&1 | tail -n 1
timeout -k 10 300 python tests/tools/fuzz_small.py 7 120 2>&1 | tail -n 1
timeout -k 10 300 python tools/dev.py shapes 2>&1 | tail -n 1
timeout -k 10 100 python tools/dev.py seeds cfg3_qp 0 4 2>&1 | tail -n 1 | cut -c1-200
timeout -k 10 100 python tools/dev.py seeds cfg2_qp 0 8 2>&1 | tail -n 1 | cut -c1-200
