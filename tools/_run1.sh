cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_lp.py tests/test_gpu_ecp.py tests/test_distributed.py -m gpu -x -q -p no:cacheprovider -k "tiled or million or cfg4" 2>&1 | tail -n 3 | cut -c1-300
timeout -k 10 300 python tools/spmv_bench.py 1000000 40 2>/dev/null | tail -n 1 | cut -c1-1500
timeout -k 10 300 python tools/dev.py seeds cfg4 0 4 2>&1 | tail -n 1 | cut -c1-200
