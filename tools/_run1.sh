cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/dev.py seeds cfg3 0 16 2>&1 | tail -n 1 | cut -c1-160
timeout -k 10 200 python tools/dev.py seeds cfg2_qp 0 4 2>&1 | tail -n 1 | cut -c1-160
timeout -k 10 200 python tools/dev.py seeds cfg3_qp 0 2 2>&1 | tail -n 1 | cut -c1-160
KTN_IPC_TIMEOUT_S=5 timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 2
