cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_gpu_lzss.py -m gpu -x -q -k "front_end or decode" 2>&1 | tail -8) 2>&1
