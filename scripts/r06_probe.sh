cd $GRAFT_REPO_ROOT
RSN_FUZZ=4 timeout 2000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "periodic_tail_fuzz" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_huffman_small.py -m gpu -x -q 2>&1 | tail -3
