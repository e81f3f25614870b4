cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_huffman_decode.py -m gpu -x -q 2>&1 | tail -12) 2>&1
