cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_lzss.py tests/test_gpu_full_size.py -m gpu -x -q -k "not huffman_1GiB and not config2 and not config5 and not beyond and not bench" 2>&1 | tail -4) 2>&1
timeout 300 python scripts/quick_lzss.py period 1024 2>&1 | grep -B14 "^decode" | grep -v amdgpu
