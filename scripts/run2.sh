cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_gpu_lzss.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -m gpu -x -q -k "not huffman and not config2 and not config5 and not bench and not beyond" 2>&1 | tail -4) 2>&1
