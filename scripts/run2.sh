cd $GRAFT_REPO_ROOT
timeout 120 python scripts/dbg_dec.py 2>&1 | grep -v amdgpu.ids | tail -9
(timeout 1200 python -m pytest tests/test_gpu_lzss.py tests/test_gpu_fuzz.py tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -8) 2>&1
for k in text period random; do timeout 300 python scripts/quick_lzss.py $k 1024 2>&1 | grep -A12 "^decode"; done
