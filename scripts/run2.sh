cd $GRAFT_REPO_ROOT
timeout 120 python scripts/dbg_dec.py 2>&1 | grep -v amdgpu.ids | tail -12
(timeout 900 python -m pytest tests/test_gpu_lzss.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -15) > gpurun_out/t2.log 2>&1
tail -5 gpurun_out/t2.log
for k in text period random; do timeout 300 python scripts/quick_lzss.py $k 1024 2>&1 | grep -A30 "^decode"; done
