cd $GRAFT_REPO_ROOT
gcc -O1 -shared -fPIC tests/pthread_fail_shim.c -o /tmp/shim.so -ldl 2>/dev/null
gcc -O1 -std=gnu11 -pthread -I include tests/abi_shim_test.c -o /tmp/abi -L raisin_amd -lrsn -ldl -Wl,-rpath,$PWD/raisin_amd 2>/dev/null
echo "--- threadfail"; LD_PRELOAD=/tmp/shim.so timeout 600 /tmp/abi threadfail 2>&1 | tail -5
echo "--- config 3 quick"; timeout 600 python bench.py --profile-only 3 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['profile_only']['3']
print({k:d[k] for k in ('encode_ms','decode_ms','encode_ms_all','decode_ms_all','kernels_encode_ms','kernels_decode_ms','lossless','bit_exact_vs_oracle_on_sample')})"
echo "--- periodic tests"; timeout 900 python -m pytest tests/test_gpu_lzss.py -m gpu -x -q -k "periodic_tail or run_tiles or decode_paths or decode_errors" 2>&1 | tail -15
