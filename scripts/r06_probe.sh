cd $GRAFT_REPO_ROOT
echo '--- tests'; timeout 1500 python -m pytest tests/test_gpu_huffman_decode.py tests/test_gpu_huffman_encode.py tests/test_gpu_fuzz.py tests/test_gpu_host_pipeline.py tests/test_gpu_shapes.py -m gpu -x -q 2>&1 | tail -6
for e in 0 1; do
if [ $e = 1 ]; then export RSN_DEC_SYNC1=1; fi
for w in skewed 4 2b; do
echo "--- $w sync1=$e"; timeout 600 python bench.py --profile-only $w 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['profile_only']['$w']
print({k:d[k] for k in ('decode_ms','decode_ms_all','kernels_decode_ms','lossless','bit_exact_vs_oracle_on_sample')})"
done; done
