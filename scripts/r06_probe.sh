cd $GRAFT_REPO_ROOT
for w in 3 4; do
echo "--- config $w quick"; timeout 600 python bench.py --profile-only $w 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['profile_only']['$w']
print({k:d[k] for k in ('encode_ms','decode_ms','encode_ms_all','kernels_encode_ms','lossless','bit_exact_vs_oracle_on_sample')})"
done
echo "--- lzss tests"; timeout 1200 python -m pytest tests/test_gpu_lzss.py -m gpu -x -q -k "periodic or period" 2>&1 | tail -4
