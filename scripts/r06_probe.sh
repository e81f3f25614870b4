cd $GRAFT_REPO_ROOT
echo '--- tests'; timeout 1500 python -m pytest tests/test_gpu_lzss.py -m gpu -x -q -k "escape or alphabets or periodic or fixtures" 2>&1 | tail -4
for w in 3 4; do
echo "--- config $w quick"; timeout 600 python bench.py --profile-only $w 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['profile_only']['$w']
print({k:d[k] for k in ('encode_ms','encode_ms_all','lossless','bit_exact_vs_oracle_on_sample')}, d['kernels_encode_ms'].get('lzss_esc_check'))"
done
