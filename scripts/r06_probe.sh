cd $GRAFT_REPO_ROOT
for e in 0 1; do
echo "--- config 3 quick, no overlap = $e"; if [ $e = 1 ]; then export RSN_LZSS_NO_HEAD_OVERLAP=1; fi; timeout 600 python bench.py --profile-only 3 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['profile_only']['3']
print({k:d[k] for k in ('encode_ms','decode_ms','encode_ms_all','kernels_encode_ms','lossless','bit_exact_vs_oracle_on_sample')})"
done
