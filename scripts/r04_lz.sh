#!/bin/bash
# GPU box: LZSS tests + config 3 / config 4 timings through bench.py's own path
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_lzss.py tests/test_gpu_fuzz.py tests/test_gpu_huffman_decode.py tests/test_gpu_engine.py -x -q 2>&1 | tail -8) > gpurun_out/r04_lz_tests.log 2>&1
tail -8 gpurun_out/r04_lz_tests.log
(timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -k "config3 or config4 or 1GiB_text" 2>&1 | tail -5)
timeout 600 python bench.py --profile-only 3,4 > gpurun_out/r04_lz_bench.json 2> gpurun_out/r04_lz_bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r04_lz_bench.json").read().strip().splitlines()[-1])["profile_only"]
for k,v in j.items():
    print(k, {x:v.get(x) for x in ("encode_ms","decode_ms","lossless","bit_exact_vs_oracle_on_sample","error")})
    print("   enc", v.get("kernels_encode_ms")); print("   dec", v.get("kernels_decode_ms"))
PY
echo "== no fused periodic"; RSN_LZSS_NO_FUSED_PERIODIC=1 timeout 600 python bench.py --profile-only 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])['profile_only']['3']; print(j['encode_ms'], j['kernels_encode_ms'])"
