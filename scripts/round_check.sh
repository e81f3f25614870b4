#!/bin/bash
# runs on the GPU box (gpurun): from the snapshot's root, or from the current directory when started by hand
# usage: scripts/round_check.sh <tag> [notests]   -- the -m gpu suite, bench.py's line, the per-workload profiles (headline, 2b, config3, config4, skewed)
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
mkdir -p gpurun_out
TAG=${1:-r05a}
if [ "$2" != "notests" ]; then
  (timeout 2400 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -12) > gpurun_out/t_full.log 2>&1
  tail -12 gpurun_out/t_full.log
fi
(timeout 1500 python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo rc=$?)
python - $TAG <<'PY'
import json,sys
j=json.loads(open("gpurun_out/%s_bench.json" % sys.argv[1]).read().strip().splitlines()[-1])
print({k:j[k] for k in ("value","ms_per_step","encode_ms","decode_ms","roofline")})
print({k:(v.get("ms"),v.get("frac_of_hbm_peak")) for k,v in j["kernels"].items()})
for k,v in j["other_configs"].items():
    print(k, {x:v.get(x) for x in ("encode_ms","decode_ms","encode_ms_min","decode_ms_min","lossless","bit_exact_vs_oracle_on_sample","encode_frac_of_hbm_peak","decode_frac_of_hbm_peak","error","huffman_2a","lzss_text")})
for k in ("general_path_2a","cold_start"):
    print(k, j.get(k))
PY
for L in headline 2b config3 config4 skewed; do bash scripts/profile.sh $TAG $L > gpurun_out/prof_${TAG}_${L}.txt 2>&1; tail -2 gpurun_out/prof_${TAG}_${L}.txt; done
cp gpurun_out/${TAG}_bench.json profiles/${TAG}_bench.json
python3 scripts/traffic_report.py $TAG > profiles/${TAG}_traffic.txt 2>&1
mkdir -p gpurun_out/profiles_new && cp profiles/${TAG}_* gpurun_out/profiles_new/ && cp gpurun_out/${TAG}_bench.json gpurun_out/profiles_new/
ls gpurun_out/profiles_new | grep $TAG
