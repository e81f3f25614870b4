cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25) > gpurun_out/t_full.log 2>&1
tail -25 gpurun_out/t_full.log
(timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_r03b.json 2> gpurun_out/bench_r03b.err; echo rc=$?)
python - <<'PY'
import json
j=json.loads(open("gpurun_out/bench_r03b.json").read().strip().splitlines()[-1])
print({k:j[k] for k in ("value","ms_per_step","encode_ms","decode_ms","roofline")})
print({k:(v.get("ms"),v.get("frac_of_hbm_peak")) for k,v in j["kernels"].items()})
for k,v in j["other_configs"].items():
    print(k, {x:v.get(x) for x in ("encode_ms","decode_ms","lossless","bit_exact_vs_oracle_on_sample","per_chunk_ms","pass_ms","error")})
PY
