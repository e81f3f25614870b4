#!/bin/bash
# For the first box with more than one GPU (none has been available to this repository yet): everything that exists for N > 1, in one go.
#   1. bench.py --gpus N over RCCL (one rank per GPU; the line's value is one chunk per rank, config5_dealt is configs[4]'s eight chunks
#      dealt over the ranks + the segment gather) for N = 2, 4, 8 as far as the box goes
#   2. tests/test_gpu_full_size.py::test_bench_two_ranks_over_rccl (skipped on one GPU)
#   3. one Huffman stream from slices on DIFFERENT devices (RSN_BATCH_DEVICES=all) against the single call, and the batch over devices
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
G=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $G"
[ "$G" -lt 2 ] && { echo "one GPU: nothing to check here"; exit 0; }
mkdir -p gpurun_out
for N in 2 4 8; do
  [ "$N" -le "$G" ] || continue
  timeout 900 python3 bench.py --gpus $N --steps 10 --warmup 2 > gpurun_out/multi_bench_$N.json 2> gpurun_out/multi_bench_$N.err; echo "bench --gpus $N: rc=$?"
  python3 - $N <<'PY'
import json, sys
j = json.loads(open("gpurun_out/multi_bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("value", "n_gpus", "ms_per_step", "lossless", "frac_of_hbm_peak_all_gpus", "per_rank", "gather_ms", "gather_warmup_ms", "gather_GBps")}, j.get("config5_dealt"))
PY
done
timeout 900 python3 -m pytest tests/test_gpu_full_size.py -q -k over_rccl 2>&1 | tail -3
RSN_BATCH_DEVICES=all timeout 900 python3 - <<'PY'
import os, sys, hashlib
sys.path.insert(0, os.getcwd())
import workloads as W
from raisin_amd import huffman
for kind in ("2a", "skewed", "2b"):
    d = bytes(W.config_input(kind, 256 << 20).numpy())
    ref = huffman.Compress(d)
    for g in (2, 3, 8):
        assert huffman.CompressSharded(d, g) == ref, (kind, g)
    print(kind, "sharded over devices == single call", hashlib.sha256(ref).hexdigest()[:16])
chunks = [bytes(W.config_input("5", 64 << 20, chunk=k).numpy()) for k in range(8)]
assert huffman.CompressBatch(chunks) == [huffman.Compress(c) for c in chunks]
print("batch over devices == single calls")
PY
