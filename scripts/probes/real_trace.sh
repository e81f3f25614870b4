export RSN_DEBUG=1
for k in "sparse: a byte in 100" "sparse: a byte in 3000" "runs up to 1000" "records 256 B" "records 100 B" "bits as"; do
  timeout 120 python scripts/probes/real_shapes.py 4 "$k" trace 2>&1 | grep -v amdgpu.ids | grep -v "^kind"
done
