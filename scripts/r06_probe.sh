cd $GRAFT_REPO_ROOT
g++ -O2 -o /tmp/probe scripts/probes/host_call_probe.cpp -Lraisin_amd -lrsn -Wl,-rpath,$PWD/raisin_amd
python - <<'PY'
import sys; sys.path.insert(0, '.')
import workloads as W
W.config_input("4", 1 << 30).numpy().tofile('/tmp/text.bin')
PY
echo '--- C process, pipelined'; RSN_HOST_TIMING=1 /tmp/probe 1024 @/tmp/text.bin 2>&1 | grep -v "^host call" | tail -4
