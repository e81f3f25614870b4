cd $GRAFT_REPO_ROOT
for k in text period; do timeout 300 python scripts/quick_lzss.py $k 1024 2>&1 | grep -A12 "^decode" | grep -E "decode|resolve"; done
