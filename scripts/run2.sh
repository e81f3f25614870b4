cd $GRAFT_REPO_ROOT
timeout 300 python scripts/quick_huff.py skewed 1024 2>&1 | grep -A8 "decode"
RSN_DEC_EMIT256=1 timeout 300 python scripts/quick_huff.py skewed 1024 2>&1 | grep -A3 "decode"
