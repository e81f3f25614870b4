cd $GRAFT_REPO_ROOT
timeout 300 python scripts/quick_cfg4_lzss.py 2>&1 | grep -E "layer|chain_tail |tok_emit"
