cd $GRAFT_REPO_ROOT
RSN_LIB_PATH=$PWD/scripts/ab/librsn_stats.so timeout 300 python scripts/quick_cfg4_lzss.py 2>&1 | grep -v amdgpu.ids | head -40
