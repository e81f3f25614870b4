#!/bin/bash
# SQ counters of k_match_chain with a wavefront / a row per chain (Zipf text, 1 GiB); runs on the GPU box
R=${GRAFT_REPO_ROOT:-$(pwd)}
for L in ${LANES:-64 16}; do
  export RSN_LZSS_CHAIN_LANES=$L
  bash $R/scripts/pmc.sh rows${L}_1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $R/scripts/quick_lzss.py text 1024 | grep -A8 "^k_match_chain"
  bash $R/scripts/pmc.sh rows${L}_2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -- $R/scripts/quick_lzss.py text 1024 | grep -A8 "^k_match_chain"
  bash $R/scripts/pmc.sh rows${L}_3 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH -- $R/scripts/quick_lzss.py text 1024 | grep -A8 "^k_match_chain"
done
