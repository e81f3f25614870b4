#!/bin/bash
# SQ counters of the general Huffman decode kernels (skewed 1 GiB), three passes; runs on the GPU box
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/scripts/pmc.sh dec1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $R/scripts/quick_huff.py skewed 1024
bash $R/scripts/pmc.sh dec2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -- $R/scripts/quick_huff.py skewed 1024
bash $R/scripts/pmc.sh dec3 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- $R/scripts/quick_huff.py skewed 1024
