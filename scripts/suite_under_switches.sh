#!/bin/bash
# The whole -m gpu suite under this round's A/B switches (all paths must be bit-identical): the r02 token emitter and decode front end,
# the batch split over two workers, devices handed out round-robin.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export RSN_LZSS_NO_LIST=1 RSN_LZSS_DEC_3PASS=1 RSN_BATCH_WORKERS=2 RSN_DEVICE=rr RSN_MAX_PARKED=2
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
