#!/bin/bash
# The whole -m gpu suite under the switches that select a second formulation (all paths must be bit-identical): the general parse
# instead of the chain walk's own, one codeword per Huffman lookup, no run tiles in the LZSS decoder, the batch split over two workers,
# devices handed out round-robin, W-periodic inputs and token runs through the whole pipeline / the ordinary decoder, the chain walk's
# instance for streams with runs of a byte whatever the stream, the parse's redo rounds instead of strips flagged ahead (r06).
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
export RSN_LZSS_NO_FUSED_PARSE=1 RSN_NO_MULTI=1 RSN_LZSS_DEC_NO_RUNS=1 RSN_BATCH_WORKERS=2 RSN_DEVICE=rr RSN_MAX_PARKED=2 RSN_LZSS_NO_PERIODIC_TAIL=1 RSN_LZSS_DEC_NO_RUN_TAIL=1 RSN_LZSS_RUNS=1 RSN_LZSS_NO_PREFLAG=1 RSN_LZSS_SWEEP_WIDE=0
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
