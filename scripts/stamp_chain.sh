#!/bin/bash
# Rebuilds the library with per-phase cycle stamps in the chain kernel (GPU box, scratch copy only) and dumps them.
cd $GRAFT_REPO_ROOT/dnn-compression-tensor-admm_amd/csrc && touch chain.hip && make CXXEXTRA=-DTADMM_CHAIN_STAMPS > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 scripts/stamp_chain.py
