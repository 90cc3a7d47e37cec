#!/bin/bash
# Developer A/B (knob build): zero-copy limit of the host-buffer API against batch size.
cd "$GRAFT_REPO_ROOT" || exit 1
export TABCORR_AMD_LIBRARY=$PWD/build/ab/dev.so
for kb in 64 128 256 512 1024; do
  echo "== TC_ZERO_COPY_KB=$kb"
  TC_ZERO_COPY_KB=$kb python tools/archive/host_sizes.py 500 1000 1500 2000 3000 4000 6000
done
