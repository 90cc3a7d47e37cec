#!/bin/bash
# Developer A/B: host-buffer API per batch size, build/ab/old.so against the in-tree library.
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
echo "== old"; TABCORR_AMD_LIBRARY=$PWD/build/ab/old.so python tools/archive/host_sizes.py 3000 4000 6000 10000 12000 20000
echo "== new"; python tools/archive/host_sizes.py 3000 4000 6000 10000 12000 20000
done
