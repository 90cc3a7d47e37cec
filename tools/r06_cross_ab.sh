# Round 6: A/B of a variant of the mode-cross kernels -- build/ab/base.so (before) against the
# library in the tree: parity tests of mode cross, then the AbacusSummit interpolator (ds4), its
# first table (ds1) and the default step, pipelined, twice each in alternation.
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "cross" 2>&1 | tail -3
F="--cpu-seconds 0 --detail 0"
one() { python bench.py $F --only-config $1 2>/dev/null | grep -o "\"us_per_step\": [0-9.]*" | head -1; }
for tag in ds4 ds1; do
  for rep in 1 2; do
    echo -n "$tag base: "; TABCORR_AMD_LIBRARY=build/ab/base.so one $tag
    echo -n "$tag new:  "; one $tag
  done
done
