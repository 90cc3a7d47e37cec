# Round 6: A/B of a kernel variant of the latency form -- build/ab/base.so (before) against the
# library in the tree: parity tests of the form, the form alone on the chip (twice each in
# alternation), the pipelined step, then the stamps of the developer build.
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "latency or spread or 40 or defer or degenerate or golden" 2>&1 | tail -3
F="--cpu-seconds 0 --detail 0 --steps 2000 --warmup 200"
run() { python bench.py $* $F 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f' % (r['ms_per_step']*1e3), end=' ')"; }
for opts in "--lanes 1 --option fused=2" ""; do
  echo "== [$opts] us per step, base / new, alternating"
  for rep in 1 2; do
    echo -n "   base "; TABCORR_AMD_LIBRARY=build/ab/base.so run $opts
    echo -n "  new "; run $opts
    echo
  done
done
TABCORR_AMD_LIBRARY=build/ab/dev.so TC_FUSED_STAMPS=1 python3 tools/r06_stamps.py 40
