cd $GRAFT_REPO_ROOT
F="--cpu-seconds 0 --detail 0 --steps 2000 --warmup 200"
for opts in "" "--option fused_waves=16" "--option fused_draws=32" "--option fused_defer=0" "--option fused_defer=1"; do
  echo "== alone: $opts"
  python bench.py --lanes 1 --option fused=2 $opts $F 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step']*1e3, r['roofline']['kernel'])"
done
for opts in "" "--option fused_draws=32"; do
  echo "== pipelined: $opts"
  python bench.py $opts $F 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step']*1e3, r['roofline']['kernel'])"
done
