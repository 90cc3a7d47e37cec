# Round 6: more lanes than four with more hardware queues (GPU_MAX_HW_QUEUES) for the one-launch
# kernel: sustained step and the driver's 20-step shape.
cd $GRAFT_REPO_ROOT
F="--cpu-seconds 0 --detail 0"
run() { python bench.py $F $* 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f' % (r['ms_per_step']*1e3), end=' ')"; }
for q in 4 8; do
  for lanes in 4 5 6 8; do
    export GPU_MAX_HW_QUEUES=$q
    echo -n "GPU_MAX_HW_QUEUES=$q lanes=$lanes: sustained "; run --steps 3000 --warmup 300 --lanes $lanes
    echo -n " driver shape "; run --steps 20 --warmup 5 --lanes $lanes; run --steps 20 --warmup 5 --lanes $lanes; run --steps 20 --warmup 5 --lanes $lanes
    echo
  done
done
