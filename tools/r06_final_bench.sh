#!/bin/bash
# Round 6: the bench records of the FINAL library, after tools/profile_round.sh +
# tools/install_profiles.sh put this round's PMC files under profiles/ (so that every `traffic`
# is quoted from them), plus the round's own measurement logs.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/profile
mkdir -p $OUT
python bench.py > $OUT/bench.log 2> $OUT/bench.err
tail -1 $OUT/bench.log > $OUT/bench.json
cp bench_detail.json $OUT/bench_detail.json
: > $OUT/bench_driver_command.json
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $OUT/bench_driver_command.json
done
python tools/r06_latency.py > gpurun_out/r06_latency.log 2>&1
python tools/r06_two_tables.py > gpurun_out/r06_two_tables.log 2>&1
bash tools/r06_alone_forms.sh > gpurun_out/r06_alone_forms.log 2>&1
[ -f build/ab/dev.so ] && bash tools/r06_phases.sh > gpurun_out/r06_phases.log 2>&1
[ -f build/ab/dev.so ] && TABCORR_AMD_LIBRARY=build/ab/dev.so TC_FUSED_STAMPS=1 \
  python tools/r06_stamps.py 40 > gpurun_out/r06_stamps.log 2>&1
[ -f build/ab/dev.so ] && TABCORR_AMD_LIBRARY=build/ab/dev.so TC_FUSED_STAMPS=1 \
  python tools/r06_stamps.py 64 >> gpurun_out/r06_stamps.log 2>&1
tail -2 $OUT/bench.json | cut -c1-400
