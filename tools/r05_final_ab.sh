#!/bin/bash
# Round 5, evidence in one gpurun call (one box): the round-4 library (tools/ab_build.sh efabbad r04)
# against the final one on the headline and the reference's table shapes, and the options of the
# deferring one-launch instance.  ->  gpurun_out/r05_final_ab.txt
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r05_final_ab.txt
mkdir -p gpurun_out
{
  echo "== headline (BASELINE configs[1], sustained): round-4 library / final"
  python tools/ab_bench.py --rounds 2 r04=tools/ab/libtabcorr_hip_r04.so final=tree | tail -2
  echo "== headline: node loops in place / satellites: expansion up to 12 + 4 cap terms in place (no records) / the bin's shortest from its record / the centrals from records too (default)"
  python tools/ab_bench.py --rounds 2 nodefer=tree,fused_defer=0 cap5=tree,fused_defer=1,fused_sat_cap=5 \
    cap4=tree,fused_defer=1,fused_sat_cap=4 cap3=tree,fused_defer=1,fused_sat_cap=3 \
    cap2=tree,fused_defer=1,fused_sat_cap=2 cap1=tree,fused_defer=1,fused_sat_cap=1 \
    sats=tree,fused_defer=1 default=tree | tail -8
  for tag in ds4 ds1 wp cfg3; do
    echo "== $tag: round-4 library / final"
    python tools/ab_bench.py --rounds 2 --args "--only-config $tag --cpu-seconds 0 --detail 0" \
      r04=tools/ab/libtabcorr_hip_r04.so final=tree | tail -2
  done
  echo "== ds4: deferred pairs off / on"
  python tools/ab_bench.py --rounds 2 --args "--only-config ds4 --cpu-seconds 0 --detail 0" \
    nodefer=tree,cross_defer=0 defer=tree | tail -2
  echo "== ds1: register form / chunk form"
  python tools/ab_bench.py --rounds 2 --args "--only-config ds1 --cpu-seconds 0 --detail 0" \
    registers=tree,cross_wide_min_draws=0 chunks=tree | tail -2
} 2>&1 | tee $OUT
