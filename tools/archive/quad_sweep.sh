run() { python bench.py --cpu-seconds 0 --steps 4000 --warmup 400 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.2f us/step  isolated %.2f us  overlapped %.2f us' % (d['ms_per_step']*1e3, d['roofline']['mean_launch_ms']*1e3, d['roofline']['overlapped_launch_ms']*1e3))"; }
# (the TC_* knobs are read by developer builds only: tools/build_dev.sh)
export TABCORR_AMD_LIBRARY=$PWD/build/ab/dev.so
export TC_PRIO_O=0 TC_PRIO_C=1 TC_PRIO_F=3
echo -n "base (O0 C1 F3): "; run
for lanes in 2 3; do echo -n "lanes $lanes: "; TC_LANES=$lanes run; done
for qw in 1 3; do echo -n "quad waves $qw: "; TC_QUAD_WAVES=$qw run; done
for sp in 1 2 3 5 10 13; do echo -n "occ splits $sp: "; TC_OCC_SPLITS=$sp run; done
for ft in 256 512 1024; do for rb in 1 2 5; do echo -n "finalize threads $ft row blocks $rb: "; TC_FINALIZE_THREADS=$ft TC_FINALIZE_ROW_BLOCKS=$rb run; done; done
echo -n "base again: "; run
