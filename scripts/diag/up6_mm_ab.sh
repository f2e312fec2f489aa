#!/bin/bash
# Run ON THE GPU BOX: 1001 taps at UP = 6 (1 / 5 / 7 MS/s) -- matrix-core form from 1 (mm1) / 3 (main) sub-receivers or never (mm0) --
# and the steady loop's task loop against the two-slot form (oldloop) on ft8tri / test2rx.  Variant libraries built beforehand.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_call_lengths or long_prototype_does_not or multi_rx_long or steady or launch_scripts" 2>&1 | tail -3
for rep in 1 2; do for pt in "1 1" "1 2" "5 2" "7 3" "5 4" "1 3"; do for v in main mm1 mm0; do
  if [ $v = main ]; then e="PYSDR_X=0"; else e="PYSDR_TUNING=1 PYSDR_LIB_VARIANT=$v"; fi
  echo "$v $pt: $(env $e python3 scripts/launch_script_rates.py $pt | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('front %.4f ms frac %.3f; %.0f GS/s job %.3f' % (j['front_ms'], j['frac'], j['gsps'], j['job_frac']))")"
done; done; done
VARIANTS="main oldloop" WLS="ft8tri test2rx c3_1001" REPS=2 bash scripts/diag/long_multirx_ab.sh 2>&1 | tail -12
