cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_call_lengths or long_prototype_does_not or operating_points or many_thousand" 2>&1 | tail -3
for rep in 1 2 3; do for pt in "1 1" "5 1"; do for v in main nosv; do
  if [ $v = main ]; then e="PYSDR_X=0"; else e="PYSDR_TUNING=1 PYSDR_LIB_VARIANT=$v"; fi
  echo "$v $pt: $(env $e python3 scripts/launch_script_rates.py $pt | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('front %.4f ms frac %.3f; %.0f GS/s job %.3f' % (j['front_ms'], j['frac'], j['gsps'], j['job_frac']))")"
done; done; done
