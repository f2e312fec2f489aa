# round 4, first GPU call: LDS-DMA alignment probe, the tests that reach the matrix-core mix+decimate, C1 / C4 with it on and off
O=gpurun_out/r04_first
mkdir -p $O
timeout 60 scripts/diag/glds_align_test.bin > $O/glds_align.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c1 or long_prototype or random_call or c4 or wbfm or batch_equals or ragged" > $O/pytest_sel.txt 2>&1
tail -5 $O/pytest_sel.txt
for w in c1 c4; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-host-fed > $O/bench_${w}_mfma.json 2> $O/bench_${w}_mfma.err
  PYSDR_TUNING=1 PYSDR_MIXDEC_MFMA=0 timeout 300 python bench.py --workload $w --no-cpu-baseline --no-host-fed > $O/bench_${w}_valu.json 2> $O/bench_${w}_valu.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_first/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], {k:(v.get('frac'),v.get('kernel_ms')) if isinstance(v,dict) else v for k,v in d.items() if k.startswith('roofline')})
    except Exception as e:
        print(f, 'ERR', e)
PY
