O=gpurun_out/mfma_check
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c1 or long_prototype or random_call or c4 or wbfm or batch_equals or ragged or full_size_batch" > $O/pytest_sel.txt 2>&1
tail -3 $O/pytest_sel.txt
for w in c1 c4; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-host-fed --steps 15 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', 'GS/s %.1f' % (d['value'] / 1e3), 'ms/step %.4f' % d['ms_per_step'], 'front ms %.4f' % d['kernel_ms']['front'], 'mixdec frac %.3f' % d['roofline_mixdec']['frac'])
"
done
