python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for t in 512 768 1024; do
 for f in 0 6 7; do
  echo "threads=$t flags=$f"; PYSDR_DEBUG_FLAGS=$f python bench.py --steps 10 --warmup 3 --chunks 256 --no-cpu-baseline --no-psd --threads $t 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'], d.get('kernels_ms'))"
 done
done
