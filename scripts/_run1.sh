for f in 0 8 16; do
  echo "flags=$f"; PYSDR_DEBUG_FLAGS=$f python bench.py --steps 10 --warmup 3 --chunks 256 --no-cpu-baseline --no-psd 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'])"
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
