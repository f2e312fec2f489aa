cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "ft8tri 1" "ft8tri 3" "ft8tri 17" "test2rx 5" "c1 1" "c1 7" "c4 2" "c4 3" "c2 1" "c3 3" "rx6 2" "c1synch 3" "c4mono 1"; do
  set -- $spec
  out=$(python3 bench.py --workload $1 --chunks $2 --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed --verify 2>&1 | tail -1)
  echo "$1 chunks=$2: $(echo "$out" | python3 -c "
import sys,json
try:
    j=json.loads(sys.stdin.read().strip()); print('ok', round(j['value']/1e3,1),'GS/s verify', j.get('verify_worst_rel'), j.get('verified_ranks'))
except Exception as e: print('FAILED', e)")"
done
