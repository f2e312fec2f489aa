#!/bin/bash
# Run ON THE GPU BOX: streams that run past 2^32 OUTPUT samples (25 hours of audio at 48 kHz; 2^32 input samples are passed by every
# default bench run) -- the last step of each verified against the oracle primed at that absolute position.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "c1 1100" "c2 2200" "ft8tri 2200" "c1synch 1100" "c4 2200" "rx6 2200"; do
  set -- $spec
  python3 bench.py --workload $1 --steps $2 --warmup 5 --no-cpu-baseline --no-host-fed --no-other-configs --full-line --verify 2>&1 | tail -1 | python3 -c "
import sys,json
try:
    j=json.loads(sys.stdin.read().strip())
    n=j['config']['chunks_per_step']*j['config']['in_chunk']*($2+5)
    print('$1 steps $2: %.3g input samples, ~%.3g outputs per RX; %.1f GS/s; verify' % (n, n*48e3/j['config'].get('srate', 0) if j['config'].get('srate') else -1, j['value']/1e3), j.get('verify_worst_rel'), j.get('verified_ranks'))
except Exception as e: print('$1 FAILED', e)"
done
