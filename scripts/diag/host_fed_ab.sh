#!/bin/bash
# host_fed (ingest ring, PCIe-inclusive) of the r04 tree against HEAD on ONE box, twice each: is the drop 5.14 -> 4.00 GS/s
# across driver runs the library's or the box's?  (needs the r04 worktree: git worktree add _r04 13a8f9a && (cd _r04 && python -m pysdr_amd.build))
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
for t in _r04 .; do
  (cd $t && python3 bench.py --no-cpu-baseline --no-other-configs --steps 5 --warmup 2 > /tmp/o.json 2>/tmp/o.err; python3 - $t <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
h=d['host_fed']
print(sys.argv[1], "host_fed %.0f MS/s  %.4f ms/chunk" % (h['value'], h['ms_per_chunk']), {k:(round(v,2) if isinstance(v,float) else v) for k,v in h.items() if k in ('memcpy_GBps','h2d_GBps','pcie_ceiling_MSps','per_slot_ms')}, "prefilled", (h.get('slots_prefilled') or {}).get('value'))
PY
  )
done
done
