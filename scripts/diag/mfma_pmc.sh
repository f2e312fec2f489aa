# SQ counters of the matrix-core mix+decimate kernel on C1: how busy the matrix pipe is, what the waves wait for, the clock
O=gpurun_out/r04_pmc
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $O && mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" $O/counters.txt | sort -u > $O/mfma_counters.txt
cat $O/mfma_counters.txt | head -40
A="--workload c1 --no-cpu-baseline --no-host-fed --steps 3 --warmup 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 bench.py $A > $O/p1.json 2> $O/p1.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $O/p2 -- python3 bench.py $A > $O/p2.json 2> $O/p2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py $A > $O/kt.json 2> $O/kt.err
python3 - <<'PY'
import csv, glob, collections
for p in ('p1', 'p2'):
    for f in glob.glob(f'gpurun_out/r04_pmc/{p}/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, d in acc.items():
            if 'mixdec' in k:
                print(p, k, {c: v for c, v in d.items()})
for f in glob.glob('gpurun_out/r04_pmc/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'mixdec' in r['Name']:
            print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +4M -delete
