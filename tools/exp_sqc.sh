#!/bin/bash
# scalar data cache counters of k_primary_only (mesh_1m, monkey): one PMC pass
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/sqc; rm -rf $out; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL --output-format csv -d $out -o r -- python3 tools/exp_primary.py 3 > $out/stdout.txt 2> $out/stderr.txt
echo rc=$?
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/sqc/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row['Kernel_Name'][:40]
    acc[k][row['Counter_Name']] += float(row['Counter_Value']); 
    if row['Counter_Name'] == 'SQC_DCACHE_REQ': n[(k, row.get('Grid_Size'))] += 1
for k, v in acc.items():
    if 'primary' in k: print(k, dict(v))
print(n)
PY
