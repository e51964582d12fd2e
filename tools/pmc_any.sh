#!/bin/bash
# one rocprofv3 counter pass per quoted counter group over a command; prints per-kernel sums for kernels matching a pattern
#   tools/pmc_any.sh <kernel-substring> "<command>" "<CTR CTR ...>" ["<CTR ...>" ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_any
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PAT=$1; shift
CMD=$1; shift
i=0
for G in "$@"; do
  rm -rf $OUT/raw
  timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/raw -o p -- $CMD > $OUT/pass$i.log 2>&1
  f=$(find $OUT/raw -name '*counter_collection.csv' | head -1)
  python - "$f" "$PAT" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r['Kernel_Name']:
        continue
    k = r['Kernel_Name'][:70]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, c in acc.items():
    n = len(next(iter(c.values())))
    print(k, 'launches', n, 'dur_us(avg)', round(sum(dur[k]) / len(dur[k]), 2))
    for name, v in c.items():
        print('    %-36s %.6g per launch' % (name, sum(v) / len(v)))
PY
  rm -rf $OUT/raw
  i=$((i+1))
done
