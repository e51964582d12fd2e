#!/bin/bash
# SQ counters (one rocprofv3 pass, 8 SQ slots + GRBM) over tools/front_time.py: where do the front-end / dJPEG kernels spend
# their wave time?   gpurun ... 'bash tools/pmc_sq.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/raw
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/raw -o p -- python $ROOT/tools/front_time.py 2 > $OUT/run.log 2>&1
find $OUT/raw -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} $OUT/sq.csv
rm -rf $OUT/raw
cd $ROOT
python - <<'PY'
import csv, collections, os
p = 'gpurun_out/pmc_sq/sq.csv'
rows = list(csv.DictReader(open(p)))
per = collections.OrderedDict()
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]
    if 'at::native' in k:
        continue
    per.setdefault((k, r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
last = collections.OrderedDict()
for (k, d), v in per.items():
    last[k] = v
print('%-60s %9s %6s %6s %6s %6s %9s %9s' % ('kernel (last dispatch)', 'wavecyc', 'wait%', 'stall%', 'act%', 'valu%', 'instVALU', 'instLDS'))
for k, v in last.items():
    wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    print('%-60s %9.3g %6.1f %6.1f %6.1f %6.1f %9.3g %9.3g' % (k, wc, 100 * v.get('SQ_WAIT_ANY', 0) / wc, 100 * v.get('SQ_WAIT_INST_ANY', 0) / wc,
          100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc, 100 * v.get('SQ_ACTIVE_INST_VALU', 0) / wc, v.get('SQ_INSTS_VALU', 0), v.get('SQ_INSTS_LDS', 0)))
    if 'GRBM_GUI_ACTIVE' in v:
        print('    GRBM_GUI_ACTIVE %.4g  -> waves per SIMD ~ %.2f' % (v['GRBM_GUI_ACTIVE'], wc * 4 / (1024 * v['GRBM_GUI_ACTIVE'])))
PY
