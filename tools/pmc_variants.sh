#!/bin/bash
# SQ counters of the 5x5 ring kernels under several builds / dispatch switches, one rocprofv3 --pmc pass each (kernel-trace only):
#   tools/pmc_variants.sh <out dir> "<label>:<ENV=VAL or lib path>" ...
# -> <out dir>/<label>.csv (counter_collection) and a summary table on stdout: per kernel name, duration, cycles (-> clock),
#    MFMA-busy share, wait shares.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
[[ "$OUT" != /* ]] && OUT=$ROOT/$OUT
mkdir -p $OUT
export TMPDIR=/tmp
for V in "$@"; do
  L=${V%%:*}; S=${V#*:}
  E=""
  for T in ${S//,/ }; do           # comma-separated: NAME=VALUE settings and / or one library path
    if [[ "$T" == *.so ]]; then E="$E NIMG_LIBPATH=$ROOT/$T"; else E="$E $T"; fi
  done
  (cd /tmp && env $E timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
     --kernel-trace --output-format csv -d $OUT/raw_$L -o p -- python $ROOT/tools/ring_time.py --child --reps 3 > $OUT/$L.log 2>&1)
  find $OUT/raw_$L -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} $OUT/$L.csv
  rm -rf $OUT/raw_$L
done
python - "$OUT" <<'PY'
import csv, glob, os, sys
for path in sorted(glob.glob(os.path.join(sys.argv[1], '*.csv'))):
    per = {}
    for r in csv.DictReader(open(path)):
        if 'conv5_ring' not in r['Kernel_Name']:
            continue
        k = r['Kernel_Name'].split('conv5_ring_kernel')[1].split('(')[0] + ' grid ' + r.get('Grid_Size', '?')
        e = per.setdefault(k, {})
        e.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        e.setdefault('dur', []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print('==', os.path.basename(path))
    for k, e in per.items():
        m = {c: sum(v) / len(v) for c, v in e.items()}
        cyc = m['GRBM_GUI_ACTIVE'] / 8.0
        print('%-34s %7.1f us  clock %.2f GHz  mfma busy %.3f  waves/simd %.2f  wait %.3f  wait_inst %.3f  active %.3f' % (
            k, m['dur'], cyc / m['dur'] / 1e3, m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc), m['SQ_WAVE_CYCLES'] * 4 / (1024 * cyc),
            m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']))
PY
