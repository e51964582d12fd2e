#!/bin/bash
# SQ counter passes (each in its OWN rocprofv3 run, kernel-trace only) over tools/wgrad5_time.py for the all-taps kernel and the
# 8-wave kernel:   gpurun ... 'bash tools/pmc_wgrad5.sh'  ->  gpurun_out/pmc_wgrad5/summary.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_wgrad5
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {  # name, env, counters...
  local name=$1; shift
  local ev=$1; shift
  rm -rf $OUT/raw_$name
  env $ev timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/raw_$name -o p -- python $ROOT/tools/wgrad5_time.py 2 > $OUT/$name.log 2>&1
  find $OUT/raw_$name -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} $OUT/$name.csv
  rm -rf $OUT/raw_$name
}
for v in ${PMC_SET:-new:A=1 old:NIMG_NO_WGRAD5_ALLTAPS=1} ${PMC_EXTRA}; do
  run ${v%%:*}_sq1 ${v#*:} SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  run ${v%%:*}_sq2 ${v#*:} SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
done
cd $ROOT
python - <<'PY'
import csv, glob, json
res = {}
for path in sorted(glob.glob('gpurun_out/pmc_wgrad5/*.csv')):
    tag = path.split('/')[-1].split('_')[0]
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name']
        if 'wgrad' not in k:
            continue
        key = tag + ' ' + k.replace('(anonymous namespace)::', '').split('(')[0][-60:] + ' grid' + r.get('Grid_Size', r.get('Grid_Size_X', ''))
        e = res.setdefault(key, {})
        e[r['Counter_Name']] = float(r['Counter_Value'])          # last dispatch wins
for k, o in res.items():
    d = {}
    if o.get('GRBM_GUI_ACTIVE'):
        cyc = o['GRBM_GUI_ACTIVE'] / 8.0
        d['kernel_cycles_per_xcd'] = cyc
        d['waves_per_simd'] = o['SQ_WAVE_CYCLES'] * 4 / (1024 * cyc)
        d['mfma_pipe_busy_frac'] = o['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc)
        d['wait_any_frac'] = o['SQ_WAIT_ANY'] / o['SQ_WAVE_CYCLES']
        d['wait_inst_frac'] = o['SQ_WAIT_INST_ANY'] / o['SQ_WAVE_CYCLES']
        d['active_frac'] = o['SQ_ACTIVE_INST_ANY'] / o['SQ_WAVE_CYCLES']
    if o.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_bank_conflict_frac_of_lds_cycles'] = o['SQ_LDS_BANK_CONFLICT'] / o['SQ_LDS_IDX_ACTIVE']
        d['wait_inst_lds'] = o.get('SQ_WAIT_INST_LDS')
    o['derived'] = d
json.dump(res, open('gpurun_out/pmc_wgrad5/summary.json', 'w'), indent=1)
for k, o in res.items():
    print(k, json.dumps(o['derived']), 'LDS insts', o.get('SQ_INSTS_LDS'), 'VALU', o.get('SQ_INSTS_VALU'), 'ldsidx', o.get('SQ_LDS_IDX_ACTIVE'), 'conf', o.get('SQ_LDS_BANK_CONFLICT'))
PY
