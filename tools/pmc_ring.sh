#!/bin/bash
# PMC passes over the dominant kernel as the FAN runs it (tools/dominant_kernel.py --dtype bf16 --store-bf16 --pool =
# conv5_ring_kernel<128>): FETCH_SIZE, WRITE_SIZE and two SQ groups, each in its OWN rocprofv3 run (kernel-trace only).
#   gpurun ... 'bash tools/pmc_ring.sh'  ->  gpurun_out/pmc_ring/summary.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_ring
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/tools/dominant_kernel.py --dtype bf16 --store-bf16 --pool"
run() {  # name, counters...
  local name=$1; shift
  rm -rf $OUT/raw_$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/raw_$name -o p -- $CMD > $OUT/$name.log 2>&1
  find $OUT/raw_$name -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} $OUT/$name.csv
  rm -rf $OUT/raw_$name
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
cd $ROOT
python - <<'PY'
import csv, glob, json, os, sys
sys.path.insert(0, 'tools')
from src_stamp import csrc_sha16
out = {}
for path in sorted(glob.glob('gpurun_out/pmc_ring/*.csv')):
    rows = list(csv.DictReader(open(path)))
    per = {}
    for r in rows:
        if 'conv5_ring_kernel' not in r['Kernel_Name']:
            continue
        per.setdefault(r['Counter_Name'], []).append((float(r['Counter_Value']), (int(r.get('End_Timestamp', 0) or 0) - int(r.get('Start_Timestamp', 0) or 0)) / 1e3))
    for c, vals in per.items():
        out[c] = vals[-1][0]
        out.setdefault('dur_us_under_pmc', {})[c] = vals[-1][1]
out['kernel'] = 'conv5_ring_kernel<128, false> (FAN conv3 forward + LReLU + pool, 320 x 64x64x64 -> 128, bf16 in / pooled bf16 out)'
out['images'] = 320
if 'FETCH_SIZE' in out and 'WRITE_SIZE' in out:
    out['fetch_bytes_corrected_x2'] = 2 * out['FETCH_SIZE'] * 1024
    out['write_bytes'] = out['WRITE_SIZE'] * 1024
    out['traffic_bytes_per_launch'] = out['fetch_bytes_corrected_x2'] + out['write_bytes']
    out['algorithmic_bytes_per_launch'] = 320 * (64 * 64 * 64 * 2 + 32 * 32 * 128 * 3) + 25 * 64 * 128 * 2
d = {}
if out.get('GRBM_GUI_ACTIVE'):
    cyc = out['GRBM_GUI_ACTIVE'] / 8.0                       # summed over the 8 XCDs
    d['kernel_cycles_per_xcd'] = cyc
    d['waves_per_simd'] = out['SQ_WAVE_CYCLES'] * 4 / (1024 * cyc)
    d['mfma_pipe_busy_frac'] = out['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc)
    d['wait_any_frac'] = out['SQ_WAIT_ANY'] / out['SQ_WAVE_CYCLES']
    d['wait_inst_frac'] = out['SQ_WAIT_INST_ANY'] / out['SQ_WAVE_CYCLES']
    d['active_frac'] = out['SQ_ACTIVE_INST_ANY'] / out['SQ_WAVE_CYCLES']
if out.get('SQ_LDS_IDX_ACTIVE'):
    d['lds_bank_conflict_frac_of_lds_cycles'] = out['SQ_LDS_BANK_CONFLICT'] / out['SQ_LDS_IDX_ACTIVE']
out['derived'] = d
json.dump({'csrc_sha16': csrc_sha16('.'), 'bf16_stored_input_pooled': out}, open('gpurun_out/pmc_ring/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:2500])
PY
