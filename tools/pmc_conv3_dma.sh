#!/bin/bash
# SQ counter passes (each its own rocprofv3 run) over ONE conv3_dma_kernel layer (tools/conv3_probe.py), per environment variant:
#   tools/pmc_conv3_dma.sh "8 512 512" "NIMG_CONV3_PIPE=0" "NIMG_CONV3_PIPE=1"   -> gpurun_out/pmc_conv3_dma/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_conv3_dma
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SHAPE=$1; shift
run() { local name=$1; shift; local envv=$1; shift
  rm -rf $OUT/raw_$name
  env $envv timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/raw_$name -o p -- python $ROOT/tools/conv3_probe.py $SHAPE > $OUT/$name.log 2>&1
  find $OUT/raw_$name -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} $OUT/$name.csv; rm -rf $OUT/raw_$name; }
i=0
for V in "$@"; do
  run v${i}_sq1 "$V" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  run v${i}_sq2 "$V" SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
  run v${i}_sq3 "$V" SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_SMEM
  echo "v$i = $V" >> $OUT/variants.txt
  i=$((i+1))
done
cd $ROOT
python - <<'PY' | tee gpurun_out/pmc_conv3_dma/summary.txt
import csv, glob, json
res = {}
for path in sorted(glob.glob('gpurun_out/pmc_conv3_dma/*.csv')):
    tag = path.split('/')[-1].split('_')[0]
    for r in csv.DictReader(open(path)):
        if 'conv3_dma_kernel' not in r['Kernel_Name'] and 'conv3_big' not in r['Kernel_Name']:
            continue
        e = res.setdefault(tag, {})
        e[r['Counter_Name']] = float(r['Counter_Value'])
        e['dur_us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        e['grid'] = r.get('Grid_Size'); e['vgpr'] = r.get('VGPR_Count'); e['lds'] = r.get('LDS_Block_Size'); e['kernel'] = r['Kernel_Name'][:80]
print(open('gpurun_out/pmc_conv3_dma/variants.txt').read())
for k, o in res.items():
    wc = o.get('SQ_WAVE_CYCLES', 1)
    cyc = o.get('GRBM_GUI_ACTIVE', 8) / 8.0
    print(k, o.get('kernel'), 'dur_us', o.get('dur_us'), 'grid', o.get('grid'), 'vgpr', o.get('vgpr'), 'lds', o.get('lds'))
    print('   waves/SIMD %.2f  mfma busy %.3f  wait_any %.3f  wait_inst %.3f  active %.3f' % (wc * 4 / (1024 * cyc), o.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc), o.get('SQ_WAIT_ANY', 0) / wc, o.get('SQ_WAIT_INST_ANY', 0) / wc, o.get('SQ_ACTIVE_INST_ANY', 0) / wc))
    print('   insts: VALU %.3g LDS %.3g VMEM_RD %.3g SALU %.3g SMEM %.3g | active: LDS %.3f VMEM %.3f VALU %.3f SCA %.3f | lds conflict %.3f wait_inst_lds %.3f  lds_idx_active/cyc/CU %.3f' % (
        o.get('SQ_INSTS_VALU', 0), o.get('SQ_INSTS_LDS', 0), o.get('SQ_INSTS_VMEM_RD', 0), o.get('SQ_INSTS_SALU', 0), o.get('SQ_INSTS_SMEM', 0),
        o.get('SQ_ACTIVE_INST_LDS', 0) / wc, o.get('SQ_ACTIVE_INST_VMEM', 0) / wc, o.get('SQ_ACTIVE_INST_VALU', 0) / wc, o.get('SQ_ACTIVE_INST_SCA', 0) / wc,
        o.get('SQ_LDS_BANK_CONFLICT', 0) / max(o.get('SQ_LDS_IDX_ACTIVE', 1), 1), o.get('SQ_WAIT_INST_LDS', 0) / wc, o.get('SQ_LDS_IDX_ACTIVE', 0) / (256 * cyc)))
json.dump(res, open('gpurun_out/pmc_conv3_dma/summary.json', 'w'), indent=1)
PY
