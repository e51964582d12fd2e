#!/bin/bash
# PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 runs, kernel-trace only) over the stand-alone kernel drivers:
#   front = tools/front_time.py (front-end kernels, un-pool, dJPEG), dcn = tools/dcn_kernel.py, dom = tools/dominant_kernel.py
# Results: gpurun_out/pmc/<driver>_<counter>.csv (+ tools/pmc_summary.py turns them into JSON)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  for D in front dcn dom; do
    case $D in
      front) CMD="python $ROOT/tools/front_time.py 2" ;;
      dcn) CMD="python $ROOT/tools/dcn_kernel.py 80 bf16" ;;
      dom) CMD="python $ROOT/tools/dominant_kernel.py --dtype bf16 --store-bf16 --pool" ;;
    esac
    rm -rf $OUT/raw_${D}_$C
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/raw_${D}_$C -o p -- $CMD > $OUT/${D}_$C.log 2>&1
    find $OUT/raw_${D}_$C -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} $OUT/${D}_$C.csv
    rm -rf $OUT/raw_${D}_$C
  done
done
cd $ROOT
python tools/pmc_summary.py $OUT > $OUT/summary.json 2> $OUT/summary.err
cat $OUT/summary.json | head -c 3000
