#!/bin/bash
# One GPU-box session, made of the stages named on the command line; everything lands under gpurun_out/<tag>/.
#   gpurun --timeout 900 -- 'bash tools/gpu_check.sh r02_b tests c4 prof'
# stages: tests | tests:<pytest -k expression> | c4 | c4nocpu | c4eager | c3 | c5 | n2 | prof | trace | pmc:<counter> | pmcstep | pmcring | py:<script and args>
TAG=${1:-r02}
shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
note() { echo "$1" | tee -a $OUT/summary.txt; }
for ST in "$@"; do
  case $ST in
    tests) timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; note "pytest rc=$?"; tail -15 $OUT/pytest.log ;;
    tests:*) timeout 900 python -m pytest tests -m gpu -q -k "${ST#tests:}" > $OUT/pytest_k.log 2>&1; note "pytest -k rc=$?"; tail -30 $OUT/pytest_k.log ;;
    c4) timeout 500 python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err; note "bench c4 rc=$?"; tail -c 1500 $OUT/bench_c4.json ;;
    c4nocpu) timeout 300 python bench.py --no-cpu-baseline --no-parity-mode > $OUT/bench_c4_nocpu.json 2> $OUT/bench_c4_nocpu.err; note "bench c4 (no cpu) rc=$?"; head -c 400 $OUT/bench_c4_nocpu.json ;;
    c4old) NIMG_OLD_FRONTEND=1 timeout 300 python bench.py --no-cpu-baseline --no-parity-mode > $OUT/bench_c4_old.json 2> $OUT/bench_c4_old.err; note "bench c4 (old front end) rc=$?"; head -c 400 $OUT/bench_c4_old.json ;;
    c4eager) timeout 300 python bench.py --no-graph --no-cpu-baseline --no-parity-mode > $OUT/bench_c4_eager.json 2> $OUT/bench_c4_eager.err; note "bench c4 eager rc=$?"; head -c 400 $OUT/bench_c4_eager.json ;;
    c3) timeout 400 python bench.py --workload c3 --steps 50 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; note "bench c3 rc=$?"; tail -c 900 $OUT/bench_c3.json ;;
    c5) timeout 400 python bench.py --workload c5 --steps 50 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; note "bench c5 rc=$?"; tail -c 900 $OUT/bench_c5.json ;;
    n2) timeout 300 python bench.py --gpus 2 --backend gloo --single-device --steps 5 --warmup 2 --batch 16 --no-cpu-baseline \
          > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err; note "bench n2 gloo rc=$?" ;;
    prof) (cd /tmp && NIMG_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o c4 -- \
            python $ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-parity-mode > $ROOT/$OUT/prof.log 2>&1)
          find $OUT/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
          find $OUT/prof -name '*kernel_trace.csv' -delete; note "prof done"; head -25 $OUT/kernel_stats.csv | cut -c1-160 ;;
    trace) (cd /tmp && NIMG_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/trace -o c4 -- \
            python $ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-parity-mode > $ROOT/$OUT/trace.log 2>&1)
          find $OUT/trace -name '*kernel_trace.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_trace.csv; rm -rf $OUT/trace; note "trace done" ;;
    pmc:*) C=${ST#pmc:}
          (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$C -o p -- \
            python $ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-parity-mode > $ROOT/$OUT/pmc_$C.log 2>&1)
          note "pmc $C done" ;;
    ltrace) (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/ltrace -o c4 -- \
            python $ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-parity-mode > $ROOT/$OUT/ltrace.log 2>&1)
          find $OUT/ltrace -name '*kernel_trace.csv' | head -1 | xargs -I{} python tools/launch_trace.py {} > $OUT/launch_trace.txt
          rm -rf $OUT/ltrace; note "ltrace done"; tail -1 $OUT/launch_trace.txt ;;
    pmcstep) # counter-based HBM bytes of one whole C4 step on THIS build: two separate passes + tools/pmc_step_total.py (stamped)
          for C in FETCH_SIZE WRITE_SIZE; do
            (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$C -o p -- \
              python $ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-parity-mode > $ROOT/$OUT/pmc_$C.log 2>&1)
            find $OUT/pmc_$C -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} $OUT/pmc_$C.csv; rm -rf $OUT/pmc_$C
          done
          python tools/pmc_step_total.py $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv 64 c4 > $OUT/pmc_step_total.json 2> $OUT/pmc_step_total.err
          note "pmcstep rc=$?"; head -c 600 $OUT/pmc_step_total.json; rm -f $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv ;;
    pmcring) bash tools/pmc_ring.sh > $OUT/pmc_ring.log 2>&1; cp gpurun_out/pmc_ring/summary.json $OUT/pmc_ring_kernel.json; note "pmcring rc=$?"
          tail -c 900 $OUT/pmc_ring_kernel.json ;;
    py:*) timeout 600 python ${ST#py:} > $OUT/py_$(echo "${ST#py:}" | tr ' /' '__' | cut -c1-40).log 2>&1; note "py ${ST#py:} rc=$?"
          tail -40 $OUT/py_$(echo "${ST#py:}" | tr ' /' '__' | cut -c1-40).log ;;
    *) note "unknown stage $ST" ;;
  esac
done
