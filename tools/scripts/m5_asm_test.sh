#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
C=$R/dynamicsparsearrays.jl_amd/csrc
O=$R/gpurun_out/m5asm; mkdir -p $O
FUZZ_ONLY=append timeout -k 10 200 python tools/fuzz.py 50 6000 > $O/fuzz_append.log 2>&1; echo "fuzz append (asm) rc=$?"; tail -1 $O/fuzz_append.log
DSA_DEV=1 DSA_MODEL5=2 FUZZ_ONLY=append timeout -k 10 200 python tools/fuzz.py 25 6500 > $O/fuzz_append_cpp.log 2>&1; echo "fuzz append (generic path) rc=$?"; tail -1 $O/fuzz_append_cpp.log
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "c5 or append or column_generation or streaming" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
DSA_DEV=1 DSA_DBG_RUN=1 timeout -k 10 300 python tools/c5bench.py --full > $O/c5bench_dbg.log 2>&1; grep "append model" $O/c5bench_dbg.log | tail -3
DSA_DEV=1 DSA_DBG_TIME=1 timeout -k 10 300 python tools/c5bench.py --full > $O/c5_time.log 2>&1; tail -3 $O/c5_time.log | head -1
grep "mat_apply_sets" $O/c5_time.log | awk 'NR>1{tot+=$5; n++; if(NR<=11){f10+=$5}} END{print n, "batches total", tot, "ms; first10", f10}'
echo "== bug3 lib must fail in mode 1 on leafmat seeds 9215.."
DSA_LIBRARY=$C/libdsa_hip_fpcheck_bug3.so DSA_FP_MODE=1 FUZZ_ONLY=leafmat timeout -k 10 300 python tools/fuzz.py 20 9101 > $O/bug3_mode1.log 2>&1; echo "rc=$?"; grep -c DSA_FP_CHECK $O/bug3_mode1.log
DSA_LIBRARY=$C/libdsa_hip_fpcheck.so DSA_FP_MODE=1 FUZZ_ONLY=leafmat timeout -k 10 300 python tools/fuzz.py 60 9101 > $O/fixed_mode1.log 2>&1; echo "fixed rc=$?"; tail -1 $O/fixed_mode1.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python3 -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'frac',d['roofline']['frac']); print('c5',d.get('c5_streaming')); print('c5_per_call',d.get('c5_per_call')); print('ins',d.get('inserts_per_s')); print('ic3',{k:v for k,v in d.get('inserts_on_c3',{}).items() if 'runs' in k or k=='error'}); print('reb',d.get('roofline_rebalance')); print('fill',d.get('buffered_writes')); print('err',d.get('extras_error'))"
