#!/bin/bash
# interleaved comparison of several engine builds: [BENCH_ARGS="--kind harvest"] tools/ab_many.sh ROUNDS lib1.so lib2.so ...
N=$1; shift
for i in $(seq $N); do for L in "$@"; do echo -n "$(basename $L) "; CONTRACTS_AMD_LIB=$L python bench.py --full --steps 1000 --warmup 50 --no-cpu-baseline $BENCH_ARGS | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1000,2), 'us/step', round(d['value']/1e6), 'M/s')"; done; done
