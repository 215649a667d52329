#!/bin/bash
# interleaved comparison of several engine builds: tools/ab_many.sh ROUNDS lib1.so lib2.so ...
N=$1; shift
for i in $(seq $N); do for L in "$@"; do echo -n "$(basename $L) "; CONTRACTS_AMD_LIB=$L python bench.py --steps 1000 --warmup 50 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms']*1000,2), 'us', round(d['value']/1e6), 'M/s')"; done; done
