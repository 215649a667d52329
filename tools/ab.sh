#!/bin/bash
# A/B of two engine builds in one process sequence, interleaved (same box, same clocks)
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do for L in $A $B; do echo -n "$(basename $L) "; CONTRACTS_AMD_LIB=$L python bench.py --steps 400 --warmup 50 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms']*1000,2), 'us', round(d['value']/1e6), 'M/s')"; done; done
