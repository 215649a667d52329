#!/bin/bash
# interleaved A/B of engine builds on the GPU box:  tools/ab.sh ROUNDS "SPECS..." lib1.so lib2.so ...   (see tools/rate.py)
N=$1; SPECS=$2; shift 2
for i in $(seq $N); do for L in "$@"; do CONTRACTS_AMD_LIB=$L python3 tools/rate.py $SPECS; done; done
