#!/bin/bash
# env slices / streams for the SMALL configs' per-step rows (C2: 4 096 envs, C1 / C5: cheap steps): bench.py cuts every config into
# three slices on three streams because that is the headline's optimum; is it theirs?  tools/rate.py protocol, S = 1 2 3 4 6
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06_streams_small.txt; cd $R; : > $OUT
for round in 1 2; do
  for S in 1 2 3 4 6; do
    echo "== round $round, $S slice(s)" >> $OUT
    RATE_STREAMS=$S RATE_PREROLL=${PRE:-600} timeout 600 python3 tools/rate.py C2 C2:fused C1 C1:fused C5 C5:fused C3 2>&1 | grep -v amdgpu.ids >> $OUT
  done
done
cat $OUT
