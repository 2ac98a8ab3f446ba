#!/bin/bash
# on the GPU box: wall clock of the probe process against its own clock, for nothing / pinned only / device only / both (the command line's footprint);
# EXIT_COST_CASES="800 13000 3 12 0 0;800 13000 3 0 1500 0" gives other argument sets (see exit_cost.hip)
cd "$(dirname "$0")" && hipcc --offload-arch=gfx950 -O2 -o /tmp/exit_cost exit_cost.hip || exit 1
IFS=';' read -ra CASES <<< "${EXIT_COST_CASES:-0 0;800 0;0 13000;800 13000;100 13000}"
for rep in 1 2; do for args in "${CASES[@]}"; do
  s=$(date +%s.%N); /tmp/exit_cost $args 2>/tmp/ec.log; e=$(date +%s.%N)
  python3 -c "import sys; s,e=float('$s'),float('$e'); l=open('/tmp/ec.log').read().strip(); w=l.split(); left=float(w[-1]); m=float(w[-4]); print('%-28s wall %.3f s = exec->main %.3f + own %.3f + _exit->gone %.3f | %s' % ('$args', e-s, m-s, left-m, e-left, ' '.join(w[:-6])))"
done; done
