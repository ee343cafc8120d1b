#!/bin/bash
# Interleaved A/B of this tree's bench.py against worktrees of older commits, on ONE box in one gpurun call (rule 24):
#   bash tools/ab_commits.sh <rounds> .cmp_r02 [.cmp_xxx ...]      ("git worktree add .cmp_r02 <commit>" + make in its csrc first)
R=$1; shift
for i in $(seq $R); do for d in . "$@"; do
  (cd $d && timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-parity 2>/dev/null) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', d['value'], d['config'].get('value_with_two_steps_in_flight'))" || exit 1
done; done
