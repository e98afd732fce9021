#!/bin/bash
# same-box A/B of the C3 line: the committed tree (tools/scratch/old, `git worktree`) against the working tree, interleaved
root=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for which in old new; do
    dir=$root; [ $which = old ] && dir=$root/tools/scratch/old
    (cd $dir && python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extra "$@" 2>/dev/null) | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$which', d['value'], 'fps', d['ms_per_step'], 'ms', {k: round(v*1e3,1) for k,v in d['stage_ms'].items()})"
  done
done
