#!/bin/bash
# same-box A/B of the render() frame: the committed tree (tools/scratch/old) against the working tree, interleaved
root=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for which in old new; do
    dir=$root; [ $which = old ] && dir=$root/tools/scratch/old
    (cd $dir && python tools/render_stage_ab.py 2>/dev/null | tail -1 | sed "s/^/$which /")
  done
done
