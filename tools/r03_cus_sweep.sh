#!/bin/bash
# pipelined period of a 1.25M-row shard vs the screen's CU count (ablation build), lanes on private streams (round 3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for C in 256 240 224 208 192 176; do
  echo -n "screen CUs=$C  "
  OI_LIB=ablation OI_SCREEN_CUS=$C python3 $R/tools/shard_step_bench.py 1250000 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lists %.3f  pipelined %.3f  tried %s' % (d['lists_ms'], d['pipelined']['period_ms'], [round(t['ms'],3) for t in d['pipelined']['tried']]))"
done
done
python3 -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; g.smoke()"
