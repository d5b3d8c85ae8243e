#!/bin/bash
# round 6, lease 15: the whole GPU suite once (timing against the driver's 20-minute step), smoke, then the new stated-length tests' output
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06o; mkdir -p $O
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > $O/suite.log 2>&1; echo "suite rc=$?"; tail -32 $O/suite.log
grep -h "config 4 chain\|drifted trajectory\|config 5 at T=200" $O/suite.log | cut -c1-900
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
