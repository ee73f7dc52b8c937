#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for g in 0 2 4 8 16 32; do echo "== XCD_GROUP=$g"; XCD_GROUP=$g timeout 600 python tools/sconv_sweep.py -1 2>&1 | grep "^(32,64\|^(64,"; done
