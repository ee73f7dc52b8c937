#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for a in 0 0x200 0x100; do echo "== ABLATE=$a"; ABLATE=$a ONLY6464=1 GEMM=1 NW=8 python tools/sconv_tiles.py 2>&1 | grep -v amdgpu.ids | grep "layer\|duration\|phase\|setup" | head -4; done
