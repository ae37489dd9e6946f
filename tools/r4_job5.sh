#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/job5; mkdir -p $o
{
echo "== k_z_ob roles, headline"; python tools/zob_roles.py 100 100 4
echo "== k_z_ob roles, config 4"; python tools/zob_roles.py 500 500 1
echo "== chains split over the XCDs at 100x100 (k_tiles forced): T=1, T=2"
for T in 1 2; do OCC_FORCE_TILES=$T python tools/sizes.py 100,100,4,1000 100,100,1,1000 100,100,2,1000; done
echo "== default forms"; python tools/sizes.py 100,100,4,1000 100,100,1,1000 100,100,2,1000
} > $o/out.txt 2>&1
cat $o/out.txt
