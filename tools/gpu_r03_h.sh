#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for sg in 0 4 8 12 16 24; do echo "stagger $sg"; PERCEPNET_FP_STAGGER=$sg python tools/kernel_times.py 65536 10 2>&1 | grep ms/step | sed 's/.*| //' ; done
