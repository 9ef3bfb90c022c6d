#!/bin/bash
# which kernel boundary needs a host sync for the fp16 bench to be reproducible?  families: 0 frontend 1 fc 2 conv1 3 conv2 4 gru512 5 gru_rb 6 fc_gb 7 fc_rb 8 backend
for m in "$@"; do
  a=$(PERCEPNET_SYNC_EACH=$m timeout 200 python bench.py --fp16 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['checksum'])")
  b=$(PERCEPNET_SYNC_EACH=$m timeout 200 python bench.py --fp16 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['checksum'])")
  echo "mask $m: $a $b $([ "$a" == "$b" ] && echo SAME || echo DIFF)"
done
