#!/bin/bash
# reproducibility of bench.py's checksum for a given set of extra flags (three runs each); env assignments may precede flags
for spec in "$@"; do
  r=""
  for i in 1 2 3; do
    c=$(env $(echo "$spec" | tr ' ' '\n' | grep '=' | tr '\n' ' ') timeout 200 python bench.py --no-cpu-baseline --no-profile $(echo "$spec" | tr ' ' '\n' | grep -v '=' | tr '\n' ' ') 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['checksum'])")
    r="$r $c"
  done
  echo "[$spec]:$r"
done
