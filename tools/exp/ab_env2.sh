#!/bin/bash
# scan kernel alone under several values of one switch (second switch fixed via the environment)
var=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $var=$v timeout -k 5 200 python tools/profile_scan.py --batch 262144 --reps 30 2>/dev/null | sed "s/^/$var=$v /"
  env $var=$v timeout -k 5 200 python tools/profile_scan.py --batch 4096 --reps 200 2>/dev/null | sed "s/^/$var=$v /"
done; done
