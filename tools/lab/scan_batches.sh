#!/bin/bash
# lab: the shipped scan kernel by batch size (does the per-row time at large batches come from waves drifting apart?)
for b in 8192 16384 32768 65536 131072 262144 524288; do
    a=$(timeout -k 5 200 python tools/profile_scan.py --batch $b --reps 40 2>/dev/null | tail -1)
    echo "$a"
done
