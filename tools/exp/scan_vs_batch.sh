#!/bin/bash
for b in 64 256 1024 2048 4096 8192 16384; do timeout -k 5 100 python tools/profile_scan.py --batch $b --reps 300 2>/dev/null; done
