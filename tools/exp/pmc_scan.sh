#!/bin/bash
# issue/stall counters of the scan kernel alone at batch 262144 (separate --pmc passes, kernel-trace only)
out=gpurun_out/pmc_scan; mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "VALUBusy SALUBusy" "MemUnitStalled WriteUnitStalled" "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/profile_scan.py --batch ${BATCH:-262144} --reps 6 > $out/p$i.log 2>&1 || echo "pass $i ($set) failed"
done
python3 - <<'P'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_scan/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "music_scan_" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = sorted(v); print(f"{k:32s} n={len(v):3d} median={v[len(v)//2]:.4g}")
P
