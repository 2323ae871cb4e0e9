#!/bin/bash
# Counter passes over the kernels a python driver launches (each pass its own run; --pmc never combined with tracing).
# usage: tools/pmc_kernel.sh <kernel-name-substring> <outdir> <driver.py> [driver args]
set -e
export TMPDIR=/tmp
K=$1; O=$2; shift 2
mkdir -p $O
P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS"
P3="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr"
P4="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES"
P5="SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS SQ_IFETCH"
i=1
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  # a pass that dies (a GPU fault or hang included) ends the script: no further launches on a card that just failed
  rocprofv3 --pmc $P --output-format csv -d $O/p$i -o p -- python3 "$@" > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; echo "pass $i failed: see $O/p$i.log; no counters aggregated"; exit 1; }
  i=$((i+1))
done
python3 - "$O" "$K" <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44], r["Counter_Name"])
        a = acc.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k in sorted(acc):
    print(f"{k[0]:44s} {k[1]:36s} launches {acc[k][0]:3d}  per-launch {acc[k][1] / acc[k][0]:18.1f}")
PY
