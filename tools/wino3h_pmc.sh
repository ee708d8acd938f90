#!/bin/bash
# PMC passes of rocprofv3 over tools/_build/wino3h_bench (APZ_PROFILE=1: 12 launches of each trunk kernel at 512 boards).
# Run on the GPU box from the repo root; writes gpurun_out/w3h_pmc/pass*/ and a per-kernel summary.
set -e
cd "$(dirname "$0")/.."
export APZ_PROFILE=1
out=gpurun_out/w3h_pmc; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
B=$OLDPWD/tools/_build/wino3h_bench
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OLDPWD/$out/pass$i -o p -- $B > $OLDPWD/$out/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $OLDPWD
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/w3h_pmc/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open("gpurun_out/w3h_pmc/summary.txt", "w") as out:
    for k in sorted(tot):
        out.write(k + "\n")
        for c in sorted(tot[k]):
            out.write("   %-34s %16.0f per launch (%d launches)\n" % (c, tot[k][c] / cnt[k][c], cnt[k][c]))
print(open("gpurun_out/w3h_pmc/summary.txt").read())
PY
