#!/bin/bash
# Vector-memory / LDS latency counters of the trunk kernels (tools/_build/wino3h_bench, APZ_PROFILE=1).  Run on the GPU box
# from the repo root; writes gpurun_out/w3h_pmc2/ and a per-kernel summary (counter values summed over all dimensions).
set -e
cd "$(dirname "$0")/.."
export APZ_PROFILE=1
out=gpurun_out/w3h_pmc2; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
B=$OLDPWD/tools/_build/wino3h_bench
i=0
for set in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL"; do




    i=$((i+1))
    rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OLDPWD/$out/pass$i -o p -- $B > $OLDPWD/$out/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $OLDPWD
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("gpurun_out/w3h_pmc2/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "trunk15" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
with open("gpurun_out/w3h_pmc2/summary.txt", "w") as out:
    for k in sorted(tot):
        out.write(k + "\n")
        for c in sorted(tot[k]):
            out.write("   %-34s %16.0f per launch (%d launches)\n" % (c, tot[k][c] / len(disp[k][c]), len(disp[k][c])))
print(open("gpurun_out/w3h_pmc2/summary.txt").read())
PY
