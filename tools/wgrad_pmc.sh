#!/bin/bash
# PMC passes of rocprofv3 over tools/_build/wgrad_kernel_bench (APZ_WGKB_N=512: wgrad_wino3_kernel at 512 boards, both load forms in
# alternating rounds: grid 256 workgroups x 512 threads).  Run on the GPU box from the repo root; writes
# gpurun_out/wgrad_pmc/pass*/ and a per-kernel summary.  (Counters in their own passes, kernel trace only: no other tracing.)
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/wgrad_pmc; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
B=$OLDPWD/tools/_build/wgrad_kernel_bench
export APZ_WGKB_N=512
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OLDPWD/$out/pass$i -o p -- $B > $OLDPWD/$out/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $OLDPWD
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/wgrad_pmc/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_wino3_kernel" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0][:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open("gpurun_out/wgrad_pmc/summary.txt", "w") as out:
    for k in sorted(tot):
        out.write(k + "\n")
        for c in sorted(tot[k]):
            out.write("   %-34s %16.0f per launch (%d launches)\n" % (c, tot[k][c] / cnt[k][c], cnt[k][c]))
print(open("gpurun_out/wgrad_pmc/summary.txt").read())
PY
