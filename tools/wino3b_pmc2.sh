set -e
cd /root/repo
export APZ_PROFILE=1
out=gpurun_out/w3b_pmc2; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
B=/root/repo/tools/_build/wino3b_bench
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --output-format csv --pmc $set -d /root/repo/$out/pass$i -o p -- $B > /root/repo/$out/pass$i.log 2>&1 || echo "pass $i failed"
done
cd /root/repo
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/w3b_pmc2/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        print("   %-34s %16.0f per launch (%d launches)" % (c, tot[k][c] / cnt[k][c], cnt[k][c]))
PY
