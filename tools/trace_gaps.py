#!/usr/bin/env python3
"""GPU idle gaps from a rocprofv3 kernel trace (`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3
bench.py ...`): busy fraction of the steady state, histogram of the gaps between consecutive kernels and the
kernels in front of which the long gaps sit.  This is how the CPU-quota throttling stalls of the host scheduler
were found (2+ ms gaps in front of encode_planes_kernel, GPU busy 88 %).  usage: trace_gaps.py DIR"""
import csv,sys,glob,statistics
f=sorted(glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:40]) for r in rows)
ev=ev[len(ev)//3:]                      # steady state
span=ev[-1][1]-ev[0][0]; busy=sum(e-s for s,e,_ in ev)
gaps=[(ev[i+1][0]-ev[i][1],ev[i+1][2]) for i in range(len(ev)-1)]
print("events",len(ev),"span ms %.1f busy ms %.1f frac %.4f"%(span/1e6,busy/1e6,busy/span))
for lo,hi in ((0,2e3),(2e3,2e4),(2e4,2e5),(2e5,2e6),(2e6,1e12)):
    g=[x for x,_ in gaps if lo<=x<hi]
    print("gaps %8.0f..%8.0f ns: count %6d total ms %.2f"%(lo,hi,len(g),sum(g)/1e6))
from collections import Counter
print(Counter(k for x,k in gaps if x>2e4).most_common(4))
