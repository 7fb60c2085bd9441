import csv,re,sys
from collections import defaultdict
rows=list(csv.DictReader(open(sys.argv[1])))
ntile=int(sys.argv[2])
g=defaultdict(lambda:[0,0.0])
t0=min(int(r["Start_Timestamp"]) for r in rows); t1=max(int(r["End_Timestamp"]) for r in rows)
for r in rows:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    nm=r["Kernel_Name"]
    m=re.search(r'(k_\w+(<\d+>)?)',nm)
    key=(m.group(1) if m else nm[:30], int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]))
    g[key][0]+=1; g[key][1]+=d
tot=sum(v[1] for v in g.values())
print(f"kernel time {tot:.1f} ms, span {(t1-t0)/1e6:.1f} ms, {len(rows)} launches")
for k,v in sorted(g.items(),key=lambda kv:-kv[1][1])[:int(sys.argv[3]) if len(sys.argv)>3 else 25]:
    print(f"{k[0]:18s} wgs {k[1]:8d} ({k[1]/ntile:7.2f} states) calls {v[0]:4d} total {v[1]:8.2f} ms avg {v[1]/v[0]*1e3:9.1f} us")
