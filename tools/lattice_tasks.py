import re, collections, sys
run=open(sys.argv[1]).read().splitlines()
tl=[l.split() for l in run if l.startswith('launch ')]
starts=[i for i,t in enumerate(tl) if t[1]=='0' and t[3]=='keys' and t[5]=='0']
last=tl[starts[-1]:]
show=[l for l in open(sys.argv[2]) if 'k_lattice_fused' in l][-len(last):]
rows=[]
for t,s in zip(last,show):
    m=re.search(r'\+\s*([\d.]+) us\s+grid\s+(\d+)',s)
    rows.append((int(t[1]),t[3],int(t[5]),int(t[9]),float(m.group(1))))
agg=collections.defaultdict(float)
for r in rows: agg[r[1]]+=r[4]
print('total task time %.1f'%sum(agg.values()), {k:round(v,1) for k,v in sorted(agg.items(), key=lambda x:-x[1])})
by=collections.defaultdict(float)
for r in rows: by[r[0]]=max(by[r[0]], r[4])
print('critical path (sum of per-launch maxima) %.1f us over %d launches'%(sum(by.values()), len(by)))
for t in sorted(by): print(t, round(by[t],1), [(r[1],r[2],round(r[4],1)) for r in rows if r[0]==t and r[4]>0.6*by[t]])
