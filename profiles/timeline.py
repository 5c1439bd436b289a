import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# pick steady-state window: last 30% of trace
t0=int(rows[0]['Start_Timestamp']); t1=int(rows[-1]['End_Timestamp'])
lo=t0+(t1-t0)*0.6; hi=t0+(t1-t0)*0.9
sel=[r for r in rows if lo<=int(r['Start_Timestamp'])<hi]
span=(hi-lo)/1e6
by=collections.defaultdict(lambda:[0,0.0])
streams=collections.defaultdict(float)
for r in sel:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    k=(r.get('Stream_Id','?'),r['Kernel_Name'][:70])
    by[k][0]+=1; by[k][1]+=d
    streams[r.get('Stream_Id','?')]+=d
print("window ms",span, "streams busy ms",dict(streams))
for k,v in sorted(by.items(), key=lambda kv:-kv[1][1])[:45]:
    print("%6.2f%% %5d %8.3fms  s%s %s"%(100*v[1]/span, v[0], v[1], k[0], k[1]))
# gaps on busiest stream
main=max(streams,key=streams.get)
ms=[r for r in sel if r.get('Stream_Id','?')==main]
gap=0;prev=None;big=[]
for r in ms:
    s=int(r['Start_Timestamp']);e=int(r['End_Timestamp'])
    if prev is not None and s>prev:
        gap+=s-prev
        if s-prev>20000: big.append(((s-prev)/1e3, r['Kernel_Name'][:50]))
    prev=max(prev or 0,e)
print("main stream",main,"gap total ms",gap/1e6, "n big gaps",len(big))
cnt=collections.Counter(); tot=collections.Counter()
for g,n in big: cnt[n]+=1; tot[n]+=g
for n,c in tot.most_common(15): print("  gap before %-50s n=%d total=%.1fus"%(n,cnt[n],c))
