import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows: r['s']=float(r['start_us']); r['d']=float(r['dur_us']); r['e']=r['s']+r['d']
starts=[i for i,r in enumerate(rows) if 'pack_image' in r['name']]
a,b=starts[-2],starts[-1]
step=rows[a:b]
T0=step[0]['s']; T1=max(r['e'] for r in step)
print('step span ms', round((T1-T0)/1e3,2), 'kernels', len(step), 'sum ms', round(sum(r['d'] for r in step)/1e3,2))
ev=sorted([(r['s'],1) for r in step]+[(r['e'],-1) for r in step])
cur=0; last=T0; hist={}
for t,dl in ev:
    hist[cur]=hist.get(cur,0)+(t-last); cur+=dl; last=t
print('concurrency ms:', {k:round(v/1e3,2) for k,v in sorted(hist.items())})
def span(name):
    xs=[r for r in step if name in r['name']]
    return (round(xs[0]['s']-T0), round(xs[-1]['e']-T0), len(xs), round(sum(x['d'] for x in xs))) if xs else None
for n in ['maxpool','msdeform','attn_kernel<64','attn_kernel<32','m2f_mask_kernel','resize','pts3d','gaussian_adapter','pp_class']:
    print(n, span(n))
queues={}
for r in step: queues.setdefault(r['queue'],[]).append(r)
for q,rs in sorted(queues.items()): print('queue',q,len(rs),'first',round(rs[0]['s']-T0),'last',round(rs[-1]['e']-T0),'sum',round(sum(x['d'] for x in rs)))
if len(sys.argv)>2:
    lo,hi=float(sys.argv[2]),float(sys.argv[3])
    for r in step:
        if lo<=r['s']-T0<=hi: print(round(r['s']-T0,1), r['d'], r['queue'], r['grid'], r['name'][:50])
