import csv, re, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'logprob_fwd_kernel' in n]
pairs=[(a,b) for a,b in zip(idx,idx[1:]) if b-a>50]
a,b=pairs[-2]
step=rows[a:b]
t0=int(step[0]['Start_Timestamp']); t1=int(step[-1]['End_Timestamp'])
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in step)/1e3
print("kernels/step",len(step),"span us",(t1-t0)/1e3,"busy us",busy)
mine=('gt_fwd','gt_bwd','attn_fwd','attn_bwd','votes_fwd','votes_bwd','likelihood_fwd','likelihood_bwd','render_fwd','render_bwd','logprob_fwd')
cnt=0;dur=0.0
for r in step:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    hit=[k for k in mine if k in r['Kernel_Name']]
    if hit:
        print(f"  [{cnt:3d} kernels {dur:7.1f} us]  -> {hit[0]} {d:.1f}"); cnt=0; dur=0.0
    else: cnt+=1; dur+=d
print(f"  [{cnt:3d} kernels {dur:7.1f} us]  -> END")
if len(sys.argv)>2:
    for r in step:
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        n=r['Kernel_Name'].replace('void ','').replace('at::native::','').replace('(anonymous namespace)::','')[:100]
        print(f"{d:8.2f} {n}")
