import csv, sys, statistics as st
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
steps=[r for r in rows if 'connector_step_gen_kernel' in r['Kernel_Name']]
refs=[r for r in rows if 'stash_refill_gen_kernel' in r['Kernel_Name']]
roll=steps[-int(sys.argv[2]):]
t0=int(roll[0]['Start_Timestamp'])
dur=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in roll]
gaps=[(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3 for a,b in zip(roll,roll[1:])]
print('steps: median dur %.1f mean %.1f ; gaps mean %.2f, n>50us %d (sum %.0f us); span/step %.1f'%(st.median(dur),sum(dur)/len(dur),sum(gaps)/len(gaps),sum(g>50 for g in gaps),sum(g for g in gaps if g>50),(int(roll[-1]['End_Timestamp'])-t0)/1e3/len(roll)))
rr=[r for r in refs if int(r['Start_Timestamp'])>t0]
iv=[((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-t0)/1e3) for r in rr]
print('refills in window',len(iv),'mean dur %.0f'%(sum(b-a for a,b in iv)/max(1,len(iv))))
# overlap count over time
ov=0.0; tot=0.0
ev=sorted([(a,1) for a,b in iv]+[(b,-1) for a,b in iv])
n=0; last=None; hist={}
for t,d in ev:
    if last is not None: hist[n]=hist.get(n,0)+t-last
    n+=d; last=t
print('time with k refills running (us):',{k:round(v) for k,v in hist.items()})
print('first starts:',[round(a) for a,b in iv[:8]])
