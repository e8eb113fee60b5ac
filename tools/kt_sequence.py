import csv,glob,sys
f=glob.glob('/tmp/kt/**/*_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
out=[]
for r in rows:
    n=r['Kernel_Name']
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    tag='L' if 'lookup_fwd' in n else ('B' if 'build16' in n else ('F' if 'lookup_convc1' in n else None))
    if tag: out.append((tag,d,int(r['Start_Timestamp'])))
# print last 3 steps of the timed region: find builds
idx=[i for i,o in enumerate(out) if o[0]=='B']
for bi in idx[20:23]:
    seg=out[bi:bi+33]
    print(' '.join(f"{t}{d:.1f}" for t,d,_ in seg))
    gaps=[(seg[i+1][2]-seg[i][2])/1e3 for i in range(len(seg)-1)]
    print('start-to-start:',' '.join(f"{g:.1f}" for g in gaps))
