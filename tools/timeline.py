import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_ref_encode' in r['Kernel_Name'] or 'k_ref_prep' in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
t0=int(rows[a]['Start_Timestamp'])
prev_end=t0
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=r['Kernel_Name'].split('(')[0].replace('mia::','').replace('void ','')[:30]
    print(f"{(s-t0)/1e3:8.1f}us dur {(e-s)/1e3:6.1f} gap {(s-prev_end)/1e3:6.1f} {name:30s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d}x{r['Workgroup_Size_X']:4s} st {r['Stream_Id']}")
    prev_end=max(prev_end,e)
print('step span us', (int(rows[b]['Start_Timestamp'])-t0)/1e3, 'n kernels', b-a)
