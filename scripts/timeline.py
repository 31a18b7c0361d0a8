import csv,glob,sys
rows=[]
for f in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True): rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'prescan_kernel' in r['Kernel_Name']]
j=idx[-4]-6
t0=int(rows[j]['Start_Timestamp'])
for r in rows[j:idx[-2]+6]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-t0)/1e3:9.1f}  q{r.get('Queue_Id','?')}  {r['Kernel_Name'].replace('vers::','')[:50]}")
