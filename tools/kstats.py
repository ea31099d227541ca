import csv,sys,glob,collections
for d in sys.argv[1:]:
    f=glob.glob(d+'/**/*kernel_trace.csv',recursive=True)
    if not f: print(d,'no trace'); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        agg[r['Kernel_Name'][:60]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    print(d)
    for k,v in sorted(agg.items(),key=lambda kv:-sum(kv[1]))[:9]:
        v=sorted(v); print('  %-60s n=%4d median %8.1f us'%(k,len(v),v[len(v)//2]/1e3))
