import csv, collections, glob, sys
f=glob.glob(f'/root/repo/gpurun_out/{sys.argv[1]}/*/*counter_collection.csv')[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'tg::' in k:
        w=sum(v['SQ_WAVES'])/len(v['SQ_WAVES'])
        print(k,{c: round(sum(x)/len(x)/w,1) for c,x in v.items() if c!='SQ_WAVES'}, 'waves',w,'n',len(v['SQ_WAVES']))
