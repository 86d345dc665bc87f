"""Per-dispatch FETCH_SIZE of the NT GEMM launches of the last step, grouped by kernel instance and fetch volume:
    python tools/pmc_fetch_by_shape.py <rocprofv3 --pmc FETCH_SIZE counter_collection.csv>   (fetch bytes = 2 x FETCH_SIZE x 1024 on gfx950)"""
import csv, collections, re, sys
rows=list(csv.DictReader(open(sys.argv[1])))
nt=[r for r in rows if 'gemm_bf16_nt' in r['Kernel_Name']][-99:]
agg=collections.OrderedDict()
for r in nt:
    name=re.search(r'gemm_bf16_nt\w*<([^>]*)>', r['Kernel_Name']).group(1)
    agg.setdefault(name, []).append(2*float(r['Counter_Value'])*1024/1e6)
tot=0
for k,v in agg.items():
    v.sort(); cl=[]
    for f in v:
        if cl and abs(cl[-1][-1]-f) < 0.15*f: cl[-1].append(f)
        else: cl.append([f])
    print(k, ' | '.join('n=%d %.0f MB'%(len(c), sum(c)/len(c)) for c in cl))
    tot+=sum(v)
print('NT total fetched per step: %.1f GB'%(tot/1e3))
