import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
cat={'own':[0,0],'hipblaslt':[0,0],'torch':[0,0],'other':[0,0]}
for r in rows:
    n=r['Name']; t=float(r['TotalDurationNs'])/1e6/14; c=int(r['Calls'])/14
    k='own' if (n.startswith('k_') or 'void k_' in n or n.startswith('_Z') and 'k_' in n) else 'hipblaslt' if n.startswith('Cijk') else 'torch' if 'at::native' in n else 'other'
    cat[k][0]+=t; cat[k][1]+=c
for k,v in cat.items(): print("%-10s %.3f ms/step  %.0f launches/step"%(k,v[0],v[1]))
