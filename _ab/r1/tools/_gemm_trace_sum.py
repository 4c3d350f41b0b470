import sqlite3,glob,sys,collections
db=glob.glob(sys.argv[1]+'/**/*.db',recursive=True)[0]
c=sqlite3.connect(db)
d=collections.defaultdict(list)
for n,s,e,gx in c.execute("select name,start,end,grid_x from kernels order by start"):
    if 'gemm' in n: d[gx].append((e-s)/1e3)
for gx,v in sorted(d.items()):
    v=sorted(v); print(f"grid_x={gx} (WGs={gx//512}) n={len(v)} median={v[len(v)//2]:.1f}us min={v[0]:.1f}")
