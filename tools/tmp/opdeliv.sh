export TMPDIR=/tmp PG_ONLY_BF16=1
for cfg in base K1; do
  unset PG_DEBUG_ONE_KTILE
  case $cfg in K1) export PG_DEBUG_ONE_KTILE=1;; esac
  OUT=/tmp/prof_$cfg; rm -rf $OUT; mkdir -p $OUT
  PG_NO_SIDE_STREAM=1 PG_NS_ITERS=10 rocprofv3 --kernel-trace -d $OUT -o ns -- python tools/gen_fwd_bwd_bench.py 32 > $OUT/stdout.log 2>&1 || true
  python - $OUT/*results.db <<'PY'
import sqlite3,sys,collections
db=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows=db.execute(f"select s.kernel_name,d.start,d.end,d.grid_size_x,d.grid_size_y,d.grid_size_z,d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
agg=collections.OrderedDict()
for n,s,e,gx,gy,gz,wx in rows:
    if 'conv_bf16_big' not in n: continue
    key=(n[9:40],gx//wx,gy,gz)
    agg.setdefault(key,[]).append((e-s)/1e3)
for k,v in agg.items(): print(k, len(v), "avg us %.1f"%(sum(v)/len(v)), "WGs", k[1]*k[2]*k[3], "us per WG-round %.2f"%((sum(v)/len(v))/max(1,(k[1]*k[2]*k[3])/256)))
PY
done
