cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export WGRAD_ONLY=7,0
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" ; do
rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/_pw -o pw -- python $R/scripts/wgrad_bench.py > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$R/gpurun_out/_pw/*.db")[0])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = f"select s.kernel_name, i.name, count(*), avg(e.value) from {pmc} e join {info} i on e.pmc_id=i.id join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, i.name"
for r in cur.execute(q):
    if "wgrad" in r[0] or "splitk" in r[0]:
        print(r[0][:60], r[1], r[2], "%.4g" % r[3])
PY
rm -rf $R/gpurun_out/_pw
done
