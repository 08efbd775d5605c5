"""Print one train step's kernel timeline from the rocpd database scripts/prof_timeline.sh wrote (diagnostic)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_tl/tl_results.db")
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
wgx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "256")
gz = "grid_z" if "grid_z" in cols else "1"
ks = sorted(db.execute("select start,end,name,stream_id,grid_x,grid_y,%s,%s from kernels" % (gz, wgx)))
adam = [k[1] for k in ks if "clip_adam" in k[2]]
t0, t1 = adam[-3], adam[-2]
sel = [k for k in ks if t0 <= k[0] < t1]
def short(n):
    return re.sub(r"\(.*", "", n.replace("void ", "").replace("asr::", ""))[:58]
print("step span us", (t1 - t0) / 1e3)
end0 = t0
for s, e, n, st, gx, gy, gz, wx in sel:
    gap = (s - end0) / 1e3 if st == sel[0][3] else 0.0
    print("%8.1f %7.1f  s%d  %-58s g=%dx%dx%d %s" % ((s - t0) / 1e3, (e - s) / 1e3, st, short(n), gx // max(wx, 1), gy, gz,
                                                 ("   <-- main-stream gap %.1f" % gap) if gap > 8 else ""))
    if st == sel[0][3]:
        end0 = max(end0, e)
