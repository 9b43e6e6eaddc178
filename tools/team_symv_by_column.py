import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else "grid_size_x"
wx = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
rows = db.execute("select name, start, duration, %s, %s from kernels where name like '%%symv_kernel<true>%%' or name like '%%yreduce%%' or name like '%%colupd_kernel<true>%%' order by start" % (gx, wx)).fetchall()
# last repetition only: take the last third
n = len(rows)
rows = rows[2 * n // 3:]
# per column there are 8 symv, 8 yreduce, 8 colupd (team of 8); print the member-0 symv, yreduce, colupd of every 1024th column
sym = [r for r in rows if 'symv' in r[0]]
yr = [r for r in rows if 'yreduce' in r[0]]
cu = [r for r in rows if 'colupd' in r[0]]
print(len(sym), len(yr), len(cu))
for c in range(0, len(sym) // 8, 1024):
    for m in (0, 7):
        r = sym[8 * c + m]
        print("col %5d member %d symv %7.2f us grid %5d wgs | " % (c, m, r[2] / 1e3, r[3] // r[4]), end="")
    print("yreduce %6.2f us  colupd %6.2f us" % (yr[8 * (c + c // 64) + 0][2] / 1e3 if 8 * (c + c // 64) < len(yr) else -1, cu[8 * (c + c // 64)][2] / 1e3 if 8 * (c + c // 64) < len(cu) else -1))
