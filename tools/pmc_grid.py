import sqlite3, sys
db, ctr = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
gcol = 'grid_size' if 'grid_size' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
q = "select %s, count(*), avg(value) from counters_collection where kernel_name like '%%k_gconv<64, 128, 2, 4, true, 15>%%' and counter_name=? group by 1 order by 1" % (gcol or "'all'")
print(ctr, c.execute(q, (ctr,)).fetchall())
