import sqlite3, glob, sys
db = glob.glob(sys.argv[1] + "/*.db")[0]
con = sqlite3.connect(db)
print("name,calls,total_us,avg_us,pct")
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 10"):
    print('"%s",%d,%.1f,%.3f,%.2f' % (r[0][:70], r[1], r[2], r[3], r[4]))
