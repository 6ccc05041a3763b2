#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (.db) outputs into small text files for profiles/.
usage: prof_summary.py <results.db> [--counters]"""
import sqlite3
import sys


def short(name):
    name = name.replace(",", ";")               # kernel signatures contain commas; the output is CSV
    return name if len(name) < 70 else name[:40] + "..." + name[-24:]


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db); cur = con.cursor()
    if "--counters" in sys.argv:
        print("kernel,counter,dispatches,avg_value,sum_value")
        q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) "
             "from counters_collection group by kernel_name, counter_name order by sum(value) desc")
        for r in cur.execute(q):
            print("%s,%s,%d,%.6g,%.6g" % (short(r[0]), r[1], r[2], r[3], r[4]))
    else:
        # avg_us is what rocprofv3 --stats prints (the top_kernels view); min / median / max from the dispatches themselves:
        # the first launch of a kernel in a process (cold code, first touch of its buffers) can be several times the others
        print("kernel,calls,total_us,avg_us,percent,min_us,median_us,max_us")
        per = {}
        try:
            q = ("select S.display_name, K.end - K.start from rocpd_kernel_dispatch K "
                 "inner join rocpd_info_kernel_symbol S on S.id = K.kernel_id and S.guid = K.guid")
            for name, dur in con.cursor().execute(q):
                per.setdefault(name, []).append(dur / 1000.0)
        except sqlite3.Error:
            per = {}
        for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            d = sorted(per.get(r[0], []))
            extra = ",%.3f,%.3f,%.3f" % (d[0], d[len(d) // 2], d[-1]) if d else ",,,"
            print("%s,%d,%.3f,%.3f,%.2f%s" % (short(r[0]), r[1], r[2], r[3], r[4], extra))


if __name__ == "__main__":
    main()
