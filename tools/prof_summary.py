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
        print("kernel,calls,total_us,avg_us,percent")
        for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print("%s,%d,%.3f,%.3f,%.2f" % (short(r[0]), r[1], r[2], r[3], r[4]))


if __name__ == "__main__":
    main()
