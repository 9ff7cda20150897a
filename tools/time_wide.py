"""the float64 operator records of bench.py alone (bench.wide_records): key, ms, frac of 8 TB/s"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
for r in bench.wide_records(0):
    print("%-16s %8.3f ms  frac %.3f  %s" % (r.get("key"), r.get("kernel_ms"), r.get("frac"), r.get("kernel")))
