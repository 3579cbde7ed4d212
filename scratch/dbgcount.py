import sys, os, ctypes, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NS_K"] = "50"
sys.argv = ["ns.py", "P64b", "16384", "1"]
exec(open(os.path.join(os.path.dirname(__file__), "ns.py")).read())
lib = ctypes.CDLL(os.environ["RECOMETRICS_HIP_LIB"])
buf = (ctypes.c_ulonglong * 8)()
print("rc", lib.rm_debug_counters(buf), "compactions", buf[0], "entries compacted", buf[1], "tiles with events (per wave)", buf[2])
