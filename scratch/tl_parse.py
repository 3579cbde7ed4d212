"""prints the last step of a rocprofv3 kernel trace: python3 scratch/tl_parse.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_classify" in r["Kernel_Name"]]
i0 = starts[-1]; t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f q%s/s%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id"), r.get("Stream_Id"), r["Kernel_Name"][:70]))
