#!/bin/bash
# FETCH_SIZE of a KNOWN byte count in the sweeps' LDS-DMA access pattern: bash scratch/fetch_calib.sh [out file]
OUT=${1:-gpurun_out/r6_fetch_calibration.txt}
R=$(pwd)
hipcc --offload-arch=gfx950 -O3 scratch/fetch_calib.hip -o /tmp/fetch_calib || exit 1
/tmp/fetch_calib > $OUT 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/fetch_calib_pmc
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fetch_calib_pmc -- /tmp/fetch_calib > /dev/null 2>&1
cd $R
python3 - >> $OUT <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/fetch_calib_pmc/*/*counter_collection.csv")[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    key = (int(r["Dispatch_Id"]), r["Kernel_Name"][:40])
    agg[key] = agg.get(key, 0.0) + float(r["Counter_Value"])
known = 2 << 30
print("\nrocprofv3 --pmc FETCH_SIZE (KiB summed over the XCDs' counters), known bytes per dispatch = %d" % known)
for (d, name), v in agg.items():
    print("dispatch %2d %-40s FETCH_SIZE %.0f KiB = %.4f x the bytes streamed" % (d, name, v, v * 1024 / known))
PY
cat $OUT
