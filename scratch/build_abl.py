"""Ablation builds of the three-sub-tile fp32 sweep (timing only, wrong results): python3 scratch/build_abl.py NAME=-DFLAG[,-DFLAG2] ...
The unit is rm_sweep32_n3_s1.hip -- the specialisation BASELINE C2 runs (dense train rows); ABL_UNIT=<file> picks another."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recometrics_amd import build as B
B.build(incremental=True)
os.makedirs("scratch/libs", exist_ok=True)
UNIT = os.environ.get("ABL_UNIT", "rm_sweep32_n3_s1.hip")
objs = [os.path.join(B.CSRC, os.path.splitext(s)[0] + ".o") for s in B.SOURCES if s != UNIT]
procs = []
for spec in sys.argv[1:]:
    name, flags = spec.split("=", 1)
    obj = "scratch/libs/n3_%s.o" % name
    procs.append((name, obj, subprocess.Popen([B._hipcc()] + B.FLAGS + B.SWEEP_FLAGS + flags.split(",") + ["-c", os.path.join(B.CSRC, UNIT), "-o", obj])))
for name, obj, p in procs:
    assert p.wait() == 0, name
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-o", "scratch/libs/lib_abl_%s.so" % name, obj] + objs)
    print("built", name)
