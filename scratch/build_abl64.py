"""Ablation builds of the fp64 sweep (timing only, wrong results): python3 scratch/build_abl64.py NAME=-DFLAG[,-DFLAG2] ..."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recometrics_amd import build as B
B.build()
os.makedirs("scratch/libs", exist_ok=True)
objs = [os.path.join(B.CSRC, os.path.splitext(s)[0] + ".o") for s in B.SOURCES if s != "rm_sweep64_large.hip"]
procs = []
for spec in sys.argv[1:]:
    name, flags = spec.split("=", 1)
    obj = "scratch/libs/l64_%s.o" % name
    procs.append((name, obj, subprocess.Popen([B._hipcc()] + B.FLAGS + B.SWEEP_FLAGS + flags.split(",") + ["-c", os.path.join(B.CSRC, "rm_sweep64_large.hip"), "-o", obj])))
for name, obj, p in procs:
    assert p.wait() == 0, name
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-o", "scratch/libs/lib_abl64_%s.so" % name, obj] + objs)
    print("built", name)
