"""Ablation builds of rm_lib.hip (timing only): python3 scratch/build_abl_lib.py NAME=-DFLAG[,-DFLAG2] ..."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recometrics_amd import build as B
B.build()
os.makedirs("scratch/libs", exist_ok=True)
objs = [os.path.join(B.CSRC, os.path.splitext(s)[0] + ".o") for s in B.SOURCES if s != "rm_lib.hip"]
procs = []
for spec in sys.argv[1:]:
    name, flags = spec.split("=", 1)
    obj = "scratch/libs/lib_%s.o" % name
    procs.append((name, obj, subprocess.Popen([B._hipcc()] + B.FLAGS + flags.split(",") + ["-c", os.path.join(B.CSRC, "rm_lib.hip"), "-o", obj])))
for name, obj, p in procs:
    assert p.wait() == 0, name
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-o", "scratch/libs/lib_abl_%s.so" % name, obj] + objs)
    print("built", name)
