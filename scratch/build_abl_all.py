"""Ablation builds of every fp32 sweep translation unit (timing only): python3 scratch/build_abl_all.py NAME=-DFLAG[,-DFLAG2] ..."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recometrics_amd import build as B
B.build()
os.makedirs("scratch/libs", exist_ok=True)
sweeps = [s for s in B.SOURCES if s.startswith("rm_sweep32")]
others = [os.path.join(B.CSRC, os.path.splitext(s)[0] + ".o") for s in B.SOURCES if s not in sweeps]
for spec in sys.argv[1:]:
    name, flags = spec.split("=", 1)
    procs = []
    for src in sweeps:
        obj = "scratch/libs/%s_%s.o" % (os.path.splitext(src)[0], name)
        procs.append((obj, subprocess.Popen([B._hipcc()] + B.FLAGS + B.SWEEP_FLAGS + flags.split(",") + ["-c", os.path.join(B.CSRC, src), "-o", obj])))
    objs = []
    for obj, p in procs:
        assert p.wait() == 0, obj
        objs.append(obj)
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-o", "scratch/libs/lib_abl_%s.so" % name] + objs + others)
    print("built", name)
