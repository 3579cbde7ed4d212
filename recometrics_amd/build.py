"""Builds recometrics_amd/csrc/librecometrics_hip.so for gfx950 (in-tree; hipcc cross-compiles without a GPU).

The library is several translation units (host + prep/finalize kernels, fp32 sweep, fp64 sweeps) compiled in parallel."""
import concurrent.futures
import glob
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "librecometrics_hip.so")
SOURCES = ["rm_lib.hip", "rm_sweep32.hip", "rm_sweep32_large.hip", "rm_sweep64_small.hip", "rm_sweep64_small_s1.hip", "rm_sweep64_large.hip", "rm_sweep64_large_s1.hip",
           # the fp32 sweep families x the specialisations of the epilogue's switches (rm_sweep.hpp k_sweep SPEC): one unit each
           "rm_sweep32_n3.hip", "rm_sweep32_n3_s1.hip", "rm_sweep32_n3_s2.hip", "rm_sweep32_lds.hip", "rm_sweep32_lds_s1.hip", "rm_sweep32_lds_s2.hip",
           "rm_sweep32_hbm.hip", "rm_sweep32_hbm_s1.hip", "rm_sweep32_hbm_s2.hip", "rm_split.cpp", "rm_csr.cpp"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-inline-asm"]   # m0 is clobbered by the LDS-DMA asm on purpose


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _deps():
    root = os.path.dirname(os.path.dirname(CSRC))
    return (glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(CSRC, "*.cpp"))
            + glob.glob(os.path.join(root, "include", "*.h")))


STAMP = os.path.join(CSRC, "build_stamp.json")


def sources_digest():
    """sha256 over the contents of everything the library is compiled from (file names included), and over the flags"""
    import hashlib
    h = hashlib.sha256()
    for p in sorted(_deps()):
        h.update(os.path.basename(p).encode() + b"\0")
        h.update(open(p, "rb").read())
    h.update(repr((SOURCES, FLAGS, SWEEP_FLAGS)).encode())
    return h.hexdigest()


def read_stamp():
    import json
    try:
        return json.load(open(STAMP))
    except Exception:      # noqa: BLE001
        return None


def needs_build(extra_flags=()):
    """The library is rebuilt when it is missing or when it was not built from the sources as they are now, with the flags asked
    for: the decision is made on CONTENT (the digest build() records next to the library, csrc/build_stamp.json), not on
    modification times -- a checkout, a copy to another box or a touched file must neither force nor hide a rebuild.  A library
    WITHOUT a stamp (a prebuilt .so copied without it: the stamp is git-ignored) is judged by the old rule instead of being
    rebuilt unconditionally -- it is kept when it is newer than every source."""
    if not os.path.exists(LIB):
        return True
    st = read_stamp()
    if not st:
        return os.path.getmtime(LIB) < max(os.path.getmtime(p) for p in _deps()) or bool(extra_flags)
    return (st.get("sources_sha256") != sources_digest() or st.get("library_bytes") != os.path.getsize(LIB)
            or list(st.get("extra_flags", [])) != list(extra_flags))


# the sweeps: no SLP vectorisation -- it pairs independent f32 adds into v_pk_add_f32, which issue at less than half the rate of
# two plain instructions on gfx950 (profiles/r3_coexec4_issue_rates.txt)
SWEEP_FLAGS = ["-fno-slp-vectorize"]


def _headers():
    root = os.path.dirname(os.path.dirname(CSRC))
    return glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(root, "include", "*.h"))


def _compile(src, extra, incremental=False):
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
    # (development only, `build(incremental=True)`: an object newer than its source and every header is kept; the library's own
    # stamp stays content-based)
    if incremental and os.path.exists(obj):
        hdrs = _headers()
        if src.startswith("rm_sweep"):       # the sweeps' units include neither the preparation nor the finalisation kernels
            hdrs = [h for h in hdrs if os.path.basename(h) not in ("rm_finalize.hpp", "rm_prep.hpp", "rm_noise.hpp")]
        if os.path.getmtime(obj) > max(os.path.getmtime(p) for p in [os.path.join(CSRC, src)] + hdrs):
            return obj, ""
    if src.startswith("rm_sweep"):
        extra = list(extra) + SWEEP_FLAGS
    if src.endswith(".cpp"):        # host-only translation unit
        cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-pthread", "-c", os.path.join(CSRC, src), "-o", obj]
    else:
        cmd = [_hipcc()] + FLAGS + list(extra) + ["-c", os.path.join(CSRC, src), "-o", obj]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, res.stdout, res.stderr))
    return obj, res.stderr


def build(force=False, verbose=False, extra_flags=(), out=None, incremental=False):
    out = out or LIB
    if not force and out == LIB and not needs_build(extra_flags):
        return LIB
    import time
    t_start = time.time()
    digest_at_start = sources_digest()        # (what the objects are compiled FROM: an edit during the build must not be stamped as built)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        results = list(ex.map(lambda s: _compile(s, extra_flags, incremental), SOURCES))
    if verbose:
        for _, err in results:
            print(err)
    res = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-o", out] + [o for o, _ in results],
                         capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    if out == LIB:                  # every build into LIB leaves its stamp, debug / ablation flags included (`extra_flags`)
        import json
        ver = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout.splitlines()
        json.dump({"sources_sha256": digest_at_start, "library_bytes": os.path.getsize(LIB), "built_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
                   "translation_units": SOURCES, "flags": FLAGS, "sweep_flags": SWEEP_FLAGS, "extra_flags": list(extra_flags), "hipcc": ver[0] if ver else "?",
                   "seconds": round(time.time() - t_start, 1)}, open(STAMP, "w"), indent=1)
    return out


CY_SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cy.pyx")


def cython_module_path():
    import sysconfig
    return os.path.join(os.path.dirname(CY_SRC), "_cy" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_cython(force=False):
    """Compiles recometrics_amd/_cy.pyx (the Cython binding over include/recometrics_hip.h) in-tree and links it against
    csrc/librecometrics_hip.so.  Returns the path of the extension module."""
    import sysconfig
    build()
    out = cython_module_path()
    root = os.path.dirname(os.path.dirname(CSRC))
    if not force and os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(CY_SRC), os.path.getmtime(os.path.join(root, "include", "recometrics_hip.h"))):
        return out
    csrc = os.path.join(os.path.dirname(CY_SRC), "_cy.c")
    if not shutil.which("cython"):
        raise RuntimeError("cython executable not found")
    res = subprocess.run([shutil.which("cython"), "-3", CY_SRC, "-o", csrc], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("cython failed:\n" + res.stdout + res.stderr)
    cmd = [shutil.which("gcc") or "gcc", "-O2", "-fPIC", "-shared", "-w", csrc, "-o", out,
           "-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(root, "include"),
           "-L" + CSRC, "-lrecometrics_hip", "-Wl,-rpath,$ORIGIN/csrc"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("compiling the Cython binding failed:\n" + res.stdout + res.stderr)
    os.remove(csrc)
    return out


R_SHIM_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "r", "src")


def build_r_shim(force=False):
    """Compiles r/src/rm_r_shim.c (the R-independent half of the Rcpp glue, r/src/Rwrapper_hip.cpp) into r/src/librm_r_shim.so,
    linked against csrc/librecometrics_hip.so -- the object an R package would link, minus Rcpp.  Plain gcc, no R needed."""
    build()
    src, out = os.path.join(R_SHIM_DIR, "rm_r_shim.c"), os.path.join(R_SHIM_DIR, "librm_r_shim.so")
    root = os.path.dirname(os.path.dirname(CSRC))
    deps = [src, os.path.join(R_SHIM_DIR, "rm_r_shim.h"), os.path.join(root, "include", "recometrics_hip.h")]
    if not force and os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(d) for d in deps):
        return out
    cmd = [shutil.which("gcc") or "gcc", "-O2", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", src, "-o", out,
           "-I" + os.path.join(root, "include"), "-L" + CSRC, "-lrecometrics_hip", "-Wl,-rpath," + CSRC]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("compiling the R shim failed:\n" + res.stdout + res.stderr)
    return out


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose="-v" in sys.argv))
    print(build_cython(force=True))
    print(build_r_shim(force=True))
