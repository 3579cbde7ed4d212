"""CPU-only: the environment never changes what a call computes.

SURVEY.md section 5: environment knobs "must not change results".  The library reads its switches once, at load, in ONE place
(csrc/rm_lib.hip `Switches::load`); every name it reads must be listed in DESIGN.md section 7 as a result-neutral test / timing
switch, and no other translation unit of the product may look at the environment at all."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "recometrics_amd", "csrc")


def _sources():
    return sorted(p for ext in ("*.hip", "*.hpp", "*.inc", "*.cpp") for p in glob.glob(os.path.join(CSRC, ext)))


def _design_switches():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    sec = text[text.index("## 7."):text.index("## 8.")]
    return set(re.findall(r"`(RM_[A-Z0-9_]+)`", sec))


def test_every_environment_switch_is_documented_as_result_neutral():
    allowed = _design_switches()
    assert "RM_NOISE_OFF" in allowed or True          # (mentioned there as gone; it must not be READ, below)
    read = set()
    for path in _sources():
        text = open(path).read()
        text = re.sub(r"//[^\n]*", "", text)             # comments may mention names
        for name in re.findall(r'getenv\(\s*"([^"]+)"\s*\)', text):
            read.add(name)
        # names handed to the two helpers of Switches::load
        for name in re.findall(r'\b(?:on|num)\(\s*"([^"]+)"\s*\)', text):
            read.add(name)
    assert read, "the scan found no switch at all: the pattern is stale"
    assert "RM_NOISE_OFF" not in read, "a run-time switch that drops the tie noise changes results"
    undocumented = read - allowed
    assert not undocumented, "switches read by the library but not listed in DESIGN.md section 7: %s" % sorted(undocumented)


def test_the_environment_is_read_in_one_place_only():
    """getenv appears in rm_lib.hip's `Switches` and nowhere else in the product's sources; nothing else (secure_getenv,
    environ) reads the environment"""
    for path in _sources():
        text = re.sub(r"//[^\n]*", "", open(path).read())
        if os.path.basename(path) != "rm_lib.hip":
            assert "getenv" not in text and "environ" not in text, path
            continue
        body = text[text.index("struct Switches {"):text.index("Switches g_sw;")]
        assert text.count("getenv") == body.count("getenv"), "getenv outside struct Switches"
        assert "environ" not in text.replace("environment", "")


def test_python_side_reads_only_the_library_path_overrides():
    """the Python package looks at the environment for two things: where the library is (RECOMETRICS_HIP_LIB) and, in the
    sanitizer test, which build of the host-only units to load (RECOMETRICS_SPLIT_LIB)"""
    names = set()
    for path in glob.glob(os.path.join(ROOT, "recometrics_amd", "*.py")):
        names |= set(re.findall(r'environ(?:\.get)?\(?\[?\s*"([^"]+)"', open(path).read()))
    assert names <= {"RECOMETRICS_HIP_LIB", "RECOMETRICS_SPLIT_LIB"}, names
