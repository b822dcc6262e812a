"""The C++ face's model views (print / report / simplereport / histogram) against the text the REAL reference prints for the same
model file (tests/golden/views/, made by tests/golden/make_views.py from reference include/patternmodel.h:2294-2601, :2907-2959, :3390-3450).

report / histogram: byte-identical. print: identical header, identical rows as a multiset (row order is unordered_map iteration order).
The CPU tests load the reference-written model; the GPU tests train the same corpus on the device through the CLI and must print the same text.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
VIEWS = os.path.join(GOLD, "views")
CLI = os.path.join(ROOT, "colibri-core_amd", "bin", "colibri-patternmodeller")

CASES = {  # case -> (corpus, CLI training flags, class file)
    "hamlet.u": ("hamlet.v2", ["-u", "-l", "5", "-t", "2"], "hamlet.colibri.cls"),
    "hamlet.us": ("hamlet.v2", ["-u", "-s", "-l", "5", "-t", "2"], "hamlet.colibri.cls"),
    "hamlet.i": ("hamlet.v2", ["-l", "5", "-t", "2"], "hamlet.colibri.cls"),
    "hamlet.is": ("hamlet.v2", ["-s", "-l", "5", "-t", "2"], "hamlet.colibri.cls"),
    "zipf20k.us": ("zipf20k", ["-u", "-s", "-l", "3", "-t", "2", "-y", "3"], "synthetic.colibri.cls"),
    "zipf20k.is": ("zipf20k", ["-s", "-l", "3", "-t", "2"], "synthetic.colibri.cls"),
}
VIEW_FLAGS = {"print": "-P", "report": "-R", "simplereport": "-r", "histogram": "-H"}


def golden(case, view):
    return open(os.path.join(VIEWS, f"{case}.{view}.txt"), "rb").read()


def run_cli(args):
    if not os.path.exists(CLI):
        pytest.fail(f"{CLI} is not built (python -c 'import __graft_entry__ as g; g.build()')")
    p = subprocess.run([CLI] + args, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stdout


def check(case, view, out):
    want = golden(case, view)
    if view == "print":
        got_lines, want_lines = out.split(b"\n"), want.split(b"\n")
        assert got_lines[0] == want_lines[0]
        assert sorted(got_lines[1:]) == sorted(want_lines[1:])
    else:
        assert out == want


@pytest.mark.parametrize("view", list(VIEW_FLAGS))
@pytest.mark.parametrize("case", list(CASES))
def test_view_of_reference_model(case, view):
    _, flags, cls = CASES[case]
    unindexed = ["-u"] if "-u" in flags else []
    out = run_cli(["-i", os.path.join(VIEWS, f"{case}.colibri.patternmodel"), "-c", os.path.join(GOLD, cls), VIEW_FLAGS[view]] + unindexed)
    check(case, view, out)


def test_views_in_one_call_share_the_stream_state():
    """-P -R -H together: the reference prints them in that order on one stream (src/patternmodeller.cpp:246-261)."""
    case = "hamlet.is"
    out = run_cli(["-i", os.path.join(VIEWS, f"{case}.colibri.patternmodel"), "-c", os.path.join(GOLD, "hamlet.colibri.cls"), "-P", "-R", "-H"])
    p, r, h = golden(case, "print"), golden(case, "report"), golden(case, "histogram")
    assert len(out) == len(p) + len(r) + len(h)
    assert out.endswith(r + h)


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
def test_views_after_device_training(case):
    corpus, flags, cls = CASES[case]
    for view, vf in VIEW_FLAGS.items():
        out = run_cli(["-f", os.path.join(GOLD, f"{corpus}.colibri.dat"), "-c", os.path.join(GOLD, cls), vf] + flags)
        check(case, view, out)
