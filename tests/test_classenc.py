"""Class encoder (ClassEncoder::build / save / encodefile, reference src/classencoder.cpp:134-277, :369-600).
Goldens under tests/golden/classenc/ come from the REAL reference (tests/golden/make_classenc_golden.py).
CPU: the oracle (oracle/classenc_oracle.cpp) against every golden — byte-identical .colibri.dat, identical class maps, same status.
GPU: the HIP path (colibri_text_* in libcolibri_hip.so, through the C++ face's ClassEncoder and the colibri-classencode CLI) against the
same goldens and against the oracle on random texts; the encoded corpus feeds colibri-patternmodeller unchanged."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402

G = os.path.join(GOLDEN, "classenc")
CLI = os.path.join(ROOT, "colibri-core_amd", "bin", "colibri-classencode")

CASES = {  # case -> (text, threshold, allowunknown, class file of which case, extend)
    "apology": ("apology.txt", 0, False, None, False),
    "apology.t3U": ("apology.txt", 3, True, None, False),
    "apology.t3": ("apology.txt", 3, False, None, False),
    "quirks": ("quirks.txt", 0, False, None, False),
    "quirks.U": ("quirks.txt", 0, True, None, False),
    "zipf": ("zipf.txt", 0, False, None, False),
    "zipf.t2U": ("zipf.txt", 2, True, None, False),
    "quirks.c_apologyU": ("quirks.txt", 0, True, "apology", False),
    "quirks.c_apology": ("quirks.txt", 0, False, "apology", False),
    "quirks.c_apology.e": ("quirks.txt", 0, False, "apology", True),
    "zipf.c_quirksU.e.t2": ("zipf.txt", 2, False, "quirks.U", True),
}


def golden(case):
    rc = int(open(os.path.join(G, case + ".rc")).read())
    if rc != 0:
        return rc, None, None
    return rc, open(os.path.join(G, case + ".colibri.cls"), "rb").read(), open(os.path.join(G, case + ".colibri.dat"), "rb").read()


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_matches_reference_goldens(case):
    textfile, threshold, allowunknown, clsof, extend = CASES[case]
    text = open(os.path.join(G, textfile), "rb").read()
    cls = open(os.path.join(G, clsof + ".colibri.cls"), "rb").read() if clsof else None
    st, ocls, odat = oracle.classencode(text, threshold, allowunknown, cls, extend)
    rc, gcls, gdat = golden(case)
    assert (st != 0) == (rc != 0)
    if rc == 0:
        assert odat == gdat
        assert oracle.parse_cls(ocls) == oracle.parse_cls(gcls)
        assert ocls == gcls  # even the line order: the same libstdc++ containers, filled in the same order


def random_text(rng, nlines=60):
    alphabet = [b"a", b"b", b"ab", b"the", b"cat", b"\t", b"\r", b"x\t", b"\tx", b"{*}", b"{**}", b"{?}", b"{*2*}", b"\b", b"y\b", "é".encode(), b"zz" * 120]
    lines = []
    for _ in range(nlines):
        n = int(rng.integers(0, 12))
        toks = [alphabet[int(i)] for i in np.minimum(rng.pareto(0.8, size=n).astype(np.int64), len(alphabet) - 1)]
        sep = [b" " * int(rng.integers(1, 3)) for _ in toks]
        lines.append(b"".join(t + s for t, s in zip(toks, sep)))
    return b"\n".join(lines) + (b"\n" if rng.random() < 0.7 else b"")


@pytest.mark.skipif(not oracle.have_ref(), reason="needs oracle/_ref/ref_driver (build container)")
def test_oracle_matches_reference_on_random_texts(tmp_path):
    rng = np.random.default_rng(7)
    for k in range(25):
        text = random_text(rng)
        p = tmp_path / f"t{k}.txt"
        p.write_bytes(text)
        allowunknown = bool(k % 2)
        rc = oracle.ref_classencode(str(p), str(tmp_path / f"r{k}"), threshold=k % 3, allowunknown=allowunknown)
        st, ocls, odat = oracle.classencode(text, k % 3, allowunknown)
        assert (st != 0) == (rc != 0), (k, st, rc)
        if rc == 0:
            assert odat == (tmp_path / f"r{k}.colibri.dat").read_bytes(), k
            assert ocls == (tmp_path / f"r{k}.colibri.cls").read_bytes(), k
