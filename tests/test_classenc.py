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


# ---------------------------------------------------------------------------------------------------------------------------------
# GPU: the HIP path behind the C++ face (ClassEncoder) and the colibri-classencode CLI
# ---------------------------------------------------------------------------------------------------------------------------------
def run_cli(args, cwd):
    if not os.path.exists(CLI):
        pytest.fail(f"{CLI} is not built (python -c 'import __graft_entry__ as g; g.build()')")
    return subprocess.run([CLI] + args, capture_output=True, cwd=cwd, timeout=600)


def cli_flags(case):
    textfile, threshold, allowunknown, clsof, extend = CASES[case]
    flags = ["-t", str(threshold)] if threshold else []
    if allowunknown:
        flags.append("-U")
    if clsof:
        flags += ["-c", os.path.join(G, clsof + ".colibri.cls")]
    if extend:
        flags.append("-e")
    return flags, os.path.join(G, textfile)


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
def test_cli_matches_reference_goldens(case, tmp_path):
    flags, text = cli_flags(case)
    p = run_cli(flags + ["-o", "out", text], str(tmp_path))
    rc, gcls, gdat = golden(case)
    assert (p.returncode != 0) == (rc != 0), p.stderr.decode()[-1500:]
    if rc == 0:
        stem = os.path.basename(text)[:-4]
        assert (tmp_path / f"{stem}.colibri.dat").read_bytes() == gdat
        if CASES[case][3] is None or CASES[case][4]:  # with -c and without -e colibri-classencode writes no class file (reference src/classencode.cpp:137-146)
            got_cls = (tmp_path / "out.colibri.cls").read_bytes()
            assert oracle.parse_cls(got_cls) == oracle.parse_cls(gcls)
            assert got_cls == gcls  # same containers filled in the same order: even the line order of the class file agrees
    else:
        assert p.returncode == 4  # UnknownTokenError


@pytest.mark.gpu
def test_cli_matches_oracle_on_random_texts(tmp_path):
    rng = np.random.default_rng(11)
    for k in range(30):
        text = random_text(rng, nlines=int(rng.integers(1, 120)))
        (tmp_path / f"t{k}.txt").write_bytes(text)
        allowunknown, threshold = bool(k % 2), k % 3
        st, ocls, odat = oracle.classencode(text, threshold, allowunknown)
        p = run_cli((["-U"] if allowunknown else []) + (["-t", str(threshold)] if threshold else []) + [f"t{k}.txt"], str(tmp_path))
        assert (p.returncode != 0) == (st != 0), (k, p.stderr.decode()[-800:])
        if st == 0:
            assert (tmp_path / f"t{k}.colibri.dat").read_bytes() == odat, k
            assert (tmp_path / f"t{k}.colibri.cls").read_bytes() == ocls, k


@pytest.mark.gpu
def test_text_to_model_end_to_end(tmp_path):
    """colibri-classencode then colibri-patternmodeller, both on the device: the model of the reference's own apology.txt equals the
    oracle's model of the reference-encoded corpus."""
    p = run_cli([os.path.join(G, "apology.txt")], str(tmp_path))
    assert p.returncode == 0, p.stderr.decode()
    dat = (tmp_path / "apology.colibri.dat").read_bytes()
    assert dat == golden("apology")[2]
    modeller = os.path.join(ROOT, "colibri-core_amd", "bin", "colibri-patternmodeller")
    out = subprocess.run([modeller, "-f", "apology.colibri.dat", "-u", "-t", "2", "-l", "5", "-o", "apology.colibri.patternmodel"], capture_output=True, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr.decode()
    want = oracle.train(dat[2:], 2, 5)
    from test_host_face import parse_model
    mtype, tokens, types, counts, _ = parse_model(str(tmp_path / "apology.colibri.patternmodel"))
    assert (mtype, tokens, types) == (10, want.tokens, want.types) and counts == want.counts


@pytest.mark.gpu
def test_c_abi_text_path_feeds_train_without_host_round_trip():
    """colibri_text_upload / count / words / encode / as_corpus through ctypes: the encoded stream never leaves the device before train()"""
    import ctypes as C
    from colibri_amd import capi
    text = open(os.path.join(G, "apology.txt"), "rb").read()
    _, _, gdat = golden("apology")
    L = capi.load()
    with capi.Context(0) as ctx:
        h = ctx.h
        nwords, nd = C.c_uint64(), C.c_uint64()
        assert L.colibri_text_upload(h, text, C.c_uint64(len(text))) == 0
        assert L.colibri_text_count(h, 1, C.byref(nwords), C.byref(nd)) == 0
        n = nd.value
        start, length, count = (np.zeros(n, dtype=np.uint32) for _ in range(3))
        assert L.colibri_text_words(h, start.ctypes.data_as(C.c_void_p), length.ctypes.data_as(C.c_void_p), count.ctypes.data_as(C.c_void_p)) == 0
        assert int(count.sum()) == nwords.value == len(text.split())
        gcls = oracle.parse_cls(golden("apology")[1])
        cls = np.array([gcls[text[s:s + ln]] for s, ln in zip(start.tolist(), length.tolist())], dtype=np.uint32)
        rep = np.ones(n, dtype=np.uint32)
        ob, nt, nl = C.c_uint64(), C.c_uint64(), C.c_uint64()
        assert L.colibri_text_encode(h, cls.ctypes.data_as(C.c_void_p), rep.ctypes.data_as(C.c_void_p), C.byref(ob), C.byref(nt), C.byref(nl)) == 0
        assert (ob.value, nt.value, nl.value) == (len(gdat) - 2, nwords.value, text.count(b"\n"))
        out = np.zeros(ob.value, dtype=np.uint8)
        assert L.colibri_text_fetch(h, out.ctypes.data_as(C.c_void_p)) == 0
        assert out.tobytes() == gdat[2:]
        assert L.colibri_text_as_corpus(h, 1) == 0
        st = ctx.train(mintokens=2, maxlength=5)
        got, _ = ctx.export_dict()
    want = oracle.train(gdat[2:], 2, 5)
    assert st.totaltokens == want.tokens and got == want.counts


@pytest.mark.gpu
def test_cli_matches_oracle_on_a_two_million_word_text(tmp_path):
    """Zipf text, 2 M words, 200 k word forms, every frequency tie in it: the class file and the encoded corpus equal the oracle's byte for byte"""
    rng = np.random.default_rng(12)
    vocab = 200_000
    p = 1.0 / np.arange(1, vocab + 1)
    ranks = np.searchsorted(np.cumsum(p / p.sum()), rng.random(2_000_000))
    lens = rng.integers(1, 30, size=150_000)
    ends = np.cumsum(lens)
    ends = ends[ends < ranks.size]
    words = np.array([f"w{r:x}" for r in range(vocab)], dtype=object)[ranks]
    lines, start = [], 0
    for e in ends.tolist():
        lines.append(" ".join(words[start:e]))
        start = e
    text = ("\n".join(lines) + "\n").encode()
    (tmp_path / "big.txt").write_bytes(text)
    st, ocls, odat = oracle.classencode(text)
    assert st == 0
    out = run_cli(["big.txt"], str(tmp_path))
    assert out.returncode == 0, out.stderr.decode()[-800:]
    assert (tmp_path / "big.colibri.dat").read_bytes() == odat
    assert (tmp_path / "big.colibri.cls").read_bytes() == ocls
