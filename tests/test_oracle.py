"""CPU tests (no GPU): pin the oracle (oracle/colibri_oracle.c) against
  (1) the reference's own fixtures and known answers (tests/golden/hamlet.v1.*; src/test.cpp:1214-1221, :1246, :1258,
      :1268-1283, :1327-1337),
  (2) golden dumps produced by the real reference (tests/golden/*.txt, made by tests/golden/make_golden.py),
  (3) where oracle/_ref/ref_driver is available, the real reference run live on fresh seeded corpora.
"""
import glob
import json
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN

import oracle
from colibri_amd import synth

MODES = {
    "u": {}, "us": dict(doskipgrams_exhaustive=True), "usy3": dict(doskipgrams_exhaustive=True, mintokens_skipgrams=3),
    "i": dict(indexed=True), "is": dict(indexed=True, doskipgrams=True), "isT1": dict(indexed=True, doskipgrams=True, minskiptypes=1),
    "ut1": dict(mintokens=1), "it1": dict(indexed=True, mintokens=1),  # MINTOKENS = 1: the reference's single pass over all lengths
    "ust1": dict(doskipgrams_exhaustive=True, mintokens=1), "usy3t1": dict(doskipgrams_exhaustive=True, mintokens_skipgrams=3, mintokens=1),  # ... with skipgrams
    "ist1": dict(indexed=True, doskipgrams=True, mintokens=1), "isT1t1": dict(indexed=True, doskipgrams=True, minskiptypes=1, mintokens=1),
}


def read_payload(name):
    data = open(os.path.join(GOLDEN, name + ".colibri.dat"), "rb").read()
    if data[:1] == b"\xa2":
        assert data[1] == 2
        return data[2:]
    return oracle.v1_to_v2(data)  # v1 fixture (reference classdecoder.cpp:259-284 decides by the first byte)


def test_spooky_known_answers():
    kat = json.load(open(os.path.join(GOLDEN, "spooky_kat.json")))
    assert len(kat) > 60
    for hx, want in kat.items():
        assert f"{oracle.spooky64(bytes.fromhex(hx)):016x}" == want, hx
    # the three values quoted in SURVEY.md §8 a-5
    assert oracle.spooky64(bytes([6])) == 0x5D3553AC0AA134FA
    assert oracle.spooky64(bytes([6, 7, 8])) == 0x6EE4E90E1C0B57C9


def test_skip_configurations():
    masks = json.load(open(os.path.join(GOLDEN, "masks.json")))
    for key, want in masks.items():
        n, ms = (int(x) for x in key.split(","))
        assert oracle.skip_configurations(n, ms) == want, key
    assert len(oracle.skip_configurations(6, 3)) == 15  # src/test.cpp:1258


def golden_cases():
    for path in sorted(glob.glob(os.path.join(GOLDEN, "*.l*.txt"))):
        base = os.path.basename(path)[:-4]
        corpus, tag, l = base.rsplit(".", 2)
        yield pytest.param(corpus, tag, int(l[1:]), path, id=base)


@pytest.mark.parametrize("corpus,tag,maxlength,path", list(golden_cases()))
def test_oracle_matches_reference_goldens(corpus, tag, maxlength, path):
    kw = dict(MODES[tag])
    want = oracle.parse_dump(open(path).read(), indexed=kw.get("indexed", False))
    got = oracle.train(read_payload(corpus), kw.pop("mintokens", 2), maxlength, **kw)
    assert (got.tokens, got.types) == (want.tokens, want.types)
    assert got.counts == want.counts
    if want.refs is not None:
        assert got.refs == want.refs


@pytest.mark.parametrize("corpus", ["hamlet.v2", "zipf20k"])
@pytest.mark.parametrize("mode", ["u", "i", "us", "is"])
def test_oracle_matches_reference_word_threshold_goldens(corpus, mode):
    """-W 4 (MINTOKENS_UNIGRAMS) in all four kinds of model, skipgrams included: dumps of the real reference"""
    kw = {"u": {}, "i": dict(indexed=True), "us": dict(doskipgrams_exhaustive=True), "is": dict(indexed=True, doskipgrams=True)}[mode]
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"wordthreshold.{corpus}.{mode}.W4.txt")).read(), indexed=kw.get("indexed", False))
    got = oracle.train(read_payload(corpus), 2, 5, mintokens_unigrams=4, **kw)
    assert (got.tokens, got.types, got.counts) == (want.tokens, want.types, want.counts)
    if want.refs is not None:
        assert got.refs == want.refs


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "patternlist.*.txt"))), ids=os.path.basename)
def test_patternlist_restatement_matches_reference_dumps(path):
    """DOPATTERNPERLINE (patternmodeller -L, implies -t 1): dumps of the real reference"""
    name, l = os.path.basename(path)[len("patternlist."):-len(".txt")].rsplit(".", 1)
    want = oracle.parse_dump(open(path).read())
    got = oracle.train_patternperline(read_payload(name), int(l[1:]))
    assert (got.tokens, got.types, got.counts) == (want.tokens, want.types, want.counts)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "backoff.*.txt"))), ids=os.path.basename)
def test_oracle_matches_reference_backoff_goldens(path):
    """MAXBACKOFFLENGTH (-b) 1 and 2 at MAXLENGTH 8: dumps of the real reference"""
    name, mode, b = os.path.basename(path)[len("backoff."):-len(".txt")].rsplit(".", 2)
    want = oracle.parse_dump(open(path).read(), indexed=mode == "i")
    got = oracle.train(read_payload(name), 2, 8, indexed=mode == "i", maxbackofflength=int(b[1:]))
    assert (got.tokens, got.types, got.counts) == (want.tokens, want.types, want.counts)
    if mode == "i":
        assert got.refs == want.refs


def test_hamlet_fixture_model_file():
    """exp/hamlet.v1.colibri.patternmodel (the reference's only committed golden model): 111 patterns, tokens 354, types 186."""
    raw = open(os.path.join(GOLDEN, "hamlet.v1.colibri.patternmodel"), "rb").read()
    assert raw[0] == 0 and raw[1] == 10 and raw[2] == 1  # null, UNINDEXEDPATTERNMODEL, model version 1 (v1 class encoding)
    tokens, types, npat = struct.unpack_from("<QQQ", raw, 3)
    assert (tokens, types, npat) == (354, 186, 111)
    # v1 model: patterns are v1-encoded (length-prefixed tokens), each followed by 00 and a u32 count
    pos, model = 27, {}
    for _ in range(npat):
        start = pos
        while raw[pos] != 0:
            pos += 1 + raw[pos] if raw[pos] < 128 else 1
        key = oracle.v1_to_v2(raw[start:pos])
        pos += 1
        model[key] = struct.unpack_from("<I", raw, pos)[0]
        pos += 4
    assert pos == len(raw)
    got = oracle.train(read_payload("hamlet.v1"), 2, 100)
    assert got.counts == model
    assert (got.tokens, got.types, len(got)) == (354, 186, 111)
    by_n = {}
    for k in got.counts:
        n = sum(1 for b in k if b < 128)
        by_n[n] = by_n.get(n, 0) + 1
    assert [by_n[n] for n in range(1, 8)] == [45, 22, 14, 12, 9, 6, 3]


def test_hamlet_known_answers_from_reference_tests():
    p = read_payload("hamlet.v2")
    m = oracle.train(p, -1, 100)  # default options: MINTOKENS -1 -> 2
    assert (len(m), m.types, m.tokens) == (111, 186, 354)  # src/test.cpp:1214-1221
    assert m.stats[1] == (186, 141, 45) and m.stats[2] == (71, 49, 22) and m.stats[3] == (15, 1, 14)
    m = oracle.train(p, -1, 100, doskipgrams_exhaustive=True)
    assert len(m) == 385  # src/test.cpp:1268-1283, test.py:236
    m = oracle.train(p, -1, 100, indexed=True, doskipgrams=True)
    assert len(m) == 133  # src/test.cpp:1327-1337, test.py:289
    ngrams = oracle.train(p, -1, 100, indexed=True)
    assert len(ngrams) == 111 and 133 == 111 + sum(1 for k in m.counts if k not in ngrams.counts)
    # config 1 of BASELINE.json
    m = oracle.train(p, 2, 3)
    assert len(m) == 81 and [m.stats[n][2] for n in (1, 2, 3)] == [45, 22, 14]


def test_v1_to_v2_roundtrip_tokens():
    v1 = open(os.path.join(GOLDEN, "hamlet.v1.colibri.dat"), "rb").read()
    v2 = oracle.v1_to_v2(v1)
    assert v2.count(b"\x00") == 40  # 40 sentences (src/test.cpp:1549)
    assert sum(1 for b in v2 if b < 128) - 40 == 354


def test_edge_cases():
    assert len(oracle.train(b"", 2, 5)) == 0
    assert len(oracle.train(b"\x00\x00", 2, 5)) == 0
    m = oracle.train(b"\x06\x07\x06\x07", 2, 5)  # no trailing delimiter
    assert m.counts == {b"\x06": 2, b"\x07": 2, b"\x06\x07": 2} and m.tokens == 4
    m = oracle.train(b"\x00\x06\x07\x00\x00\x06\x07\x00", 2, 5, indexed=True)
    assert m.refs[b"\x06\x07"] == [(2, 0), (4, 0)]  # empty sentences are numbered (pattern.cpp:1947-1958)
    m = oracle.train(b"\x06\x07\x00\x06\x07\x00", 2, 5, indexed=True, firstsentence=11)
    assert m.refs[b"\x06"] == [(11, 0), (12, 0)]


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/ref_driver not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_matches_live_reference(tmp_path, seed):
    rng = np.random.default_rng(seed)
    payload = synth.random_corpus(rng, nsent=250, maxlen=14, vocab=10 + seed * 5)
    path = str(tmp_path / "c.colibri.dat")
    open(path, "wb").write(synth.HEADER + payload)
    for mode, kw in [("u", {}), ("U", {}), ("us", dict(doskipgrams_exhaustive=True)), ("i", dict(indexed=True))]:
        want, _ = oracle.ref_train(path, mode, 6, 2, dump_path=str(tmp_path / "d.txt"))
        got = oracle.train(payload, 2, 6, **kw)
        assert (got.tokens, got.types, got.counts, got.refs) == (want.tokens, want.types, want.counts, want.refs), mode


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/ref_driver not built (needs /root/reference)")
def test_indexed_skipgrams_reference_hazard_documented(tmp_path):
    """IndexedPatternModel::trainskipgrams inserts into the unordered_map it iterates (patternmodel.h:2986-2991).
    When that rehashes mid-loop the reference revisits/skips n-grams (duplicate refs, missing skipgrams). The
    specification followed by the oracle and the HIP path is the clean semantics; this test
    documents that the divergence is exactly that: duplicate references in the reference's own output."""
    rng = np.random.default_rng(1)
    payload = synth.random_corpus(rng, nsent=200, maxlen=14, vocab=12)
    path = str(tmp_path / "c.colibri.dat")
    open(path, "wb").write(synth.HEADER + payload)
    want, _ = oracle.ref_train(path, "is", 5, 2, dump_path=str(tmp_path / "d.txt"), minskiptypes=1)
    got = oracle.train(payload, 2, 5, indexed=True, doskipgrams=True, minskiptypes=1)
    if got.counts != want.counts:
        assert any(len(set(r)) != len(r) for r in want.refs.values()), "reference differs without duplicate refs: not the known hazard"
    ngr = {k: v for k, v in got.counts.items() if 3 not in k}
    assert ngr == {k: v for k, v in want.counts.items() if 3 not in k}


# ---- constrained training: the restatement against the real reference's dumps ------------------------------------------
def _model_keys(path, mincount=1):
    """key bytes of every pattern of a .colibri.patternmodel (v2) file that occurs at least `mincount` times"""
    import struct
    raw = open(path, "rb").read()
    mtype = raw[1]
    npat = struct.unpack_from("<Q", raw, 19)[0]
    pos, keys = 27, []
    for _ in range(npat):
        start, prevhigh = pos, False
        while prevhigh or raw[pos] != 0:
            prevhigh = raw[pos] >= 128
            pos += 1
        (c,) = struct.unpack_from("<I", raw, pos + 1)
        if c >= mincount:
            keys.append(raw[start:pos])
        pos += 5 + (6 * c if mtype == 20 else 0)
    return keys


CONSTRAINED_GOLDENS = {  # tag -> (corpus, constraint model, indexed, mintokens, maxlength, minlength)
    "j_zipf.u.t1": ("phrases15k", "constraint.zipf20k.u.l5.patternmodel", False, 1, 4, 1),
    "j_zipf.u.t2": ("phrases15k", "constraint.zipf20k.u.l5.patternmodel", False, 2, 4, 1),
    "j_zipf.u.t3": ("phrases15k", "constraint.zipf20k.u.l5.patternmodel", False, 3, 5, 1),
    "j_zipf.u.t1m2": ("phrases15k", "constraint.zipf20k.u.l5.patternmodel", False, 1, 4, 2),
    "j_zipf.i.t1": ("phrases15k", "constraint.zipf20k.u.l5.patternmodel", True, 1, 4, 1),
    "j_zipf.i.t2": ("phrases15k", "constraint.zipf20k.u.l5.patternmodel", True, 2, 5, 1),
    "j_hamlet.u.t1": ("edge", "constraint.hamlet.i.l5.patternmodel", False, 1, 5, 1),
    "j_self.u.t2": ("zipf20k", "constraint.zipf20k.u.l5.patternmodel", False, 2, 5, 1),
    "I_zipf.u.t2": ("phrases15k", "constraint.zipf20k.u.l5.patternmodel", False, 2, 5, 1),
    "I_zipf.i.t1": ("phrases15k", "constraint.zipf20k.u.l5.patternmodel", True, 1, 5, 1),
    "I_self.i.t2": ("hamlet.v2", "constraint.hamlet.i.l5.patternmodel", True, 2, 5, 1),
    # with skipgrams (ref_driver modes us / is with -j): only a run at MINTOKENS = 1 computes any; extra = (MINTOKENS_SKIPGRAMS, MINSKIPTYPES)
    "js_hamlet.us.t1": ("hamlet.v2", "constraint.hamlet.us.l5.patternmodel", False, 1, 5, 1, (-1, 2)),
    "js_hamlet.us.t1T1": ("hamlet.v2", "constraint.hamlet.us.l5.patternmodel", False, 1, 5, 1, (-1, 1)),
    "js_hamlet.us.t1y3": ("hamlet.v2", "constraint.hamlet.us.l5.patternmodel", False, 1, 5, 1, (3, 2)),
    "js_hamlet.is.t1": ("hamlet.v2", "constraint.hamlet.us.l5.patternmodel", True, 1, 5, 1, (-1, 2)),
    "js_hamlet.is.t1T1": ("hamlet.v2", "constraint.hamlet.us.l5.patternmodel", True, 1, 5, 1, (-1, 1)),
    "js_edge.us.t1": ("edge", "constraint.hamlet.us.l5.patternmodel", False, 1, 5, 1, (-1, 2)),
    "js_hamlet.us.t2": ("hamlet.v2", "constraint.hamlet.us.l5.patternmodel", False, 2, 5, 1, (-1, 2)),
    "js_hamlet.is.t2": ("hamlet.v2", "constraint.hamlet.us.l5.patternmodel", True, 2, 4, 1, (-1, 2)),
    "js_zipf.us.t1": ("phrases15k", "constraint.zipf20k.us.l4t3.patternmodel", False, 1, 4, 1, (-1, 2)),
    "js_zipf.us.t1y4": ("phrases15k", "constraint.zipf20k.us.l4t3.patternmodel", False, 1, 4, 1, (4, 2)),
    "js_zipf.is.t1": ("phrases15k", "constraint.zipf20k.us.l4t3.patternmodel", True, 1, 4, 1, (-1, 2)),
    "js_zipf.is.t1y4": ("phrases15k", "constraint.zipf20k.us.l4t3.patternmodel", True, 1, 4, 1, (4, 2)),
    "js_zipf.is.t1y4T1": ("phrases15k", "constraint.zipf20k.us.l4t3.patternmodel", True, 1, 4, 1, (4, 1)),
}


@pytest.mark.parametrize("tag", sorted(CONSTRAINED_GOLDENS))
def test_constrained_restatement_matches_reference_dumps(tag):
    """oracle.train_constrained against ref_driver train ... -j / -I (pattern set, counts, reference lists; the totals are the C++ face's)"""
    import oracle
    corpus, cmodel, indexed, mintokens, maxlength, minlength = CONSTRAINED_GOLDENS[tag][:6]
    skip = CONSTRAINED_GOLDENS[tag][6] if len(CONSTRAINED_GOLDENS[tag]) > 6 else None
    payload = open(os.path.join(GOLDEN, corpus + ".colibri.dat"), "rb").read()[2:]
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"constrained.{tag}.txt")).read(), indexed=indexed)
    # the constraint model is loaded under the run's own options (src/patternmodeller.cpp:712-718): its patterns below MINTOKENS are not read
    kw = dict(doskipgrams=True, mintokens_skipgrams=skip[0], minskiptypes=skip[1]) if skip else {}
    got = oracle.train_constrained(payload, _model_keys(os.path.join(GOLDEN, cmodel), mintokens), mintokens, maxlength, minlength, indexed=indexed, **kw)
    assert got.counts == want.counts
    if indexed:
        assert got.refs == want.refs


CONTINUED = {  # golden tag -> (corpus, loaded model fixture, indexed, mintokens, maxlength)
    "E_zipf.u": ("zipf20k", "continued.zipf20k.u.t3l2.patternmodel", False, 2, 5),
    "E_hamlet.i": ("hamlet.v2", "continued.hamlet.i.t2l3.patternmodel", True, 2, 6),
    "E_cross.u": ("phrases15k", "continued.zipf20k.u.t3l2.patternmodel", False, 2, 4),
}


def load_model_file(path):
    """a .colibri.patternmodel written by the reference -> oracle.Model (file format: reference include/patternmodel.h:781-861)"""
    import struct
    raw = open(path, "rb").read()
    assert raw[0] == 0 and raw[2] == 2
    mtype = raw[1]
    tokens, types, npat = struct.unpack_from("<QQQ", raw, 3)
    pos, counts, refs = 27, {}, {}
    for _ in range(npat):
        start, prevhigh = pos, False
        while prevhigh or raw[pos] != 0:
            prevhigh = raw[pos] >= 128
            pos += 1
        key = raw[start:pos]
        pos += 1
        (c,) = struct.unpack_from("<I", raw, pos)
        pos += 4
        counts[key] = c
        if mtype == 20:
            refs[key] = [struct.unpack_from("<IH", raw, pos + 6 * k) for k in range(c)]
            pos += 6 * c
    assert pos == len(raw)
    return oracle.Model(tokens, types, counts, refs if mtype == 20 else None)


@pytest.mark.parametrize("tag", list(CONTINUED))
def test_continued_training_restatement_matches_the_reference(tag):
    """train(..., continued = true) (include/patternmodel.h:983-995): the restatement against dumps of the real reference continuing models it wrote itself"""
    corpus, fixture, indexed, mintokens, maxlength = CONTINUED[tag]
    payload = open(os.path.join(GOLDEN, corpus + ".colibri.dat"), "rb").read()[2:]
    loaded = load_model_file(os.path.join(GOLDEN, fixture))
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"continued.{tag}.txt")).read(), indexed=indexed)
    got = oracle.train_continued(payload, loaded, mintokens, maxlength, indexed=indexed)
    assert (got.tokens, got.types) == (want.tokens, want.types)
    assert got.counts == want.counts
    if indexed:
        assert got.refs == want.refs
    assert len(got.counts) > len(loaded.counts)  # the run did add longer patterns


FILTERED = {  # golden tag -> (corpus, filter fixture, indexed, mintokens, maxlength)
    "f_ngrams.u.t2": ("zipf20k", "ngrams", False, 2, 5), "f_ngrams.i.t2": ("zipf20k", "ngrams", True, 2, 4), "f_skip.u.t2": ("zipf20k", "skipgrams", False, 2, 5),
    "f_mixed.u.t2": ("zipf20k", "mixed", False, 2, 6), "f_mixed.i.t3": ("phrases15k", "mixed", True, 3, 5), "f_mixed.u.t1": ("hamlet.v2", "mixed", False, 1, 4),
    "f_ngrams.u.t1": ("zipf20k", "ngrams", False, 1, 3), "f_skip.u.t1": ("zipf20k", "skipgrams", False, 1, 4), "f_skip.i.t1": ("zipf20k", "skipgrams", True, 1, 4),
}


@pytest.mark.parametrize("tag", sorted(FILTERED))
def test_filtered_training_restatement_matches_the_reference(tag):
    """train(..., filter) (include/patternmodel.h:899-914, :1106-1137): the restatement against dumps of the real reference (ref_driver train -f)"""
    corpus, flt, indexed, mintokens, maxlength = FILTERED[tag]
    payload = open(os.path.join(GOLDEN, corpus + ".colibri.dat"), "rb").read()[2:]
    keys = _model_keys(os.path.join(GOLDEN, f"filter.{flt}.patternmodel"))
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"filtered.{tag}.txt")).read(), indexed=indexed)
    got = oracle.train_filtered(payload, keys, mintokens, maxlength, indexed=indexed)
    assert (got.tokens, got.types) == (want.tokens, want.types)
    assert got.counts == want.counts
    if indexed:
        assert got.refs == want.refs
