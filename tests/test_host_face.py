"""The C++ face (colibri-core_amd/host: PatternModelOptions / PatternModel<uint32_t> / IndexedPatternModel<> / IndexedCorpus)
and the colibri-patternmodeller-compatible CLI.
CPU part: key types, SpookyHash on the host, reading the reference's own golden model (v1), v2 write/read round trip.
GPU part: the CLI builds models through the C ABI; the written .colibri.patternmodel is parsed here AND loaded by the real
reference (oracle/_ref/ref_driver load) — the on-disk format is what makes the build a drop-in."""
import os
import struct
import subprocess

import pytest

from conftest import GOLDEN, ROOT

BIN = os.path.join(ROOT, "colibri-core_amd", "bin")
SELFTEST = os.path.join(BIN, "host_selftest")
CLI = os.path.join(BIN, "colibri-patternmodeller")


def parse_model(path):
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
            r = []
            for _ in range(c):
                s, t = struct.unpack_from("<IH", raw, pos)
                pos += 6
                r.append((s, t))
            refs[key] = r
    assert pos == len(raw)
    return mtype, tokens, types, counts, refs


def test_binaries_are_built():
    assert os.access(SELFTEST, os.X_OK) and os.access(CLI, os.X_OK), "run __graft_entry__.build()"


def test_host_selftest_cpu():
    out = subprocess.run([SELFTEST, "cpu", os.path.join(GOLDEN, "hamlet.v2.colibri.dat"), os.path.join(GOLDEN, "hamlet.v1.colibri.patternmodel")],
                         capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "OK", out.stdout + out.stderr


def test_cli_rejects_out_of_scope_flags():
    for flag in (["-g"], ["-F", "0.5"], ["-Q"]):  # relations, flexgrams from co-occurrence, query mode
        out = subprocess.run([CLI, "-f", os.path.join(GOLDEN, "hamlet.v2.colibri.dat")] + flag, capture_output=True, text=True)
        assert out.returncode == 2 and "not part of the MI355X-accelerated build" in out.stderr


def test_cli_prints_the_references_golden_model(tmp_path):
    """-i <reference's own v1 model> -c <its class file> -P: loads and prints 111 patterns under the reference's header;
    without a class file the reference refuses to print (src/patternmodeller.cpp:247-249) and so does this CLI."""
    model = os.path.join(GOLDEN, "hamlet.v1.colibri.patternmodel")
    out = subprocess.run([CLI, "-i", model, "-u", "-P", "-c", os.path.join(GOLDEN, "hamlet.colibri.cls")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert lines[0] == "PATTERN\tCOUNT\tTOKENS\tCOVERAGE\tCATEGORY\tSIZE\tFREQUENCY"
    assert len(lines) == 112
    assert ",\t27\t27\t0.0762712\tngram\t1\t0.126761" in lines
    out = subprocess.run([CLI, "-i", model, "-u", "-P"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout == ""
    assert "ERROR: Unable to print model, no class file specified (--classfile)" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("corpus,maxlength", [("hamlet.v1", 3), ("hamlet.v1", 100), ("hamlet.v2", 5), ("edge", 5), ("zipf20k", 5), ("phrases15k", 5)])
def test_cli_builds_unindexed_model_reference_can_load(tmp_path, corpus, maxlength):
    import oracle
    model = str(tmp_path / "m.colibri.patternmodel")
    data = os.path.join(GOLDEN, corpus + ".colibri.dat")
    out = subprocess.run([CLI, "-f", data, "-u", "-t", "2", "-l", str(maxlength), "-o", model], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    golden = os.path.join(GOLDEN, f"{corpus}.u.l{maxlength}.txt")
    want = oracle.parse_dump(open(golden).read())
    mtype, tokens, types, counts, _ = parse_model(model)
    assert (mtype, tokens, types) == (10, want.tokens, want.types)
    assert counts == want.counts
    # the progress lines mirror the reference's (found / pruned / kept per order)
    assert "Counting 1-grams" in out.stderr and "total kept" in out.stderr
    if oracle.have_ref():  # the real reference reads the file this build wrote
        dump = str(tmp_path / "d.txt")
        subprocess.check_call([oracle.REF_DRIVER, "load", model, "u", dump])
        got = oracle.parse_dump(open(dump).read())
        assert (got.tokens, got.types, got.counts) == (want.tokens, want.types, want.counts)


@pytest.mark.gpu
def test_cxx_api_preloaded_corpus(tmp_path):
    """PatternModel<uint32_t> model(&corpus); model.train(file, options) — src/benchmarks.cpp test 5 / src/test.cpp:1211-1221."""
    model = str(tmp_path / "m.colibri.patternmodel")
    out = subprocess.run([SELFTEST, "gpu", os.path.join(GOLDEN, "hamlet.v2.colibri.dat"), model, "U", "100", "-1"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.split() == ["111", "354", "186", "7", "27"]


@pytest.mark.gpu
@pytest.mark.parametrize("corpus,kind,l,t", [("hamlet.v2", "u", 5, 2), ("zipf20k", "u", 5, 1), ("phrases15k", "i", 4, 2), ("zipf20k", "i", 3, 1)])
def test_cxx_api_lookups_on_a_fresh_model_match_the_node_map(corpus, kind, l, t):
    """has() / occurrencecount() right after train() (reference include/patternmodel.h:1994-2050; its callers: src/test.cpp:1214-1232) are answered from the flat result
    arrays (host/include/patternmodel.h FlatIndex) — every pattern and two absent keys per pattern against the same model as unordered_map nodes."""
    out = subprocess.run([SELFTEST, "lookup", os.path.join(GOLDEN, corpus + ".colibri.dat"), kind, str(l), str(t)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    verdict, checked, absent = out.stdout.split()
    assert verdict == "OK" and int(checked) > 100 and int(absent) > 0


@pytest.mark.gpu
def test_cxx_api_errors_are_internalerror(tmp_path):
    out = subprocess.run([SELFTEST, "gpu", os.path.join(GOLDEN, "hamlet.v2.colibri.dat"), str(tmp_path / "m"), "is", "5", "2", "b2"], capture_output=True, text=True)
    assert out.returncode == 1 and "EXCEPTION" in out.stdout  # MAXBACKOFFLENGTH < MAXLENGTH with skipgrams is outside the accelerated subset: loud failure, no fallback


@pytest.mark.gpu
@pytest.mark.parametrize("corpus", ["hamlet.v2", "edge"])
@pytest.mark.parametrize("mode,flags", [("u", ["-u"]), ("i", [])])
@pytest.mark.parametrize("b", [1, 2])
def test_cli_max_backoff_length(tmp_path, corpus, mode, flags, b):
    """-b: goldens by the real reference (the model; the per-order candidate counts are compared through the C ABI in test_gpu_parity.py)"""
    import oracle
    model = str(tmp_path / "m.colibri.patternmodel")
    out = subprocess.run([CLI, "-f", os.path.join(GOLDEN, corpus + ".colibri.dat"), "-t", "2", "-l", "8", "-b", str(b), "-o", model] + flags, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"backoff.{corpus}.{mode}.b{b}.txt")).read(), indexed=mode == "i")
    mtype, tokens, types, counts, refs = parse_model(model)
    assert (tokens, types, counts) == (want.tokens, want.types, want.counts)
    if mode == "i":
        assert refs == want.refs


@pytest.mark.gpu
@pytest.mark.parametrize("corpus,flags,tag", [("hamlet.v2", [], "i"), ("hamlet.v2", ["-s"], "is"), ("hamlet.v2", ["-s", "-T", "1"], "isT1"), ("zipf20k", [], "i"),
                                              ("phrases15k", ["-s"], "is"), ("phrases15k", ["-u", "-s"], "us"), ("zipf20k", ["-u", "-s", "-y", "3"], "usy3")])
def test_cli_indexed_and_skipgram_models_reference_can_load(tmp_path, corpus, flags, tag):
    """colibri-patternmodeller [-u] [-s] ... writes model types 10 / 20 that the real reference loads back identically."""
    import oracle
    model = str(tmp_path / "m.colibri.patternmodel")
    data = os.path.join(GOLDEN, corpus + ".colibri.dat")
    out = subprocess.run([CLI, "-f", data, "-t", "2", "-l", "5", "-o", model] + flags, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    indexed = "-u" not in flags
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"{corpus}.{tag}.l5.txt")).read(), indexed=indexed)
    mtype, tokens, types, counts, refs = parse_model(model)
    assert (mtype, tokens, types) == (20 if indexed else 10, want.tokens, want.types)
    assert counts == want.counts
    if indexed:
        assert refs == want.refs
    if oracle.have_ref():
        dump = str(tmp_path / "d.txt")
        subprocess.check_call([oracle.REF_DRIVER, "load", model, "i" if indexed else "u", dump])
        got = oracle.parse_dump(open(dump).read(), indexed=indexed)
        assert (got.tokens, got.types, got.counts, got.refs) == (want.tokens, want.types, want.counts, want.refs)


@pytest.mark.gpu
@pytest.mark.parametrize("gpus,devices", [(2, "0,0"), (3, "0,0,0"), (1, None)], ids=["two-ranks-one-device", "three-ranks-one-device", "one-rank-rccl"])
@pytest.mark.parametrize("corpus,flags,tag", [("hamlet.v2", ["-u"], "u"), ("hamlet.v2", [], "i"), ("hamlet.v2", ["-s"], "is"), ("zipf20k", [], "i"), ("phrases15k", ["-s"], "is"),
                                              ("phrases15k", ["-u", "-s"], "us"), ("zipf20k", ["-u", "-s", "-y", "3"], "usy3")])
def test_cli_sharded_across_gpus(tmp_path, corpus, flags, tag, gpus, devices):
    """colibri-patternmodeller --gpus N: the C++ sharded driver (host/src/sharded.cpp) — N rank threads, each with its own device context and sentence
    range; candidates and replies exchanged between the contexts. On a one-GPU box the ranks share device 0 (COLIBRI_DEVICES=0,0: device-to-device
    copies between the contexts); --gpus 1 with COLIBRI_GPUS_FORCE_SHARDED runs the same protocol through RCCL (ncclCommInitAll on one device). The
    model must be the reference's (goldens by ref_driver train) and load back in the reference."""
    import oracle
    model = str(tmp_path / "m.colibri.patternmodel")
    data = os.path.join(GOLDEN, corpus + ".colibri.dat")
    env = dict(os.environ)
    if devices:
        env["COLIBRI_DEVICES"] = devices
    else:
        env["COLIBRI_GPUS_FORCE_SHARDED"] = "1"
    out = subprocess.run([CLI, "-f", data, "-t", "2", "-l", "5", "-o", model, "--gpus", str(gpus)] + flags, capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr
    assert f"sentence-sharded over {gpus} GPU" in out.stderr and ("RCCL" in out.stderr) == (devices is None), out.stderr
    indexed = "-u" not in flags
    golden = os.path.join(GOLDEN, f"{corpus}.{tag}.l5.txt")
    if not os.path.exists(golden):
        golden = os.path.join(GOLDEN, f"{corpus}.{tag}.txt")
    want = oracle.parse_dump(open(golden).read(), indexed=indexed)
    mtype, tokens, types, counts, refs = parse_model(model)
    assert (mtype, tokens, types) == (20 if indexed else 10, want.tokens, want.types)
    assert counts == want.counts
    if indexed:
        assert refs == want.refs
    if oracle.have_ref():
        dump = str(tmp_path / "d.txt")
        subprocess.check_call([oracle.REF_DRIVER, "load", model, "i" if indexed else "u", dump])
        got = oracle.parse_dump(open(dump).read(), indexed=indexed)
        assert (got.tokens, got.types, got.counts, got.refs) == (want.tokens, want.types, want.counts, want.refs)


@pytest.mark.gpu
@pytest.mark.parametrize("corpus,flags,tag", [("hamlet.v2", [], "is"), ("phrases15k", [], "is"), ("zipf20k", [], "is"), ("zipf20k", ["-T", "1"], "isT1")])
def test_cli_flexgrams_from_skipgrams(tmp_path, corpus, flags, tag):
    """-F S (implies -s): the model gains the flexgrams its skipgrams abstract to (reference computeflexgrams_fromskipgrams,
    include/patternmodel.h:3724-3744; goldens = the reference's own output where its loop stayed stable). The written model is loaded
    back by the real reference."""
    import oracle
    model = str(tmp_path / "m.colibri.patternmodel")
    data = os.path.join(GOLDEN, corpus + ".colibri.dat")
    out = subprocess.run([CLI, "-f", data, "-t", "2", "-l", "5", "-F", "S", "-o", model] + flags, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"flex.{corpus}.{tag}.txt")).read(), indexed=True)
    before = oracle.parse_dump(open(os.path.join(GOLDEN, f"{corpus}.{tag}.l5.txt")).read(), indexed=True)
    assert f"{len(want) - len(before)} flexgrams found" in out.stderr
    mtype, tokens, types, counts, refs = parse_model(model)
    assert (mtype, tokens, types) == (20, want.tokens, want.types)
    assert counts == want.counts and refs == want.refs
    if oracle.have_ref():
        dump = str(tmp_path / "d.txt")
        subprocess.check_call([oracle.REF_DRIVER, "load", model, "i", dump])
        got = oracle.parse_dump(open(dump).read(), indexed=True)
        assert (got.tokens, got.types, got.counts, got.refs) == (want.tokens, want.types, want.counts, want.refs)


def test_cli_flexgrams_need_a_fresh_build():
    out = subprocess.run([CLI, "-i", os.path.join(GOLDEN, "hamlet.v1.colibri.patternmodel"), "-F", "S"], capture_output=True, text=True)
    assert out.returncode == 2 and "-F S on a loaded model" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("corpus,flags,tag", [("hamlet.v2", [], "i2"), ("hamlet.v2", ["-s"], "is2"), ("zipf20k", [], "i2"), ("zipf20k", ["-s"], "is2")])
def test_cli_two_stage_build_matches_the_references(tmp_path, corpus, flags, tag):
    """colibri-patternmodeller -2 [-s]: the model the reference's two-stage build leaves (goldens by ref_driver train ... i2 / is2): the
    indexed n-grams with their references, no skipgrams, and a type count equal to the number of patterns; <out>.stage1 is the unindexed model."""
    import oracle
    model = str(tmp_path / "m.colibri.patternmodel")
    out = subprocess.run([CLI, "-f", os.path.join(GOLDEN, corpus + ".colibri.dat"), "-2", "-t", "2", "-l", "5", "-o", model] + flags, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"twostage.{corpus}.{tag}.txt")).read(), indexed=True)
    mtype, tokens, types, counts, refs = parse_model(model)
    assert (mtype, tokens, types) == (20, want.tokens, want.types)
    assert counts == want.counts and refs == want.refs
    s1 = parse_model(model + ".stage1")
    assert s1[0] == 10 and s1[3] == want.counts


def test_cli_two_stage_needs_an_output_model():
    out = subprocess.run([CLI, "-f", os.path.join(GOLDEN, "hamlet.v2.colibri.dat"), "-2"], capture_output=True, text=True)
    assert out.returncode == 2 and "mandatory for two-stage building" in out.stderr


CONSTRAINED = {  # golden tag -> (corpus, unindexed?, CLI flags)
    "j_zipf.u.t1": ("phrases15k", True, ["-l", "4", "-t", "1", "-j", "constraint.zipf20k.u.l5.patternmodel"]),
    "j_zipf.u.t2": ("phrases15k", True, ["-l", "4", "-t", "2", "-j", "constraint.zipf20k.u.l5.patternmodel"]),
    "j_zipf.u.t3": ("phrases15k", True, ["-l", "5", "-t", "3", "-j", "constraint.zipf20k.u.l5.patternmodel"]),
    "j_zipf.u.t1m2": ("phrases15k", True, ["-l", "4", "-t", "1", "-m", "2", "-j", "constraint.zipf20k.u.l5.patternmodel"]),
    "j_zipf.i.t1": ("phrases15k", False, ["-l", "4", "-t", "1", "-j", "constraint.zipf20k.u.l5.patternmodel"]),
    "j_zipf.i.t2": ("phrases15k", False, ["-l", "5", "-t", "2", "-j", "constraint.zipf20k.u.l5.patternmodel"]),
    "j_hamlet.u.t1": ("edge", True, ["-l", "5", "-t", "1", "-j", "constraint.hamlet.i.l5.patternmodel"]),
    "j_self.u.t2": ("zipf20k", True, ["-l", "5", "-t", "2", "-j", "constraint.zipf20k.u.l5.patternmodel"]),
    "I_zipf.u.t2": ("phrases15k", True, ["-l", "5", "-t", "2", "-I", "-i", "constraint.zipf20k.u.l5.patternmodel"]),
    "I_zipf.i.t1": ("phrases15k", False, ["-l", "5", "-t", "1", "-I", "-i", "constraint.zipf20k.u.l5.patternmodel"]),
    "I_self.i.t2": ("hamlet.v2", False, ["-l", "5", "-t", "2", "-I", "-i", "constraint.hamlet.i.l5.patternmodel"]),
    # -s with -j: the masked forms of member windows that the constraint model holds — computed only by a run at -t 1 (reference include/patternmodel.h:1163)
    "js_hamlet.us.t1": ("hamlet.v2", True, ["-l", "5", "-t", "1", "-s", "-j", "constraint.hamlet.us.l5.patternmodel"]),
    "js_hamlet.us.t1T1": ("hamlet.v2", True, ["-l", "5", "-t", "1", "-s", "-T", "1", "-j", "constraint.hamlet.us.l5.patternmodel"]),
    "js_hamlet.us.t1y3": ("hamlet.v2", True, ["-l", "5", "-t", "1", "-s", "-y", "3", "-j", "constraint.hamlet.us.l5.patternmodel"]),
    "js_hamlet.is.t1": ("hamlet.v2", False, ["-l", "5", "-t", "1", "-s", "-j", "constraint.hamlet.us.l5.patternmodel"]),
    "js_edge.us.t1": ("edge", True, ["-l", "5", "-t", "1", "-s", "-j", "constraint.hamlet.us.l5.patternmodel"]),
    "js_hamlet.us.t2": ("hamlet.v2", True, ["-l", "5", "-t", "2", "-s", "-j", "constraint.hamlet.us.l5.patternmodel"]),
    "js_hamlet.is.t2": ("hamlet.v2", False, ["-l", "4", "-t", "2", "-s", "-j", "constraint.hamlet.us.l5.patternmodel"]),
    "js_zipf.us.t1": ("phrases15k", True, ["-l", "4", "-t", "1", "-s", "-j", "constraint.zipf20k.us.l4t3.patternmodel"]),
    "js_zipf.us.t1y4": ("phrases15k", True, ["-l", "4", "-t", "1", "-s", "-y", "4", "-j", "constraint.zipf20k.us.l4t3.patternmodel"]),
    "js_zipf.is.t1": ("phrases15k", False, ["-l", "4", "-t", "1", "-s", "-j", "constraint.zipf20k.us.l4t3.patternmodel"]),
    "js_zipf.is.t1y4": ("phrases15k", False, ["-l", "4", "-t", "1", "-s", "-y", "4", "-j", "constraint.zipf20k.us.l4t3.patternmodel"]),
    "js_zipf.is.t1y4T1": ("phrases15k", False, ["-l", "4", "-t", "1", "-s", "-y", "4", "-T", "1", "-j", "constraint.zipf20k.us.l4t3.patternmodel"]),
}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CONSTRAINED))
def test_cli_constrained_training_matches_the_references(tmp_path, tag):
    """colibri-patternmodeller -j <model> / -I -i <model>: a membership-filtered single pass on the device — any threshold incl. 1,
    any minimum length, unindexed and indexed; goldens by the real reference (ref_driver train ... -j / -I), incl. its totals (tokens of the
    constraint model + the corpus', types of the constraint model; in place: the number of loaded patterns)."""
    import oracle
    corpus, unindexed, flags = CONSTRAINED[tag]
    flags = [os.path.join(GOLDEN, f) if f.endswith(".patternmodel") else f for f in flags]
    model = str(tmp_path / "m.colibri.patternmodel")
    out = subprocess.run([CLI, "-f", os.path.join(GOLDEN, corpus + ".colibri.dat"), "-o", model] + (["-u"] if unindexed else []) + flags, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"constrained.{tag}.txt")).read(), indexed=not unindexed)
    mtype, tokens, types, counts, refs = parse_model(model)
    assert (mtype, tokens, types) == (10 if unindexed else 20, want.tokens, want.types)
    assert counts == want.counts
    if not unindexed:
        assert refs == want.refs


FILTERED = {  # golden tag -> (corpus, filter fixture, kind, maxlength, mintokens)
    "f_ngrams.u.t2": ("zipf20k", "ngrams", "u", 5, 2), "f_ngrams.i.t2": ("zipf20k", "ngrams", "i", 4, 2), "f_skip.u.t2": ("zipf20k", "skipgrams", "u", 5, 2),
    "f_mixed.u.t2": ("zipf20k", "mixed", "u", 6, 2), "f_mixed.i.t3": ("phrases15k", "mixed", "i", 5, 3), "f_mixed.u.t1": ("hamlet.v2", "mixed", "u", 4, 1),
    "f_ngrams.u.t1": ("zipf20k", "ngrams", "u", 3, 1), "f_skip.u.t1": ("zipf20k", "skipgrams", "u", 4, 1), "f_skip.i.t1": ("zipf20k", "skipgrams", "i", 4, 1),
}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(FILTERED))
def test_cxx_api_filtered_training_matches_the_references(tmp_path, tag):
    """model.train(file, options, NULL, &filter) through the C++ face (the argument the reference's Python binding passes, include/patternmodel.h:880): only the
    windows that contain a filter n-gram or instantiate a filter skipgram are counted, on the device. Goldens: the real reference (ref_driver train -f)."""
    import oracle
    corpus, flt, kind, l, t = FILTERED[tag]
    model = str(tmp_path / "m.colibri.patternmodel")
    out = subprocess.run([SELFTEST, "gpu", os.path.join(GOLDEN, corpus + ".colibri.dat"), model, kind, str(l), str(t), "f" + os.path.join(GOLDEN, f"filter.{flt}.patternmodel")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"filtered.{tag}.txt")).read(), indexed=kind == "i")
    mtype, tokens, types, counts, refs = parse_model(model)
    assert (mtype, tokens, types) == (20 if kind == "i" else 10, want.tokens, want.types)
    assert counts == want.counts
    if kind == "i":
        assert refs == want.refs


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [["-u"], []], ids=["unindexed", "indexed"])
def test_cli_sharded_matches_the_single_device_model_on_a_larger_corpus(tmp_path, flags):
    """--gpus 3 (three rank threads sharing device 0) against the single-device run of the same CLI on 3 M tokens: exchange buffers of millions of candidates,
    every order on the radix path, the forward index merged from the ranks' runs."""
    from colibri_amd import synth
    data = str(tmp_path / "c.colibri.dat")
    with open(data, "wb") as f:
        f.write(synth.zipf_corpus(3_000_000, 50_000, 9, phrases=True))
    one, three = str(tmp_path / "one.patternmodel"), str(tmp_path / "three.patternmodel")
    a = subprocess.run([CLI, "-f", data, "-t", "2", "-l", "5", "-o", one] + flags, capture_output=True, text=True)
    assert a.returncode == 0, a.stderr
    b = subprocess.run([CLI, "-f", data, "-t", "2", "-l", "5", "-o", three, "--gpus", "3"] + flags, capture_output=True, text=True, env=dict(os.environ, COLIBRI_DEVICES="0,0,0"))
    assert b.returncode == 0, b.stderr
    ma, mb = parse_model(one), parse_model(three)
    assert ma[:3] == mb[:3] and len(ma[3]) > 100_000
    assert ma[3] == mb[3]
    assert ma[4] == mb[4]


CONTINUED = {  # golden tag -> (corpus, loaded model, unindexed?, -l, -t)
    "E_zipf.u": ("zipf20k", "continued.zipf20k.u.t3l2.patternmodel", True, 5, 2),
    "E_hamlet.i": ("hamlet.v2", "continued.hamlet.i.t2l3.patternmodel", False, 6, 2),
    "E_cross.u": ("phrases15k", "continued.zipf20k.u.t3l2.patternmodel", True, 4, 2),
}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CONTINUED))
def test_cli_continued_training_matches_the_references(tmp_path, tag):
    """colibri-patternmodeller -i <model> -f <corpus> -e 1 -E: train(..., continued = true) — the pattern lengths the loaded model lacks are counted on the
    device, with a look-back that finds the loaded patterns; the model written is the loaded one plus the new lengths, totals untouched. Goldens: the real
    reference continuing models it wrote itself (ref_driver train ... -E), on the same corpus under another threshold and on a different corpus."""
    import oracle
    corpus, fixture, unindexed, l, t = CONTINUED[tag]
    model = str(tmp_path / "m.colibri.patternmodel")
    out = subprocess.run([CLI, "-i", os.path.join(GOLDEN, fixture), "-f", os.path.join(GOLDEN, corpus + ".colibri.dat"), "-e", "1", "-E", "-l", str(l), "-t", str(t), "-o", model] +
                         (["-u"] if unindexed else []), capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"continued.{tag}.txt")).read(), indexed=not unindexed)
    mtype, tokens, types, counts, refs = parse_model(model)
    assert (mtype, tokens, types) == (10 if unindexed else 20, want.tokens, want.types)
    assert counts == want.counts
    if not unindexed:
        assert refs == want.refs


@pytest.mark.gpu
@pytest.mark.parametrize("corpus", ["hamlet.v2", "zipf20k"])
@pytest.mark.parametrize("mode,flags", [("u", ["-u"]), ("i", []), ("is", ["-s"]), ("us", ["-u", "-s"])])
def test_cli_minimum_length(tmp_path, corpus, mode, flags):
    """-m 3: the model without its patterns of fewer than three tokens, same totals (goldens by the real reference, all four kinds of model)"""
    import oracle
    model = str(tmp_path / "m.colibri.patternmodel")
    out = subprocess.run([CLI, "-f", os.path.join(GOLDEN, corpus + ".colibri.dat"), "-t", "2", "-l", "5", "-m", "3", "-o", model] + flags, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    indexed = "-u" not in flags
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"minlength.{corpus}.{mode}.m3.txt")).read(), indexed=indexed)
    mtype, tokens, types, counts, refs = parse_model(model)
    assert (mtype, tokens, types) == (20 if indexed else 10, want.tokens, want.types)
    assert counts == want.counts
    if indexed:
        assert refs == want.refs


@pytest.mark.gpu
@pytest.mark.parametrize("corpus", ["hamlet.v2", "zipf20k"])
@pytest.mark.parametrize("mode,flags", [("u", ["-u"]), ("i", []), ("us", ["-u", "-s"]), ("is", ["-s"])])
def test_cli_word_threshold(tmp_path, corpus, mode, flags):
    """-W 4 (MINTOKENS_UNIGRAMS): unigrams stay at the pattern threshold, longer patterns need every word to occur four times (goldens by the reference)"""
    import oracle
    model = str(tmp_path / "m.colibri.patternmodel")
    out = subprocess.run([CLI, "-f", os.path.join(GOLDEN, corpus + ".colibri.dat"), "-t", "2", "-l", "5", "-W", "4", "-o", model] + flags, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    indexed = "-u" not in flags
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"wordthreshold.{corpus}.{mode}.W4.txt")).read(), indexed=indexed)
    mtype, tokens, types, counts, refs = parse_model(model)
    assert (mtype, tokens, types) == (20 if indexed else 10, want.tokens, want.types)
    assert counts == want.counts
    if indexed:
        assert refs == want.refs


@pytest.mark.gpu
@pytest.mark.parametrize("corpus", ["hamlet.v2", "zipf20k"])
@pytest.mark.parametrize("kind", ["u", "i"])
@pytest.mark.parametrize("tag", ["p4", "S3"])
def test_cxx_api_subsumption_prunes(tmp_path, corpus, kind, tag):
    """PRUNENONSUBSUMED = 4 (colibri-patternmodeller -p 4) / PRUNESUBSUMED = 3 through the C++ face: goldens by the real reference"""
    import oracle
    model = str(tmp_path / "m.colibri.patternmodel")
    out = subprocess.run([SELFTEST, "gpu", os.path.join(GOLDEN, corpus + ".colibri.dat"), model, "u" if kind == "u" else "i", "5", "2", tag], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    indexed = kind == "i"
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"subsumption.{corpus}.{kind}.{tag}.txt")).read(), indexed=indexed)
    mtype, tokens, types, counts, refs = parse_model(model)
    assert (tokens, types) == (want.tokens, want.types)
    assert counts == want.counts
    if indexed:
        assert refs == want.refs
    if tag == "p4":  # the CLI flag
        out = subprocess.run([CLI, "-f", os.path.join(GOLDEN, corpus + ".colibri.dat"), "-t", "2", "-l", "5", "-p", "4", "-o", model] + (["-u"] if kind == "u" else []), capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert parse_model(model)[3] == want.counts


# ---- relation queries on indexed skipgram models -------------------------------------------------------------------------
RELATIONS = [("hamlet.v2", 1, "isT1"), ("hamlet.v2", 2, "is"), ("phrases15k", 2, "is"), ("zipf20k", 2, "is")]


def _cls_for(corpus):
    return os.path.join(GOLDEN, "hamlet.colibri.cls" if corpus.startswith("hamlet") else "synthetic.colibri.cls")


@pytest.mark.gpu
@pytest.mark.parametrize("corpus,T,tag", RELATIONS)
@pytest.mark.parametrize("flt", ["skipcontent", "instances_api", "templates_api"])
def test_cxx_api_relation_queries(tmp_path, corpus, T, tag, flt):
    """getskipcontent / getinstances / gettemplates + outputrelations of the C++ face on a model trained on the device, against the real
    reference walking its own model the same way (ref_driver relations); the reference iterates an unordered_map: lines are compared sorted"""
    out = str(tmp_path / "rel.txt")
    p = subprocess.run([SELFTEST, "relations", os.path.join(GOLDEN, corpus + ".colibri.dat"), _cls_for(corpus), "5", "2", str(T), flt, out], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip() == "OK", p.stdout + p.stderr
    want = open(os.path.join(GOLDEN, f"relations.{corpus}.{tag}.{flt}.txt")).read().splitlines()
    assert sorted(open(out).read().splitlines()) == want


@pytest.mark.gpu
def test_cli_skipcontent_and_the_silent_relation_flags(tmp_path):
    """--skipcontent after building: every pattern, then its skip content rows; --instances / --templates print the patterns only, as the
    reference does from its CLI (its call resolves to the base class's empty getters, include/patternmodel.h:2635-2640)"""
    data = os.path.join(GOLDEN, "hamlet.v2.colibri.dat")
    base = [CLI, "-f", data, "-c", _cls_for("hamlet.v2"), "-s", "-T", "1", "-t", "2", "-l", "5"]
    out = subprocess.run(base + ["--skipcontent"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    want = open(os.path.join(GOLDEN, "relations.hamlet.v2.isT1.skipcontent.txt")).read().splitlines()
    assert sorted(out.stdout.splitlines()) == want
    for flag in ("--instances", "--templates"):
        out = subprocess.run(base + [flag], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        lines = out.stdout.splitlines()
        assert len(lines) == 202 and lines.count("#\tPATTERN1\tRELATION\tPATTERN2\tREL.COUNT\tREL.FREQUENCY\tCOUNT2") == 1
        assert not [ln for ln in lines if ln.startswith("\t")]


def test_cli_relations_need_a_class_file():
    out = subprocess.run([CLI, "-f", os.path.join(GOLDEN, "hamlet.v2.colibri.dat"), "-s", "--skipcontent"], capture_output=True, text=True)
    assert out.returncode == 2 and "needs a class file" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("corpus,l", [("shortlines", 2), ("shortlines", 100), ("edge", 3), ("edge", 100), ("zipf20k", 100), ("hamlet.v2", 100)])
def test_cli_pattern_list(tmp_path, corpus, l):
    """-L: the data file is a list of one pattern per line (implies -t 1, unindexed); goldens by the real reference, which also loads the model back"""
    import oracle
    model = str(tmp_path / "m.colibri.patternmodel")
    out = subprocess.run([CLI, "-f", os.path.join(GOLDEN, corpus + ".colibri.dat"), "-L", "-l", str(l), "-o", model], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "Counting patterns from list, one per line" in out.stderr
    want = oracle.parse_dump(open(os.path.join(GOLDEN, f"patternlist.{corpus}.L{l}.txt")).read())
    mtype, tokens, types, counts, _ = parse_model(model)
    assert (mtype, tokens, types, counts) == (10, want.tokens, want.types, want.counts)
    if oracle.have_ref():
        dump = str(tmp_path / "d.txt")
        subprocess.check_call([oracle.REF_DRIVER, "load", model, "u", dump])
        got = oracle.parse_dump(open(dump).read())
        assert (got.tokens, got.counts) == (want.tokens, want.counts)
