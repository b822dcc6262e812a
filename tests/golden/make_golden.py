"""Generates the golden fixtures in this directory by running the REAL reference (oracle/_ref/ref_driver, built
from /root/reference by oracle/Makefile) — run in the build container only:  python tests/golden/make_golden.py

Inputs (data, committed): hamlet.v1.colibri.dat (the reference's own fixture exp/hamlet.v1.colibri.dat),
edge.colibri.dat / zipf20k.colibri.dat / phrases15k.colibri.dat (written by this script from seeded generators).
Outputs: <corpus>.<mode>.l<maxlength>.txt — canonical dumps (sorted hex key, count[, refs]) of the reference's
models; spooky_kat.json — SpookyHash::Hash64 values; masks.json — compute_skip_configurations outputs.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402
from colibri_amd import synth  # noqa: E402

DRIVER = oracle.REF_DRIVER


def corpora():
    rng = np.random.default_rng(11)
    edge = (b"\x06\x07\x08\x00" b"\x00" b"\x06\x07\x08\x09\x0a\x00" b"\x06\x00" b"\x00\x00" b"\x06\x07\x00"
            + synth.encode_v2(np.array([200, 300, 20000, 200, 300, 20000, 3000000, 0, 200, 300, 20000, 3000000, 0], dtype=np.uint32)).tobytes()
            + synth.random_corpus(rng, nsent=60, maxlen=9, vocab=7, big_classes=False))
    return {
        "edge": synth.HEADER + edge,
        "zipf20k": synth.zipf_corpus(20000, 500, 5),
        "phrases15k": synth.zipf_corpus(15000, 800, 9, phrases=True),
    }


def tokens_of(key):
    out, start = [], 0
    for j, b in enumerate(key):
        if b < 128:
            out.append(key[start:j + 1])
            start = j + 1
    return out


def reference_is_dump_is_stable(path, minskiptypes, maxlength):
    """IndexedPatternModel::trainskipgrams inserts into the unordered_map it is iterating (reference
    include/patternmodel.h:2986-2991); when that rehashes mid-loop the reference revisits / skips n-grams. A dump is
    kept as a golden only if it is self-consistent: every skipgram's index equals the union of the indices of the
    n-grams (present in the same dump) that it abstracts, no duplicates, and it has >= MINSKIPTYPES sources."""
    m = oracle.parse_dump(open(path).read(), indexed=True)
    ngrams = {k: r for k, r in m.refs.items() if b"\x03" not in tokens_of(k)}
    skip = {k: r for k, r in m.refs.items() if b"\x03" in tokens_of(k)}
    exp, nsrc = {}, {}
    for k, refs in ngrams.items():
        toks = tokens_of(k)
        n = len(toks)
        if n < 3 or n > maxlength:
            continue
        for mask in oracle.skip_configurations(n, 3):
            sk = b"".join(b"\x03" if (mask >> i) & 1 else t for i, t in enumerate(toks))
            exp.setdefault(sk, []).extend(refs)
            nsrc[sk] = nsrc.get(sk, 0) + 1
    exp = {k: sorted(v) for k, v in exp.items() if nsrc[k] >= minskiptypes}
    return exp == skip


def longspan():
    """patterns beyond 13 tokens (the device path's former limit; the reference's gap masks reach 31 tokens, include/pattern.h:368): a 14-token sentence
    four times — twice verbatim, twice with one word changed, so that skipgrams with two distinct fillers exist — among short ones"""
    span = list(range(6, 20))
    near = list(span)
    near[6] = 40
    syms = []
    for sent in (span, [30, 31, 32], near, [33, 30, 31], span, [6, 7, 8, 50], near, [30, 31, 32, 33]):
        syms += sent + [0]
    with open(os.path.join(HERE, "longspan.colibri.dat"), "wb") as f:
        f.write(synth.HEADER + synth.encode_v2(np.array(syms, dtype=np.uint32)).tobytes())
    out = []
    for mode, l, extra in [("us", 14, []), ("is", 14, ["-T", "1"]), ("is", 14, []), ("u", 14, [])]:
        tag = mode + "".join(extra).replace("-", "")
        path = os.path.join(HERE, f"longspan.{tag}.l{l}.txt")
        subprocess.check_call([DRIVER, "train", os.path.join(HERE, "longspan.colibri.dat"), mode, str(l), "2", "-q", "-d", path] + extra, stdout=subprocess.DEVNULL)
        if mode == "is" and not reference_is_dump_is_stable(path, 1 if extra else 2, l):
            out.append(os.path.basename(path))
            os.remove(path)
    return out


def main():
    if sys.argv[1:] == ["longspan"]:
        print("unstable:", longspan())
        return
    unstable = []
    for name, data in corpora().items():
        with open(os.path.join(HERE, f"{name}.colibri.dat"), "wb") as f:
            f.write(data)
    unstable += longspan()
    jobs = []
    for name in ["hamlet.v1", "edge", "zipf20k", "phrases15k"]:
        path = os.path.join(HERE, f"{name}.colibri.dat")
        modes = [("u", 3, []), ("u", 5, []), ("u", 100, [])] if name == "hamlet.v1" else [("u", 5, [])]
        if name != "hamlet.v1":  # v1 input cannot be preloaded into an IndexedCorpus (reference pattern.cpp:1936-1940)
            modes += [("us", 5, []), ("us", 5, ["-y", "3"]), ("i", 5, []), ("is", 5, []), ("is", 5, ["-T", "1"])]
        for mode, l, extra in modes:
            tag = mode + "".join(extra).replace("-", "")
            out = os.path.join(HERE, f"{name}.{tag}.l{l}.txt")
            jobs.append(out)
            subprocess.check_call([DRIVER, "train", path, mode, str(l), "2", "-q", "-d", out] + extra, stdout=subprocess.DEVNULL)
            if mode == "is" and not reference_is_dump_is_stable(out, 1 if extra else 2, l):
                unstable.append(os.path.basename(out))
                os.remove(out)
    # hamlet as v2 (converted by the oracle's v1->v2, which is itself checked against the reference reading v1 directly)
    v1 = open(os.path.join(HERE, "hamlet.v1.colibri.dat"), "rb").read()
    with open(os.path.join(HERE, "hamlet.v2.colibri.dat"), "wb") as f:
        f.write(synth.HEADER + oracle.v1_to_v2(v1))
    for mode, l, extra in [("u", 5, []), ("us", 5, []), ("i", 5, []), ("is", 5, []), ("is", 5, ["-T", "1"]), ("us", 100, []), ("is", 100, [])]:
        tag = mode + "".join(extra).replace("-", "")
        out = os.path.join(HERE, f"hamlet.v2.{tag}.l{l}.txt")
        subprocess.check_call([DRIVER, "train", os.path.join(HERE, "hamlet.v2.colibri.dat"), mode, str(l), "2", "-q", "-d", out] + extra, stdout=subprocess.DEVNULL)
        if mode == "is" and not reference_is_dump_is_stable(out, 1 if extra else 2, l):
            unstable.append(os.path.basename(out))
            os.remove(out)
    # MINTOKENS = 1 (the reference's single pass over all lengths)
    for name, l in [("hamlet.v2", 5), ("edge", 5), ("zipf20k", 3), ("zipf20k", 5)]:  # (zipf20k at l = 5: tests/test_kshard.py's key-sharded indexed threshold-1 cases)
        for mode in ("u", "i"):
            out = os.path.join(HERE, f"{name}.{mode}t1.l{l}.txt")
            subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{name}.colibri.dat"), mode, str(l), "1", "-q", "-d", out], stdout=subprocess.DEVNULL)
    # ... and with skipgrams: every window of three or more tokens counts all its masked forms (indexed dumps only where the reference's loop stayed stable)
    for name, l, mode, extra, tag in [("hamlet.v2", 4, "us", [], "ust1"), ("hamlet.v2", 4, "us", ["-y", "3"], "usy3t1"), ("edge", 4, "us", [], "ust1"),
                                      ("edge", 4, "us", ["-y", "3"], "usy3t1"), ("hamlet.v2", 3, "is", [], "ist1"), ("hamlet.v2", 3, "is", ["-T", "1"], "isT1t1")]:
        out = os.path.join(HERE, f"{name}.{tag}.l{l}.txt")
        subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{name}.colibri.dat"), mode, str(l), "1", "-q", "-d", out] + extra, stdout=subprocess.DEVNULL)
        if mode == "is" and not reference_is_dump_is_stable(out, 1 if extra else 2, l):
            unstable.append(os.path.basename(out))
            os.remove(out)
    # one pattern per line (patternmodeller -L, implies -t 1): a list of short lines with many repeats, and the ordinary corpora
    rng = np.random.default_rng(23)
    with open(os.path.join(HERE, "shortlines.colibri.dat"), "wb") as f:
        f.write(synth.HEADER + synth.random_corpus(rng, nsent=400, maxlen=4, vocab=4, big_classes=True, empty_rate=0.2))
    for name, l in [("shortlines", 2), ("shortlines", 100), ("edge", 3), ("edge", 100), ("zipf20k", 100), ("hamlet.v2", 100)]:
        out = os.path.join(HERE, f"patternlist.{name}.L{l}.txt")
        subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{name}.colibri.dat"), "u", str(l), "1", "-q", "-L", "-d", out], stdout=subprocess.DEVNULL)
    # MAXBACKOFFLENGTH (-b) below the longest pattern: same model, other candidate counts per order
    for name in ["hamlet.v2", "edge"]:
        for mode in ("u", "i"):
            for b in (1, 2):
                out = os.path.join(HERE, f"backoff.{name}.{mode}.b{b}.txt")
                subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{name}.colibri.dat"), mode, "8", "2", "-q", "-b", str(b), "-d", out], stdout=subprocess.DEVNULL)
    # MINLENGTH = 3: the shorter orders are counted for the look-back and pruned away afterwards
    for name in ["hamlet.v2", "zipf20k"]:
        for mode in ("u", "i", "is", "us"):
            out = os.path.join(HERE, f"minlength.{name}.{mode}.m3.txt")
            subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{name}.colibri.dat"), mode, "5", "2", "-q", "-m", "3", "-d", out], stdout=subprocess.DEVNULL)
    # MINTOKENS_UNIGRAMS = 4 (-W): longer patterns need every word to occur at least four times
    for name in ["hamlet.v2", "zipf20k"]:
        for mode in ("u", "i", "us", "is"):
            out = os.path.join(HERE, f"wordthreshold.{name}.{mode}.W4.txt")
            subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{name}.colibri.dat"), mode, "5", "2", "-q", "-W", "4", "-d", out], stdout=subprocess.DEVNULL)
    # PRUNENONSUBSUMED = 4 (-p) / PRUNESUBSUMED = 3: post-hoc passes over the finished model
    for name in ["hamlet.v2", "zipf20k"]:
        for mode in ("u", "i"):
            for flag, tag in (("-p", "p4"), ("-S", "S3")):
                out = os.path.join(HERE, f"subsumption.{name}.{mode}.{tag}.txt")
                subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{name}.colibri.dat"), mode, "5", "2", "-q", flag, tag[1:], "-d", out], stdout=subprocess.DEVNULL)
    # two-stage build (patternmodeller -2): what the reference's constrained in-place second stage leaves (with and without -s)
    for name in ["hamlet.v2", "zipf20k"]:
        for mode in ("i2", "is2"):
            out = os.path.join(HERE, f"twostage.{name}.{mode}.txt")
            subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{name}.colibri.dat"), mode, "5", "2", "-q", "-d", out], stdout=subprocess.DEVNULL)
    # constrained training (patternmodeller -j / -I): constraint models written by the reference itself, kept as fixtures
    cj = os.path.join(HERE, "constraint.zipf20k.u.l5.patternmodel")
    subprocess.check_call([DRIVER, "train", os.path.join(HERE, "zipf20k.colibri.dat"), "u", "5", "2", "-q", "-o", cj], stdout=subprocess.DEVNULL)
    ch = os.path.join(HERE, "constraint.hamlet.i.l5.patternmodel")
    subprocess.check_call([DRIVER, "train", os.path.join(HERE, "hamlet.v2.colibri.dat"), "i", "5", "2", "-q", "-o", ch], stdout=subprocess.DEVNULL)
    for tag, corpus, mode, args in [("j_zipf.u.t1", "phrases15k", "u", ["4", "1", "-j", cj]), ("j_zipf.u.t2", "phrases15k", "u", ["4", "2", "-j", cj]),
                                    ("j_zipf.u.t3", "phrases15k", "u", ["5", "3", "-j", cj]), ("j_zipf.u.t1m2", "phrases15k", "u", ["4", "1", "-m", "2", "-j", cj]),
                                    ("j_zipf.i.t1", "phrases15k", "i", ["4", "1", "-j", cj]), ("j_zipf.i.t2", "phrases15k", "i", ["5", "2", "-j", cj]),
                                    ("j_hamlet.u.t1", "edge", "u", ["5", "1", "-j", ch]), ("j_self.u.t2", "zipf20k", "u", ["5", "2", "-j", cj]),
                                    ("I_zipf.u.t2", "phrases15k", "u", ["5", "2", "-I", cj]), ("I_zipf.i.t1", "phrases15k", "i", ["5", "1", "-I", cj]),
                                    ("I_self.i.t2", "hamlet.v2", "i", ["5", "2", "-I", ch])]:
        out = os.path.join(HERE, f"constrained.{tag}.txt")
        subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{corpus}.colibri.dat"), mode] + args + ["-q", "-d", out], stdout=subprocess.DEVNULL)
    # skipgrams in a constrained run (-j with -s): constraint models WITH skipgrams, written by the reference; only a run at -t 1 computes any (patternmodel.h:1163)
    chs = os.path.join(HERE, "constraint.hamlet.us.l5.patternmodel")
    subprocess.check_call([DRIVER, "train", os.path.join(HERE, "hamlet.v2.colibri.dat"), "us", "5", "2", "-q", "-o", chs], stdout=subprocess.DEVNULL)
    czs = os.path.join(HERE, "constraint.zipf20k.us.l4t3.patternmodel")
    subprocess.check_call([DRIVER, "train", os.path.join(HERE, "zipf20k.colibri.dat"), "us", "4", "3", "-q", "-o", czs], stdout=subprocess.DEVNULL)
    for tag, corpus, mode, args in [("js_hamlet.us.t1", "hamlet.v2", "us", ["5", "1", "-j", chs]), ("js_hamlet.us.t1T1", "hamlet.v2", "us", ["5", "1", "-T", "1", "-j", chs]),
                                    ("js_hamlet.us.t1y3", "hamlet.v2", "us", ["5", "1", "-y", "3", "-j", chs]), ("js_hamlet.is.t1", "hamlet.v2", "is", ["5", "1", "-j", chs]),
                                    ("js_hamlet.is.t1T1", "hamlet.v2", "is", ["5", "1", "-T", "1", "-j", chs]), ("js_edge.us.t1", "edge", "us", ["5", "1", "-j", chs]),
                                    ("js_hamlet.us.t2", "hamlet.v2", "us", ["5", "2", "-j", chs]), ("js_hamlet.is.t2", "hamlet.v2", "is", ["4", "2", "-j", chs]),
                                    ("js_zipf.us.t1", "phrases15k", "us", ["4", "1", "-j", czs]), ("js_zipf.us.t1y4", "phrases15k", "us", ["4", "1", "-y", "4", "-j", czs]),
                                    ("js_zipf.is.t1", "phrases15k", "is", ["4", "1", "-j", czs]), ("js_zipf.is.t1y4", "phrases15k", "is", ["4", "1", "-y", "4", "-j", czs]),
                                    ("js_zipf.is.t1y4T1", "phrases15k", "is", ["4", "1", "-y", "4", "-T", "1", "-j", czs])]:
        out = os.path.join(HERE, f"constrained.{tag}.txt")
        subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{corpus}.colibri.dat"), mode] + args + ["-q", "-d", out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # filtered training (train(..., filter), the Python binding's argument): filter sets as small hand-made model files (type 10), goldens by ref_driver train -f
    def write_model(path, keys):
        import struct
        with open(path, "wb") as f:
            f.write(bytes([0, 10, 2]) + struct.pack("<QQQ", 0, 0, len(keys)))
            for k in keys:
                f.write(k + b"\x00" + struct.pack("<I", 1))
    write_model(os.path.join(HERE, "filter.ngrams.patternmodel"), [bytes([9]), bytes([6, 12]), bytes([7, 8, 10])])
    write_model(os.path.join(HERE, "filter.skipgrams.patternmodel"), [bytes([6, 3, 8]), bytes([7, 3, 3, 10]), bytes([11, 3, 6])])
    write_model(os.path.join(HERE, "filter.mixed.patternmodel"), [bytes([15]), bytes([6, 7]), bytes([6, 3, 9]), bytes([8, 3, 6, 3, 7]), bytes([10, 4, 11])])
    for tag, corpus, mode, l, t, flt in [("f_ngrams.u.t2", "zipf20k", "u", 5, 2, "ngrams"), ("f_ngrams.i.t2", "zipf20k", "i", 4, 2, "ngrams"), ("f_skip.u.t2", "zipf20k", "u", 5, 2, "skipgrams"),
                                         ("f_mixed.u.t2", "zipf20k", "u", 6, 2, "mixed"), ("f_mixed.i.t3", "phrases15k", "i", 5, 3, "mixed"), ("f_mixed.u.t1", "hamlet.v2", "u", 4, 1, "mixed"),
                                         ("f_ngrams.u.t1", "zipf20k", "u", 3, 1, "ngrams"), ("f_skip.u.t1", "zipf20k", "u", 4, 1, "skipgrams"), ("f_skip.i.t1", "zipf20k", "i", 4, 1, "skipgrams")]:
        out = os.path.join(HERE, f"filtered.{tag}.txt")
        subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{corpus}.colibri.dat"), mode, str(l), str(t), "-q", "-f", os.path.join(HERE, f"filter.{flt}.patternmodel"), "-d", out],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # continued training (patternmodeller -i <model> -f <corpus> -e 1 -E = train(..., continued = true)): models written by the reference, kept as fixtures, then
    # continued by the reference to longer patterns — on the same corpus under another threshold, and on a different corpus
    ez = os.path.join(HERE, "continued.zipf20k.u.t3l2.patternmodel")
    subprocess.check_call([DRIVER, "train", os.path.join(HERE, "zipf20k.colibri.dat"), "u", "2", "3", "-q", "-o", ez], stdout=subprocess.DEVNULL)
    eh = os.path.join(HERE, "continued.hamlet.i.t2l3.patternmodel")
    subprocess.check_call([DRIVER, "train", os.path.join(HERE, "hamlet.v2.colibri.dat"), "i", "3", "2", "-q", "-o", eh], stdout=subprocess.DEVNULL)
    for tag, corpus, mode, args in [("E_zipf.u", "zipf20k", "u", ["5", "2", "-E", ez]), ("E_hamlet.i", "hamlet.v2", "i", ["6", "2", "-E", eh]),
                                    ("E_cross.u", "phrases15k", "u", ["4", "2", "-E", ez])]:
        out = os.path.join(HERE, f"continued.{tag}.txt")
        subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{corpus}.colibri.dat"), mode] + args + ["-q", "-d", out], stdout=subprocess.DEVNULL)
    # flexgrams from skipgrams (ref_driver -F = computeflexgrams_fromskipgrams after training, src/patternmodeller.cpp:790-794). The loop
    # inserts into the map it iterates (patternmodel.h:3727-3738): a dump is kept only where it equals the hazard-free restatement
    # applied to the reference's own model before the call
    for name, tag, extra in [("hamlet.v2", "is", []), ("hamlet.v2", "isT1", ["-T", "1"]), ("phrases15k", "is", []), ("phrases15k", "isT1", ["-T", "1"]),
                             ("zipf20k", "is", []), ("zipf20k", "isT1", ["-T", "1"])]:
        before = os.path.join(HERE, f"{name}.{tag}.l5.txt")
        if not os.path.exists(before):
            continue
        out = os.path.join(HERE, f"flex.{name}.{tag}.txt")
        subprocess.check_call([DRIVER, "train", os.path.join(HERE, f"{name}.colibri.dat"), "is", "5", "2", "-q", "-F", "-d", out] + extra, stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        mine, _ = oracle.flexgrams_from_skipgrams(oracle.parse_dump(open(before).read(), indexed=True))
        got = oracle.parse_dump(open(out).read(), indexed=True)
        if mine.refs != got.refs:
            unstable.append(os.path.basename(out))
            os.remove(out)
    # relation queries on indexed skipgram models: skipcontent as colibri-patternmodeller --skipcontent walks the model, instances / templates
    # through the C++ API proper (getinstances / gettemplates(const Pattern&)); kept only where the model itself is a stable golden
    for name, T in [("hamlet.v2", 1), ("hamlet.v2", 2), ("phrases15k", 2), ("zipf20k", 2)]:
        tag = "isT1" if T == 1 else "is"
        if not os.path.exists(os.path.join(HERE, f"{name}.{tag}.l5.txt")):
            continue
        for flt in ("skipcontent", "instances_api", "templates_api"):
            out = os.path.join(HERE, f"relations.{name}.{tag}.{flt}.txt")
            subprocess.check_call([DRIVER, "relations", os.path.join(HERE, f"{name}.colibri.dat"), os.path.join(HERE, "hamlet.colibri.cls" if name.startswith("hamlet") else "synthetic.colibri.cls"),
                                   "5", "2", str(T), flt, out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            lines = sorted(open(out).read().splitlines())  # canonical form: the reference walks its unordered_map
            with open(out, "w") as f:
                f.write("\n".join(lines) + "\n")
    with open(os.path.join(HERE, "unstable_reference_outputs.json"), "w") as f:
        json.dump({"note": "indexed+skipgram / flexgram dumps of the reference that were NOT kept because the reference's insert-while-iterating "
                           "hazard (patternmodel.h:2986-2991, :3727-3738) corrupted them (self-consistency checks in make_golden.py)", "dropped": unstable}, f, indent=1)
    # SpookyHash known answers from the reference's own implementation
    rng = np.random.default_rng(5)
    keys = [bytes([6]), bytes([6, 7, 8]), bytes.fromhex("8601904e07")] + [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in range(1, 192, 3)]
    out = subprocess.run([DRIVER, "hash"] + [k.hex() for k in keys], check=True, capture_output=True, text=True).stdout
    kat = {ln.split("\t")[0]: ln.split("\t")[1] for ln in out.strip().splitlines()}
    with open(os.path.join(HERE, "spooky_kat.json"), "w") as f:
        json.dump(kat, f, indent=0)
    masks = {}
    for n in range(3, 10):
        for ms in (1, 2, 3):
            o = subprocess.run([DRIVER, "masks", str(n), str(ms)], check=True, capture_output=True, text=True).stdout.split()
            masks[f"{n},{ms}"] = [int(x) for x in o]
    with open(os.path.join(HERE, "masks.json"), "w") as f:
        json.dump(masks, f)
    print("golden fixtures written:", len(os.listdir(HERE)), "files")


if __name__ == "__main__":
    main()
