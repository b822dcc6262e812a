"""oracle/oracle.py — TEST INFRASTRUCTURE: ctypes loader for oracle/liboracle.so and a runner for
oracle/_ref/ref_driver (the real reference, built from its own sources by oracle/Makefile).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only
as the checker. The product (colibri-core_amd/) never does.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_DRIVER = os.path.join(HERE, "_ref", "ref_driver")


class Options(C.Structure):
    _fields_ = [
        ("mintokens", C.c_int32),
        ("maxlength", C.c_int32),
        ("mintokens_skipgrams", C.c_int32),
        ("minskiptypes", C.c_int32),
        ("maxskips", C.c_int32),
        ("doskipgrams", C.c_int32),
        ("doskipgrams_exhaustive", C.c_int32),
        ("indexed", C.c_int32),
        ("mintokens_unigrams", C.c_int32),
        ("maxbackofflength", C.c_int32),
    ]


def build():
    """(Re)build liboracle.so with gcc; cheap (one C file)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.co_spooky64.restype = C.c_uint64
        L.co_spooky64.argtypes = [C.c_void_p, C.c_uint64]
        L.co_skip_configurations.restype = C.c_int
        L.co_skip_configurations.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.co_v1_to_v2.restype = C.c_int
        L.co_v1_to_v2.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
        L.co_train.restype = C.c_void_p
        L.co_train.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Options), C.c_uint32]
        L.co_free.argtypes = [C.c_void_p]
        for name in ("co_npatterns", "co_totaltokens", "co_totaltypes", "co_keybytes", "co_nrefs", "co_windows"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [C.c_void_p]
        L.co_maxn.restype = C.c_int
        L.co_maxn.argtypes = [C.c_void_p]
        L.co_order_stat.restype = C.c_int64
        L.co_order_stat.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.co_export.argtypes = [C.c_void_p] * 7
        _lib = L
    return _lib


def spooky64(data: bytes) -> int:
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    return int(lib().co_spooky64(buf, len(data)))


def skip_configurations(n: int, maxskips: int = 3):
    k = lib().co_skip_configurations(n, maxskips, None, 0)
    out = np.zeros(max(1, k), dtype=np.uint32)
    lib().co_skip_configurations(n, maxskips, out.ctypes.data, out.size)
    return [int(x) for x in out[:k]]


def v1_to_v2(data: bytes) -> bytes:
    src = np.frombuffer(data, dtype=np.uint8)
    nout = C.c_uint64(0)
    assert lib().co_v1_to_v2(src.ctypes.data, src.size, None, C.byref(nout)) == 0
    out = np.zeros(max(1, nout.value), dtype=np.uint8)
    assert lib().co_v1_to_v2(src.ctypes.data, src.size, out.ctypes.data, C.byref(nout)) == 0
    return out[: nout.value].tobytes()


class Model:
    """Result of a training run in canonical form: dict key-bytes -> count (and -> [(sentence, token)])."""

    def __init__(self, tokens, types, counts, refs=None, stats=None, maxn=0, windows=0):
        self.tokens, self.types, self.counts, self.refs = tokens, types, counts, refs
        self.stats, self.maxn, self.windows = stats or {}, maxn, windows

    def __len__(self):
        return len(self.counts)


def train(payload: bytes, mintokens=2, maxlength=100, *, mintokens_unigrams=0, maxbackofflength=0, indexed=False, doskipgrams=False, doskipgrams_exhaustive=False,
          minskiptypes=2, maxskips=3, mintokens_skipgrams=-1, firstsentence=1) -> Model:
    """PatternModel::train on a v2 payload (header stripped) with the C restatement."""
    L = lib()
    src = np.frombuffer(payload, dtype=np.uint8) if len(payload) else np.zeros(0, dtype=np.uint8)
    opt = Options(mintokens, maxlength, mintokens_skipgrams, minskiptypes, maxskips, int(doskipgrams), int(doskipgrams_exhaustive),
                  int(indexed), mintokens_unigrams, maxbackofflength)
    ptr = src.ctypes.data if src.size else None
    m = L.co_train(ptr, src.size, C.byref(opt), firstsentence)
    if not m:
        raise ValueError("options outside the restated subset")
    try:
        npat = L.co_npatterns(m)
        key_off = np.zeros(npat + 1, dtype=np.uint64)
        key_bytes = np.zeros(max(1, L.co_keybytes(m)), dtype=np.uint8)
        counts = np.zeros(max(1, npat), dtype=np.uint32)
        nrefs = L.co_nrefs(m)
        ref_off = np.zeros(npat + 1, dtype=np.uint64)
        ref_s = np.zeros(max(1, nrefs), dtype=np.uint32)
        ref_t = np.zeros(max(1, nrefs), dtype=np.uint16)
        L.co_export(m, key_off.ctypes.data, key_bytes.ctypes.data, counts.ctypes.data, ref_off.ctypes.data if indexed else None,
                    ref_s.ctypes.data, ref_t.ctypes.data)
        kb = key_bytes.tobytes()
        cd, rd = {}, ({} if indexed else None)
        for j in range(npat):
            k = kb[int(key_off[j]): int(key_off[j + 1])]
            cd[k] = int(counts[j])
            if indexed:
                a, b = int(ref_off[j]), int(ref_off[j + 1])
                rd[k] = list(zip(ref_s[a:b].tolist(), ref_t[a:b].tolist()))
        stats = {n: tuple(int(L.co_order_stat(m, n, w)) for w in range(3)) for n in range(1, min(maxlength, 127) + 1)}
        return Model(int(L.co_totaltokens(m)), int(L.co_totaltypes(m)), cd, rd, stats, int(L.co_maxn(m)), int(L.co_windows(m)))
    finally:
        L.co_free(m)


def train_timed(payload: bytes, mintokens=2, maxlength=5):
    """Time co_train only (no export); returns (seconds, windows, npatterns)."""
    import time
    L = lib()
    src = np.frombuffer(payload, dtype=np.uint8)
    opt = Options(mintokens, maxlength, -1, 2, 3, 0, 0, 0)
    t0 = time.perf_counter()
    m = L.co_train(src.ctypes.data, src.size, C.byref(opt), 1)
    dt = time.perf_counter() - t0
    w, n = int(L.co_windows(m)), int(L.co_npatterns(m))
    L.co_free(m)
    return dt, w, n


# ---------------------------------------------------------------------------------------------
# the real reference (only where oracle/_ref/ref_driver exists: the build container, and the GPU box
# because the prebuilt binary travels with the snapshot)
# ---------------------------------------------------------------------------------------------
def have_ref() -> bool:
    return os.path.exists(REF_DRIVER) and os.access(REF_DRIVER, os.X_OK)


def parse_dump(text: str, indexed=False) -> Model:
    lines = text.splitlines()
    tokens = int(lines[0].split()[1])
    types = int(lines[1].split()[1])
    npat = int(lines[2].split()[1])
    cd, rd = {}, ({} if indexed else None)
    for ln in lines[3:]:
        parts = ln.split("\t")
        k = bytes.fromhex(parts[0])
        cd[k] = int(parts[1])
        if indexed:
            refs = []
            if len(parts) > 2 and parts[2]:
                for r in parts[2].split(" "):
                    s, t = r.split(":")
                    refs.append((int(s), int(t)))
            rd[k] = refs
    assert len(cd) == npat, (len(cd), npat)
    return Model(tokens, types, cd, rd)


def ref_train(corpus_path: str, mode: str, maxlength: int, mintokens: int, *, minskiptypes=None, mintokens_skipgrams=None, dump_path=None,
              model_path=None):
    """Run the real reference through ref_driver. Returns (Model or None, info dict with train_s)."""
    cmd = [REF_DRIVER, "train", corpus_path, mode, str(maxlength), str(mintokens), "-q"]
    if minskiptypes is not None:
        cmd += ["-T", str(minskiptypes)]
    if mintokens_skipgrams is not None:
        cmd += ["-y", str(mintokens_skipgrams)]
    if dump_path:
        cmd += ["-d", dump_path]
    if model_path:
        cmd += ["-o", model_path]
    out = subprocess.run(cmd, check=True, capture_output=True, text=True)
    info = json.loads(out.stdout.strip().splitlines()[-1])
    model = None
    if dump_path:
        with open(dump_path) as f:
            model = parse_dump(f.read(), indexed=mode in ("i", "is"))
    return model, info


# ---- constrained training ------------------------------------------------------------------------------------------
def train_constrained(payload: bytes, constraint, mintokens=2, maxlength=100, minlength=1, indexed=False, firstsentence=1, doskipgrams=False, mintokens_skipgrams=-1,
                      minskiptypes=2, maxskips=3) -> "Model":
    """PatternModel::train(..., constrainbymodel) restated (reference include/patternmodel.h:1062-1072: every n-gram of every length
    MINLENGTH..MAXLENGTH of every sentence in ONE pass; :1088-1089: counted iff the constraint model has it, no look-back; :1209-1217:
    prune(MINTOKENS) regardless of size). `constraint` = the key bytes of the constraint model's patterns. Pure Python: small inputs.
    tokens = the corpus' tokens, types = 0 (what the C ABI reports; the totals quirks of the reference are the C++ face's).
    doskipgrams (either kind: :941-956 makes it the exhaustive one): at MINTOKENS = 1 — the only threshold at which the single pass reaches computeskipgrams
    (:1163) — every member window of three or more tokens also counts each of its masked forms the constraint model holds (:1410-1411); they are pruned by the
    BASE pruneskipgrams for either model type (:1236, :2167-2186): below MINTOKENS_SKIPGRAMS when MINSKIPTYPES > 1, not at all otherwise."""
    constraint = set(constraint)
    thr = 2 if mintokens == -1 else max(1, mintokens)
    counts, refs, skips = {}, {}, {}
    sentence, tokens = firstsentence - 1, 0
    toks, start, prevhigh = [], 0, False
    sentences = []
    for j, b in enumerate(payload):
        if b >= 128:
            continue
        tok = payload[start:j + 1]
        start = j + 1
        if tok == b"\x00":
            sentences.append(toks)
            toks = []
        else:
            toks.append(tok)
    if toks:
        sentences.append(toks)
    for toks in sentences:
        sentence += 1
        tokens += len(toks)
        for n in range(max(1, minlength), min(maxlength, len(toks)) + 1):
            for i in range(len(toks) - n + 1):
                k = b"".join(toks[i:i + n])
                if k in constraint:
                    counts[k] = counts.get(k, 0) + 1
                    if indexed:
                        refs.setdefault(k, []).append((sentence, i))
                    if doskipgrams and thr == 1 and n >= 3:
                        for mask in skip_configurations(n, maxskips):
                            sk = b"".join(b"\x03" if (mask >> j) & 1 else toks[i + j] for j in range(n))
                            if sk in constraint:
                                skips[sk] = skips.get(sk, 0) + 1
                                if indexed:
                                    refs.setdefault(sk, []).append((sentence, i))
    counts = {k: c for k, c in counts.items() if c >= thr}
    skipthr = max(thr, mintokens_skipgrams) if minskiptypes > 1 else 1
    counts.update({k: c for k, c in skips.items() if c >= skipthr})
    return Model(tokens, 0, counts, {k: sorted(refs[k]) for k in counts} if indexed else None)


def _sentences(payload: bytes):
    toks, start, out = [], 0, []
    for j, b in enumerate(payload):
        if b >= 128:
            continue
        tok = payload[start:j + 1]
        start = j + 1
        if tok == b"\x00":
            out.append(toks)
            toks = []
        else:
            toks.append(tok)
    if toks:
        out.append(toks)
    return out


def key_ntokens(k: bytes) -> int:
    return sum(1 for b in k if b < 128)


# ---- continued training (train(..., continued = true), colibri-patternmodeller -E) ----------------------------------------------------
def train_continued(payload: bytes, loaded: "Model", mintokens=2, maxlength=100, indexed=False, firstsentence=1) -> "Model":
    """PatternModel::train(in, options, NULL, NULL, continued = true) on a model that already holds patterns, restated for MINTOKENS > 1 without
    skipgrams (reference include/patternmodel.h:983-995: an order the model already has n-grams of is skipped — "Skipping n-grams, already in model";
    :1139-1152: the look-back of every other order asks the model, i.e. the loaded patterns and the new survivors alike; :1047-1048: the token total
    is not touched; :1189-1194: "None found" does not end a continued run; :1197: neither is the type total). Returns the whole model: the loaded
    patterns unchanged plus the new orders. Pure Python: small inputs."""
    thr = 2 if mintokens == -1 else max(1, mintokens)
    assert thr > 1
    counts = dict(loaded.counts)
    refs = {k: list(v) for k, v in loaded.refs.items()} if indexed else None
    have = {}
    for k in counts:
        if not any(t in (b"\x03", b"\x04") for t in key_tokens(k)):  # n-grams only (the one-byte tokens 03 / 04 are the skip / flex classes, classencoder.h:61-62)
            have[key_ntokens(k)] = True
    sentences = _sentences(payload)
    for n in range(1, maxlength + 1):
        if have.get(n):
            continue
        new, newrefs = {}, {}
        sentence = firstsentence - 1
        for toks in sentences:
            sentence += 1
            for i in range(len(toks) - n + 1):
                if n > 1 and not (b"".join(toks[i:i + n - 1]) in counts and b"".join(toks[i + 1:i + n]) in counts):
                    continue
                k = b"".join(toks[i:i + n])
                new[k] = new.get(k, 0) + 1
                if indexed:
                    newrefs.setdefault(k, []).append((sentence, i))
        for k, c in new.items():
            if c >= thr:
                counts[k] = c
                if indexed:
                    refs[k] = sorted(newrefs[k])
    return Model(loaded.tokens, loaded.types, counts, refs)


# ---- filtered training (train(..., filter)) --------------------------------------------------------------------------------------------
def train_filtered(payload: bytes, filterkeys, mintokens=2, maxlength=100, indexed=False, firstsentence=1) -> "Model":
    """PatternModel::train(in, options, NULL, filter) restated for MINLENGTH = 1 without skipgram options (reference include/patternmodel.h:899-914: which kinds of
    patterns the filter holds; :1106-1133: a window passes iff one of its sub-n-grams of any length is in the filter, or it is an instance of one of the filter's
    skipgrams — src/pattern.cpp:1760-1785: same length, every non-gap token equal; flexgrams match nothing; :1137-1152: a filtered order has NO look-back;
    :1189-1194: the order loop ends at the first order that finds nothing; MINTOKENS = 1: one pass over all lengths, :1069-1072). Pure Python: small inputs."""
    thr = 2 if mintokens == -1 else max(1, mintokens)
    F = set(filterkeys)
    ngrams = {k for k in F if key_category(k) == 1}
    skipgrams = [key_tokens(k) for k in F if key_category(k) == 2]
    has_ngrams, has_skip = bool(ngrams), any(key_category(k) != 1 for k in F)

    def matches(toks):
        n = len(toks)
        if has_ngrams:
            for m in range(1, n + 1):
                for j in range(n - m + 1):
                    if b"".join(toks[j:j + m]) in ngrams:
                        return True
        if has_skip:
            for sk in skipgrams:
                if len(sk) == n and all(a == b or a == b"\x03" for a, b in zip(sk, toks)):
                    return True
        return False

    sentences = _sentences(payload)
    counts, refs, tokens, types = {}, {}, sum(len(t) for t in sentences), 0
    for n in range(1, maxlength + 1):
        new, newrefs = {}, {}
        sentence = firstsentence - 1
        for toks in sentences:
            sentence += 1
            for i in range(len(toks) - n + 1):
                w = toks[i:i + n]
                if not matches(w):
                    continue
                k = b"".join(w)
                new[k] = new.get(k, 0) + 1
                if indexed:
                    newrefs.setdefault(k, []).append((sentence, i))
        if n == 1:
            types = len(new)  # the unigrams that were counted, before pruning (:1199-1201)
        if not new and thr > 1:
            break
        for k, c in new.items():
            if c >= thr:
                counts[k] = c
                if indexed:
                    refs[k] = sorted(newrefs[k])
    return Model(tokens, types, counts, refs if indexed else None)


# ---- one pattern per line (PatternModelOptions::DOPATTERNPERLINE, colibri-patternmodeller -L) -----------------------------------------
def train_patternperline(payload: bytes, maxlength=100) -> "Model":
    """PatternModel::train with DOPATTERNPERLINE at MINTOKENS = 1 (the CLI's -L implies -t 1, src/patternmodeller.cpp:677-678) restated:
    every non-empty line of at most MAXLENGTH tokens is one pattern — the whole line, no sub-n-grams (include/patternmodel.h:1055-1058);
    tokens = all tokens of the corpus (:1047-1048, counted before the length check), types = the distinct one-token lines
    (totalwordtypesingroup(NGRAM, 1) at :1201-1207). Pure Python: small inputs."""
    counts, tokens, toks, start = {}, 0, [], 0
    lines = []
    for j, b in enumerate(payload):
        if b >= 128:
            continue
        tok = payload[start:j + 1]
        start = j + 1
        if tok == b"\x00":
            lines.append(toks)
            toks = []
        else:
            toks.append(tok)
    if toks:
        lines.append(toks)
    for toks in lines:
        tokens += len(toks)
        if 1 <= len(toks) <= maxlength:
            k = b"".join(toks)
            counts[k] = counts.get(k, 0) + 1
    types = sum(1 for k in counts if len(key_tokens(k)) == 1)
    return Model(tokens, types, counts, None)


# ---- flexgrams from skipgrams ---------------------------------------------------------------------------------------
def key_tokens(key: bytes):
    """The tokens of a pattern key (each a varint: bytes >= 128 continue, a byte < 128 ends the token)."""
    out, start = [], 0
    for j, b in enumerate(key):
        if b < 128:
            out.append(key[start:j + 1])
            start = j + 1
    return out


def key_category(key: bytes) -> int:
    """Pattern::category (reference src/pattern.cpp:107-127): 3 = flexgram (any {**}), 2 = skipgram (any {*}), 1 = n-gram."""
    toks = key_tokens(key)
    return 3 if b"\x04" in toks else 2 if b"\x03" in toks else 1


def toflexgram(key: bytes) -> bytes:
    """Pattern::toflexgram (reference src/pattern.cpp:145-180): every run of {*} tokens becomes one {**}."""
    out, gap = [], False
    for t in key_tokens(key):
        if t == b"\x03":
            if not gap:
                out.append(b"\x04")
            gap = True
        else:
            out.append(t)
            gap = False
    return b"".join(out)


def flexgrams_from_skipgrams(m: "Model"):
    """IndexedPatternModel::computeflexgrams_fromskipgrams (reference include/patternmodel.h:3724-3744) without its
    insert-while-iterating hazard: every skipgram's references are appended to the flexgram it abstracts to (IndexedData::insert
    is a push_back: duplicates stay). Canonical form: references ascending. Returns (model with the flexgrams added, number of
    new flexgrams)."""
    counts, refs = dict(m.counts), {k: list(v) for k, v in m.refs.items()}
    found = 0
    for k in m.refs:
        if key_category(k) != 2:
            continue
        f = toflexgram(k)
        if f not in refs:
            found += 1
            refs[f] = []
        refs[f].extend(m.refs[k])
    for k in refs:
        if key_category(k) == 3:
            refs[k].sort()
        counts[k] = len(refs[k])
    return Model(m.tokens, m.types, counts, refs, m.stats, m.maxn, m.windows), found


# ---- class encoder: oracle/classenc_oracle.cpp and the real reference -----------------------------------------------
CLASSENC_LIB_PATH = os.path.join(HERE, "libclassenc_oracle.so")
_cel = None


def classenc_lib():
    global _cel
    if _cel is None:
        if not os.path.exists(CLASSENC_LIB_PATH):
            build()
        L = C.CDLL(CLASSENC_LIB_PATH)
        L.classenc_oracle_run.restype = C.c_int
        L.classenc_oracle_run.argtypes = [C.c_char_p, C.c_uint64, C.c_uint, C.c_int, C.c_char_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.classenc_oracle_free.argtypes = [C.c_void_p]
        _cel = L
    return _cel


def classencode(text: bytes, threshold=0, allowunknown=False, cls: bytes = None, extend=False):
    """-> (status, class file text, .colibri.dat bytes incl. the A2 02 header). status 1 = unknown token (strict mode)."""
    L = classenc_lib()
    co, cn, do, dn = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
    rc = L.classenc_oracle_run(text, len(text), threshold, int(allowunknown), cls, len(cls) if cls is not None else 0, int(extend), C.byref(co), C.byref(cn), C.byref(do),
                               C.byref(dn))
    out = (rc, C.string_at(co, cn.value), C.string_at(do, dn.value))
    L.classenc_oracle_free(co)
    L.classenc_oracle_free(do)
    return out


def ref_classencode(text_path: str, prefix: str, threshold=0, allowunknown=False, cls_path=None, extend=False):
    """the REAL reference's class encoder (build container only) -> exit status (4 = unknown token); writes <prefix>.colibri.{cls,dat}"""
    cmd = [REF_DRIVER, "encode", text_path, prefix, "-t", str(threshold)]
    if cls_path:
        cmd += ["-c", cls_path]
    if extend:
        cmd.append("-e")
    if allowunknown:
        cmd.append("-U")
    return subprocess.run(cmd, capture_output=True).returncode


def parse_cls(text: bytes) -> dict:
    """class file -> {word bytes: class}; line order in the file is unordered_map order and carries no meaning"""
    out = {}
    for ln in text.split(b"\n"):
        if b"\t" in ln:
            c, w = ln.split(b"\t", 1)
            out[w] = int(c)
    return out
