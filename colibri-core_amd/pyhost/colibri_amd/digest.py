"""Order-independent digests of a pattern model — test / bench infrastructure, not part of the product path.

A model is a multiset of rows (key bytes, count[, reference list]). Two independent 64-bit row hashes are computed per row
(vectorised), and the model's digest is (sum, xor) of each over all rows: four 64-bit numbers that do not depend on the order
in which an implementation lists its patterns (the reference's order is libstdc++ unordered_map order). The committed
full-size fixtures (tests/golden/fullsize/*.json, made by tests/golden/make_fullsize_golden.py from models the real
reference wrote) hold these numbers; tests/test_gpu_fullsize.py and bench.py's self-check compare the HIP path's model
against them.
"""
import struct

import numpy as np


def row_hashes(key_off, key_bytes, extra=None):
    """two independent 64-bit hashes per key: polynomials in the key bytes over Z/2^64 with odd multipliers; `extra` (u64 per
    row: the count, or count mixed with a reference-list digest) is folded in last"""
    n = key_off.size - 1
    lens = (key_off[1:] - key_off[:-1]).astype(np.int64)
    maxlen = int(lens.max()) if n else 0
    h1 = np.full(n, 0x9E3779B97F4A7C15, dtype=np.uint64)
    h2 = np.full(n, 0xC2B2AE3D27D4EB4F, dtype=np.uint64)
    starts = key_off[:-1].astype(np.int64)
    m1, m2 = np.uint64(0x100000001B3), np.uint64(0xD6E8FEB86659FD93)
    with np.errstate(over="ignore"):
        for b in range(maxlen):
            sel = lens > b
            v = key_bytes[starts[sel] + b].astype(np.uint64) + np.uint64(1)
            h1[sel] = (h1[sel] ^ v) * m1
            h2[sel] = (h2[sel] + v) * m2
        h1 ^= lens.astype(np.uint64) << np.uint64(56)
        if extra is not None:
            h1 = (h1 ^ extra.astype(np.uint64)) * m1
            h2 = (h2 + extra.astype(np.uint64)) * m2
    return h1, h2


def refs_digest(ref_off, ref_sentence, ref_token):
    """per pattern: an order-dependent 64-bit digest of its reference list (sentence, token)"""
    v = (ref_sentence.astype(np.uint64) << np.uint64(16)) | ref_token.astype(np.uint64)
    ref_off = ref_off.astype(np.uint64)
    pos = np.arange(v.size, dtype=np.uint64) - np.repeat(ref_off[:-1], (ref_off[1:] - ref_off[:-1]).astype(np.int64))
    with np.errstate(over="ignore"):
        w = (v + np.uint64(0x9E3779B97F4A7C15)) * (np.uint64(2) * pos + np.uint64(0x100000001B3))
        csum = np.concatenate([[np.uint64(0)], np.cumsum(w, dtype=np.uint64)])
    return csum[ref_off[1:].astype(np.int64)] - csum[ref_off[:-1].astype(np.int64)]


def row_extra(counts, refs=None):
    """the u64 folded into a row's hashes: the count, mixed with the digest of the whole reference list for indexed models"""
    extra = counts.astype(np.uint64)
    if refs is not None:
        with np.errstate(over="ignore"):
            extra = extra * np.uint64(0xD6E8FEB86659FD93) + refs_digest(*refs)
    return extra


def model_digest(key_off, key_bytes, counts, refs=None):
    """{'sum1','xor1','sum2','xor2'} (hex strings) + totals: the multiset digest of (key bytes, count[, refs]) rows"""
    h1, h2 = row_hashes(key_off, key_bytes, row_extra(counts, refs))
    with np.errstate(over="ignore"):
        out = {
            "sum1": "%016x" % int(np.add.reduce(h1, dtype=np.uint64)) if h1.size else "0" * 16,
            "xor1": "%016x" % int(np.bitwise_xor.reduce(h1)) if h1.size else "0" * 16,
            "sum2": "%016x" % int(np.add.reduce(h2, dtype=np.uint64)) if h2.size else "0" * 16,
            "xor2": "%016x" % int(np.bitwise_xor.reduce(h2)) if h2.size else "0" * 16,
        }
    term = key_bytes < 128
    ntok = np.add.reduceat(term.astype(np.int64), key_off[:-1].astype(np.int64)) if counts.size else np.zeros(0, dtype=np.int64)
    out["npatterns"] = int(counts.size)
    out["occurrences"] = int(counts.astype(np.uint64).sum())
    out["keybytes"] = int(key_bytes.size)
    out["patterns_by_length"] = {str(int(n)): int(c) for n, c in zip(*np.unique(ntok, return_counts=True))}
    if refs is not None:
        out["nrefs"] = int(refs[1].size)
    return out


def combine(parts):
    """the digest of a model given as disjoint shares (one per rank of a sharded trainer): sums add mod 2^64, xors xor, totals add"""
    out = {"sum1": 0, "xor1": 0, "sum2": 0, "xor2": 0, "npatterns": 0, "occurrences": 0, "keybytes": 0, "patterns_by_length": {}}
    for d in parts:
        for k in ("sum1", "sum2"):
            out[k] = (out[k] + int(d[k], 16)) & 0xFFFFFFFFFFFFFFFF
        for k in ("xor1", "xor2"):
            out[k] ^= int(d[k], 16)
        for k in ("npatterns", "occurrences", "keybytes"):
            out[k] += d[k]
        for n, c in d["patterns_by_length"].items():
            out["patterns_by_length"][n] = out["patterns_by_length"].get(n, 0) + c
    for k in ("sum1", "xor1", "sum2", "xor2"):
        out[k] = "%016x" % out[k]
    return out


def parse_model_file(path):
    """.colibri.patternmodel (reference include/patternmodel.h:670-760 writer; types 10 = unindexed, 20 = indexed) -> flat arrays
    (mtype, tokens, types, key_off u64[n+1], key_bytes u8[], counts u32[n], refs or None) in file order. Keys without the
    trailing 00, as colibri_export_* gives them."""
    raw = np.fromfile(path, dtype=np.uint8)
    assert raw[0] == 0 and raw[2] == 2, "not a v2 pattern model"
    mtype = int(raw[1])
    tokens, types, npat = struct.unpack_from("<QQQ", raw[:27].tobytes(), 3)
    buf = raw.tobytes()
    kstart = np.empty(npat, dtype=np.int64)
    kend = np.empty(npat, dtype=np.int64)
    counts = np.empty(npat, dtype=np.uint32)
    pos = 27
    indexed = mtype == 20
    unpack = struct.unpack_from
    for i in range(npat):
        start = pos
        while True:  # a key ends at the first 00 that does not continue a varint (a byte < 128 ends a token; 00 right after one ends the key)
            z = buf.find(b"\x00", pos)
            if z == start or buf[z - 1] < 128:
                break
            pos = z + 1
        kstart[i], kend[i] = start, z
        (c,) = unpack("<I", buf, z + 1)
        counts[i] = c
        pos = z + 5 + (6 * c if indexed else 0)
    assert pos == len(buf), (pos, len(buf))
    lens = kend - kstart
    key_off = np.zeros(npat + 1, dtype=np.uint64)
    np.cumsum(lens, out=key_off[1:])
    idx = np.repeat(kstart - key_off[:-1].astype(np.int64), lens) + np.arange(int(key_off[-1]), dtype=np.int64)
    key_bytes = raw[idx]
    refs = None
    if indexed:
        ref_off = np.zeros(npat + 1, dtype=np.uint64)
        np.cumsum(counts.astype(np.uint64), out=ref_off[1:])
        nrefs = int(ref_off[-1])
        rstart = kend + 5  # first reference of each pattern
        base = np.repeat(rstart - 6 * ref_off[:-1].astype(np.int64), counts.astype(np.int64)) + 6 * np.arange(nrefs, dtype=np.int64)
        sent = np.zeros(nrefs, dtype=np.uint32)
        for b in range(4):
            sent |= raw[base + b].astype(np.uint32) << np.uint32(8 * b)
        tok = raw[base + 4].astype(np.uint16) | (raw[base + 5].astype(np.uint16) << np.uint16(8))
        refs = (ref_off, sent, tok)
    return mtype, tokens, types, key_off, key_bytes, counts, refs


def source_digest(root):
    """sha256 over the device library's sources (colibri-core_amd/csrc/* and include/colibri_hip.h, by name): what ties a committed profile to the library it was taken from
    (profiles/pmc_dominant_kernel.json carries it; bench.py prints roofline.traffic only when the tree it runs from has the same one)"""
    import hashlib
    import os
    h = hashlib.sha256()
    csrc = os.path.join(root, "colibri-core_amd", "csrc")
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc))] + [os.path.join(root, "include", "colibri_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()
