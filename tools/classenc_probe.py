"""Throughput of the class encoder path on one MI355X, with the reference's ClassEncoder timed beside it on a sample.
Synthetic tokenised text: Zipf(1.0) over V word forms "w<base36 rank>", lines of 5..35 words. Prints one JSON object.
  device: colibri_text_upload (H2D + line/segment scan) | count under the frequency-list rules | words D2H | host: classes |
          count under the encoder's rules | encode | (fetch)            — the steps colibri-classencode performs
  cpu:    oracle/_ref/ref_driver encode (ClassEncoder::build + save + encodefile) on the first `--cpu-words` words, 1 core
Not the bench contract (bench.py measures the north-star path); numbers go into DESIGN.md."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def make_text(nwords, vocab, seed):
    from colibri_amd import synth
    rng = np.random.default_rng(seed)
    toks = synth.zipf_tokens(nwords, vocab, rng) - 6
    lens = synth.sentence_lengths(nwords, rng)
    forms = np.array([("w" + np.base_repr(int(r), 36).lower()).encode() for r in range(vocab)], dtype=object)
    maxlen = max(len(f) for f in forms)
    table = np.zeros((vocab, maxlen + 1), dtype=np.uint8)
    flen = np.zeros(vocab, dtype=np.int64)
    for r, f in enumerate(forms):
        table[r, :len(f)] = np.frombuffer(f, dtype=np.uint8)
        flen[r] = len(f)
    sep = np.full(nwords, ord(" "), dtype=np.uint8)
    sep[np.cumsum(lens)[:-1] - 1] = ord("\n")
    sep[-1] = ord("\n")
    wl = flen[toks] + 1
    off = np.concatenate([[0], np.cumsum(wl)])
    out = np.zeros(int(off[-1]), dtype=np.uint8)
    for k in range(maxlen):
        m = flen[toks] > k
        out[off[:-1][m] + k] = table[toks[m], k]
    out[off[1:] - 1] = sep
    return out.tobytes()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--words", type=int, default=100_000_000)
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--cpu-words", type=int, default=4_000_000)
    a = ap.parse_args()
    from colibri_amd import capi
    import oracle
    t0 = time.time()
    text = make_text(a.words, a.vocab, 48)
    gen_s = time.time() - t0
    L = capi.load()
    res = {"words": a.words, "vocab": a.vocab, "text_bytes": len(text), "generation_s": round(gen_s, 2)}
    with capi.Context(0) as ctx:
        h = ctx.h
        nw, nd, ob, nt, nl = (C.c_uint64() for _ in range(5))
        best = {}
        for rep in range(3):
            t = {}
            s = time.perf_counter(); assert L.colibri_text_upload(h, text, C.c_uint64(len(text))) == 0; t["upload_scan"] = time.perf_counter() - s
            s = time.perf_counter(); assert L.colibri_text_count(h, 0, C.byref(nw), C.byref(nd)) == 0; t["count_freqlist"] = time.perf_counter() - s
            n = nd.value
            start, length, count = (np.zeros(n, dtype=np.uint32) for _ in range(3))
            s = time.perf_counter(); assert L.colibri_text_words(h, start.ctypes.data_as(C.c_void_p), length.ctypes.data_as(C.c_void_p), count.ctypes.data_as(C.c_void_p)) == 0; t["words_d2h"] = time.perf_counter() - s
            s = time.perf_counter()
            order = np.lexsort((start, -count.astype(np.int64)))  # stand-in for buildclasses (the C++ face uses the reference's containers)
            cls_of = np.empty(n, dtype=np.uint32); cls_of[order] = np.arange(6, 6 + n, dtype=np.uint32)
            t["host_classes"] = time.perf_counter() - s
            s = time.perf_counter(); assert L.colibri_text_count(h, 1, C.byref(nw), C.byref(nd)) == 0; t["count_encoder_rules"] = time.perf_counter() - s
            start2, length2, count2 = (np.zeros(nd.value, dtype=np.uint32) for _ in range(3))
            assert L.colibri_text_words(h, start2.ctypes.data_as(C.c_void_p), length2.ctypes.data_as(C.c_void_p), count2.ctypes.data_as(C.c_void_p)) == 0
            # same text, same rules for these word forms: map through first occurrence
            lut = dict(zip(start.tolist(), cls_of.tolist()))
            cls2 = np.array([lut[x] for x in start2.tolist()], dtype=np.uint32)
            rep1 = np.ones(nd.value, dtype=np.uint32)
            s = time.perf_counter(); assert L.colibri_text_encode(h, cls2.ctypes.data_as(C.c_void_p), rep1.ctypes.data_as(C.c_void_p), C.byref(ob), C.byref(nt), C.byref(nl)) == 0; t["encode"] = time.perf_counter() - s
            for k, v in t.items():
                best[k] = min(best.get(k, 1e9), v)
        res.update({"distinct": nd.value, "encoded_bytes": ob.value, "lines": nl.value, "device_s": {k: round(v, 4) for k, v in best.items()}})
        dev = best["count_freqlist"] + best["count_encoder_rules"] + best["encode"]
        res["device_kernels_Mwords_per_s"] = round(a.words / dev / 1e6, 1)
        res["device_incl_upload_Mwords_per_s"] = round(a.words / (dev + best["upload_scan"] + best["words_d2h"]) / 1e6, 1)
    if oracle.have_ref() and a.cpu_words:
        cut = 0
        seen = 0
        # first cpu_words words: cut at a line end
        arr = np.frombuffer(text, dtype=np.uint8)
        seps = np.flatnonzero((arr == 32) | (arr == 10))
        cut = int(seps[min(a.cpu_words, seps.size) - 1]) + 1
        nl = arr[:cut].tobytes().rfind(b"\n") + 1
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "s.txt")
            open(p, "wb").write(text[:nl])
            nwords_s = int(np.count_nonzero((arr[:nl] == 32) | (arr[:nl] == 10)))
            s = time.perf_counter()
            subprocess.check_call([oracle.REF_DRIVER, "encode", p, os.path.join(d, "o")])
            cpu = time.perf_counter() - s
        res["cpu_reference"] = {"words": nwords_s, "seconds": round(cpu, 2), "Mwords_per_s": round(nwords_s / cpu / 1e6, 2), "cores": 1}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
