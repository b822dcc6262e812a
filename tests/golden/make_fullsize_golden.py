#!/usr/bin/env python
"""Full-size fixtures from the REAL reference (build container only: needs oracle/_ref/ref_driver, i.e. /root/reference).

  python tests/golden/make_fullsize_golden.py [case ...]        (no argument: every case; cases run one after the other)

For each case: generate the bench corpus (colibri_amd.synth, fixed seed), run the reference's own PatternModel::train /
IndexedPatternModel::train on it through oracle/_ref/ref_driver (reference include/patternmodel.h:880-1345, :2828-2844, :2969-3010;
the benchmark the reference runs this way is src/benchmarks.cpp:217-237), let it WRITE its model file (patternmodel.h write()),
parse that file, and commit a summary under tests/golden/fullsize/<case>.json:
  tokens, types, patterns, patterns by length, occurrences, references, per-order found / pruned / kept as the reference printed them on
  stderr, and the four 64-bit multiset checksums of colibri_amd.digest.model_digest over the rows (key bytes, count[, reference list]).
A summary is data about the reference's output — no reference source is stored. tests/test_gpu_fullsize.py and bench.py's self-check
compare the HIP path's model against these numbers; the z100m case takes the reference ~7 min and ~5 GB here.
"""
import json
import os
import platform
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
OUT = os.path.join(ROOT, "tests", "golden", "fullsize")
REF_DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")

# name -> (synth.zipf_corpus arguments, ref_driver mode, extra ref_driver arguments); n <= 5, threshold 2 everywhere (BASELINE.json)
CASES = {
    # BASELINE.json configs[1]: what bench.py times
    "z100m_seed44_plain": (dict(ntok=100_000_000, vocab=1_000_000, seed=44), "U", []),
    # the same size with injected repeated phrases, so that orders 4-5 do real work
    "z100m_seed44_phrases_plain": (dict(ntok=100_000_000, vocab=1_000_000, seed=44, phrases=True), "U", []),
    # the id-keeping modes of configs[3] / configs[4] on the corpus of tests/test_gpu_fullsize.py::test_id_keeping_modes_*
    "z20m_seed7_phrases_indexed": (dict(ntok=20_000_000, vocab=300_000, seed=7, phrases=True), "i", []),
    "z20m_seed7_phrases_exhaustive_skipgrams": (dict(ntok=20_000_000, vocab=300_000, seed=7, phrases=True), "us", []),
    "z20m_seed7_phrases_indexed_skipgrams_T1": (dict(ntok=20_000_000, vocab=300_000, seed=7, phrases=True), "is", ["-T", "1"]),
    # BASELINE.json configs[2]: the 1 B-token corpus bench.py's other_configs.z1b_* build — the eight 125 M-token shards (seeds 44..51) an 8-GPU run holds,
    # concatenated in rank order (one header). The reference needs ~35-50 GB and 1-2 h for it; the address space is capped so that it fails instead of
    # taking the container down. Its model FILE is not kept (1 GB): only the digest.
    "z1b_seeds44_51_plain": (dict(ntok=125_000_000, vocab=1_000_000, seeds=list(range(44, 52))), "U", []),
    # round 5: the id-keeping kinds on the TIMED corpus (bench.py other_configs.indexed / .exhaustive_skipgrams carry a self_check against these), and the three shards
    # of other_configs.z375m_single_device
    "z100m_seed44_indexed": (dict(ntok=100_000_000, vocab=1_000_000, seed=44), "i", []),
    "z100m_seed44_exhaustive_skipgrams": (dict(ntok=100_000_000, vocab=1_000_000, seed=44), "us", []),
    "z375m_seeds44_46_plain": (dict(ntok=125_000_000, vocab=1_000_000, seeds=list(range(44, 47))), "U", []),
    # round 6: the reference lists at 375 M tokens (IndexedPatternModel::train on the three shards: ~650 M references), so that other_configs.z375m_single_device.indexed
    # is held to the reference's own forward index, not only to the plain model's (key, count) rows
    "z375m_seeds44_46_indexed": (dict(ntok=125_000_000, vocab=1_000_000, seeds=list(range(44, 47))), "i", []),
    # the fallback the round-3 review names if the container cannot hold the above: four shards
    "z500m_seeds44_47_plain": (dict(ntok=125_000_000, vocab=1_000_000, seeds=list(range(44, 48))), "U", []),
}

LINE = re.compile(r"Found (\d+) (ngrams|skipgrams)\.\.\.pruned (\d+)(?: plus (\d+) extra skipgrams)?\.*total kept: (\d+)")


def run_case(name):
    from colibri_amd import digest, synth
    kw, mode, extra = CASES[name]
    kw = dict(kw)
    t0 = time.time()
    seeds = kw.pop("seeds", None)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        corpus = os.path.join(td, "c.colibri.dat")
        model = os.path.join(td, "m.colibri.patternmodel")
        with open(corpus, "wb") as f:
            if seeds is None:
                f.write(synth.zipf_corpus(kw.pop("ntok"), kw.pop("vocab"), kw.pop("seed"), **kw))
            else:  # shards of an N-GPU run, concatenated in rank order under one header
                ntok, vocab = kw.pop("ntok"), kw.pop("vocab")
                f.write(synth.HEADER)
                for seed in seeds:
                    f.write(synth.zipf_corpus(ntok, vocab, seed, header=False, **kw))
        gen_s = time.time() - t0
        cmd = [REF_DRIVER, "train", corpus, mode, "5", "2", "-o", model] + extra
        if seeds is not None:
            cmd = ["prlimit", "--as=%d" % (int(os.environ.get("REF_AS_GB", "57")) << 30)] + cmd
        t0 = time.time()
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            sys.stderr.write(p.stderr[-4000:])
            raise SystemExit("%s: the reference exited with %d after %.0f s" % (name, p.returncode, time.time() - t0))
        wall_s = time.time() - t0
        info = json.loads(p.stdout.strip().splitlines()[-1])
        orders, order, kind = [], None, None
        for line in p.stderr.splitlines():
            m = re.match(r"Counting (\d+)-(grams|skipgrams)", line.strip())
            if m:
                order, kind = int(m.group(1)), m.group(2)
                continue
            m = LINE.search(line)
            if m:
                orders.append({"n": order, "kind": "ngrams" if kind == "grams" else "skipgrams", "found": int(m.group(1)), "pruned": int(m.group(3)),
                               "pruned_extra": int(m.group(4) or 0), "kept": int(m.group(5))})
        mtype, tokens, types, key_off, key_bytes, counts, refs = digest.parse_model_file(model)
        summary = digest.model_digest(key_off, key_bytes, counts, refs)
    assert tokens == info["tokens"] and types == info["types"]
    summary.update({
        "case": name, "corpus": CASES[name][0], "generator": "colibri_amd.synth.zipf_corpus", "reference_mode": mode, "reference_args": ["5", "2"] + extra,
        "model_type": mtype, "tokens": int(tokens), "types": int(types), "orders": orders,
        "reference_train_s": info["train_s"], "reference_load_s": info["load_s"], "reference_wall_s": round(wall_s, 1),
        "host": {"cpu": platform.processor() or platform.machine(), "cores": os.cpu_count(), "threads_used": 1},
        "corpus_generation_s": round(gen_s, 1),
    })
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
        f.write("\n")
    print(name, "->", summary["npatterns"], "patterns,", summary.get("nrefs", 0), "references, reference train", info["train_s"], "s", flush=True)


if __name__ == "__main__":
    if not os.access(REF_DRIVER, os.X_OK):
        sys.exit("oracle/_ref/ref_driver is not built (make -C oracle ref; needs /root/reference)")
    for case in (sys.argv[1:] or list(CASES)):
        run_case(case)
