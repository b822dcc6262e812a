#!/usr/bin/env python
"""bench.py — the driver's benchmark contract for the hot path (PatternModel::train, n <= 5, thr = 2).

  python bench.py --gpus N --steps K --warmup W                    (any N: the N ranks are host threads of this one process, RCCL linked by the C++ trainer)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...   (one process per rank)

A "step" is one complete train() (all five orders: scan + key + count + prune + look-back state for the next order) over one synthetic class-encoded corpus that
is already resident in HBM when the timed region starts.
  N = 1 : BASELINE.json configs[1] — 100M-token Zipf(1.0, V = 1e6) corpus, unindexed, n <= 5, threshold 2, one device (colibri_train).
  N > 1 : BASELINE.json configs[2] — the corpus is sharded by sentence, 100M tokens PER RANK (weak scaling; N = 8: 125M per rank = the 1B-token corpus), trained by
          the product's own multi-GPU trainer (colibri-core_amd/host/src/sharded.cpp through include/colibri_sharded.h — what colibri-patternmodeller --gpus N
          runs): order 1 an RCCL all-reduce of the dense class counts, orders >= 2 key-sharded counting (records to the owner of their key, csrc/kshard.hpp).
          --force-shard runs that trainer with one rank (prices the exchange machinery against the single-device step on the same corpus).
The run checks what it timed: the model of the default corpus (seed 44) must be the one the REAL reference built from the same bytes — per-order kept counts, pattern
total and the multiset digest of (key, count) rows in tests/golden/fullsize/z100m_seed44_plain.json (made by tests/golden/make_fullsize_golden.py) — or the exit
status is non-zero. `other_configs` reports the other model kinds of BASELINE.json (configs[3], configs[4]) and the phrase-injected corpus of SURVEY 8(d) from
extra, untimed steps.
Metric (BASELINE.json): M patterns counted / s, patterns counted = sum_{n<=5} W_n = the n-token windows inside sentences that the reference enumerates in
line.ngrams() (include/patternmodel.h:1063) — a property of the input.
Rank 0 prints ONE JSON line. `roofline` prices the dominant kernel (count) against HBM peak with the algorithmic bytes of SURVEY.md 8(d) (stated in DESIGN.md 4);
`cpu_baseline` times the real reference (oracle/_ref/ref_driver, built from the reference's own sources) on a bounded sample of the same distribution on this
box's host cores.
"""
import argparse
import concurrent.futures
import json
import multiprocessing
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
XGMI_A2A_GBS = 270.0   # assumed all-to-all rate per GPU and direction for the N = 8 prediction: half of 7 links x 76.8 GB/s (153.6 GB/s bidirectional each)
MAXLENGTH, MINTOKENS = 5, 2
FIXTURES = os.path.join(ROOT, "tests", "golden", "fullsize")


def make_corpus(spec):
    """(tokens, vocab, seed, phrases) -> v2 payload bytes (worker process: numpy only)"""
    from colibri_amd import synth
    tokens, vocab, seed, phrases = spec
    return synth.zipf_corpus(tokens, vocab, seed, phrases=phrases, header=False)


def algorithmic_bytes(nbytes, npos, stats, maxlength):
    """SURVEY.md 8(d), per order n, for the counting stage (scan + key + table build):
         scan share   B + 4*(T+S)            corpus bytes + token-start vector, read once
                    + [n>1] * W_n * 2/8      two survivor bits per window (look-back)
         build share  P_n * (8 + 4 + 4)      per admitted window: key read, count read, count write
                    + D_n * (8 + 4)          per distinct candidate: key + count written once
       (the T/8 survivor-bitmap write of the formula belongs to the resolve kernel and is left out).
       Returns per-order lists (scan bytes, build bytes), index 0 = order 1."""
    scan, build = [], []
    for n in range(1, maxlength + 1):
        scan.append(nbytes + 4.0 * npos + (stats.windows[n] * 2.0 / 8.0 if n > 1 else 0.0))
        build.append(stats.admitted[n] * 16.0 + stats.found[n] * 12.0)
    return scan, build


def load_fixture(name):
    try:
        with open(os.path.join(FIXTURES, name + ".json")) as f:
            return json.load(f)
    except Exception:
        return None


def check_against_reference(fixture, kept, npatterns, arrays=None):
    """per-order kept counts, pattern total and — when the exported model is given — the multiset digest of its (key, count) rows, against what the real
    reference left for the same corpus (tests/golden/fullsize/*.json)"""
    from colibri_amd import digest
    want_kept = [o["kept"] for o in fixture["orders"] if o["kind"] == "ngrams"]
    ok = kept[:len(want_kept)] == want_kept and npatterns == fixture["npatterns"]
    if ok and arrays is not None:
        d = digest.model_digest(arrays[0], arrays[1], arrays[2])
        ok = all(d[k] == fixture[k] for k in ("sum1", "xor1", "sum2", "xor2", "npatterns", "occurrences", "keybytes"))
    return ok


def cpu_baseline(sample_tokens, vocab):
    """Time the reference's PatternModel<uint32_t>::train on a preloaded IndexedCorpus (src/benchmarks.cpp test 5 style), 1 thread (the reference is
    single-threaded), on a bounded sample of the bench distribution; beside it the committed figure of the same reference on the FULL corpus of the timed run."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    from colibri_amd import synth
    data = synth.zipf_corpus(sample_tokens, vocab, 45)
    arr = np.frombuffer(data, dtype=np.uint8)[2:]
    term = arr < 128
    prev_low = np.concatenate([[True], term[:-1]])
    delim = (arr == 0) & prev_low
    dpos = np.flatnonzero(delim[term])  # delimiter positions in position space
    lens = np.diff(np.concatenate([[-1], dpos])) - 1
    windows = int(sum(np.maximum(0, lens - n + 1).sum() for n in range(1, MAXLENGTH + 1)))
    sample = f"{sample_tokens}-token Zipf(1.0,V={vocab}) corpus, seed 45, same generator as the GPU workload; train() only, corpus preloaded"
    full = None
    fx = load_fixture("z100m_seed44_plain")
    if fx is not None:
        full = {"value": round(450005710 / fx["reference_train_s"] / 1e6, 4), "unit": "M patterns counted/s", "seconds": fx["reference_train_s"], "cores": 1,
                "host": f"build container ({fx['host']['cores']} cores), tests/golden/make_fullsize_golden.py — not this box",
                "workload": "the timed run's own 100M-token corpus (seed 44), the real reference's train() on a preloaded IndexedCorpus"}
    if oracle.have_ref():
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "sample.colibri.dat")
            with open(path, "wb") as f:
                f.write(data)
            _, info = oracle.ref_train(path, "U", MAXLENGTH, MINTOKENS)
        return {"value": round(windows / info["train_s"] / 1e6, 4), "unit": "M patterns counted/s", "cores": 1, "kind": "reference",
                "sample": sample, "seconds": round(info["train_s"], 3), "host_cores": os.cpu_count(), "full_size_reference": full}
    dt, w, _ = oracle.train_timed(data[2:], MINTOKENS, MAXLENGTH)
    return {"value": round(w / dt / 1e6, 4), "unit": "M patterns counted/s", "cores": 1, "kind": "port", "sample": sample,
            "seconds": round(dt, 3), "host_cores": os.cpu_count(), "full_size_reference": full}


def other_configs(ctx, capi, nbytes, default_corpus=False):
    """The other model kinds of BASELINE.json on the corpus that is resident: configs[3] (skipgrams: the exhaustive unindexed variant and the
    indexed one with MINSKIPTYPES = 2) and configs[4] (indexed model = forward index on the device). Untimed steps after the timed region: best of
    three train() calls each, with the kernel classes bracketed by HIP events in a fourth. Algorithmic bytes: the counting stage of the n-gram
    passes as in `roofline` plus, for indexed models, 8 bytes per reference written once and read once per 8-bit sort pass."""
    res = {}
    all_ok = True
    kinds = (("exhaustive_skipgrams", dict(doskipgrams_exhaustive=1)), ("indexed", dict(indexed=1)), ("indexed_skipgrams_T2", dict(indexed=1, doskipgrams=1, minskiptypes=2)))
    for name, kw in kinds:
        best, st = None, None
        for _ in range(3):
            st = ctx.train(maxlength=MAXLENGTH, mintokens=MINTOKENS, **kw)
            best = st.train_ms if best is None else min(best, st.train_ms)
        ctx.train(maxlength=MAXLENGTH, mintokens=MINTOKENS, profile=1, **kw)
        kms = {capi.KERNEL_CLASSES[k]: round(ctx.kernel_time(k)[0], 3) for k in range(len(capi.KERNEL_CLASSES)) if ctx.kernel_time(k)[1]}
        dom = max(kms, key=kms.get) if kms else None
        scan_n, build_n = algorithmic_bytes(nbytes, ctx.positions(), st, MAXLENGTH)
        algo = sum(scan_n) + sum(build_n) + (8.0 * st.nrefs * 2 if kw.get("indexed") else 0.0)
        res[name] = {"ms_per_step": round(best, 3), "patterns_in_model": int(st.npatterns), "references": int(st.nrefs), "dominant_kernel_class": dom,
                     "kernel_ms_per_step": kms, "algorithmic_bytes_per_step": round(algo), "achieved_GBps": round(algo / (best * 1e-3) / 1e9, 1),
                     "frac_of_hbm_peak": round(algo / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        # round 5: the real reference's own model of THIS corpus in this kind (tests/golden/fullsize/z100m_seed44_{indexed,exhaustive_skipgrams}.json: 750 s / 887 s of
        # its train() in the build container): totals and the multiset digest of every (key, count[, reference list]) row. The indexed skipgram kind at the default
        # MINSKIPTYPES has no stable reference output (tests/golden/unstable_reference_outputs.json).
        fx = load_fixture({"exhaustive_skipgrams": "z100m_seed44_exhaustive_skipgrams", "indexed": "z100m_seed44_indexed"}.get(name, "-")) if default_corpus else None
        if fx is not None:
            from colibri_amd import digest
            key_off, key_bytes, counts, refs = ctx.export_arrays()
            d = digest.model_digest(key_off, key_bytes, counts, refs)
            ok = all(d[k] == fx[k] for k in ("sum1", "xor1", "sum2", "xor2", "npatterns", "occurrences", "keybytes")) and (refs is None or d["nrefs"] == fx["nrefs"])
            res[name]["self_check"] = ("ok" if ok else "FAILED") + ": the multiset digest of every (key, count%s) row equals the real reference's model of this corpus" % (
                ", reference list" if refs is not None else "")
            all_ok = all_ok and ok
            del key_off, key_bytes, counts, refs
        else:
            res[name]["self_check"] = "no reference model for this configuration"
    return res, all_ok


def phrases_config(ctx, capi, payload):
    """the same size with repeated phrases injected (orders 4 and 5 do real work), self-checked against the real reference's model of the same bytes"""
    ctx.upload(payload)
    best, st = None, None
    for _ in range(4):
        st = ctx.train(maxlength=MAXLENGTH, mintokens=MINTOKENS)
        best = st.train_ms if best is None else min(best, st.train_ms)
    kept = [int(st.kept[n]) for n in range(1, MAXLENGTH + 1)]
    fx = load_fixture("z100m_seed44_phrases_plain")
    ok = None
    if fx is not None:
        arrays = ctx.export_arrays()
        ok = check_against_reference(fx, kept, int(st.npatterns), arrays)
        del arrays
    scan_n, build_n = algorithmic_bytes(len(payload), ctx.positions(), st, MAXLENGTH)
    windows = sum(st.windows[1:MAXLENGTH + 1])
    algo = sum(scan_n) + sum(build_n)
    return {"workload": "the timed corpus' tokens and sentences with 15 % of the stream overwritten by copies of 2000 phrases of 3-8 tokens (synth.zipf_corpus(..., phrases=True))",
            "ms_per_step": round(best, 3), "M_patterns_counted_per_s": round(windows / best / 1e3, 1), "patterns_in_model": int(st.npatterns), "kept_per_order": kept,
            "admitted_per_order": [int(st.admitted[n]) for n in range(1, MAXLENGTH + 1)],
            "self_check": ("ok" if ok else "FAILED") if ok is not None else "no fixture", "algorithmic_bytes_per_step": round(algo),
            "frac_of_hbm_peak": round(algo / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}, ok


def z1b_configs(capi, shards):
    """BASELINE.json configs[2]'s corpus — 10^9 tokens, the eight 125 M-token shards an 8-GPU run holds — on ONE device, two ways (untimed extras, best of three):
    one context over the concatenation (the radix path in passes over key slices: an order of more than ~110 M records does not fit one pass), and the multi-GPU
    trainer with its eight ranks all on this device (key-sharded counting; the ranks exchange by device copies). Both must build the same model."""
    def sentences_of(p):
        prev_low = np.concatenate([[True], p[:-1] < 128])
        return int(((p == 0) & prev_low).sum())
    res = {}
    # three of the shards in one context: 375 M tokens, beyond what rounds 1-3 counted in one pass (110 M positions) and kept on the radix path for the id-keeping
    # kinds (128 M tokens); since round 4 one pass of the second-generation engine with 2048-slot bin tables (untimed extras, best of two)
    with capi.Context(0) as c:
        c.upload(np.concatenate(shards[:3]))
        entry = {"workload": "three of configs[2]'s shards (375 M tokens) in one context", "kinds": {}}
        for name, kw in (("plain", {}), ("indexed", dict(indexed=1)), ("exhaustive_skipgrams", dict(doskipgrams_exhaustive=1))):
            best, st = None, None
            for _ in range(2):
                st = c.train(maxlength=MAXLENGTH, mintokens=MINTOKENS, **kw)
                best = st.train_ms if best is None else min(best, st.train_ms)
            mode, passes = c.last_mode(with_passes=True)
            entry["kinds"][name] = {"ms_per_step": round(best, 2), "patterns_in_model": int(st.npatterns), "references": int(st.nrefs),
                                    "path": "radix" if mode == 2 else "global table", "passes_at_order_2": passes}
            # round 5: the reference's own model of these three shards (tests/golden/fullsize/z375m_seeds44_46_plain.json, 2490 s of its train()): the plain model row for
            # row; the indexed model holds the same patterns with the same counts (its reference lists are pinned at 10^8 tokens: other_configs.indexed)
            # round 6: ... and its own INDEXED model of them (z375m_seeds44_46_indexed.json, 2656 s): every (key, count, reference list) row, 650 M references
            fx375 = load_fixture({"plain": "z375m_seeds44_46_plain", "indexed": "z375m_seeds44_46_indexed"}.get(name, "-"))
            if fx375 is not None:
                from colibri_amd import digest
                key_off, key_bytes, counts, refs = c.export_arrays()
                d = digest.model_digest(key_off, key_bytes, counts, refs)
                ok = all(d[k] == fx375[k] for k in ("sum1", "xor1", "sum2", "xor2", "npatterns", "occurrences", "keybytes")) and (refs is None or d["nrefs"] == fx375["nrefs"])
                entry["kinds"][name]["self_check"] = ("ok" if ok else "FAILED") + ": (key, count%s) rows = the reference's model of these shards" % (", reference list" if refs is not None else "")
                del key_off, key_bytes, counts, refs
            else:
                entry["kinds"][name]["self_check"] = "no reference model (the same kind is pinned at 10^8 tokens: other_configs.exhaustive_skipgrams)"
        res["z375m_single_device"] = entry
    fx = load_fixture("z1b_seeds44_51_plain")  # what the reference's own train() left for this corpus (8030 s on one core of the build container)
    want = (fx["npatterns"], [o["kept"] for o in fx["orders"]]) if fx else None
    whole = np.concatenate(shards)
    with capi.Context(0) as c:
        c.upload(whole)
        del whole
        best, st = None, None
        for _ in range(3):
            st = c.train(maxlength=MAXLENGTH, mintokens=MINTOKENS)
            best = st.train_ms if best is None else min(best, st.train_ms)
        mode, passes = c.last_mode(with_passes=True)
    windows = sum(st.windows[1:MAXLENGTH + 1])
    one = (int(st.npatterns), [int(st.kept[n]) for n in range(1, MAXLENGTH + 1)])
    res["z1b_single_device"] = {"workload": "configs[2]'s 1 B-token corpus (8 x 125 M tokens, seeds 44..51) in one context", "ms_per_step": round(best, 2),
                                "M_patterns_counted_per_s": round(windows / best / 1e3, 1), "patterns_in_model": one[0], "kept_per_order": one[1],
                                "passes_over_key_slices_at_order_2": passes,
                                "self_check": ("ok" if one == want else "FAILED") if want else "no fixture"}
    nsent = [sentences_of(p) for p in shards]
    with capi.ShardedTrainer(8, devices=[0] * 8) as tr:
        for r, p in enumerate(shards):
            tr.upload(r, p, 1 + sum(nsent[:r]))
        best = None
        for _ in range(3):
            st = tr.train(maxlength=MAXLENGTH, mintokens=MINTOKENS)
            best = tr.info.wall_ms if best is None else min(best, tr.info.wall_ms)
        info = tr.info
        eight = (int(st.npatterns), [int(st.kept[n]) for n in range(1, MAXLENGTH + 1)])
        check8 = None
        if fx:  # every rank exports its share of the model: the multiset digest of all (key, count) rows against the reference's
            from concurrent.futures import ThreadPoolExecutor
            from colibri_amd import digest
            shares = [tr.export_arrays(r) for r in range(8)]
            with ThreadPoolExecutor(8) as ex:
                got = digest.combine(list(ex.map(lambda a: digest.model_digest(*a), shares)))
            del shares
            check8 = eight == want and all(got[k] == fx[k] for k in ("sum1", "xor1", "sum2", "xor2", "npatterns", "occurrences", "keybytes"))
    res["z1b_eight_ranks_on_one_device"] = {"workload": "the same shards, one rank each, all eight ranks on this device (what `--gpus 8` runs on eight devices, minus xGMI)",
                                            "ms_per_step": round(best, 2), "M_patterns_counted_per_s": round(windows / best / 1e3, 1), "patterns_in_model": eight[0],
                                            "kept_per_order": eight[1], "protocol": "candidate exchange" if info.protocol == 1 else "key-sharded counting",
                                            "alltoall_bytes_per_rank_and_step": int(info.alltoall_bytes), "of_which_to_self": int(info.alltoall_bytes_to_self),
                                            "same_model_as_z1b_single_device": eight == one,
                                            "self_check": ("ok" if check8 else "FAILED") if check8 is not None else "no fixture",
                                            "self_check_against": "the reference's model of the 1 B-token corpus: per-order kept, totals, multiset digest of the "
                                                                  "(key, count) rows (tests/golden/fullsize/z1b_seeds44_51_plain.json)"}
    return res


def cxx_face(payload):
    """What a drop-in user of the reference's API sees: PatternModel<uint32_t>::train() on a preloaded corpus through host/include/patternmodel.h (reference
    include/patternmodel.h:1353-1364, src/benchmarks.cpp:228-237) — a fresh device context, the H2D upload and tokenising, colibri_train, the export of keys and counts to
    host memory — and the first look-up (answered from the flat result arrays through a look-up table built by host threads; the unordered_map of heap Patterns is only
    built when a caller iterates or mutates). host_selftest bench: three runs in one process, the first one cold (context created, device memory reserved, result arrays
    mapped); the later ones reuse the process' idle context and the released model's arrays (host/src/colibri_host.cpp: CtxCache, ResultPool)."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "colibri-core_amd", "bin", "host_selftest")
    if not os.access(exe, os.X_OK):
        return {"error": "colibri-core_amd/bin/host_selftest is not built"}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        path = os.path.join(td, "c.colibri.dat")
        with open(path, "wb") as f:
            f.write(bytes([0xA2, 0x02]))
            f.write(payload.tobytes())
        try:
            p = subprocess.run([exe, "bench", path, str(MAXLENGTH), str(MINTOKENS), "3"], capture_output=True, text=True, timeout=300, env=dict(os.environ, COLIBRI_HOST_TIMING="1"))
            d = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            return {"error": f"host_selftest bench failed: {e}"}
    runs = d["runs"]
    # where a call's time goes (COLIBRI_HOST_TIMING lines of the library and the C++ face on stderr): the best call's phases, and the host -> device copy of the corpus
    # alone per call — 3.4 ms when the runtime pins the caller's pages, 10-35 ms when it stages them through its own buffers
    import re
    phases, copies = [], []
    for line in p.stderr.splitlines():
        m = re.search(r"COLIBRI_HOST_TIMING create ([0-9.]+) upload ([0-9.]+) train ([0-9.]+) sizes ([0-9.]+) alloc ([0-9.]+) export ([0-9.]+) ms", line)
        if m:
            phases.append(dict(zip(("create", "upload", "train", "sizes", "alloc", "export"), (float(x) for x in m.groups()))))
        m = re.search(r"upload: copy of [0-9.]+ MB ([0-9.]+) ms", line)
        if m:
            copies.append(float(m.group(1)))
    extra = {}
    if len(phases) == len(runs) and phases:
        best = min(range(len(runs)), key=lambda i: runs[i]["train_ms"])
        extra["phases_ms_of_the_best_call"] = phases[best]
    if copies:
        extra["corpus_copy_ms_per_call"] = copies
    return {**extra, "workload": "PatternModel<uint32_t>::train(corpusfile, options) on a preloaded IndexedCorpus of the timed corpus, C++ face (host_selftest bench): context + upload + "
                        "colibri_train + export to host vectors, per call",
            "cxx_face_train_ms": round(min(r["train_ms"] for r in runs), 1), "cxx_face_train_ms_first_call": round(runs[0]["train_ms"], 1),
            "first_lookup_ms": round(min(r["first_lookup_ms"] for r in runs), 1), "first_lookup_ms_worst": round(max(r["first_lookup_ms"] for r in runs), 1), "patterns": runs[-1]["patterns"], "corpus_load_ms_host": round(d["corpus_load_ms"], 1)}


def measured_traffic(workload_tokens, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), or None — NOT measured by this run"""
    path = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")
    try:
        with open(path) as f:
            d = json.load(f)
        from colibri_amd import digest
        if int(d.get("tokens", 0)) == int(workload_tokens) and d.get("kernel") == kernel and d.get("csrc_sha256") == digest.source_digest(ROOT):
            return d.get("hbm_bytes_per_launch"), d.get("source", "profiles/pmc_dominant_kernel.json")  # (the passes were taken from THIS source tree's library)
    except Exception:
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tokens", type=int, default=0, help="tokens per GPU (default: 100M — the 100M-token config —, 125M with --gpus 8 = the 1B-token corpus of config 3)")
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="tokens of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the untimed steps of the other model kinds (skipgrams, indexed) and of the phrase corpus")
    ap.add_argument("--share-gpu", action="store_true", help="debugging: all ranks of --gpus N on device 0 (they exchange by device copies, not RCCL)")
    ap.add_argument("--force-shard", action="store_true", help="run the multi-GPU trainer even with one rank (prices the exchange machinery)")
    ap.add_argument("--candidates", action="store_true", help="multi-GPU runs: the candidate-exchange protocol instead of key-sharded counting (comparison)")
    ap.add_argument("--no-z1b", action="store_true", help="skip other_configs.z1b_*: the 1 B-token corpus of configs[2] on this ONE device (two untimed extras, ~1 minute)")
    args = ap.parse_args()
    if args.tokens <= 0:
        args.tokens = 125_000_000 if args.gpus == 8 else 100_000_000

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    per_process = world_env > 1 or (os.environ.get("COLIBRI_BENCH_PER_PROCESS") == "1" and "RANK" in os.environ)  # launched by torch.distributed.run: one process per rank
    if per_process and world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} under a launcher with WORLD_SIZE={world_env}")
    sharded = args.gpus > 1 or args.force_shard
    nlocal = 1 if per_process else args.gpus

    # ---- N ranks in this process over RCCL: checked first, in a child process with a deadline (a small corpus, one step) — a multi-GPU RCCL run cannot be rehearsed
    # on the one-GPU boxes this is developed on, and a hang inside a collective cannot be recovered from in-process. If the child fails or does not return, the
    # ranks of this run exchange by peer copies between the devices instead (COLIBRI_NO_RCCL: the trainer's other backend) and the line says so.
    preflight = None
    if sharded and not per_process and args.gpus > 1 and not args.share_gpu and not os.environ.get("COLIBRI_NO_RCCL") and not os.environ.get("COLIBRI_BENCH_NO_PREFLIGHT"):
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", "1", "--warmup", "1", "--tokens", "2000000", "--vocab", "100000", "--cpu-sample", "0"]
        t0 = time.time()
        try:
            child = subprocess.run(cmd, env=dict(os.environ, COLIBRI_BENCH_NO_PREFLIGHT="1"), capture_output=True, text=True, timeout=240)
            ok = child.returncode == 0 and '"metric"' in child.stdout
            why = "" if ok else (child.stderr.strip().splitlines() or ["rc %d" % child.returncode])[-1][:300]
        except subprocess.TimeoutExpired:
            ok, why = False, "no answer within 240 s"
        preflight = {"rccl_ok": ok, "seconds": round(time.time() - t0, 1)}
        if not ok:
            preflight["failure"] = why
            os.environ["COLIBRI_NO_RCCL"] = "1"
            print(f"bench.py: the RCCL rehearsal with {args.gpus} ranks failed ({why}); this run exchanges by peer copies between the devices", file=sys.stderr)

    # ---- synthetic input: generated by worker processes (numpy) before this process touches the GPU ----------------
    my_ranks = [rank] if per_process else list(range(args.gpus))
    specs = [(args.tokens, args.vocab, 44 + r, False) for r in my_ranks]
    want_phrases = not sharded and not args.no_other_configs and args.tokens == 100_000_000 and args.vocab == 1_000_000
    if want_phrases:
        specs.append((args.tokens, args.vocab, 44, True))
    # configs[2]'s corpus (8 x 125 M tokens, the shards an 8-GPU run would hold) on this one device: only where generating it is a matter of seconds
    want_z1b = want_phrases and not args.no_z1b and (os.cpu_count() or 1) >= 16
    if want_z1b:
        specs += [(125_000_000, args.vocab, 44 + r, False) for r in range(8)]
    t0 = time.time()
    pool = concurrent.futures.ProcessPoolExecutor(max_workers=min(len(specs), max(1, (os.cpu_count() or 2) - 1)), mp_context=multiprocessing.get_context("fork"))
    futures = [pool.submit(make_corpus, s) for s in specs]
    payloads = [np.frombuffer(f.result(), dtype=np.uint8) for f in futures[:len(my_ranks)]]
    phrase_payload = futures[len(my_ranks)].result() if want_phrases else None  # (every worker is done before anything is timed: a busy host core next to the timed loop cost 2 ms per step)
    z1b_shards = [np.frombuffer(f.result(), dtype=np.uint8) for f in futures[len(my_ranks) + 1:]] if want_z1b else None
    pool.shutdown(wait=True)
    gen_s = time.time() - t0

    import torch
    from colibri_amd import capi
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU path to measure")
    ndev = torch.cuda.device_count()
    if sharded and not per_process and not args.share_gpu and args.gpus > ndev:
        raise SystemExit(f"--gpus {args.gpus}: only {ndev} device(s) visible (--share-gpu puts all ranks on device 0, exchanging by device copies)")
    if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
        os.environ["NCCL_DEBUG"] = "NONE"  # RCCL's version banner goes to stdout through C stdio and would land beside the JSON line
    dist = None
    if per_process:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world_env)  # host values only (unique id, sentence counts, timing); the data path is the trainer's own RCCL

    def sentences_of(p):
        prev_low = np.concatenate([[True], p[:-1] < 128])
        return int(((p == 0) & prev_low).sum())

    # a multi-rank run that has not finished its warm-up steps after ten minutes is stuck in a collective (a rank died, a link is down): say so and leave,
    # instead of holding the GPUs until somebody's timeout
    import threading
    started = threading.Event()
    if sharded and args.gpus > 1:
        def watchdog():
            if not started.wait(600):
                sys.stderr.write(f"bench.py: rank {rank}: uploads and warm-up not finished after 600 s: giving up\n")
                sys.stderr.flush()
                os._exit(3)
        threading.Thread(target=watchdog, daemon=True).start()

    opt = capi.Options.defaults(mintokens=MINTOKENS, maxlength=MAXLENGTH, profile=2)
    opt_all = capi.Options.defaults(mintokens=MINTOKENS, maxlength=MAXLENGTH, profile=1)
    ctx = tr = None
    t0 = time.time()
    if not sharded:
        torch.cuda.set_device(0)
        ctx = capi.Context(0)
        dev_payload = torch.from_numpy(payloads[0].copy()).cuda()  # H2D outside the timed region
        torch.cuda.synchronize()
        t0 = time.time()
        ctx.upload_device(dev_payload.data_ptr(), payloads[0].size, 1)  # tokenise on device
        step = lambda o=opt: ctx.train(o)
        ktime = ctx.kernel_time
        npos0 = None
    else:
        if per_process:
            uid = [capi.sharded_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            counts = [None] * world_env
            dist.all_gather_object(counts, sentences_of(payloads[0]))
            firsts = [1 + sum(counts[:rank])]
            tr = capi.ShardedTrainer(args.gpus, nlocal=1, first_rank=rank, devices=[0 if args.share_gpu else local_rank], unique_id=uid[0])
        else:
            nsent = [sentences_of(p) for p in payloads]
            firsts = [1 + sum(nsent[:r]) for r in range(args.gpus)]
            tr = capi.ShardedTrainer(args.gpus, devices=[0] * args.gpus if args.share_gpu else None)
        if args.candidates:
            tr.set_protocol(1)
        for lr, (p, first) in enumerate(zip(payloads, firsts)):
            tr.upload(lr, p, first)
        step = lambda o=opt: tr.train(o)
        ktime = lambda k: tr.kernel_time(k, 0)
    tokenise_ms = (time.time() - t0) * 1e3

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        st = step()
    barrier()
    started.set()
    t0 = time.perf_counter()
    kclasses = (capi.K_CLEAR, capi.K_COUNT, capi.K_PRUNE, capi.K_RESOLVE, capi.K_EMIT, capi.K_SCATTER, capi.K_BINCOUNT, capi.K_EMIT2, capi.K_LEVELB2, capi.K_COUNT2, capi.K_LISTS2)
    kms = {k: 0.0 for k in kclasses}
    kn = {k: 0 for k in kclasses}
    for _ in range(args.steps):
        st = step()
        for k in (capi.K_COUNT2, capi.K_BINCOUNT, capi.K_COUNT):  # HIP events on the library's own stream, this step (profile = 2: only the dominant kernel's class has any)
            ms, n = ktime(k)
            kms[k] += ms
            kn[k] += n
    barrier()
    elapsed = time.perf_counter() - t0
    info = tr.info if tr is not None else None
    # untimed: the per-class breakdown of a step (every kernel class bracketed with events)
    EXTRA = 2
    kms_all = {k: 0.0 for k in kclasses}
    kn_all = {k: 0 for k in kclasses}
    for _ in range(EXTRA):
        st_all = step(opt_all)
        for k in kclasses:
            ms, n = ktime(k)
            kms_all[k] += ms
            kn_all[k] += n
    windows = sum(st.windows[1:MAXLENGTH + 1])
    npatterns = int(st.npatterns)
    if dist is not None:
        t = torch.tensor([elapsed, float(windows), float(npatterns)], dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, windows, npatterns = float(tmax[0]), float(t[1]), int(t[2])
    if rank != 0:
        tr.close()
        dist.destroy_process_group()
        return

    value = windows * args.steps / elapsed / 1e6
    nbytes_all = sum(p.size for p in payloads) * (args.gpus if per_process else 1)
    # positions = tokens + sentence delimiters of the WHOLE job (per-process launches: this rank's, times N — the shards are the same size)
    npos_all = sum(int((p < 128).sum()) for p in payloads) * (args.gpus if per_process else 1)
    scan_n, build_n = algorithmic_bytes(nbytes_all, npos_all, st, MAXLENGTH)
    scan_b, build_b = sum(scan_n), sum(build_n)
    binned = kn_all[capi.K_BINCOUNT] > 0 or kn_all[capi.K_COUNT2] > 0
    # Which kernels ran. Global-table path: count_kernel does scan + key + build for every order. Radix path: order 1 is the class-indexed count (class K_COUNT),
    # order 2 the second-generation pipeline (emit2 / levelB2 / count2 / lists2: bigram2.hpp), orders >= 3 emit / scatter / bincount. The dominant kernel (largest
    # total time in the rocprofv3 stats) is then bi2_count_kernel — one launch per step and rank, the table build of order 2 — priced by what it PROCESSES: the
    # 8-byte records it reads (the windows whose two classes are both < 64 are counted in the scan's LDS histogram and never reach it) and the distinct keys it
    # writes; in a multi-GPU run rank 0's launch, which counts the keys rank 0 owns (1 / N of the job's).
    uni = binned and kn_all[capi.K_COUNT] > 0
    second = kn_all[capi.K_COUNT2] > 0
    records = head_windows = None
    if second and ctx is not None:
        records, head_windows = ctx.order2_records()
    if second:
        dom = capi.K_COUNT2
        if records is not None:
            dom_bytes = records * 16.0 + st.found[2] * 12.0
        else:
            dom_bytes = build_n[1] / args.gpus
    elif binned:
        dom, dom_bytes = capi.K_BINCOUNT, (sum(build_n[1:]) if uni else build_b) / args.gpus
    else:
        dom, dom_bytes = capi.K_COUNT, (scan_b + build_b) / args.gpus
    launches_per_step = kn[dom] / max(1, args.steps)
    avg_launch_ms = kms[dom] / max(1, kn[dom])
    achieved = (dom_bytes / max(1.0, launches_per_step)) / (avg_launch_ms * 1e-3) / 1e9 if kn[dom] else 0.0
    stage = (capi.K_COUNT, capi.K_EMIT, capi.K_SCATTER, capi.K_BINCOUNT, capi.K_EMIT2, capi.K_LEVELB2, capi.K_COUNT2) if binned else (capi.K_COUNT,)
    stage_ms = sum(kms_all[k] for k in stage) / EXTRA
    stage_gbs = (scan_b + build_b) / args.gpus / (stage_ms * 1e-3) / 1e9 if stage_ms else 0.0
    kept = [int(st.kept[n]) for n in range(1, MAXLENGTH + 1)]
    # ---- self-check against the real reference's model of the same corpus (one shard of seed 44: N = 1, sharded or not) ----
    fixture = load_fixture("z100m_seed44_plain") if (args.tokens, args.vocab, args.gpus) == (100_000_000, 1_000_000, 1) else None
    check_ok = None
    export_ms = None
    t0 = time.perf_counter()
    arrays = ctx.export_arrays()[:3] if ctx is not None else (tr.export_arrays(0) if (tr is not None and args.gpus == 1) else None)
    if arrays is not None:
        export_ms = (time.perf_counter() - t0) * 1e3
    if fixture is not None:
        check_ok = check_against_reference(fixture, kept, npatterns, arrays)
    del arrays
    # what a caller of an idle context pays for one model: upload (device to device here: the payload is resident) + tokenise + train + the result sizes, as ONE timed call
    # sequence; and the upload + tokenise alone on the warm context (the first upload of a context also allocates ~1 GB of corpus arrays: tokenise_ms_first_upload_untimed)
    tokenise_warm_ms = cold_step_ms = None
    if ctx is not None:
        opt_plain = capi.Options.defaults(mintokens=MINTOKENS, maxlength=MAXLENGTH, profile=0)
        best_u, best_c = 1e9, 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ctx.upload_device(dev_payload.data_ptr(), payloads[0].size, 1)
            t2 = time.perf_counter()
            ctx.train(opt_plain)
            ctx.result_sizes()
            t3 = time.perf_counter()
            best_u, best_c = min(best_u, (t2 - t1) * 1e3), min(best_c, (t3 - t1) * 1e3)
        tokenise_warm_ms, cold_step_ms = best_u, best_c
    traffic, traffic_src = measured_traffic(args.tokens, "bi2_count_kernel" if second else "bin_count_kernel" if binned else "count_kernel") if not sharded else (None, None)
    out = {
        "metric": "M patterns counted/sec at n<=5 thr=2; identical pattern set vs reference",
        "value": round(value, 3),
        "unit": "M patterns counted/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.tokens}-token-per-GPU synthetic Zipf(1.0, V={args.vocab}) class-encoded corpus (.colibri.dat v2), sentences 5..35 tokens, "
                        f"unindexed PatternModel<uint32_t>, MAXLENGTH={MAXLENGTH}, MINTOKENS={MINTOKENS}",
            "tokens_per_gpu": args.tokens,
            "patterns_counted_per_step": int(windows),
            "patterns_in_model": npatterns,
            "kept_per_order": kept,
            "self_check": (("ok" if check_ok else "FAILED") + ": kept per order, pattern total and the multiset digest of (key, count) rows equal the real reference's model of this corpus "
                           "(tests/golden/fullsize/z100m_seed44_plain.json)") if check_ok is not None else "no reference model for this configuration",
            "export_ms_untimed": round(export_ms, 2) if export_ms is not None else None,
            "step_excludes": "the serialised key sizes / key bytes of the model (colibri_result_sizes / colibri_export_*: export_ms_untimed), H2D upload and tokenising",
            "parallelism": "single device" if not sharded else (
                f"sentence-sharded x{args.gpus}, " + ("candidate exchange" if (info is not None and info.protocol == 1) else
                                                      "key-sharded counting: all-reduce of the dense class counts (order 1), records to the owner of their key (orders >= 2)")
                + (", RCCL" if (info is not None and info.rccl) else ", device copies between contexts") + (", one process per rank" if per_process else ", one host thread per rank")),
            "tokenise_ms_untimed": round(tokenise_warm_ms if tokenise_warm_ms is not None else tokenise_ms, 3),
            "tokenise_ms_first_upload_untimed": round(tokenise_ms, 3),
            "cold_step_ms": round(cold_step_ms, 3) if cold_step_ms is not None else None,
            "cold_step_is": "upload (payload resident in HBM) + tokenise + train + result sizes of one model on an idle context, best of 3; tokenise_ms_untimed: the upload + tokenise "
                            "of it; tokenise_ms_first_upload_untimed: a fresh context's first upload, which also allocates the corpus arrays",
            "corpus_generation_s_untimed": round(gen_s, 2),
        },
        "roofline": {
            "kernel": ("colibri::bi2_count_kernel (order 2: one wave per final bin — LDS table build, threshold, survivors, positions; one launch per step and rank)" if second else
                       "colibri::bin_count_kernel (per-bin LDS hash build + threshold + survivor ids; one launch per order >= 2)" if binned else
                       "colibri::count_kernel (scan + key + global hash-table build; one launch per order)"),
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "traffic_source": (f"NOT measured by this run: {traffic_src} (separate rocprofv3 --pmc passes of the same command, committed)" if traffic is not None else None),
            "algorithmic_bytes_per_launch": round(dom_bytes / max(1.0, launches_per_step)),
            "records_per_launch": int(records) if records is not None else None,
            "head_windows_not_records": int(head_windows) if head_windows is not None else None,
            "avg_launch_ms": round(avg_launch_ms, 4),
            "launches_per_step": launches_per_step,
            "counting_stage": {"kernels": [capi.KERNEL_CLASSES[k] for k in stage], "ms_per_step": round(stage_ms, 4),
                               "algorithmic_bytes_per_step": round((scan_b + build_b) / args.gpus), "achieved_GBps": round(stage_gbs, 2),
                               "frac": round(stage_gbs / HBM_PEAK_GBS, 5)},
            "kernel_ms_per_step": {capi.KERNEL_CLASSES[k]: round(kms_all[k] / EXTRA, 4) for k in kclasses if kn_all[k]},
            "note": f"achieved / avg_launch_ms: HIP events around the dominant kernel's launches inside the {args.steps} timed steps (rank 0); counting_stage and "
                    f"kernel_ms_per_step: {EXTRA} extra untimed steps with every kernel class bracketed; per-rank figures price 1 / n_gpus of the job's algorithmic bytes",
        },
    }
    if info is not None:
        sh = {"protocol": "candidate exchange" if info.protocol == 1 else "key-sharded counting", "host_lookups_per_step": int(info.host_lookups),
              "alltoall_bytes_per_rank_and_step": int(info.alltoall_bytes), "of_which_to_self": int(info.alltoall_bytes_to_self), "allreduce_bytes_per_rank_and_step": int(info.allreduce_bytes)}
        if args.gpus == 1:
            # what one rank of the 8-GPU, 1 B-token run (125 M tokens per rank) will take: this step's device work scaled to that shard, plus 7/8 of the exchanged
            # bytes crossing xGMI at the assumed all-to-all rate (stated above; not measurable on a one-GPU box). A lower bound on the bytes: at 10^9 tokens more
            # windows survive, and the survivors' feedback grows (other_configs.z1b_eight_ranks_on_one_device: 1.19 GB per rank and step, 0.17 of it to itself, since round 4)
            ms1 = elapsed / args.steps * 1e3
            scale = 125_000_000 / args.tokens
            xg = scale * info.alltoall_bytes * 7 / 8 / (XGMI_A2A_GBS * 1e9) * 1e3
            sh["predicted_ms_per_rank_at_8"] = round(scale * ms1 + xg, 2)
            sh["prediction"] = (f"{scale:.2f} x this step ({ms1:.2f} ms at {args.tokens} tokens -> 125 M tokens per rank) + 7/8 of {scale:.2f} x {info.alltoall_bytes / 1e9:.2f} GB over xGMI at an "
                                f"assumed {XGMI_A2A_GBS:.0f} GB/s per GPU and direction ({xg:.2f} ms, not overlapped)")
        sh["exchange"] = "RCCL" if info.rccl else "device copies between the ranks' contexts"
        if preflight is not None:
            sh["rccl_rehearsal"] = preflight
        out["sharded"] = sh
    if tr is not None and "sharded" in out and not args.candidates:
        # the same trainer on an INDEXED model (configs[4]: the forward index built on the device; the references stay with the ranks, keyed by global numbers): untimed extra
        try:
            best = None
            for _ in range(3):
                sti = tr.train(capi.Options.defaults(mintokens=MINTOKENS, maxlength=MAXLENGTH, indexed=1))
                best = tr.info.wall_ms if best is None else min(best, tr.info.wall_ms)
            out["sharded"]["indexed_model"] = {"ms_per_step": round(best, 2), "protocol": "candidate exchange" if tr.info.protocol == 1 else "key-sharded counting",
                                               "references_this_process": int(sti.nrefs), "alltoall_bytes_per_rank_and_step": int(tr.info.alltoall_bytes)}
        except Exception as e:  # noqa: BLE001
            out["sharded"]["indexed_model"] = {"error": str(e)}
    if ctx is not None and not args.no_other_configs:
        out["other_configs"], kinds_ok = other_configs(ctx, capi, payloads[0].size, default_corpus=(args.tokens, args.vocab, args.gpus) == (100_000_000, 1_000_000, 1))
        if not kinds_ok:
            check_ok = False
        if want_phrases:
            ph, ph_ok = phrases_config(ctx, capi, phrase_payload)
            out["other_configs"]["phrases"] = ph
            if ph_ok is False:
                check_ok = False
    if want_z1b and "other_configs" in out:
        ctx.close()
        ctx = None
        out["other_configs"].update(z1b_configs(capi, z1b_shards))
    if "other_configs" in out and args.gpus == 1 and not args.force_shard:
        if ctx is not None:
            ctx.close()
            ctx = None
        out["other_configs"]["cxx_face"] = cxx_face(payloads[0])
    if args.gpus == 1 and args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.vocab)
    print(json.dumps(out), flush=True)
    if ctx is not None:
        ctx.close()
    if tr is not None:
        tr.close()
    if dist is not None:
        dist.destroy_process_group()
    if check_ok is False:
        sys.stderr.write("bench.py: the timed run did not produce the reference's model of this corpus: %r / %d patterns\n" % (kept, npatterns))
        sys.exit(3)


if __name__ == "__main__":
    main()
