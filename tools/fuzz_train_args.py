"""Randomised parity sweep of the round-2 arguments of train() on the GPU box: continued training, filtered training, skipgrams in constrained runs — the HIP path
through the C ABI against the Python restatements (oracle.train_continued / train_filtered / train_constrained, each pinned to dumps of the real reference in
tests/test_oracle.py). Not a test; a tool for spare GPU minutes:
    python tools/fuzz_train_args.py --seconds 300 --seed 1 > gpurun_out/fuzz_train_args.json"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def small_corpus(rng):
    from colibri_amd import synth
    kind = int(rng.integers(0, 4))
    if kind == 0:
        return synth.random_corpus(rng, nsent=int(rng.integers(1, 250)), maxlen=int(rng.integers(1, 14)), vocab=int(rng.integers(1, 30)), big_classes=bool(rng.integers(0, 2)),
                                   empty_rate=float(rng.choice([0.0, 0.1, 0.4])))
    if kind == 1:
        return synth.zipf_corpus(int(rng.integers(100, 12000)), int(rng.integers(2, 800)), int(rng.integers(0, 1 << 30)), header=False)
    if kind == 2:
        toks = rng.integers(6, 6 + int(rng.integers(1, 4)), size=int(rng.integers(1, 1500))).astype(np.uint32)
        cut = np.sort(rng.choice(toks.size + 1, size=min(toks.size, int(rng.integers(0, 20))), replace=False))
        return synth.encode_v2(np.insert(toks, cut, np.uint32(0))).tobytes() + (b"\x00" if rng.integers(0, 2) else b"")
    s = rng.integers(6, 12, size=int(rng.integers(1, 12))).astype(np.uint32)
    return (synth.encode_v2(s).tobytes() + b"\x00") * int(rng.integers(1, 30))


def masked_forms(rng, oracle, key, howmany):
    t = oracle.key_tokens(key)
    out = []
    if len(t) >= 3:
        masks = oracle.skip_configurations(len(t), 3)
        for mask in rng.choice(masks, size=min(len(masks), howmany), replace=False):
            out.append(b"".join(b"\x03" if (int(mask) >> j) & 1 else t[j] for j in range(len(t))))
    return out


def windows(rng, oracle, payload, count, maxn):
    sents = [t for t in oracle._sentences(payload) if t]
    out = []
    for _ in range(count):
        if not sents:
            break
        t = sents[int(rng.integers(0, len(sents)))]
        n = int(rng.integers(1, min(len(t), maxn) + 1))
        i = int(rng.integers(0, len(t) - n + 1))
        out.append(b"".join(t[i:i + n]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import oracle
    from colibri_amd import capi
    ctx = capi.Context(0)
    t0, case, failures, by_mode = time.time(), 0, [], {}
    while time.time() - t0 < args.seconds:
        seed = args.seed * 1000003 + case
        rng = np.random.default_rng(seed)
        case += 1
        payload = small_corpus(rng)
        mode = ["continued", "filtered", "constrained_skipgrams"][int(rng.integers(0, 3))]
        indexed = bool(rng.integers(0, 2))
        info = {"seed": seed, "mode": mode, "bytes": len(payload), "indexed": indexed}
        try:
            if mode == "continued":
                source = small_corpus(rng) if rng.integers(0, 4) == 0 else payload
                loaded = oracle.train(source, int(rng.choice([2, 3])), int(rng.choice([1, 2, 3])), indexed=indexed)
                if rng.integers(0, 4) == 0 and loaded.maxn >= 2:
                    drop = int(rng.integers(1, loaded.maxn + 1))
                    keep = {k for k in loaded.counts if oracle.key_ntokens(k) != drop}
                    loaded = oracle.Model(loaded.tokens, loaded.types, {k: loaded.counts[k] for k in keep}, {k: loaded.refs[k] for k in keep} if indexed else None)
                if not loaded.counts:
                    continue
                mt, ml = int(rng.choice([2, 2, 3, 4])), int(rng.choice([2, 3, 5, 8]))
                want = oracle.train_continued(payload, loaded, mt, ml, indexed=indexed)
                new = {k: v for k, v in want.counts.items() if k not in loaded.counts}
                ctx.upload(payload)
                try:
                    ctx.set_continuation(sorted(loaded.counts))
                    ctx.train(mintokens=mt, maxlength=ml, indexed=int(indexed))
                    got, refs = ctx.export_dict()
                finally:
                    ctx.set_continuation([])
                ok = got == new and (not indexed or refs == {k: want.refs[k] for k in new})
            elif mode == "filtered":
                keys = set()
                kinds = int(rng.integers(0, 3))
                for w in windows(rng, oracle, payload, int(rng.integers(1, 7)), 4):
                    if kinds != 1:
                        keys.add(w)
                    if kinds != 0:
                        keys.update(masked_forms(rng, oracle, w, 1))
                keys.add(b"\x7e\x7d")
                if rng.integers(0, 4) == 0:
                    keys.add(b"\x06\x04\x07")
                mt, ml = int(rng.choice([1, 2, 2, 3])), int(rng.choice([2, 4, 6]))
                want = oracle.train_filtered(payload, sorted(keys), mt, ml, indexed=indexed)
                ctx.upload(payload)
                try:
                    ctx.set_filter(sorted(keys))
                    st = ctx.train(mintokens=mt, maxlength=ml, indexed=int(indexed))
                    got, refs = ctx.export_dict()
                finally:
                    ctx.set_filter([])
                ok = got == want.counts and (not indexed or refs == want.refs) and (int(st.totaltokens), int(st.totaltypes)) == (want.tokens, want.types)
            else:
                ml = int(rng.choice([3, 4, 5, 7]))
                keys = set(windows(rng, oracle, payload, int(rng.integers(1, 300)), ml + 1))
                for k in list(keys):
                    if rng.random() < 0.6:
                        keys.update(masked_forms(rng, oracle, k, 2))
                keys.add(b"\x06\x03\x06")
                mt = 1 if rng.integers(0, 4) else 2
                y, T = int(rng.choice([-1, 2, 3])), int(rng.choice([1, 2]))
                want = oracle.train_constrained(payload, sorted(keys), mt, ml, 1, indexed=indexed, doskipgrams=True, mintokens_skipgrams=y, minskiptypes=T)
                ctx.upload(payload)
                try:
                    ctx.set_constraint(sorted(keys))
                    ctx.train(mintokens=mt, maxlength=ml, indexed=int(indexed), doskipgrams_exhaustive=1, mintokens_skipgrams=y, minskiptypes=T, table_mode=int(rng.choice([0, 0, 1])))
                    got, refs = ctx.export_dict()
                finally:
                    ctx.set_constraint([])
                ok = got == want.counts and (not indexed or refs == want.refs)
            if not ok:
                failures.append(info)
        except Exception as e:  # noqa: BLE001
            failures.append(dict(info, error=repr(e)[:300]))
        by_mode[mode] = by_mode.get(mode, 0) + 1
    print(json.dumps({"cases": case, "seconds": round(time.time() - t0, 1), "by_mode": by_mode, "failures": failures[:20], "nfailures": len(failures)}, indent=1))


if __name__ == "__main__":
    main()
