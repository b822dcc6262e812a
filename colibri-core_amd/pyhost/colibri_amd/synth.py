"""Synthetic class-encoded corpora (.colibri.dat v2) for tests and bench.py.

The distribution is the one BASELINE.md / SURVEY.md §8(d) fix: i.i.d. Zipf(1.0) class ids over V types
(rank r -> class r + 5, so the most frequent word is class 6 — real classes start at 6, reference
src/classencoder.cpp:84,220-224), sentence lengths i.i.d. uniform integer 5..35, encoded as v2
(`A2 02`, little-endian base-128 varints with the high bit on every byte but the last,
reference src/classencoder.cpp:22-42; `00` ends a sentence, :550-600).
"""
import numpy as np

HEADER = bytes([0xA2, 0x02])


def encode_v2(symbols: np.ndarray) -> np.ndarray:
    """Varint-encode an array of class ids (0 = sentence delimiter) into a v2 payload (no header)."""
    sym = np.ascontiguousarray(symbols, dtype=np.uint32)
    nb = np.ones(sym.shape, dtype=np.uint8)
    for k in (7, 14, 21, 28):
        nb += (sym >= (1 << k)).astype(np.uint8)
    off = np.zeros(sym.size + 1, dtype=np.int64)
    np.cumsum(nb, out=off[1:])
    out = np.zeros(int(off[-1]), dtype=np.uint8)
    for k in range(5):
        sel = nb > k
        if not sel.any():
            break
        b = ((sym[sel] >> np.uint32(7 * k)) & np.uint32(127)).astype(np.uint8)
        b |= ((nb[sel] - 1 > k).astype(np.uint8) << 7)
        out[off[:-1][sel] + k] = b
    return out


def sentence_lengths(ntok: int, rng, lo=5, hi=35, scalar_draws=False) -> np.ndarray:
    if scalar_draws:  # one rng.integers() call per sentence: reproduces the survey's generator stream exactly
        lens, tot = [], 0
        while tot < ntok:
            l = min(int(rng.integers(lo, hi + 1)), ntok - tot)
            lens.append(l)
            tot += l
        return np.asarray(lens, dtype=np.int64)
    est = int(ntok / ((lo + hi) / 2) * 1.05) + 16
    lens = rng.integers(lo, hi + 1, size=est).astype(np.int64)
    cs = np.cumsum(lens)
    while cs[-1] < ntok:
        more = rng.integers(lo, hi + 1, size=est).astype(np.int64)
        lens = np.concatenate([lens, more])
        cs = np.cumsum(lens)
    k = int(np.searchsorted(cs, ntok, side="left")) + 1
    lens = lens[:k].copy()
    lens[-1] -= int(cs[k - 1] - ntok)
    return lens


def zipf_tokens(ntok: int, vocab: int, rng) -> np.ndarray:
    ranks = np.arange(1, vocab + 1, dtype=np.float64)
    p = 1.0 / ranks
    p /= p.sum()
    cdf = np.cumsum(p)
    u = rng.random(ntok)
    return np.searchsorted(cdf, u).astype(np.uint32) + np.uint32(6)


def inject_phrases(toks: np.ndarray, rng, nphrases=2000, phrase_len=(3, 8), rate=0.15) -> np.ndarray:
    """Overwrite ~rate of the stream with copies of a fixed phrase inventory so that orders 4-5 do real work
    (pure i.i.d. Zipf text has almost no recurring 4/5-grams)."""
    toks = toks.copy()
    lens = rng.integers(phrase_len[0], phrase_len[1] + 1, size=nphrases)
    starts = np.concatenate([[0], np.cumsum(lens)])
    pool = toks[: int(starts[-1])].copy()
    ninj = int(toks.size * rate / lens.mean())
    which = rng.integers(0, nphrases, size=ninj)
    where = rng.integers(0, max(1, toks.size - phrase_len[1]), size=ninj)
    for w, p in zip(where.tolist(), which.tolist()):
        a, b = int(starts[p]), int(starts[p + 1])
        toks[w: w + (b - a)] = pool[a:b]
    return toks


def zipf_corpus(ntok: int, vocab: int, seed: int, *, scalar_draws=False, phrases=False, header=True) -> bytes:
    """The benchmark corpus family. scalar_draws=True replays the survey generator's RNG stream exactly
    (tokens first, then one integers() call per sentence)."""
    rng = np.random.default_rng(seed)
    toks = zipf_tokens(ntok, vocab, rng)
    lens = sentence_lengths(ntok, rng, scalar_draws=scalar_draws)
    if phrases:
        toks = inject_phrases(toks, rng)
    ends = np.cumsum(lens)
    sym = np.insert(toks, ends, np.uint32(0))  # a delimiter after every sentence
    payload = encode_v2(sym)
    return (HEADER if header else b"") + payload.tobytes()


def random_corpus(rng, nsent=50, maxlen=12, vocab=30, big_classes=True, empty_rate=0.1) -> bytes:
    """Small adversarial corpora for parity tests: empty sentences, sentences shorter than n, 1-4-byte
    tokens, heavy repetition. Returns a v2 payload (no header)."""
    syms = []
    classes = np.arange(6, 6 + vocab, dtype=np.uint64)
    if big_classes:
        extra = np.array([127, 128, 129, 16383, 16384, 16385, 2097151, 2097152, 2097153, 268435455], dtype=np.uint64)
        classes = np.concatenate([classes, extra])
    for _ in range(nsent):
        if rng.random() < empty_rate:
            syms.append(0)
            continue
        n = int(rng.integers(1, maxlen + 1))
        # zipf-ish pick so that n-grams recur
        idx = np.minimum((rng.pareto(1.0, size=n)).astype(np.int64), classes.size - 1)
        syms.extend(int(c) for c in classes[idx])
        syms.append(0)
    return encode_v2(np.asarray(syms, dtype=np.uint32)).tobytes()
