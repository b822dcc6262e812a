"""pytest configuration: the `gpu` marker, import paths and shared corpora.

`-m "not gpu"` runs here (no GPU): oracle vs golden fixtures, host logic, C-ABI symbol checks.
`-m gpu` runs on an MI355X: parity of the HIP path (through the C ABI) against the oracle.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- the big synthetic corpora of the -m gpu suite, generated once per session (and side by side when a test asks for several) ------------------------------------
# Round 5's suite took 968 s of the driver's 1200: ~25 of its corpora of 2 x 10^7 .. 10^8 tokens were drawn again by every test that used them (10^8 tokens: ~30 s of one
# host core each). The payloads are kept as read-only uint8 arrays (~2 bytes per token).
_CORPORA = {}


def _draw(spec):
    from colibri_amd import synth
    ntok, vocab, seed, phrases = spec
    return synth.zipf_corpus(ntok, vocab, seed, phrases=phrases, header=False)


def zipf_many(specs):
    """[(ntok, vocab, seed, phrases), ...] -> list of uint8 arrays (v2 payloads without header); the missing ones are drawn in parallel processes"""
    specs = [tuple(s) + (False,) * (4 - len(s)) for s in specs]
    missing = [s for s in dict.fromkeys(specs) if s not in _CORPORA]
    if len(missing) > 1 and sum(s[0] for s in missing) >= 20_000_000:
        import multiprocessing
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(min(8, len(missing)), mp_context=multiprocessing.get_context("spawn")) as pool:  # (not fork: this process may hold a HIP context)
            for s, b in zip(missing, pool.map(_draw, missing)):
                _CORPORA[s] = np.frombuffer(b, dtype=np.uint8)
    else:
        for s in missing:
            _CORPORA[s] = np.frombuffer(_draw(s), dtype=np.uint8)
    return [_CORPORA[s] for s in specs]


def zipf_cached(ntok, vocab, seed, phrases=False):
    return zipf_many([(ntok, vocab, seed, phrases)])[0]


@pytest.fixture(scope="session")
def hamlet_payload():
    """exp/hamlet.v1.colibri.dat of the reference (a data fixture, copied to tests/golden/) converted v1 -> v2."""
    import oracle
    with open(os.path.join(GOLDEN, "hamlet.v1.colibri.dat"), "rb") as f:
        return oracle.v1_to_v2(f.read())


def small_corpora():
    """Named v2 payloads (no header) covering the edge cases the reference's tests and SURVEY §8c list."""
    from colibri_amd import synth
    rng = np.random.default_rng(20240917)
    out = {}
    for i in range(4):
        out[f"rand{i}"] = synth.random_corpus(rng, nsent=150 + 50 * i, maxlen=10 + 2 * i, vocab=8 + 4 * i)
    out["rand_noempty"] = synth.random_corpus(rng, nsent=300, maxlen=9, vocab=10, empty_rate=0.0, big_classes=False)
    out["empty"] = b""
    out["only_delims"] = b"\x00\x00\x00"
    out["one_token"] = b"\x06\x00"
    out["no_trailing_delim"] = b"\x06\x07\x06\x07\x00\x06\x07\x06\x07"
    out["short_sentences"] = b"\x06\x00\x06\x07\x00\x06\x07\x08\x00\x06\x00\x06\x07\x00\x06\x07\x08\x00"
    out["repeat"] = (b"\x06\x07\x08\x09\x0a\x0b\x0c\x00") * 5
    out["one_long_sentence"] = bytes([6, 7, 8, 9] * 300) + b"\x00"
    out["multibyte"] = synth.encode_v2(np.array([200, 300, 20000, 200, 300, 20000, 0, 3000000, 200, 300, 0, 300000000, 300000000, 0], dtype=np.uint32)).tobytes()
    # class-space boundaries of the default mode's kernel choices: three classes in one key needs maxclass < 2^21 (KeyBigramCls / KeyTrigramCls),
    # the partitioned order 1 uses 2^12 / 2^13 / 2^14-class ranges up to 2^22 classes and the atomics kernel beyond
    for tag, top in (("cls_2p21m1", (1 << 21) - 1), ("cls_2p21", 1 << 21), ("cls_2p20", 1 << 20), ("cls_2p22m1", (1 << 22) - 1), ("cls_2p22", 1 << 22)):
        pool = np.array([6, 7, 8, 9, 100, 5000, top - 3, top - 1, top], dtype=np.uint32)
        toks = pool[np.minimum(rng.pareto(0.9, size=1500).astype(np.int64), pool.size - 1)]
        lens = rng.integers(1, 12, size=400)
        ends = np.cumsum(lens)
        ends = ends[ends < toks.size]
        out[tag] = synth.encode_v2(np.insert(toks, ends, np.uint32(0))).tobytes() + b"\x00"
    out["zipf20k"] = synth.zipf_corpus(20000, 500, 5, header=False)
    out["zipf200k_phrases"] = synth.zipf_corpus(200000, 5000, 7, phrases=True, header=False)
    return out
