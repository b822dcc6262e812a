"""GPU: hot keys. A final bin of the radix path is big because of ONE key — a frequent bigram outside the dense 64 x 64 head of order 2, a frequent trigram —, and
those bins take their own code: order 2 lists bins beyond 1536 records (counted first, one wave each) and bins beyond 16 384 records go to a WORKGROUP
(bi2_count_big_kernel, csrc/bigram2.hpp: plain and id-keeping position lists, and the chunk-pool lists of key-sharded owners and of sliced runs); the count kernel of orders
>= 3 streams what exceeds its register window eight rows at a time and hands its bins out from queues when one of them outweighs a block's share (csrc/binned.hpp).
At 10^8 tokens of Zipf text no bigram bin reaches 16 384 records and at 10^9 the model can only be compared with another run of the same kernels, so these corpora
are built to have such bins at a size the oracle handles: 10^7 tokens of Zipf text with tens of thousands of copies of a 4-gram and of a bigram of rare words (class
ids far beyond 63) written over it — few enough for the emit kernel's record regions (a hot key beyond ~0.5 % of the windows overflows a region and the run repeats on the
first-generation kernels: loud, exact, and not what these tests are after; they check which kernels ran). Reference semantics as everywhere: include/patternmodel.h:880-1345 through the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def hot_corpus(ntok=10_000_000, seed=9, n4=40_000, n2=24_000, vocab=200_000):
    from colibri_amd import synth
    rng = np.random.default_rng(seed)
    toks = synth.zipf_tokens(ntok, vocab, rng)
    lens = synth.sentence_lengths(ntok, rng)
    for hot, n in ((np.array([5000, 5001, 5002, 5003], dtype=np.uint32), n4), (np.array([7000, 7001], dtype=np.uint32), n2)):
        where = rng.integers(0, ntok - 8, size=n)
        toks[where[:, None] + np.arange(hot.size)[None, :]] = hot[None, :]  # (later copies overwrite earlier ones: broken copies are part of the text)
    sym = np.insert(toks, np.cumsum(lens), np.uint32(0))
    return synth.encode_v2(sym).tobytes()


@pytest.fixture(scope="module")
def corpus_and_models():
    import oracle
    payload = hot_corpus()
    return payload, {False: oracle.train(payload, 2, 5), True: oracle.train(payload, 2, 5, indexed=True)}


def _figures(st, want, maxlength=5):
    assert (st.totaltokens, st.totaltypes, st.maxn, st.npatterns) == (want.tokens, want.types, want.maxn, len(want.counts))
    for n in range(1, maxlength + 1):
        assert (st.found[n], st.kept[n]) == (want.stats[n][0], want.stats[n][2]), n


def test_the_corpus_has_huge_bins(corpus_and_models):
    """the premise: a bigram of words outside the dense head with more than 16 384 occurrences (one final bin of order 2 holds them all)"""
    from colibri_amd import synth
    _, models = corpus_and_models
    hot = bytes(synth.encode_v2(np.array([5000, 5001], dtype=np.uint32)))
    assert models[False].counts[hot] > 16384 + 8192
    assert models[False].counts[bytes(synth.encode_v2(np.array([7000, 7001], dtype=np.uint32)))] > 16384


@pytest.mark.parametrize("indexed", [False, True], ids=["plain", "indexed"])
def test_single_device(corpus_and_models, indexed):
    """plain: the wave lists of the workgroup kernel are continued by the wave kernel; indexed: every listed position also carries its (bin, rank) code"""
    from colibri_amd import capi
    payload, models = corpus_and_models
    want = models[indexed]
    with capi.Context(0) as ctx:
        ctx.upload(payload)
        st = ctx.train(mintokens=2, maxlength=5, indexed=int(indexed), profile=1)
        assert ctx.last_mode() == 2, "the radix path must have run"
        # order 2 on the second-generation kernels, and not repeated on the first-generation ones (which count order 2 in the class of orders >= 3)
        assert ctx.kernel_time(capi.K_COUNT2)[1] >= 1 and ctx.kernel_time(capi.K_BINCOUNT)[1] == 3, (ctx.kernel_time(capi.K_COUNT2), ctx.kernel_time(capi.K_BINCOUNT))
        got, refs = ctx.export_dict()
    assert got == want.counts
    if indexed:
        assert refs == want.refs
    _figures(st, want)


def test_exhaustive_skipgrams(corpus_and_models):
    import oracle
    from colibri_amd import capi
    payload, _ = corpus_and_models
    want = oracle.train(payload, 2, 4, doskipgrams_exhaustive=True)
    with capi.Context(0) as ctx:
        ctx.upload(payload)
        st = ctx.train(mintokens=2, maxlength=4, doskipgrams_exhaustive=1)
        got, _ = ctx.export_dict()
    assert got == want.counts
    assert st.npatterns == len(want.counts)


@pytest.mark.parametrize("world", [1, 2, 8])
def test_key_sharded_owners(corpus_and_models, world):
    """the owner of the hot bigram's key receives all of its records: the chunk-pool form of the workgroup kernel"""
    from colibri_amd import capi
    payload, models = corpus_and_models
    with capi.ShardedTrainer(world, devices=[0] * world if world > 1 else None) as tr:
        tr.upload_split(payload)
        st = tr.train(mintokens=2, maxlength=5)
        assert tr.info.protocol == 0, "the run did not take the key-sharded path"
        got = tr.export_dict()
    assert got == models[False].counts
    _figures(st, models[False])


@pytest.mark.parametrize("slice_positions", ["1600000", "5200000"])
def test_sliced_runs(slice_positions):
    """the same corpus counted in 8 / 2 key slices (records split once per order, csrc/colibri_hip.hip: bigram2_order_split / binned_order_split)"""
    env = dict(os.environ, COLIBRI_SLICE_POSITIONS=slice_positions, COLIBRI_SLICED_WORKER_HOT="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sliced_worker.py")], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0 and "SLICED_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


@pytest.mark.parametrize("walk", ["1", "2"], ids=["fixed-shares", "queues"])
def test_both_bin_walks_of_the_count_kernel(walk):
    """orders >= 3: the count kernel walks its bins in fixed shares or takes them from queues — chosen per launch by the largest bin; here each is forced
    (COLIBRI_BIN_WALK, read once per process) on the small corpora of tests/sliced_worker.py, unsliced, against the oracle"""
    env = dict(os.environ, COLIBRI_SLICE_POSITIONS="1000000000", COLIBRI_BIN_WALK=walk)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sliced_worker.py")], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0 and "SLICED_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
