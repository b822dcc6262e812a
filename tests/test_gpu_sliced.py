"""The radix path beyond ~128 M tokens per device: an order is counted in passes over slices of its keys (bigram2_order / binned_order in
colibri_hip.hip), survivors appended pass after pass, positions / ids resolved once. The reference has one code path for any size
(include/patternmodel.h:880-1345); here the sliced passes must give the same model as the single pass.
  * small corpora with the slice size turned down through COLIBRI_SLICE_POSITIONS (a subprocess: the library reads it once): against the oracle;
  * 300 M tokens: the sliced radix path against the global-table path as multisets of (key bytes, count) rows (two independent row hashes)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, zipf_many

pytestmark = pytest.mark.gpu


# (up to 8 slices order 2 scans the corpus once and cuts its records into the slices — bigram2_order_split —, beyond that, or with COLIBRI_RESCAN_SLICES, every
# slice re-scans: 30000 / 100000 positions per slice give 8 / 4 / 2 slices on these corpora, 7000 gives 64)
# The split itself has two forms: direct (a sweep into runs of fixed room; the default) and exact (histogram, scan, move: what a run falls back to when a run
# outgrows its room — which the small-vocabulary corpus here does on its own); COLIBRI_SPLIT_EXACT forces the second.
@pytest.mark.parametrize("slice_positions,rescan", [("30000", ""), ("100000", ""), ("7000", ""), ("30000", "1"), ("30000", "exact")])
def test_sliced_passes_match_the_oracle(slice_positions, rescan):
    env = dict(os.environ, COLIBRI_SLICE_POSITIONS=slice_positions)
    if rescan == "exact":
        env["COLIBRI_SPLIT_EXACT"] = "1"
    elif rescan:
        env["COLIBRI_RESCAN_SLICES"] = rescan
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sliced_worker.py")], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0 and "SLICED_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_sliced_radix_path_agrees_with_the_table_at_300m_tokens():
    from colibri_amd import capi, synth
    from test_gpu_fullsize import row_hashes, summary
    payload = np.concatenate(zipf_many([(100_000_000, 1_000_000, 300 + k) for k in range(3)]))
    got = {}
    with capi.Context(0) as ctx:
        ctx.upload(payload)
        for mode in (0, 1):
            st = ctx.train(mintokens=2, maxlength=5, table_mode=mode)
            assert ctx.last_mode() == (2 if mode == 0 else 1)
            key_off, key_bytes, counts, _ = ctx.export_arrays()
            got[mode] = (summary(st), row_hashes(key_off, key_bytes, counts))
    assert got[0][0] == got[1][0]
    assert got[0][0][0] == 300_000_000
    for x, y in zip(got[0][1], got[1][1]):
        assert np.array_equal(np.sort(x), np.sort(y))


@pytest.mark.parametrize("kw", [dict(indexed=1), dict(doskipgrams_exhaustive=1), dict(indexed=1, doskipgrams=1, minskiptypes=2)], ids=["indexed", "exhaustive_skipgrams", "indexed_skipgrams_T2"])
def test_id_keeping_kinds_beyond_128m_tokens_stay_on_the_radix_path(kw):
    """Rounds 1-3: an indexed or skipgram model of more than 128 M tokens per device fell to the global table (2.6 x the per-token cost). Since round 4 one pass of the
    second-generation engine holds 2 x 10^8 positions, and 4 x 10^8 with the count kernels' 2048-slot bin tables: at 300 M tokens the id-keeping kinds run it
    (last_mode 2, one pass) and must give the table path's model, reference lists included.
    The reference has one code path at any size (include/patternmodel.h:880-1345, :2789-2800, :2969-3010)."""
    from colibri_amd import capi, digest, synth
    payload = np.concatenate(zipf_many([(100_000_000, 1_000_000, 300 + k) for k in range(3)]))
    got = {}
    with capi.Context(0) as ctx:
        ctx.upload(payload)
        del payload
        for mode in (0, 1):
            st = ctx.train(mintokens=2, maxlength=5, table_mode=mode, **kw)
            assert ctx.last_mode(with_passes=True) == ((2, 1) if mode == 0 else (1, 1))
            key_off, key_bytes, counts, refs = ctx.export_arrays()
            got[mode] = ((st.totaltokens, st.npatterns, st.nrefs, [st.found[n] for n in range(1, 6)], [st.kept[n] for n in range(1, 6)]),
                         digest.model_digest(key_off, key_bytes, counts, refs))
            del key_off, key_bytes, counts, refs
    assert got[0][0][0] == 300_000_000
    assert got[0] == got[1]


def test_plain_run_of_200m_tokens_is_one_chained_pass():
    """... and the plain run: up to 2 x 10^8 positions one pass of every order on the chained engine (round 3: two key slices from 110 M positions on), the same model as
    the table path's."""
    from colibri_amd import capi, synth
    from test_gpu_fullsize import row_hashes, summary
    payload = np.concatenate(zipf_many([(100_000_000, 1_000_000, 300 + k) for k in range(3)])[:2])
    got = {}
    with capi.Context(0) as ctx:
        ctx.upload(payload)
        for mode in (0, 1):
            st = ctx.train(mintokens=2, maxlength=5, table_mode=mode)
            assert ctx.last_mode(with_passes=True) == ((2, 1) if mode == 0 else (1, 1))
            key_off, key_bytes, counts, _ = ctx.export_arrays()
            got[mode] = (summary(st), row_hashes(key_off, key_bytes, counts))
    assert got[0][0] == got[1][0] and got[0][0][0] == 200_000_000
    for x, y in zip(got[0][1], got[1][1]):
        assert np.array_equal(np.sort(x), np.sort(y))


@pytest.mark.parametrize("kw,path", [({}, (2, 2)), (dict(indexed=1), (1, 1))], ids=["plain", "indexed"])
def test_a_pass_whose_bins_overflow_repeats_with_smaller_passes(kw, path):
    """The single-pass limit (2.15 x 10^8 positions) is what the bench distribution fills the count kernels' bin tables with; a corpus with more distinct keys per window
    overflows them earlier: 140 M uniformly drawn tokens over 10^6 types put ~1000 distinct bigrams into every final bin (the tables hold 900). The run must notice
    (Bi2State.overflow 2), repeat with round 3's pass size — two key slices; an indexed model, which has no sliced form, on the global table — and give the table path's model."""
    from colibri_amd import capi, synth
    from test_gpu_fullsize import row_hashes, summary
    T = 140_000_000
    toks = np.random.default_rng(5).integers(6, 6 + 1_000_000, size=T, dtype=np.uint32)
    payload = synth.encode_v2(np.append(np.insert(toks, np.arange(20, T, 20), np.uint32(0)), np.uint32(0)))
    del toks
    got = {}
    with capi.Context(0) as ctx:
        ctx.upload(payload)
        for mode in (0, 1):
            st = ctx.train(mintokens=2, maxlength=5, table_mode=mode, **kw)
            assert ctx.last_mode(with_passes=True) == (path if mode == 0 else (1, 1))
            if mode == 0:  # (ABI 4: the repeat is reported, with what made the first attempt give up — the second-generation order 2, whose final bins overflowed)
                from colibri_amd import capi as _capi
                assert st.retries >= 1 and st.fallback_reason == _capi.FALLBACK_ORDER2
                assert st.path & (_capi.PATH_SLICED if not kw else _capi.PATH_TABLE)
            key_off, key_bytes, counts, _ = ctx.export_arrays()
            got[mode] = (summary(st) + (st.nrefs,), row_hashes(key_off, key_bytes, counts))
    assert got[0][0] == got[1][0] and got[0][0][0] == T
    for x, y in zip(got[0][1], got[1][1]):
        assert np.array_equal(np.sort(x), np.sort(y))
