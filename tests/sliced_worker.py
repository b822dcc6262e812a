"""Worker of tests/test_gpu_sliced.py: trains corpora with COLIBRI_SLICE_POSITIONS set in the environment (so that the library counts every order of a
small corpus in passes over slices of its keys, the way it treats corpora beyond ~128 M tokens per device) and compares each model with the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from colibri_amd import capi, synth  # noqa: E402


def main():
    assert os.environ.get("COLIBRI_SLICE_POSITIONS") or os.environ.get("COLIBRI_UNI_BIN_CAP") or os.environ.get("COLIBRI_UNI_TWO_PASS") or os.environ.get("COLIBRI_FORCE_WIDE_CHAIN")  # (tests/test_gpu_parity.py: order 1's routes, the wide chained orders)
    rng = np.random.default_rng(5)
    corpora = {"zipf_phrases_300k": synth.zipf_corpus(300_000, 20_000, 7, phrases=True, header=False),
               "zipf_120k_small_vocab": synth.zipf_corpus(120_000, 40, 8, header=False),  # every bigram in the dense head
               "random_long_sentences": synth.random_corpus(rng, nsent=4000, maxlen=40, vocab=300, big_classes=False)}
    runs = ((2, 5), (3, 8), (2, 2))
    if os.environ.get("COLIBRI_SLICED_WORKER_HOT"):  # tests/test_gpu_hot_bins.py: a corpus whose hot keys fill single bins with tens of thousands of records
        from test_gpu_hot_bins import hot_corpus
        corpora, runs = {"hot_keys_2m5": hot_corpus()}, ((2, 5),)
    with capi.Context(0) as ctx:
        for name, payload in corpora.items():
            ctx.upload(payload)
            for thr, maxlength in runs:
                want = oracle.train(payload, thr, maxlength)
                st = ctx.train(mintokens=thr, maxlength=maxlength)
                got, _ = ctx.export_dict()
                assert got == want.counts, (name, thr, maxlength, len(got), len(want.counts))
                assert (st.totaltokens, st.totaltypes, st.maxn) == (want.tokens, want.types, want.maxn), (name, thr, maxlength)
                for n in range(1, maxlength + 1):
                    assert (st.found[n], st.kept[n]) == (want.stats[n][0], want.stats[n][2]), (name, thr, maxlength, n)
                assert ctx.last_mode() == 2, ("the radix path must have run (not the global-table fallback)", name, thr, maxlength, ctx.last_mode(True))
            if os.environ.get("COLIBRI_FORCE_WIDE_CHAIN"):  # the wide chained orders serve indexed models too: every reference list
                want = oracle.train(payload, 2, 5, indexed=True)
                ctx.train(mintokens=2, maxlength=5, indexed=1)
                got, refs = ctx.export_dict()
                assert got == want.counts and refs == want.refs, (name, "indexed")
    print("SLICED_OK")


if __name__ == "__main__":
    main()
