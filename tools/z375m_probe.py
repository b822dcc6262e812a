"""Three of configs[2]'s shards (375 M tokens) in one context, plain model: ms per step, path, and the digest against the reference's model
(tests/golden/fullsize/z375m_seeds44_46_plain.json).   python tools/z375m_probe.py"""
import json
import multiprocessing
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))


def main():
    from colibri_amd import capi, digest, synth
    with ProcessPoolExecutor(3, mp_context=multiprocessing.get_context("spawn")) as pool:
        shards = [np.frombuffer(j.result(), dtype=np.uint8) for j in [pool.submit(synth.zipf_corpus, 125_000_000, 1_000_000, s, header=False) for s in (44, 45, 46)]]
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize", "z375m_seeds44_46_plain.json")))
    with capi.Context(0) as c:
        c.upload(np.concatenate(shards))
        del shards
        best = 1e9
        for _ in range(3):
            st = c.train(maxlength=5, mintokens=2)
            best = min(best, st.train_ms)
        print("z375m plain train ms", round(best, 2), "patterns", st.npatterns, "mode", c.last_mode(with_passes=True), flush=True)
        ko, kb, cn, _ = c.export_arrays()
        d = digest.model_digest(ko, kb, cn)
        print("digest ok:", all(d[k] == fx[k] for k in ("sum1", "xor1", "sum2", "xor2", "npatterns", "occurrences", "keybytes")))


if __name__ == "__main__":
    main()
