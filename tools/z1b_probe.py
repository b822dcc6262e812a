"""configs[2]'s 1 B-token corpus (8 x 125 M tokens, seeds 44..51) in ONE context on one device: train() wall time and the kernel classes of a profiled step
(COLIBRI_RESCAN_SLICES=1 selects the round-2 form that re-scans the corpus per key slice):
    python tools/z1b_probe.py [shards]"""
import concurrent.futures
import multiprocessing
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "colibri-core_amd", "pyhost"))
from colibri_amd import synth  # noqa: E402


def make(seed):
    return synth.zipf_corpus(125_000_000, 1_000_000, seed, header=False)


def main():
    nshards = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    pool = concurrent.futures.ProcessPoolExecutor(max_workers=nshards, mp_context=multiprocessing.get_context("fork"))
    shards = [np.frombuffer(f.result(), dtype=np.uint8) for f in [pool.submit(make, 44 + r) for r in range(nshards)]]
    pool.shutdown(wait=True)
    from colibri_amd import capi
    whole = np.concatenate(shards)
    del shards
    with capi.Context(0) as c:
        c.upload(whole)
        del whole
        ms = []
        for _ in range(3):
            st = c.train(maxlength=5, mintokens=2)
            ms.append(round(st.train_ms, 2))
        c.train(maxlength=5, mintokens=2, profile=1)
        kms = {capi.KERNEL_CLASSES[k]: (round(c.kernel_time(k)[0], 2), c.kernel_time(k)[1]) for k in range(len(capi.KERNEL_CLASSES)) if c.kernel_time(k)[1]}
        print("tokens", nshards * 125_000_000, "train ms", ms, "patterns", int(st.npatterns), "mode/passes", c.last_mode(with_passes=True))
        print("kernel classes (ms, launches):", kms, "sum", round(sum(v[0] for v in kms.values()), 2), flush=True)


if __name__ == "__main__":
    main()
