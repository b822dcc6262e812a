"""configs[2]'s corpus as the multi-GPU trainer runs it — eight ranks of 125 M tokens — with all ranks on THIS device (device copies instead of xGMI):
wall time per train(); under `rocprofv3 --kernel-trace` tools/busy.py tells how much of the step the device was busy.    python tools/z1b_ranks_probe.py [ranks] [tokens per rank]"""
import concurrent.futures, multiprocessing, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "colibri-core_amd", "pyhost"))
from colibri_amd import synth  # noqa: E402
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 125_000_000
def make(seed):
    return synth.zipf_corpus(NT, 1_000_000, seed, header=False)
def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    pool = concurrent.futures.ProcessPoolExecutor(max_workers=n, mp_context=multiprocessing.get_context("fork"))
    shards = [np.frombuffer(f.result(), dtype=np.uint8) for f in [pool.submit(make, 44 + r) for r in range(n)]]
    pool.shutdown(wait=True)
    from colibri_amd import capi
    nsent = [int(((p == 0) & np.concatenate([[True], p[:-1] < 128])).sum()) for p in shards]
    with capi.ShardedTrainer(n, devices=[0] * n) as tr:
        for r, p in enumerate(shards):
            tr.upload(r, p, 1 + sum(nsent[:r]))
        for k in range(4):
            st = tr.train(maxlength=5, mintokens=2)
            print("train", k, "wall ms %.2f" % tr.info.wall_ms, "patterns", int(st.npatterns), "protocol", tr.info.protocol, "a2a bytes/rank", int(tr.info.alltoall_bytes), flush=True)
if __name__ == "__main__":
    main()
