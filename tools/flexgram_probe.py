"""Throughput of flexgrams-from-skipgrams on one MI355X, the reference's computeflexgrams_fromskipgrams timed beside
it on a sample. The model is the indexed + skipgram model (MINSKIPTYPES as given) of a Zipf corpus with injected phrases, trained on the
device; colibri_flexgrams then takes its export arrays from host memory (that is the boundary: PCIe-inclusive) — the device-only share
comes from the HIP events of the library's kernel classes. Prints one JSON object; numbers go into DESIGN.md."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=100_000_000)
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--cpu-tokens", type=int, default=4_000_000)
    ap.add_argument("--minskiptypes", type=int, default=2)
    a = ap.parse_args()
    from colibri_amd import capi, synth
    import oracle
    payload = np.frombuffer(synth.zipf_corpus(a.tokens, a.vocab, 51, phrases=True, header=False), dtype=np.uint8)
    res = {"tokens": a.tokens, "vocab": a.vocab, "minskiptypes": a.minskiptypes}
    with capi.Context(0) as ctx:
        ctx.upload(payload)
        t0 = time.perf_counter()
        ctx.train(mintokens=2, maxlength=5, indexed=1, doskipgrams=1, minskiptypes=a.minskiptypes, profile=1)  # cold: allocates
        t0 = time.perf_counter()
        st = ctx.train(mintokens=2, maxlength=5, indexed=1, doskipgrams=1, minskiptypes=a.minskiptypes, profile=1)
        res["train_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        key_off, key_bytes, counts, (ref_off, rs, rt) = ctx.export_arrays()
        res.update(patterns=int(counts.size), references=int(rs.size), model_bytes_in=int(key_bytes.size + 8 * key_off.size + 8 * ref_off.size + 6 * rs.size))
        def timed(fn):
            best = None
            for rep in range(3):
                k0 = [ctx.kernel_time(k)[0] for k in (capi.K_SKIPGRAM, capi.K_INDEX, capi.K_EXPORT)]
                t0 = time.perf_counter()
                out = fn()
                wall = time.perf_counter() - t0
                k1 = [ctx.kernel_time(k)[0] for k in (capi.K_SKIPGRAM, capi.K_INDEX, capi.K_EXPORT)]
                # kernel_time accumulates since the last train(): take the difference
                cur = {"wall_ms": round(wall * 1e3, 2), "kernels_ms": {"group": round(k1[0] - k0[0], 3), "merge_sort": round(k1[1] - k0[1], 3), "key_bytes": round(k1[2] - k0[2], 3)}}
                if best is None or cur["wall_ms"] < best["wall_ms"]:
                    best = cur
            return out, best
        (fo, fk, fc, (fro, frs, frt)), host_in = timed(lambda: ctx.flexgrams(key_off, key_bytes, ref_off, rs, rt))
        (_, _, fc2, (_, frs2, _)), resident = timed(ctx.flexgrams_resident)
        assert fc2.size == fc.size and frs2.size == frs.size
        nskiprefs = int(frs.size)
        res.update(flexgrams=int(fc.size), flexgram_references=nskiprefs, model_from_host_memory=host_in, model_resident_in_hbm=resident,
                   model_refs_per_s_resident=round(rs.size / (resident["wall_ms"] * 1e-3)), model_refs_per_s_from_host=round(rs.size / (host_in["wall_ms"] * 1e-3)))
    if a.cpu_tokens and oracle.have_ref():
        sample = synth.zipf_corpus(a.cpu_tokens, a.vocab, 52, phrases=True, header=True)
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "s.colibri.dat")
            with open(path, "wb") as f:
                f.write(sample)
            out = subprocess.run([oracle.REF_DRIVER, "train", path, "is", "5", "2", "-q", "-F", "-T", str(a.minskiptypes)], capture_output=True, text=True)
            line = [ln for ln in out.stderr.splitlines() if ln.startswith("flexgram_timing")]
            if line:
                t = json.loads(line[0][len("flexgram_timing "):])
                res["cpu_reference"] = dict(t, tokens=a.cpu_tokens, cores=1, skipgram_refs_per_s=round(t["skipgram_refs"] / max(t["seconds"], 1e-9)),
                                            patterns_per_s=round(t["patterns_before"] / max(t["seconds"], 1e-9)),
                                            note="reference computeflexgrams_fromskipgrams alone (its loop may skip / revisit skipgrams when the map rehashes)")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
