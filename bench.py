#!/usr/bin/env python
"""bench.py — the driver's benchmark contract for the hot path (PatternModel::train, n <= 5, thr = 2).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one complete train() (all five orders: scan + SpookyHash + hash build + prune + resolve) over one
synthetic class-encoded corpus that is already resident in HBM when the timed region starts.
  N = 1 : BASELINE.json configs[1] — 100M-token Zipf(1.0, V = 1e6) corpus, unindexed, n <= 5, threshold 2.
  N > 1 : the corpus is sharded by sentence, 100M tokens PER RANK (weak scaling; N = 8: 125M per rank = the 1B-token corpus of
          configs[2]), with the per-order exchange of candidate counts over RCCL (colibri_amd.dist).
The run checks what it timed: for the default corpus (seed 44) the model must be the known one (tests/test_gpu_fullsize.py) or the exit
status is non-zero. `other_configs` reports the other model kinds of BASELINE.json (configs[3], configs[4]) on the same corpus from extra,
untimed steps.
Metric (BASELINE.json): M patterns counted / s, patterns counted = sum_{n<=5} W_n = the n-token windows inside
sentences that the reference enumerates in line.ngrams() (include/patternmodel.h:1063) — a property of the input.
Rank 0 prints ONE JSON line. `roofline` prices the dominant kernel (count) against HBM peak with the algorithmic
bytes of SURVEY.md §8(d) (stated in DESIGN.md §4); `cpu_baseline` times the real reference (oracle/_ref/ref_driver,
built from the reference's own sources) on a bounded sample of the same distribution on this box's host cores.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
MAXLENGTH, MINTOKENS = 5, 2


def algorithmic_bytes(nbytes, npos, stats, maxlength):
    """SURVEY.md §8(d), per order n, for the counting stage (scan + hash + table build):
         scan share   B + 4*(T+S)            corpus bytes + token-start vector, read once
                    + [n>1] * W_n * 2/8      two survivor bits per window (look-back)
         build share  P_n * (8 + 4 + 4)      per admitted window: key read, count read, count write
                    + D_n * (8 + 4)          per distinct candidate: key + count written once
       (the T/8 survivor-bitmap write of the formula belongs to the resolve kernel and is left out).
       Returns per-order lists (scan bytes, build bytes), index 0 = order 1."""
    scan, build = [], []
    for n in range(1, maxlength + 1):
        scan.append(nbytes + 4.0 * npos + (stats.windows[n] * 2.0 / 8.0 if n > 1 else 0.0))
        build.append(stats.admitted[n] * 16.0 + stats.found[n] * 12.0)
    return scan, build


def cpu_baseline(sample_tokens, vocab):
    """Time the reference's PatternModel<uint32_t>::train on a preloaded IndexedCorpus (src/benchmarks.cpp test 5
    style), 1 thread (the reference is single-threaded), on a bounded sample of the bench distribution."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    from colibri_amd import synth
    data = synth.zipf_corpus(sample_tokens, vocab, 45)
    arr = np.frombuffer(data, dtype=np.uint8)[2:]
    term = arr < 128
    prev_low = np.concatenate([[True], term[:-1]])
    delim = (arr == 0) & prev_low
    dpos = np.flatnonzero(delim[term])  # delimiter positions in position space
    lens = np.diff(np.concatenate([[-1], dpos])) - 1
    windows = int(sum(np.maximum(0, lens - n + 1).sum() for n in range(1, MAXLENGTH + 1)))
    sample = f"{sample_tokens}-token Zipf(1.0,V={vocab}) corpus, seed 45, same generator as the GPU workload; train() only, corpus preloaded"
    if oracle.have_ref():
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "sample.colibri.dat")
            with open(path, "wb") as f:
                f.write(data)
            _, info = oracle.ref_train(path, "U", MAXLENGTH, MINTOKENS)
        return {"value": round(windows / info["train_s"] / 1e6, 4), "unit": "M patterns counted/s", "cores": 1, "kind": "reference",
                "sample": sample, "seconds": round(info["train_s"], 3), "host_cores": os.cpu_count()}
    dt, w, _ = oracle.train_timed(data[2:], MINTOKENS, MAXLENGTH)
    return {"value": round(w / dt / 1e6, 4), "unit": "M patterns counted/s", "cores": 1, "kind": "port", "sample": sample,
            "seconds": round(dt, 3), "host_cores": os.cpu_count()}


def other_configs(ctx, capi, nbytes, tokens):
    """The other model kinds of BASELINE.json on the corpus that is resident: configs[3] (skipgrams: the exhaustive unindexed variant and the
    indexed one with MINSKIPTYPES = 2) and configs[4] (indexed model = forward index on the device). Untimed steps after the timed region: best of
    three train() calls each, with the kernel classes bracketed by HIP events in a fourth. Algorithmic bytes: the counting stage of the n-gram
    passes as in `roofline` plus, for indexed models, 8 bytes per reference written once and read once per 8-bit sort pass."""
    res = {}
    kinds = (("exhaustive_skipgrams", dict(doskipgrams_exhaustive=1)), ("indexed", dict(indexed=1)), ("indexed_skipgrams_T2", dict(indexed=1, doskipgrams=1, minskiptypes=2)))
    for name, kw in kinds:
        best, st = None, None
        for _ in range(3):
            st = ctx.train(maxlength=MAXLENGTH, mintokens=MINTOKENS, **kw)
            best = st.train_ms if best is None else min(best, st.train_ms)
        stp = ctx.train(maxlength=MAXLENGTH, mintokens=MINTOKENS, profile=1, **kw)
        kms = {capi.KERNEL_CLASSES[k]: round(ctx.kernel_time(k)[0], 3) for k in range(len(capi.KERNEL_CLASSES)) if ctx.kernel_time(k)[1]}
        dom = max(kms, key=kms.get) if kms else None
        scan_n, build_n = algorithmic_bytes(nbytes, ctx.positions(), st, MAXLENGTH)
        algo = sum(scan_n) + sum(build_n) + (8.0 * st.nrefs * 2 if kw.get("indexed") else 0.0)
        res[name] = {"ms_per_step": round(best, 3), "patterns_in_model": int(st.npatterns), "references": int(st.nrefs), "dominant_kernel_class": dom,
                     "kernel_ms_per_step": kms, "algorithmic_bytes_per_step": round(algo), "achieved_GBps": round(algo / (best * 1e-3) / 1e9, 1),
                     "frac_of_hbm_peak": round(algo / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
    return res


def measured_traffic(workload_tokens, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if int(d.get("tokens", 0)) == int(workload_tokens) and d.get("kernel") == kernel:
            return d.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tokens", type=int, default=0, help="tokens per GPU (default: 100M — the 100M-token config —, 125M with --gpus 8 = the 1B-token corpus of config 3)")
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="tokens of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the untimed steps of the other model kinds (skipgrams, indexed)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for debugging)")
    ap.add_argument("--share-gpu", action="store_true", help="debugging: all ranks use cuda:0 (needs --backend gloo)")
    ap.add_argument("--force-shard", action="store_true", help="debugging: run the sharded trainer even with one rank (prices the exchange machinery)")
    args = ap.parse_args()
    if args.tokens <= 0:
        args.tokens = 125_000_000 if args.gpus == 8 else 100_000_000

    import torch
    from colibri_amd import capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU path to measure")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_shard:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            # RCCL's version banner goes to stdout through C stdio and would land after the JSON line: keep stdout to that one line
            if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
                os.environ["NCCL_DEBUG"] = "NONE"
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    # ---- synthetic input, resident in HBM before the timed region ---------------------------------
    seed = 44 + rank
    t0 = time.time()
    payload = np.frombuffer(synth.zipf_corpus(args.tokens, args.vocab, seed, header=False), dtype=np.uint8)
    gen_s = time.time() - t0
    ctx = capi.Context(local_rank)
    dev_payload = torch.from_numpy(payload.copy()).cuda()  # H2D outside the timed region
    torch.cuda.synchronize()
    first_sentence = 1
    if dist is not None:  # global sentence numbers: this rank's shard starts after the sentences of the lower ranks
        prev_low = np.concatenate([[True], payload[:-1] < 128])
        nsent = int(((payload == 0) & prev_low).sum())
        counts = [None] * world
        dist.all_gather_object(counts, nsent)
        first_sentence = 1 + sum(counts[:rank])
    t0 = time.time()
    ctx.upload_device(dev_payload.data_ptr(), payload.size, first_sentence)  # tokenise on device
    tokenise_ms = (time.time() - t0) * 1e3
    # timed steps bracket only the class of the dominant kernel with HIP events (two events per step); the full per-class breakdown
    # comes from extra, untimed steps afterwards, so that ~110 event records per step do not sit inside the timed region
    opt = capi.Options.defaults(mintokens=MINTOKENS, maxlength=MAXLENGTH, profile=2)
    opt_all = capi.Options.defaults(mintokens=MINTOKENS, maxlength=MAXLENGTH, profile=1)

    if dist is not None:
        from colibri_amd import dist as cdist
        trainer = cdist.ShardedTrainer(capi.HipShardEngine(ctx, torch, device), dist, torch, device)
        step = lambda o=opt: trainer.train(o)
    else:
        step = lambda o=opt: ctx.train(o)

    for _ in range(args.warmup):
        st = step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kclasses = (capi.K_CLEAR, capi.K_COUNT, capi.K_PRUNE, capi.K_RESOLVE, capi.K_EMIT, capi.K_SCATTER, capi.K_BINCOUNT, capi.K_EMIT2, capi.K_LEVELB2, capi.K_COUNT2, capi.K_LISTS2)
    kms = {k: 0.0 for k in kclasses}
    kn = {k: 0 for k in kclasses}
    for _ in range(args.steps):
        st = step()
        for k in kclasses:  # HIP events on the library's own stream, this step
            ms, n = ctx.kernel_time(k)
            kms[k] += ms
            kn[k] += n
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # untimed: the per-class breakdown of a step (every kernel class bracketed with events)
    EXTRA = 2
    kms_all = {k: 0.0 for k in kclasses}
    kn_all = {k: 0 for k in kclasses}
    for _ in range(EXTRA):
        step(opt_all)
        for k in kclasses:
            ms, n = ctx.kernel_time(k)
            kms_all[k] += ms
            kn_all[k] += n
    windows = sum(st.windows[1:MAXLENGTH + 1])
    npatterns = int(st.npatterns)
    if dist is not None:
        where = device if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed, float(windows), float(npatterns)], dtype=torch.float64, device=where)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, windows, npatterns = float(tmax[0]), float(t[1]), int(t[2])
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = windows * args.steps / elapsed / 1e6
    scan_n, build_n = algorithmic_bytes(payload.size, ctx.positions(), st, MAXLENGTH)
    scan_b, build_b = sum(scan_n), sum(build_n)
    binned = kn_all[capi.K_BINCOUNT] > 0 or kn_all[capi.K_COUNT2] > 0  # (which kernels a step runs: from the fully bracketed extra steps; the timed ones bracket one class only)
    # Which kernels ran. Global-table path: count_kernel does scan + hash + build for every order. Radix path: order 1 is the class-indexed count
    # (class K_COUNT), order 2 the second-generation pipeline (emit2 / levelB2 / count2 / lists2: bigram2.hpp), orders >= 3 emit / scatter / bincount.
    # The dominant kernel (largest total time in the rocprofv3 stats) is then bi2_count_kernel — one launch per step, the table build of order 2 —
    # and its algorithmic bytes are the build share of ITS order; without it (a corpus the second generation cannot take) bin_count_kernel with the
    # build share of orders 2..5, as in round 1.
    uni = binned and kn_all[capi.K_COUNT] > 0
    second = kn_all[capi.K_COUNT2] > 0
    if second:
        dom, dom_bytes = capi.K_COUNT2, build_n[1]
    elif binned:
        dom, dom_bytes = capi.K_BINCOUNT, (sum(build_n[1:]) if uni else build_b)
    else:
        dom, dom_bytes = capi.K_COUNT, scan_b + build_b
    launches_per_step = kn[dom] / max(1, args.steps)
    avg_launch_ms = kms[dom] / max(1, kn[dom])
    achieved = (dom_bytes / max(1.0, launches_per_step)) / (avg_launch_ms * 1e-3) / 1e9 if kn[dom] else 0.0
    stage = (capi.K_COUNT, capi.K_EMIT, capi.K_SCATTER, capi.K_BINCOUNT, capi.K_EMIT2, capi.K_LEVELB2, capi.K_COUNT2) if binned else (capi.K_COUNT,)
    stage_ms = sum(kms_all[k] for k in stage) / EXTRA
    stage_gbs = (scan_b + build_b) / (stage_ms * 1e-3) / 1e9 if stage_ms else 0.0
    kept = [int(st.kept[n]) for n in range(1, MAXLENGTH + 1)]
    # ---- self-check: the model of the default corpus is known (seed 44: tests/test_gpu_fullsize.py; the reference's CPU run gives the same counts on the 1M / 10M corpora) ----
    KNOWN = {(100_000_000, 1_000_000, 1): ([999003, 6003382, 2066683, 298710, 18788], 9386566)}
    check = KNOWN.get((args.tokens, args.vocab, args.gpus))
    check_ok = None
    if check is not None:
        check_ok = kept == check[0] and npatterns == check[1]
    # ---- export, reported separately ----
    export_ms = None
    if dist is None:
        t0 = time.perf_counter()
        arrays = ctx.export_arrays()
        export_ms = (time.perf_counter() - t0) * 1e3
        if check is not None:
            check_ok = check_ok and len(arrays[2]) == check[1] and int(np.asarray(arrays[2], dtype=np.uint64).sum()) > 0
        del arrays
    out = {
        "metric": "M patterns counted/sec at n<=5 thr=2; identical pattern set vs reference",
        "value": round(value, 3),
        "unit": "M patterns counted/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.tokens}-token-per-GPU synthetic Zipf(1.0, V={args.vocab}) class-encoded corpus (.colibri.dat v2), sentences 5..35 tokens, "
                        f"unindexed PatternModel<uint32_t>, MAXLENGTH={MAXLENGTH}, MINTOKENS={MINTOKENS}",
            "tokens_per_gpu": args.tokens,
            "patterns_counted_per_step": int(windows),
            "patterns_in_model": npatterns,
            "kept_per_order": kept,
            "self_check": ("ok" if check_ok else "FAILED") if check_ok is not None else "no known answer for this configuration",
            "export_ms_untimed": round(export_ms, 2) if export_ms is not None else None,
            "parallelism": "single device" if args.gpus == 1 else f"sentence-sharded x{args.gpus}, per-order candidate exchange over RCCL",
            "tokenise_ms_untimed": round(tokenise_ms, 3),
            "corpus_generation_s_untimed": round(gen_s, 2),
        },
        "roofline": {
            "kernel": ("colibri::bi2_count_kernel (order 2: one wave per final bin — LDS table build, threshold, survivors, positions; one launch per step)" if second else
                       "colibri::bin_count_kernel (per-bin LDS hash build + threshold + survivor ids; one launch per order >= 2)" if binned else
                       "colibri::count_kernel (scan + SpookyHash + global hash-table build; one launch per order)"),
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": measured_traffic(args.tokens, "bi2_count_kernel" if second else "bin_count_kernel" if binned else "count_kernel") if args.gpus == 1 else None,
            "algorithmic_bytes_per_launch": round(dom_bytes / max(1.0, launches_per_step)),
            "avg_launch_ms": round(avg_launch_ms, 4),
            "launches_per_step": launches_per_step,
            "counting_stage": {"kernels": [capi.KERNEL_CLASSES[k] for k in stage], "ms_per_step": round(stage_ms, 4),
                               "algorithmic_bytes_per_step": round(scan_b + build_b), "achieved_GBps": round(stage_gbs, 2),
                               "frac": round(stage_gbs / HBM_PEAK_GBS, 5)},
            "kernel_ms_per_step": {capi.KERNEL_CLASSES[k]: round(kms_all[k] / EXTRA, 4) for k in kclasses if kn_all[k]},
            "note": f"achieved / avg_launch_ms: HIP events around the dominant kernel's launches inside the {args.steps} timed steps; counting_stage and "
                    f"kernel_ms_per_step: {EXTRA} extra untimed steps with every kernel class bracketed",
        },
    }
    if dist is None and not args.no_other_configs:
        out["other_configs"] = other_configs(ctx, capi, payload.size, args.tokens)
    if args.gpus == 1 and args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.vocab)
    print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if check_ok is False:
        sys.stderr.write("bench.py: the timed run did not produce the known model of this corpus: %r / %d patterns\n" % (kept, npatterns))
        sys.exit(3)


if __name__ == "__main__":
    main()
