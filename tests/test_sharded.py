"""The N > 1 path: sentence sharding + per-order candidate exchange (colibri_amd.dist.ShardedTrainer).
CPU (gloo, world 2 and 3, numpy stand-in engine): the exchange protocol — sizes, routing by owner, global survivor ids, exporter
election, termination — against the oracle on the whole corpus. GPU (gloo staging, ranks sharing cuda:0): the same with the
real HIP engine through the colibri_shard_* C ABI."""
import os
import pickle
import subprocess
import sys

import pytest

from conftest import ROOT

WORKER = os.path.join(ROOT, "tests", "shard_worker.py")
_port = [29700]


ORACLE_MODES = {"u": {}, "ug": {}, "us": dict(doskipgrams_exhaustive=True), "i": dict(indexed=True), "is": dict(indexed=True, doskipgrams=True),
                "isT1": dict(indexed=True, doskipgrams=True, minskiptypes=1), "usy3": dict(doskipgrams_exhaustive=True, mintokens_skipgrams=3)}


def run_workers(tmp_path, world, engine, corpus, maxlength, mode="u", backend="gloo"):
    import oracle
    out = str(tmp_path / f"res_{engine}_{corpus}_{world}_{mode}_{backend}.pkl")
    _port[0] += 1
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", COLIBRI_TEST_BACKEND=backend)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_port[0]),
           WORKER, engine, corpus, str(maxlength), out, mode]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = pickle.load(open(out, "rb"))
    want = oracle.train(res["payload"], 2, maxlength, **ORACLE_MODES[mode])
    assert res["dup"] == 0, "a pattern was exported by two ranks"
    assert res["union"] == want.counts
    if want.refs is not None:
        assert res["refs"] == want.refs
    assert (res["tokens"], res["types"], res["maxn"]) == (want.tokens, want.types, want.maxn)
    for n in range(1, min(maxlength, 15) + 1):
        assert (res["found"][n], res["kept"][n]) == (want.stats[n][0], want.stats[n][2]), n


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("corpus,maxlength", [("1", 5), ("2", 9), ("tiny", 4)])
def test_exchange_protocol_gloo_cpu(tmp_path, world, corpus, maxlength):
    run_workers(tmp_path, world, "numpy", corpus, maxlength)


@pytest.mark.parametrize("world,mode,corpus,maxlength", [(w, m, "1", 5) for w in (2, 3) for m in ("us", "usy3", "i", "is", "isT1")] + [(2, "us", "zipf", 4), (2, "is", "zipf", 4)])
def test_exchange_protocol_of_every_model_kind_gloo_cpu(tmp_path, world, mode, corpus, maxlength):
    """colibri_amd.dist.ShardedTrainer over gloo with the numpy stand-in engine, one process per rank: the skipgram passes level by level (exhaustive: BASELINE configs[3];
    indexed trainskipgrams with the distinct-filler counts in the last level's exchange: configs[4]) and the forward index merged by global id in rank order — the union of
    the ranks' exports, every reference list and the per-order figures equal the oracle's on the whole corpus"""
    run_workers(tmp_path, world, "numpy", corpus, maxlength, mode=mode)


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("corpus,maxlength,thr", [("rand_noempty", 5, 2), ("zipf20k", 5, 2), ("short_sentences", 4, 2), ("repeat", 9, 3), ("one_token", 3, 2), ("empty", 3, 2), ("zipf20k", 3, 1)])
def test_key_sharded_protocol_gloo_cpu(tmp_path, world, corpus, maxlength, thr):
    """the protocol host/src/sharded.cpp drives through colibri_kshard_* (records to the owner of their key; survivors' positions and the exports back; the
    dense class counts all-reduced), on a numpy stand-in over gloo: the union of the ranks' exports is the oracle's model of the whole corpus"""
    import oracle
    out = str(tmp_path / "k.pkl")
    _port[0] += 1
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_port[0]),
           os.path.join(ROOT, "tests", "kshard_worker.py"), corpus, str(maxlength), str(thr), out]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = pickle.load(open(out, "rb"))
    want = oracle.train(res["payload"], thr, maxlength)
    assert res["dup"] == 0, "a pattern was exported by two ranks"
    assert res["union"] == want.counts
    assert res["maxn"] == want.maxn
    for n in range(1, want.maxn + 1):
        assert (res["stats"][n][0], res["stats"][n][1]) == (want.stats[n][0], want.stats[n][2]), n


def test_a_failing_rank_is_reported_on_every_rank(tmp_path):
    """A rank whose local count raises (e.g. a radix bin outgrown under table_mode = 2) must not leave the others blocked in the next all-to-all: the failure
    travels with the size exchange and every rank raises ('nothing in dist.py shares error status across ranks')."""
    out = str(tmp_path / "fail")
    _port[0] += 1
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", COLIBRI_TEST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1", "--master-port", str(_port[0]), WORKER, "numpy_fail", "1",
           "5", out, "u"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    msgs = [open(f"{out}.rank{r}").read() for r in range(3)]
    assert all("local count failed on rank(s) [1]" in m for m in msgs), msgs
    assert "simulated" in msgs[1]


def test_shard_payload_keeps_global_sentence_numbers():
    from colibri_amd.dist import shard_payload
    payload = b"\x06\x00\x00\x07\x08\x00\x09\x00\x0a\x0b"  # 5 sentences (one empty, last unterminated)
    shards = shard_payload(payload, 3)
    assert b"".join(s for s, _ in shards) == payload
    seen = 1
    for s, first in shards:
        assert first == seen
        seen += s.count(b"\x00")
    assert shard_payload(b"", 2) == [(b"", 1), (b"", 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("world,corpus,maxlength", [(1, "zipf", 5), (2, "zipf", 5), (2, "3", 8), (3, "tiny", 4), (4, "zipf", 5)])
def test_hip_shard_engine_gloo_staged(tmp_path, world, corpus, maxlength):
    run_workers(tmp_path, world, "hip", corpus, maxlength)


@pytest.mark.gpu
def test_hip_shard_engine_global_table_local_count(tmp_path):
    """table_mode = 1: local counting on the open-addressed table and a key exchange for order 1 too (the path used above 128 M tokens per rank)"""
    run_workers(tmp_path, 2, "hip", "zipf", 5, "ug")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["us", "usy3", "i", "is", "isT1"])
@pytest.mark.parametrize("world,corpus", [(1, "zipf"), (2, "zipf"), (3, "2"), (2, "tiny")])
def test_hip_shard_engine_skipgram_and_indexed_modes(tmp_path, world, corpus, mode):
    """configs 4 and 5 of BASELINE.json are multi-GPU: exhaustive skipgrams, the forward index and indexed skipgrams, sharded —
    the union over ranks equals the oracle's single-process model (counts and, for indexed models, every reference list)."""
    run_workers(tmp_path, world, "hip", corpus, 5, mode)


def test_merge_exports_concatenates_in_rank_order():
    from colibri_amd.dist import merge_exports, gap_masks, mask_parts
    a = {"patterns": {7: (b"\x06", 5)}, "index": {7: [(1, 0), (2, 3)], 9: [(2, 1)]}}
    b = {"patterns": {9: (b"\x07", 2)}, "index": {7: [(5, 0)], 9: [(6, 2)]}}
    counts, refs = merge_exports([a, b])
    assert counts == {b"\x06": 5, b"\x07": 2}
    assert refs == {b"\x06": [(1, 0), (2, 3), (5, 0)], b"\x07": [(2, 1), (6, 2)]}
    with pytest.raises(ValueError):
        merge_exports([a, a])
    import oracle
    for n in range(3, 9):
        assert gap_masks(n, 3) == oracle.skip_configurations(n, 3)
    assert mask_parts(0b01010, 5) == [(0, 1), (2, 1), (4, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 5, 8])
def test_hip_shard_engine_ranks_as_threads(world):
    """`world` ranks as threads of one process, each with its own device context on cuda:0, exchanging through an in-process stand-in
    for the process group with device tensors (no host staging): the same buffers RCCL would move (tools/fuzz_sharded.py)."""
    import numpy as np
    import oracle
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from colibri_amd import capi
    from colibri_amd import dist as cdist
    from fuzz_parity import make_corpus
    from fuzz_sharded import run_case
    for case in range(12):
        rng = np.random.default_rng(977 * world + case)
        payload = make_corpus(rng)
        mode = case % 4
        o = dict(mintokens=2 + case % 2, maxlength=5, indexed=int(mode in (2, 3)), doskipgrams=int(mode == 3), doskipgrams_exhaustive=int(mode == 1))
        if case >= 8:  # threshold 1 (everything survives on every rank) and the secondary word threshold, in every kind of model
            o.update(dict(mintokens=1, maxlength=4) if case % 2 else dict(mintokens_unigrams=o["mintokens"] + 2))
        assert run_case(case, world, payload, o, capi, oracle, torch, cdist) is None


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["u", "ug", "us", "i", "is"])
def test_hip_shard_engine_over_rccl(tmp_path, mode):
    """The exchange with torch.distributed's "nccl" backend (= RCCL): device tensors in all_to_all_single / all_reduce / all_gather, the hand-over
    between torch's stream and the library's. One rank (the GPU box has one device): every collective still runs through RCCL."""
    run_workers(tmp_path, 1, "hip", "zipf", 5, mode, backend="nccl")


def _bench(args, env_extra=None, launcher=None):
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(env_extra or {}))
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args + ["--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--no-other-configs"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
def test_bench_one_rank_of_the_multi_gpu_trainer_over_rccl_matches_oracle():
    """bench.py --force-shard: the product's multi-GPU trainer (host/src/sharded.cpp: key-sharded counting, RCCL linked directly) with one rank — every
    exchange still runs through RCCL (send / recv to itself, the all-reduces); its model against the oracle."""
    import oracle
    from colibri_amd import synth
    tokens, vocab = 2_000_000, 1_000_000
    got = _bench(["--force-shard", "--tokens", str(tokens), "--vocab", str(vocab)])
    want = oracle.train(synth.zipf_corpus(tokens, vocab, 44, header=False), 2, 5)
    assert got["config"]["kept_per_order"] == [want.stats[n][2] for n in range(1, 6)]
    assert got["config"]["patterns_in_model"] == len(want.counts)
    assert got["sharded"]["protocol"] == "key-sharded counting" and ", RCCL" in got["config"]["parallelism"]
    assert got["sharded"]["host_lookups_per_step"] <= 10


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [2, 4])
def test_bench_gpus_n_started_like_the_single_gpu_command(gpus):
    """`python bench.py --gpus N` with no launcher (it used to exit at once): the N ranks are host threads of the one process. Here all ranks share
    device 0 (--share-gpu: they exchange by device copies; with N devices the same command runs RCCL over xGMI). rc 0 and the model of the N shards (seeds
    44 .. 44 + N - 1, concatenated in rank order) must be the oracle's."""
    import oracle
    from colibri_amd import synth
    tokens, vocab = 1_000_000, 200_000
    got = _bench(["--gpus", str(gpus), "--share-gpu", "--tokens", str(tokens), "--vocab", str(vocab)])
    whole = b"".join(synth.zipf_corpus(tokens, vocab, 44 + r, header=False) for r in range(gpus))
    want = oracle.train(whole, 2, 5)
    assert got["n_gpus"] == gpus and got["config"]["kept_per_order"] == [want.stats[n][2] for n in range(1, 6)]
    assert got["config"]["patterns_in_model"] == len(want.counts)
    assert got["config"]["patterns_counted_per_step"] == want.windows  # the windows of all N shards: what the reference enumerates in line.ngrams()
    assert got["sharded"]["protocol"] == "key-sharded counting"


@pytest.mark.gpu
def test_bench_one_process_per_rank_path(tmp_path):
    """the driver's multi-GPU launch (torch.distributed.run, one process per rank): ncclCommInitRank from an id rank 0 hands out, host values through RCCL
    all-gathers. One device here, hence one rank — COLIBRI_BENCH_PER_PROCESS makes bench.py take that path for it."""
    import oracle
    from colibri_amd import synth
    tokens, vocab = 1_000_000, 200_000
    _port[0] += 1
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(_port[0])]
    got = _bench(["--gpus", "1", "--force-shard", "--tokens", str(tokens), "--vocab", str(vocab)], env_extra={"COLIBRI_BENCH_PER_PROCESS": "1"}, launcher=launcher)
    want = oracle.train(synth.zipf_corpus(tokens, vocab, 44, header=False), 2, 5)
    assert got["config"]["kept_per_order"] == [want.stats[n][2] for n in range(1, 6)] and got["config"]["patterns_in_model"] == len(want.counts)
    assert "one process per rank" in got["config"]["parallelism"]
