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


def run_workers(tmp_path, world, engine, corpus, maxlength):
    import oracle
    out = str(tmp_path / f"res_{engine}_{corpus}_{world}.pkl")
    _port[0] += 1
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_port[0]),
           WORKER, engine, corpus, str(maxlength), out]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = pickle.load(open(out, "rb"))
    want = oracle.train(res["payload"], 2, maxlength)
    assert res["dup"] == 0, "a pattern was exported by two ranks"
    assert res["union"] == want.counts
    assert (res["tokens"], res["types"], res["maxn"]) == (want.tokens, want.types, want.maxn)
    for n in range(1, min(maxlength, 15) + 1):
        assert (res["found"][n], res["kept"][n]) == (want.stats[n][0], want.stats[n][2]), n


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("corpus,maxlength", [("1", 5), ("2", 9), ("tiny", 4)])
def test_exchange_protocol_gloo_cpu(tmp_path, world, corpus, maxlength):
    run_workers(tmp_path, world, "numpy", corpus, maxlength)


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
