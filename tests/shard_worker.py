"""Worker of the multi-process sharding tests (launched by torch.distributed.run from tests/test_sharded.py).
  engine = numpy : a pure-Python/numpy stand-in for the per-rank engine (CPU, gloo) — exercises the exchange protocol
  engine = hip   : the real HIP engine through the C ABI (all ranks share cuda:0; gloo stages the exchange through the host)
Every rank trains on its sentence shard; rank 0 gathers the per-rank exports and compares the union with the oracle's
single-process result on the whole corpus (bit-exact: pattern set, counts, totaltokens, totaltypes, per-order found/kept)."""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from colibri_amd import capi, synth  # noqa: E402
from colibri_amd.dist import ShardedTrainer, shard_payload  # noqa: E402

INV = 0xFFFFFFFF


def mix64(x):
    x &= (1 << 64) - 1
    x ^= x >> 33
    x = (x * 0xFF51AFD7ED558CCD) & ((1 << 64) - 1)
    x ^= x >> 33
    x = (x * 0xC4CEB9FE1A85EC53) & ((1 << 64) - 1)
    x ^= x >> 33
    return x


class NumpyShardEngine:
    """Same interface as capi.HipShardEngine, stated position by position in Python (test-only)."""

    def __init__(self, payload):
        self.payload = payload
        self.pos, start = [], 0
        for j, b in enumerate(payload):
            if b < 128:
                self.pos.append((start, j + 1))
                start = j + 1
        self.delim = [e - s == 1 and payload[s] == 0 for s, e in self.pos]

    def local_tokens(self):
        return sum(1 for d in self.delim if not d)

    def begin(self, opt, world):
        self.world, self.thr = world, (2 if opt.mintokens == -1 else max(1, opt.mintokens))
        self.ids_prev, self.results, self.admitted = None, [], {}

    def count(self, n, mask=0, level=1):
        assert mask == 0, "the numpy stand-in covers the n-gram passes"
        self.n, self.keyof, table = n, [None] * len(self.pos), {}
        for i in range(len(self.pos)):
            if n == 1:
                if self.delim[i]:
                    continue
                s, e = self.pos[i]
                key = int.from_bytes(self.payload[s:e], "little")
            else:
                if i + 1 >= len(self.pos) or self.ids_prev[i] == INV or self.ids_prev[i + 1] == INV:
                    continue
                key = (self.ids_prev[i] << 32) | self.ids_prev[i + 1]
            self.keyof[i] = key
            ent = table.setdefault(key, [0, i])
            ent[0] += 1
        self.admitted[n] = sum(1 for k in self.keyof if k is not None)
        self.table = table
        buckets = [[] for _ in range(self.world)]
        for key, (c, _) in table.items():
            buckets[mix64(key) % self.world].append((key, c))
        self.send = [kc for b in buckets for kc in b]
        return len(self.send), [len(b) for b in buckets]

    def send_buffers(self):
        keys = np.array([k for k, _ in self.send], dtype=np.uint64).view(np.int64)
        cnts = np.array([c for _, c in self.send], dtype=np.uint32).view(np.int32)
        return torch.from_numpy(keys.copy()), torch.from_numpy(cnts.copy()), torch.zeros(len(self.send), dtype=torch.int32)

    def merge(self, keys, cnts, aux, per_src):
        self.rkeys = keys.numpy().view(np.uint64).tolist()
        rc = cnts.numpy().view(np.uint32).tolist()
        self.rsrc = [r for r, m in enumerate(per_src) for _ in range(m)]
        self.owner = {}
        for k, c, r in zip(self.rkeys, rc, self.rsrc):
            ent = self.owner.setdefault(k, [0, r])
            ent[0] += c
            ent[1] = min(ent[1], r)
        return len(self.owner), sum(1 for t, _ in self.owner.values() if t >= self.thr)

    def reply(self, base):
        gid = {}
        for k in sorted(self.owner):
            if self.owner[k][0] >= self.thr:
                gid[k] = base + len(gid)
        g = [(gid[k] | (0x80000000 if self.owner[k][1] == r else 0)) if k in gid else INV for k, r in zip(self.rkeys, self.rsrc)]
        t = [self.owner[k][0] for k in self.rkeys]
        return torch.from_numpy(np.array(g, dtype=np.uint32).view(np.int32).copy()), torch.from_numpy(np.array(t, dtype=np.uint32).view(np.int32).copy())

    def apply(self, gid, tot):
        n = self.n
        g = gid.numpy().view(np.uint32).tolist()
        t = tot.numpy().view(np.uint32).tolist()
        gmap, exported = {}, 0
        for (key, _), gg, tt in zip(self.send, g, t):
            if gg != INV:
                gmap[key] = gg & 0x7FFFFFFF
                if gg & 0x80000000:
                    self.results.append((self.table[key][1], n, tt))
                    exported += 1
        self.ids_prev = [gmap.get(k, INV) if k is not None else INV for k in self.keyof]
        return exported, self.admitted[n]

    def finish(self, found_g, kept_g, tokens_g, maxn):
        class S:
            pass
        s = S()
        s.found, s.kept, s.totaltokens, s.totaltypes, s.maxn, s.npatterns = found_g, kept_g, tokens_g, found_g[1], maxn, len(self.results)
        return s

    def export_dict(self):
        return {self.payload[self.pos[p][0]: self.pos[p + n - 1][1]]: c for p, n, c in self.results}


MODES = {"u": {}, "ug": dict(table_mode=1), "us": dict(doskipgrams_exhaustive=1), "i": dict(indexed=1), "is": dict(indexed=1, doskipgrams=1), "isT1": dict(indexed=1, doskipgrams=1, minskiptypes=1),
         "usy3": dict(doskipgrams_exhaustive=1, mintokens_skipgrams=3)}


def main():
    engine_kind, corpus_kind, maxlength, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else "u"
    backend = os.environ.get("COLIBRI_TEST_BACKEND", "gloo")
    if backend == "nccl":  # RCCL: device tensors in every collective (one rank per GPU; the GPU box has one)
        os.environ.setdefault("NCCL_DEBUG", "NONE")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if corpus_kind == "zipf":
        payload = synth.zipf_corpus(60000, 2000, 3, phrases=True, header=False)
    elif corpus_kind == "tiny":
        payload = b"\x06\x07\x00" * 3 + b"\x08\x00"  # fewer sentences than ranks can leave a shard empty
    else:
        payload = synth.random_corpus(np.random.default_rng(int(corpus_kind)), nsent=300, maxlen=12, vocab=12)
    shard, first = shard_payload(payload, world)[rank]
    opt = capi.Options.defaults(mintokens=2, maxlength=maxlength, **MODES[mode])
    if engine_kind == "numpy_fail":  # rank 1's local count of order 3 raises: every rank must learn of it instead of waiting in a collective
        class Failing(NumpyShardEngine):
            def count(self, n, mask=0, level=1):
                if n == 3 and rank == 1:
                    raise RuntimeError("radix path overflowed (simulated)")
                return super().count(n, mask, level)
        msg = ""
        try:
            ShardedTrainer(Failing(shard), dist, torch).train(opt)
        except RuntimeError as e:
            msg = str(e)
        with open(f"{out}.rank{rank}", "w") as f:
            f.write(msg)
        dist.barrier()
        dist.destroy_process_group()
        return
    if engine_kind == "numpy":
        eng = NumpyShardEngine(shard)
        trainer = ShardedTrainer(eng, dist, torch)
        st = trainer.train(opt)
        mine = {"patterns": {i: kv for i, kv in enumerate(eng.export_dict().items())}, "index": None}
        mine["patterns"] = {(rank << 40) | i: kv for i, kv in mine["patterns"].items()}
    else:
        ctx = capi.Context(0)
        ctx.upload(shard, first_sentence=first)
        eng = capi.HipShardEngine(ctx, torch, torch.device("cuda", 0))
        trainer = ShardedTrainer(eng, dist, torch, torch.device("cuda", 0) if backend == "nccl" else None)
        st = trainer.train(opt)
        mine = eng.export_local()
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        from colibri_amd.dist import merge_exports
        dup = 0
        try:
            counts, refs = merge_exports(gathered)
            seen = {}
            for ex in gathered:  # the same key bytes must not come from two ranks either
                for g, (k, c) in ex["patterns"].items():
                    dup += k in seen
                    seen[k] = c
        except ValueError:
            counts, refs, dup = {}, None, 1
        with open(out, "wb") as f:
            pickle.dump({"union": counts, "refs": refs, "dup": dup, "tokens": int(st.totaltokens), "types": int(st.totaltypes), "maxn": int(st.maxn),
                         "found": [int(x) for x in list(st.found)[:16]], "kept": [int(x) for x in list(st.kept)[:16]], "payload": payload, "mode": mode}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
