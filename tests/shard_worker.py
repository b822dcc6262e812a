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
    """Same interface as capi.HipShardEngine, stated position by position in Python (test-only): the n-gram passes, every level of a skipgram pass (n, mask, level) —
    the skipgram's identity is built by pairing the global ids of its contiguous parts left to right —, the distinct-source counts of an indexed skipgram's last level,
    the local forward index keyed by global id (include/colibri_hip.h "colibri_shard_*")."""

    def __init__(self, payload, first_sentence=1):
        self.payload = payload
        self.pos, start = [], 0
        for j, b in enumerate(payload):
            if b < 128:
                self.pos.append((start, j + 1))
                start = j + 1
        self.delim = [e - s == 1 and payload[s] == 0 for s, e in self.pos]
        self.where, sn, tn = [], first_sentence, 0  # (sentence, token) of every position
        for d in self.delim:
            self.where.append((sn, tn))
            sn, tn = (sn + 1, 0) if d else (sn, tn + 1)

    def local_tokens(self):
        return sum(1 for d in self.delim if not d)

    def begin(self, opt, world):
        self.opt, self.world, self.thr = opt, world, (2 if opt.mintokens == -1 else max(1, opt.mintokens))
        self.thr_skip = max(self.thr, int(opt.mintokens_skipgrams))
        self.ids, self.scratch, self.results, self.admitted, self.pairs = {}, {}, [], {}, []
        self.mark = [0] * (len(self.pos) + 1)

    def _id(self, arr, i):
        return arr[i] if 0 <= i < len(arr) else INV

    def count(self, n, mask=0, level=1):
        from colibri_amd.dist import mask_parts
        opt, P = self.opt, len(self.pos)
        self.n, self.mask, self.level, self.final, self.use_aux, self.pthr, self.minsrc = n, mask, level, True, False, self.thr, 0
        self.keyof, table = [None] * P, {}
        if mask:
            parts = mask_parts(mask, n)
            assert 1 <= level < len(parts)
            self.final = level + 1 == len(parts)
            if not self.final:
                self.pthr = 1  # an intermediate level only names pairs of ids
            elif opt.doskipgrams:
                self.use_aux, self.minsrc = True, (int(opt.minskiptypes) if opt.minskiptypes > 1 else 0)
            else:
                self.pthr = self.thr_skip if opt.minskiptypes > 1 else self.thr
        for i in range(P):
            if mask == 0 and n == 1:
                if self.delim[i]:
                    continue
                s, e = self.pos[i]
                key = int.from_bytes(self.payload[s:e], "little")
            elif mask == 0:
                a, b = self._id(self.ids[n - 1], i), self._id(self.ids[n - 1], i + 1)
                if a == INV or b == INV:
                    continue
                key = (a << 32) | b
            else:
                if opt.doskipgrams:
                    if self._id(self.ids[n], i) == INV:
                        continue
                elif self._id(self.ids[n - 1], i) == INV or self._id(self.ids[n - 1], i + 1) == INV:
                    continue
                left = self._id(self.ids[parts[0][1]], i + parts[0][0]) if level == 1 else self._id(self.scratch[level - 1], i)
                right = self._id(self.ids[parts[level][1]], i + parts[level][0])
                if left == INV or right == INV:
                    continue
                key = (left << 32) | right
            self.keyof[i] = key
            ent = table.setdefault(key, [0, i, 0])
            ent[0] += 1
            if self.use_aux and (self.mark[i] >> n) & 1:
                ent[2] += 1  # this occurrence is the representative of an n-gram this rank exports: one more distinct filler
        if mask == 0:
            self.admitted[n] = sum(1 for k in self.keyof if k is not None)
        self.table = table
        buckets = [[] for _ in range(self.world)]
        for key, (c, _, a) in table.items():
            buckets[mix64(key) % self.world].append((key, c, a))
        self.send = [kc for b in buckets for kc in b]
        return len(self.send), [len(b) for b in buckets]

    def send_buffers(self):
        keys = np.array([k for k, _, _ in self.send], dtype=np.uint64).view(np.int64)
        cnts = np.array([c for _, c, _ in self.send], dtype=np.uint32).view(np.int32)
        aux = np.array([a for _, _, a in self.send], dtype=np.uint32).view(np.int32)
        return torch.from_numpy(keys.copy()), torch.from_numpy(cnts.copy()), torch.from_numpy(aux.copy())

    def _kept(self, ent):
        return ent[0] >= self.pthr and ent[2] >= self.minsrc

    def merge(self, keys, cnts, aux, per_src):
        self.rkeys = keys.numpy().view(np.uint64).tolist()
        rc = cnts.numpy().view(np.uint32).tolist()
        ra = aux.numpy().view(np.uint32).tolist() if aux is not None else [0] * len(rc)
        self.rsrc = [r for r, m in enumerate(per_src) for _ in range(m)]
        self.owner = {}
        for k, c, a, r in zip(self.rkeys, rc, ra, self.rsrc):
            ent = self.owner.setdefault(k, [0, r, 0])
            ent[0] += c
            ent[1] = min(ent[1], r)
            ent[2] += a
        return len(self.owner), sum(1 for e in self.owner.values() if self._kept(e))

    def reply(self, base):
        gid = {}
        for k in sorted(self.owner):
            if self._kept(self.owner[k]):
                gid[k] = base + len(gid)
        g = [(gid[k] | (0x80000000 if self.owner[k][1] == r else 0)) if k in gid else INV for k, r in zip(self.rkeys, self.rsrc)]
        t = [self.owner[k][0] for k in self.rkeys]
        return torch.from_numpy(np.array(g, dtype=np.uint32).view(np.int32).copy()), torch.from_numpy(np.array(t, dtype=np.uint32).view(np.int32).copy())

    def apply(self, gid, tot):
        n = self.n
        g = gid.numpy().view(np.uint32).tolist()
        t = tot.numpy().view(np.uint32).tolist()
        gmap, exported = {}, 0
        for (key, _, _), gg, tt in zip(self.send, g, t):
            if gg != INV:
                gmap[key] = gg & 0x7FFFFFFF
                if gg & 0x80000000 and self.final:
                    rep = self.table[key][1]
                    self.results.append((rep, n, tt, self.mask, gg & 0x7FFFFFFF))
                    if self.mask == 0 and self.opt.doskipgrams and n < 32:
                        self.mark[rep] |= 1 << n
                    exported += 1
        out = [gmap.get(k, INV) if k is not None else INV for k in self.keyof]
        if self.mask == 0:
            self.ids[n] = out
        else:
            self.scratch[self.level] = out
        if self.opt.indexed and self.final:
            self.pairs.extend((v, i) for i, v in enumerate(out) if v != INV)
        return exported, self.admitted[n]

    def finish(self, found_g, kept_g, tokens_g, maxn):
        class S:
            pass
        s = S()
        s.found, s.kept, s.totaltokens, s.totaltypes, s.maxn, s.npatterns = found_g, kept_g, tokens_g, found_g[1], maxn, len(self.results)
        return s

    def _key(self, p, n, mask):
        return b"".join(b"\x03" if (mask >> t) & 1 else self.payload[self.pos[p + t][0]: self.pos[p + t][1]] for t in range(n))

    def export_dict(self):
        return {self._key(p, n, mask): c for p, n, c, mask, _ in self.results}

    def export_local(self):
        """this rank's share, as capi.HipShardEngine.export_local gives it: patterns by global id, the local forward index by global id"""
        index = None
        if self.opt.indexed:
            index = {}
            for g, i in sorted(self.pairs):
                index.setdefault(g, []).append(self.where[i])
        return {"patterns": {g: (self._key(p, n, mask), c) for p, n, c, mask, g in self.results}, "index": index}


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
        class Failing(NumpyShardEngine):  # noqa: E306
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
        eng = NumpyShardEngine(shard, first)
        trainer = ShardedTrainer(eng, dist, torch)
        st = trainer.train(opt)
        mine = eng.export_local()
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
