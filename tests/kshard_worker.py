"""Worker of the CPU test of the key-sharded protocol (launched by torch.distributed.run from tests/test_sharded.py, backend gloo): the steps
host/src/sharded.cpp drives through colibri_kshard_* — order 1 as an all-reduce of dense class counts; at every higher order each rank turns the windows of ITS
sentences that pass the look-back into records (key, position), every record travels to the owner of its key, the owner counts and applies the threshold to the
global count, the positions of surviving windows go back to their sources (the next order's look-back) and every surviving pattern is exported by the lowest rank
that holds an occurrence; the loop ends at the first order no rank admits a window for — stated position by position in Python on a numpy stand-in, so that the
PROTOCOL (routing, termination, export election, the sums that make the statistics) is checked against the oracle without a GPU. The HIP kernels behind the same
steps are checked on the GPU (tests/test_kshard.py)."""
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "colibri-core_amd", "pyhost"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from colibri_amd import synth  # noqa: E402
from colibri_amd.dist import shard_payload  # noqa: E402


def mix64(x):
    x &= (1 << 64) - 1
    x ^= x >> 33
    x = (x * 0xFF51AFD7ED558CCD) & ((1 << 64) - 1)
    x ^= x >> 33
    x = (x * 0xC4CEB9FE1A85EC53) & ((1 << 64) - 1)
    x ^= x >> 33
    return x


def tokens_of(payload):
    """[(class id, token bytes)] per position; delimiters are class 0"""
    out, start, val, shift = [], 0, 0, 0
    for j, b in enumerate(payload):
        val |= (b & 127) << shift
        shift += 7
        if b < 128:
            out.append((val, payload[start:j + 1]))
            start, val, shift = j + 1, 0, 0
    return out


def all_to_all(lists, world):
    """lists[d] = what this rank sends to rank d; returns what it receives, one list per source"""
    box = [None] * world
    dist.all_gather_object(box, lists)
    me = dist.get_rank()
    return [box[s][me] for s in range(world)]


def train(payload_shard, thr, maxlength):
    rank, world = dist.get_rank(), dist.get_world_size()
    toks = tokens_of(payload_shard)
    npos = len(toks)
    cls = [c for c, _ in toks]
    stats = {}
    # ---- order 1: all-reduce (SUM) of the dense class counts; the same survivors on every rank; rank 0 exports them from their class ids
    nclasses = torch.tensor([max(cls, default=0) + 1])
    dist.all_reduce(nclasses, op=dist.ReduceOp.MAX)
    cnt = torch.zeros(int(nclasses), dtype=torch.int64)
    for c in cls:
        if c:
            cnt[c] += 1
    admitted = sum(1 for c in cls if c)
    dist.all_reduce(cnt)
    alive = [bool(c and cnt[c] >= thr) for c in range(int(nclasses))]
    exports = {}
    if rank == 0:
        for c in range(1, int(nclasses)):
            if cnt[c] >= thr:
                exports[bytes(synth.encode_v2(__import__("numpy").array([c], dtype="uint32")))] = int(cnt[c])
    stats[1] = [int((cnt[1:] > 0).sum()) if rank == 0 else 0, int((cnt[1:] >= thr).sum()) if rank == 0 else 0, admitted]
    surv_prev = [alive[c] for c in cls]  # per position: the (n-1)-gram starting here survived
    maxn = 1 if int(cnt.sum()) else 0
    for n in range(2, maxlength + 1):
        # source side: the windows that pass the look-back become records (key = the window's class ids; the real records carry a bijective mix of them)
        recs = [[] for _ in range(world)]
        nadm = 0
        for i in range(npos - n + 1):
            if surv_prev[i] and surv_prev[i + 1] and all(cls[i + k] for k in range(n)):
                key = tuple(cls[i:i + n])
                recs[mix64(hash(key) & ((1 << 63) - 1)) % world].append((key, i))
                nadm += 1
        sizes = [None] * world
        dist.all_gather_object(sizes, nadm)
        if sum(sizes) == 0:  # "None found" (patternmodel.h:1189-1194): no rank has a window of this order left
            break
        maxn = n
        got = all_to_all(recs, world)
        # owner side: exact global counts of the keys this rank owns
        table = {}
        for src in range(world):
            for key, pos in got[src]:
                e = table.setdefault(key, [0, (world, 0)])
                e[0] += 1
                e[1] = min(e[1], (src, pos))
        feedback = [[] for _ in range(world)]
        exp_out = [[] for _ in range(world)]
        for src in range(world):
            for key, pos in got[src]:
                if table[key][0] >= thr:
                    feedback[src].append(pos)
        for key, (c, (src, pos)) in table.items():
            if c >= thr:
                exp_out[src].append((pos, c))
        stats[n] = [len(table), sum(1 for e in table.values() if e[0] >= thr), nadm]
        fb = all_to_all(feedback, world)
        ex = all_to_all(exp_out, world)
        surv = [False] * npos
        for part in fb:
            for pos in part:
                surv[pos] = True
        for part in ex:
            for pos, c in part:
                exports[b"".join(toks[pos + k][1] for k in range(n))] = c
        surv_prev = surv
    allstats = [None] * world
    dist.all_gather_object(allstats, stats)
    total = {n: [sum(s.get(n, [0, 0, 0])[k] for s in allstats) for k in range(3)] for n in range(1, maxn + 1)}
    return exports, total, maxn


def main():
    corpus, maxlength, thr, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from conftest import small_corpora
    payload = small_corpora()[corpus]
    shard, _ = shard_payload(payload, world)[rank]
    exports, stats, maxn = train(shard, thr, maxlength)
    box = [None] * world
    dist.all_gather_object(box, exports)
    if rank == 0:
        union, dup = {}, 0
        for e in box:
            for k, v in e.items():
                dup += k in union
                union[k] = v
        with open(out, "wb") as f:
            pickle.dump({"union": union, "dup": dup, "stats": stats, "maxn": maxn, "payload": payload}, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
