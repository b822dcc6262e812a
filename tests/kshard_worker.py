"""Worker of the CPU test of the key-sharded protocol (launched by torch.distributed.run from tests/test_sharded.py, backend gloo): the steps
host/src/sharded.cpp drives through colibri_kshard_* — order 1 as an all-reduce of dense class counts; at every higher order each rank turns the windows of ITS
sentences that pass the look-back into records (key, position), every record travels to the owner of its key, the owner counts and applies the threshold to the
global count; round 4's wire format (csrc/kshard2.hpp): KEYS travel (the positions stay at home, in send order), the owner answers one bit per key and the dense number
of every surviving key's window in the order the source sent, numbers become global by the kept counts of the owners before; every surviving pattern is exported, as
(index in the exporter's stream, count), by the lowest rank that holds an occurrence; the loop ends at the first order no rank admits a window for — stated position by position in Python on a numpy stand-in, so that the
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
    # per position: the GLOBAL number of the surviving (n-1)-gram that starts here, None if there is none (order 1: the class id names the unigram)
    num_prev = [c if alive[c] else None for c in cls]
    maxn = 1 if int(cnt.sum()) else 0
    for n in range(2, maxlength + 1):
        # source side (csrc/kshard2.hpp: emit -> level B -> ks2_move): the windows that pass the look-back become KEYS, sent in one stream per owner; the window's
        # position stays at home, in the same order. The key is exact: order 2 the class pair, order n >= 3 (global number of the leading (n-1)-gram, last class)
        # (the real stream carries the bits of its bijective mix that the (owner, bin) does not fix).
        send = [[] for _ in range(world)]
        home = [[] for _ in range(world)]
        nadm = 0
        for i in range(npos - n + 1):
            if num_prev[i] is not None and num_prev[i + 1] is not None:
                key = (cls[i], cls[i + 1]) if n == 2 else (num_prev[i], cls[i + n - 1])
                d = mix64(hash(key) & ((1 << 63) - 1)) % world
                send[d].append(key)
                home[d].append(i)
                nadm += 1
        sizes = [None] * world
        dist.all_gather_object(sizes, nadm)
        if sum(sizes) == 0:  # "None found" (patternmodel.h:1189-1194): no rank has a window of this order left
            break
        maxn = n
        got = all_to_all(send, world)
        # owner side (bi2_count_kernel<.., KEY4>): exact global counts of the keys this rank owns; a record's "position" is its place (source, index in the stream)
        table = {}
        for src in range(world):
            for j, key in enumerate(got[src]):
                e = table.setdefault(key, [0, (world, 0)])
                e[0] += 1
                e[1] = min(e[1], (src, j))
        dense = {}  # the survivors' dense numbers on this owner (any fixed order: here by first occurrence)
        for key, (c, first) in sorted(table.items(), key=lambda kv: kv[1][1]):
            if c >= thr:
                dense[key] = len(dense)
        # feedback (ks2_fb_*): per source, IN THE ORDER THE SOURCE SENT, one bit per key and then the dense number of every surviving key's window;
        # exports: (index in the exporter's stream, count) to the lowest rank holding an occurrence
        feedback = [([1 if key in dense else 0 for key in got[src]], [dense[key] for key in got[src] if key in dense]) for src in range(world)]
        exp_out = [[] for _ in range(world)]
        for key, (c, (src, j)) in table.items():
            if c >= thr:
                exp_out[src].append((j, c))
        stats[n] = [len(table), len(dense), nadm]
        kept_all = [None] * world
        dist.all_gather_object(kept_all, len(dense))  # owner d's numbers are shifted by the kept counts of the owners before it: identical on every rank
        base = [sum(kept_all[:d]) for d in range(world)]
        fb = all_to_all(feedback, world)
        ex = all_to_all(exp_out, world)
        # apply (ks2_dec_*): the source walks its own send order, counts bits and has (position, global number) of every surviving window
        num = [None] * npos
        for d in range(world):
            bits, numbers = fb[d]
            assert len(bits) == len(home[d]) and sum(bits) == len(numbers)
            k = 0
            for j, pos in enumerate(home[d]):
                if bits[j]:
                    num[pos] = base[d] + numbers[k]
                    k += 1
            for j, c in ex[d]:
                pos = home[d][j]
                exports[b"".join(toks[pos + k2][1] for k2 in range(n))] = c
        num_prev = num
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
