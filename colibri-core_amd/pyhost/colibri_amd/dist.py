"""Sentence-sharded training across the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests) for the one real exchange step the path has.

The reference is single-threaded and has no counterpart. What is distributed here is PatternModel::train's order loop
(reference include/patternmodel.h:981-1270): windows never cross sentences, so each rank counts its own contiguous range of
sentences; the only cross-shard dependency is the GLOBAL count of each candidate pattern, needed before the threshold prune of
every order. Per order n:
    1. every rank counts its shard (HIP count kernel) and partitions its distinct candidates by owner = hash(key) % world;
    2. all-to-all of (key, local count) records [sizes first];
    3. each owner sums the exact global counts, applies the threshold and — after an all-gather of the survivor counts — hands
       out GLOBAL survivor ids and names the exporting rank of each survivor;
    4. all-to-all of the replies back; every rank tags its local table and writes survivor ids per position for order n+1.
Keys are exact 64-bit identities (order 1: the token bytes; order n: the two global survivor ids of the (n-1)-grams), so counts
from different ranks are summed exactly — no approximate filter, no merged collisions. Volume per order and rank: 12 B per local
distinct candidate out, 8 B back; xGMI is point-to-point (7 links per GPU), and an all-to-all uses all of them at once.
"""
import numpy as np

MAX_ORDER = 128
FAILED = (1 << 62) - 1  # a partition size no rank can have: "my local count failed" (ShardedTrainer._pass)


class ShardedTrainer:
    def __init__(self, engine, dist, torch, device=None):
        """engine: capi.HipShardEngine (GPU) or any object with the same methods (the CPU tests use a numpy stand-in);
        dist: an initialised torch.distributed; device: where exchange tensors live (None = CPU for gloo)."""
        self.engine, self.dist, self.torch = engine, dist, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device
        # gloo moves host memory only: device tensors are staged through the host (CPU tests, and 2 ranks sharing one GPU)
        self.stage_host = dist.get_backend() == "gloo"
        if self.stage_host:
            self.device = None

    # ---- collectives ------------------------------------------------------------------------------
    def _exchange_sizes(self, sizes):
        t = self.torch
        s = t.tensor(sizes, dtype=t.int64, device=self.device)
        r = t.empty(self.world, dtype=t.int64, device=self.device)
        self.dist.all_to_all_single(r, s)
        return [int(x) for x in r.tolist()]

    def _sync(self, tensor):
        """RCCL collectives are enqueued on torch's stream; the library works on its own HIP stream and must only see finished
        buffers (and vice versa: every colibri_shard_* call returns with its stream drained)."""
        if tensor.is_cuda:
            self.torch.cuda.current_stream(tensor.device).synchronize()

    def _all_to_all_v(self, tensor, send_sizes, recv_sizes, sync=True):
        """sync=False: another collective follows on the same stream before the library reads anything (one wait for the group instead of one each)"""
        home = tensor.device
        src = tensor.contiguous().cpu() if (self.stage_host and tensor.is_cuda) else tensor.contiguous()
        out = self.torch.empty(sum(recv_sizes), dtype=src.dtype, device=src.device)
        self.dist.all_to_all_single(out, src, recv_sizes, send_sizes)
        out = out.to(home) if out.device != home else out
        if sync:
            self._sync(out)
        return out

    def _all_reduce(self, tensor, op):
        """in-place all-reduce of a dense device array (RCCL ring over xGMI; staged through the host under gloo)"""
        if self.stage_host and tensor.is_cuda:
            h = tensor.cpu()
            self.dist.all_reduce(h, op=op)
            tensor.copy_(h)
        else:
            self.dist.all_reduce(tensor, op=op)
        self._sync(tensor)

    def _unigram_pass_dense(self, state):
        """Order 1 without any key exchange: when every rank's class encoding is canonical the class id is the unigram's identity, so
        the ranks all-reduce their dense per-class count arrays (SUM) and first-seen ranks (MIN) — the north star's "RCCL all-reduce of
        the per-bucket count tables before the prune" — and apply the reduced arrays locally. Returns None if not applicable."""
        eng = self.engine
        if not hasattr(eng, "uni_info"):
            return None
        ok, maxclass = eng.uni_info()
        everyone = self._all_gather_ints([int(ok), maxclass])
        if not all(v[0] for v in everyone):
            return None
        nclasses = max(v[1] for v in everyone) + 1
        cnt, mr = eng.uni_count(nclasses, self.rank)
        self._all_reduce(cnt, self.dist.ReduceOp.SUM)
        self._all_reduce(mr, self.dist.ReduceOp.MIN)
        found, kept, _ = eng.uni_apply(cnt, mr, nclasses, self.rank)
        state["gid_total"] = max(state["gid_total"], nclasses)  # unigram ids are class ids: later passes number from nclasses on
        return found, kept

    def _all_gather_ints(self, values):
        t = self.torch
        mine = t.tensor(values, dtype=t.int64, device=self.device)
        out = [t.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [[int(x) for x in o.tolist()] for o in out]

    # ---- one pass: local count -> exchange -> owner merge -> global ids back ---------------------------------
    def _pass(self, n, mask, level, state):
        """A rank whose engine raises (a radix bin or a result buffer outgrown: loud, `table_mode = 1` / more room fixes it) must not leave the others waiting in
        a collective: its failure travels in the exchanges the pass does anyway — the size all-to-all (a sentinel size) and the (found, kept) all-gather — and every
        rank raises."""
        eng = self.engine
        err = None
        try:
            ncand, per_owner = eng.count(n, mask, level)
        except Exception as e:  # noqa: BLE001 — reported on every rank below
            err, per_owner = e, [FAILED] * self.world
        recv_sizes = self._exchange_sizes(per_owner)
        if err is not None or FAILED in recv_sizes:
            raise RuntimeError(f"sharded pass (n={n}, mask={mask}, level={level}): local count failed on rank(s) "
                               f"{[r for r, v in enumerate(recv_sizes) if v == FAILED]}" + (f": {err}" if err is not None else ""))
        keys, cnts, aux = eng.send_buffers()
        rkeys = self._all_to_all_v(keys, per_owner, recv_sizes, sync=False)
        rcnts = self._all_to_all_v(cnts, per_owner, recv_sizes, sync=aux is None)
        raux = self._all_to_all_v(aux, per_owner, recv_sizes) if aux is not None else None  # distinct-source counts: indexed skipgram passes only
        try:
            found, kept = eng.merge(rkeys, rcnts, raux, recv_sizes)
        except Exception as e:  # noqa: BLE001
            err, found, kept = e, 0, 0
        everyone = self._all_gather_ints([found, kept, int(err is not None)])
        if any(v[2] for v in everyone):
            raise RuntimeError(f"sharded pass (n={n}, mask={mask}, level={level}): owner merge failed on rank(s) {[r for r, v in enumerate(everyone) if v[2]]}"
                               + (f": {err}" if err is not None else ""))
        found_all, kept_all = sum(v[0] for v in everyone), sum(v[1] for v in everyone)
        if found_all == 0:  # nothing anywhere: every rank sees it at once (reference "None found", patternmodel.h:1189-1194)
            return 0, 0
        base = state["gid_total"] + sum(everyone[r][1] for r in range(self.rank))
        if state["gid_total"] + kept_all >= (1 << 31):
            raise OverflowError("more than 2^31 surviving patterns")
        rgid, rtot = eng.reply(base)
        gid = self._all_to_all_v(rgid, recv_sizes, per_owner, sync=False)
        tot = self._all_to_all_v(rtot, recv_sizes, per_owner)
        eng.apply(gid, tot)
        state["gid_total"] += kept_all
        return found_all, kept_all

    def _skipgram_order(self, n, opt, state):
        """all gap masks of order n, each built level by level; returns (distinct skipgrams found, kept)"""
        found_n = kept_n = 0
        for mask in gap_masks(n, int(opt.maxskips)):
            levels = len(mask_parts(mask, n)) - 1
            f = k = 0
            for level in range(1, levels + 1):
                f, k = self._pass(n, mask, level, state)
                if f == 0:
                    break
            else:
                found_n += f
                kept_n += k
        return found_n, kept_n

    # ---- one training run -------------------------------------------------------------------------
    def train(self, opt):
        eng = self.engine
        eng.begin(opt, self.world)
        maxlength = min(int(opt.maxlength), MAX_ORDER - 1)
        found_g, kept_g = [0] * MAX_ORDER, [0] * MAX_ORDER
        state, maxn = {"gid_total": 0}, 0
        tokens_g = sum(v[0] for v in self._all_gather_ints([eng.local_tokens()]))
        for n in range(1, maxlength + 1):
            dense = self._unigram_pass_dense(state) if n == 1 else None
            if n == 1 and dense is None and int(opt.mintokens_unigrams) > max(1, int(opt.mintokens)):
                raise ValueError("MINTOKENS_UNIGRAMS > MINTOKENS in a sharded run needs the class-indexed order 1 (canonical class encoding on every rank)")
            found_all, kept_all = dense if dense is not None else self._pass(n, 0, 1, state)
            if found_all == 0:
                break
            maxn = n
            found_g[n], kept_g[n] = found_all, kept_all
            if opt.doskipgrams_exhaustive and n >= 3:  # every admissible window also counts its masked forms (patternmodel.h:1163-1171)
                f, k = self._skipgram_order(n, opt, state)
                found_g[n] += f
                kept_g[n] += k
            if kept_all == 0:  # nothing can be admitted at n + 1
                break
        if opt.doskipgrams and opt.indexed:  # IndexedPatternModel::trainskipgrams: from the surviving n-grams, n = 3.. (patternmodel.h:2969-3010)
            for n in range(3, min(maxlength, maxn) + 1):
                f, k = self._skipgram_order(n, opt, state)
                found_g[n] += f
                kept_g[n] += k
                if f == 0:
                    break
        return eng.finish(found_g, kept_g, tokens_g, maxn)


def gap_masks(n, maxskips):
    """bit i = token i is a gap; never at either end; at most `maxskips` separate gaps when n - 2 >= maxskips
    (reference src/algorithms.cpp:79-94)"""
    out = []
    if n < 3:
        return out
    for i in range(1, 1 << (n - 2)):
        mask = i << 1
        runs = sum(1 for k in range(n) if (mask >> k) & 1 and not (k and (mask >> (k - 1)) & 1))
        if n - 2 >= maxskips and runs > maxskips:
            continue
        out.append(mask)
    return out


def mask_parts(mask, n):
    """contiguous runs of non-gap tokens: [(first token, length)]"""
    parts, k = [], 0
    while k < n:
        if (mask >> k) & 1:
            k += 1
            continue
        e = k
        while e < n and not (mask >> e) & 1:
            e += 1
        parts.append((k, e - k))
        k = e
    return parts


def merge_exports(per_rank):
    """Union of the ranks' exports (in rank order): {key bytes: count} and, for indexed models, {key bytes: [(sentence, token)]}.
    Every pattern is exported by exactly one rank; a pattern's index is the concatenation of the ranks' local runs — rank order is
    sentence order, so the result is sorted."""
    patterns, dup = {}, 0
    for ex in per_rank:
        for g, kc in ex["patterns"].items():
            dup += g in patterns
            patterns[g] = kc
    if dup:
        raise ValueError(f"{dup} patterns were exported by more than one rank")
    counts = {k: c for k, c in patterns.values()}
    refs = None
    if per_rank and per_rank[0]["index"] is not None:
        refs = {k: [] for k, _ in patterns.values()}
        for ex in per_rank:
            for g, r in ex["index"].items():
                if g in patterns:  # ids of interned (intermediate) pairs never reach the index; kept for safety
                    refs[patterns[g][0]].extend(r)
    return counts, refs


def shard_payload(payload, world):
    """Split a v2 payload (header stripped) into `world` contiguous sentence ranges of about equal bytes.
    Returns [(bytes, first_sentence)], first_sentence being the 1-based global index of the shard's first sentence."""
    arr = np.frombuffer(payload, dtype=np.uint8)
    term = arr < 128
    prev_low = np.concatenate([[True], term[:-1]]) if arr.size else np.zeros(0, dtype=bool)
    delim_idx = np.flatnonzero((arr == 0) & prev_low)  # byte index of every sentence delimiter
    cuts = [0]
    for r in range(1, world):
        target = arr.size * r // world
        j = int(np.searchsorted(delim_idx, target))
        cut = int(delim_idx[j]) + 1 if j < delim_idx.size else arr.size
        cuts.append(max(cut, cuts[-1]))
    cuts.append(arr.size)
    out = []
    for r in range(world):
        first = 1 + int(np.searchsorted(delim_idx, cuts[r]))  # sentences before the cut + 1
        out.append((arr[cuts[r]: cuts[r + 1]].tobytes(), first))
    return out
