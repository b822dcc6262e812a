"""ctypes binding of libcolibri_hip.so (include/colibri_hip.h). No torch types cross this boundary.

The library is the product; this module only marshals numpy buffers into the C ABI. It fails loudly if
the shared library is missing (there is no Python or CPU fallback for the hot path).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(os.path.dirname(HERE))  # colibri-core_amd/
LIB_PATH = os.environ.get("COLIBRI_HIP_LIB") or os.path.join(PKG_ROOT, "lib", "libcolibri_hip.so")  # env override: kernel experiments only

MAX_ORDER = 128
K_TOKENISE, K_CLEAR, K_COUNT, K_PRUNE, K_RESOLVE, K_SKIPGRAM, K_INDEX, K_EXPORT, K_EMIT, K_SCATTER, K_BINCOUNT, K_EMIT2, K_LEVELB2, K_COUNT2, K_LISTS2 = range(15)
KERNEL_CLASSES = ["tokenise", "clear", "count", "prune", "resolve", "skipgram", "index", "export", "emit", "scatter", "bincount", "emit2", "levelB2", "count2", "lists2"]

EXPORTED = [
    "colibri_abi_version", "colibri_create", "colibri_destroy", "colibri_last_error", "colibri_upload_corpus",
    "colibri_upload_corpus_device", "colibri_corpus_info", "colibri_train", "colibri_result_sizes", "colibri_export_unindexed",
    "colibri_export_indexed", "colibri_hash_windows", "colibri_positions", "colibri_last_mode", "colibri_hash_keys", "colibri_kernel_time",
    "colibri_shard_begin", "colibri_shard_count", "colibri_shard_send", "colibri_shard_send_view", "colibri_shard_merge", "colibri_shard_reply", "colibri_shard_apply",
    "colibri_shard_finish", "colibri_shard_export_gids", "colibri_shard_index_sizes", "colibri_shard_export_index",
    "colibri_shard_uni_info", "colibri_shard_uni_count", "colibri_shard_uni_apply",
    "colibri_order2_records", "colibri_stream", "colibri_kshard_info", "colibri_kshard_begin", "colibri_kshard_uni_count", "colibri_kshard_uni_apply", "colibri_kshard_emit", "colibri_kshard_recv_buffers",
    "colibri_kshard_count", "colibri_kshard_head_windows", "colibri_kshard_feedback_buffers", "colibri_kshard_apply", "colibri_kshard_local_stats", "colibri_kshard_finish",
    "colibri_set_constraint", "colibri_set_continuation", "colibri_set_filter", "colibri_text_upload", "colibri_text_count", "colibri_text_words", "colibri_text_encode", "colibri_text_fetch", "colibri_text_as_corpus",
    "colibri_flexgrams", "colibri_flexgrams_resident", "colibri_flexgrams_fetch",
]


class Options(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "mintokens", "maxlength", "minlength", "maxbackofflength", "mintokens_unigrams", "mintokens_skipgrams", "minskiptypes",
        "maxskips", "doskipgrams", "doskipgrams_exhaustive", "dopatternperline", "prunenonsubsumed", "prunesubsumed", "indexed",
        "profile", "table_mode")]

    @classmethod
    def defaults(cls, **kw):
        """PatternModelOptions() defaults (reference include/patternmodel.h:153-180)."""
        o = cls(mintokens=-1, maxlength=100, minlength=1, maxbackofflength=100, mintokens_unigrams=1, mintokens_skipgrams=-1,
                minskiptypes=2, maxskips=3)
        for k, v in kw.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, int(v))
        return o


class Stats(C.Structure):
    _fields_ = [
        ("totaltokens", C.c_uint64), ("totaltypes", C.c_uint64), ("npatterns", C.c_uint64), ("keybytes", C.c_uint64),
        ("nrefs", C.c_uint64), ("nsentences", C.c_uint64), ("maxn", C.c_int32), ("minn", C.c_int32),
        ("windows", C.c_uint64 * MAX_ORDER), ("admitted", C.c_uint64 * MAX_ORDER), ("found", C.c_uint64 * MAX_ORDER),
        ("pruned", C.c_uint64 * MAX_ORDER), ("kept", C.c_uint64 * MAX_ORDER), ("train_ms", C.c_double),
        ("path", C.c_int32), ("fallback_reason", C.c_int32), ("retries", C.c_int32), ("reserved_", C.c_int32),  # ABI 4: which engines counted the run / why it was repeated
    ]


# colibri_stats.path bits / fallback_reason values (include/colibri_hip.h)
PATH_TABLE, PATH_RADIX, PATH_BI2, PATH_CHAIN, PATH_WIDE, PATH_SLICED, PATH_PER_PASS = 1, 2, 4, 8, 16, 32, 64
FALLBACK_NONE, FALLBACK_REGION, FALLBACK_BIN, FALLBACK_IDS, FALLBACK_ORDER2, FALLBACK_SPLIT, FALLBACK_CHAIN, FALLBACK_RESULTS, FALLBACK_PAIRS, FALLBACK_LDS_ORDER = 0, 1, 2, 3, 4, 8, 16, 32, 64, 128


SHARDED_LIB_PATH = os.path.join(PKG_ROOT, "lib", "libcolibri_sharded.so")
SHARDED_EXPORTED = [
    "colibri_sharded_unique_id", "colibri_sharded_create", "colibri_sharded_destroy", "colibri_sharded_last_error", "colibri_sharded_upload", "colibri_sharded_upload_split",
    "colibri_sharded_set_protocol", "colibri_sharded_train", "colibri_sharded_kernel_time", "colibri_sharded_result_sizes", "colibri_sharded_export_unindexed",
    "colibri_sharded_export_gids", "colibri_sharded_index_sizes", "colibri_sharded_export_index",
]


class ShardedInfo(C.Structure):
    _fields_ = [("protocol", C.c_int32), ("rccl", C.c_int32), ("host_lookups", C.c_uint32), ("pad", C.c_uint32), ("wall_ms", C.c_double),
                ("alltoall_bytes", C.c_uint64), ("alltoall_bytes_to_self", C.c_uint64), ("allreduce_bytes", C.c_uint64)]


class ColibriError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"colibri_hip status {code}: {msg}")
        self.code = code


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no fallback path.")
        L = C.CDLL(LIB_PATH)
        L.colibri_abi_version.restype = C.c_int
        L.colibri_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.colibri_destroy.argtypes = [C.c_void_p]
        L.colibri_destroy.restype = None
        L.colibri_last_error.argtypes = [C.c_void_p]
        L.colibri_last_error.restype = C.c_char_p
        L.colibri_text_upload.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        L.colibri_text_count.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.colibri_text_words.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.colibri_text_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.colibri_text_fetch.argtypes = [C.c_void_p, C.c_void_p]
        L.colibri_text_as_corpus.argtypes = [C.c_void_p, C.c_uint32]
        L.colibri_set_constraint.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.colibri_set_continuation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.colibri_set_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.colibri_flexgrams.argtypes = [C.c_void_p] * 6 + [C.c_uint64] + [C.POINTER(C.c_uint64)] * 3
        L.colibri_flexgrams_fetch.argtypes = [C.c_void_p] * 7
        L.colibri_flexgrams_resident.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 3
        L.colibri_upload_corpus.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
        L.colibri_upload_corpus_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
        L.colibri_corpus_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 3
        L.colibri_train.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(Stats)]
        L.colibri_result_sizes.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 3
        L.colibri_export_unindexed.argtypes = [C.c_void_p] * 4
        L.colibri_export_indexed.argtypes = [C.c_void_p] * 7
        L.colibri_hash_windows.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.colibri_positions.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.colibri_last_mode.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.colibri_hash_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.colibri_kernel_time.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.colibri_order2_records.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.colibri_shard_begin.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int]
        L.colibri_shard_count.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.c_void_p]
        L.colibri_shard_send.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.colibri_shard_send_view.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.colibri_shard_merge.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.colibri_shard_reply.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.colibri_shard_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.colibri_shard_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(Stats)]
        L.colibri_shard_uni_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
        L.colibri_shard_uni_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
        L.colibri_shard_uni_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.colibri_shard_export_gids.argtypes = [C.c_void_p, C.c_void_p]
        L.colibri_shard_index_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.colibri_shard_export_index.argtypes = [C.c_void_p] * 5
        _lib = L
    return _lib


class Context:
    """One device context = one corpus shard resident in HBM + its training state."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.colibri_create(C.byref(h), device)
        if rc != 0:
            raise ColibriError(rc, "colibri_create failed (no usable HIP device?)")
        self.h = h
        self.stats = None
        self.indexed = False

    def close(self):
        if self.h:
            self.L.colibri_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise ColibriError(rc, self.L.colibri_last_error(self.h).decode())

    # -- corpus ----------------------------------------------------------------------------------
    def upload(self, payload, first_sentence=1):
        """payload: bytes / numpy uint8 of the v2 corpus WITHOUT its A2 02 header."""
        buf = np.frombuffer(payload, dtype=np.uint8) if not isinstance(payload, np.ndarray) else payload
        self._keep = buf
        self._check(self.L.colibri_upload_corpus(self.h, buf.ctypes.data if buf.size else None, buf.size, first_sentence))

    def upload_device(self, dptr, nbytes, first_sentence=1):
        self._check(self.L.colibri_upload_corpus_device(self.h, C.c_void_p(dptr), nbytes, first_sentence))

    def corpus_info(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.L.colibri_corpus_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"tokens": a.value, "sentences": b.value, "maxclass": c.value}

    def positions(self):
        n = C.c_uint64()
        self._check(self.L.colibri_positions(self.h, C.byref(n)))
        return n.value

    def last_mode(self, with_passes=False):
        """1 = the last train() counted on the global table, 2 = on the radix path (and in how many passes over key slices its order 2 ran)"""
        p = C.c_int(1)
        m = self.L.colibri_last_mode(self.h, C.byref(p))
        return (m, p.value) if with_passes else m

    def set_constraint(self, keys):
        """colibri_set_constraint: the next train() calls only count patterns whose key bytes are in `keys` (an empty list lifts it)."""
        keys = list(keys)
        off = np.zeros(len(keys) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(k) for k in keys])
        blob = np.frombuffer(b"".join(keys) or b"\0", dtype=np.uint8)
        self._check(self.L.colibri_set_constraint(self.h, off.ctypes.data, blob.ctypes.data, len(keys)))

    def set_continuation(self, keys):
        """colibri_set_continuation: the next train() calls continue the model whose patterns' key bytes are `keys` — only the orders it has no
        n-grams of are counted, and the look-back finds its patterns (an empty list ends the mode). The results are the new patterns."""
        keys = list(keys)
        off = np.zeros(len(keys) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(k) for k in keys])
        blob = np.frombuffer(b"".join(keys) or b"\0", dtype=np.uint8)
        self._check(self.L.colibri_set_continuation(self.h, off.ctypes.data, blob.ctypes.data, len(keys)))

    def set_filter(self, keys):
        """colibri_set_filter: the next train() calls count only the windows that contain one of these n-grams or are an instance of one of these skipgrams
        (train(..., filter)); an empty list ends the mode."""
        keys = list(keys)
        off = np.zeros(len(keys) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(k) for k in keys])
        blob = np.frombuffer(b"".join(keys) or b"\0", dtype=np.uint8)
        self._check(self.L.colibri_set_filter(self.h, off.ctypes.data, blob.ctypes.data, len(keys)))

    # -- training --------------------------------------------------------------------------------
    def train(self, options=None, **kw):
        opt = options if options is not None else Options.defaults(**kw)
        st = Stats()
        self._check(self.L.colibri_train(self.h, C.byref(opt), C.byref(st)))
        self.stats = st
        self.indexed = bool(opt.indexed)
        return st

    def result_sizes(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.L.colibri_result_sizes(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def export_arrays(self):
        npat, kb, nrefs = self.result_sizes()
        key_off = np.zeros(npat + 1, dtype=np.uint64)
        key_bytes = np.zeros(max(1, kb), dtype=np.uint8)
        counts = np.zeros(max(1, npat), dtype=np.uint32)
        if not self.indexed:
            self._check(self.L.colibri_export_unindexed(self.h, key_off.ctypes.data, key_bytes.ctypes.data, counts.ctypes.data))
            return key_off, key_bytes[:kb], counts[:npat], None
        ref_off = np.zeros(npat + 1, dtype=np.uint64)
        ref_s = np.zeros(max(1, nrefs), dtype=np.uint32)
        ref_t = np.zeros(max(1, nrefs), dtype=np.uint16)
        self._check(self.L.colibri_export_indexed(self.h, key_off.ctypes.data, key_bytes.ctypes.data, counts.ctypes.data, ref_off.ctypes.data,
                                                  ref_s.ctypes.data, ref_t.ctypes.data))
        return key_off, key_bytes[:kb], counts[:npat], (ref_off, ref_s[:nrefs], ref_t[:nrefs])

    def export_dict(self):
        """Canonical form for parity checks: {key bytes: count} and, for indexed models, {key bytes: [(sentence, token)]}."""
        key_off, key_bytes, counts, refs = self.export_arrays()
        kb = key_bytes.tobytes()
        off = key_off.tolist()
        cd = {kb[off[j]: off[j + 1]]: int(c) for j, c in enumerate(counts.tolist())}
        rd = None
        if refs is not None:
            ref_off, rs, rt = refs
            ro = ref_off.tolist()
            rs, rt = rs.tolist(), rt.tolist()
            rd = {kb[off[j]: off[j + 1]]: list(zip(rs[ro[j]: ro[j + 1]], rt[ro[j]: ro[j + 1]])) for j in range(len(counts))}
        return cd, rd

    # -- flexgrams from skipgrams -------------------------------------------------
    def flexgrams(self, key_off, key_bytes, ref_off, ref_s, ref_t):
        """colibri_flexgrams + colibri_flexgrams_fetch on an indexed model in export layout; returns the flexgrams in the same
        layout: (key_off, key_bytes, counts, (ref_off, ref_sentence, ref_token))."""
        npat = len(key_off) - 1
        key_off = np.ascontiguousarray(key_off, dtype=np.uint64)
        ref_off = np.ascontiguousarray(ref_off, dtype=np.uint64)
        kb_in = np.ascontiguousarray(key_bytes, dtype=np.uint8) if len(key_bytes) else np.zeros(1, dtype=np.uint8)
        rs_in = np.ascontiguousarray(ref_s, dtype=np.uint32) if len(ref_s) else np.zeros(1, dtype=np.uint32)
        rt_in = np.ascontiguousarray(ref_t, dtype=np.uint16) if len(ref_t) else np.zeros(1, dtype=np.uint16)
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.L.colibri_flexgrams(self.h, key_off.ctypes.data, kb_in.ctypes.data, ref_off.ctypes.data, rs_in.ctypes.data, rt_in.ctypes.data, npat,
                                             C.byref(a), C.byref(b), C.byref(c)))
        return self._flexgrams_fetch(a.value, b.value, c.value)

    def flexgrams_resident(self):
        """colibri_flexgrams_resident: the same on the indexed model of the last train() of this context, without leaving the device."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.L.colibri_flexgrams_resident(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return self._flexgrams_fetch(a.value, b.value, c.value)

    def _flexgrams_fetch(self, nf, kb, nr):
        fo = np.zeros(nf + 1, dtype=np.uint64)
        fk = np.zeros(max(1, kb), dtype=np.uint8)
        fc = np.zeros(max(1, nf), dtype=np.uint32)
        fro = np.zeros(nf + 1, dtype=np.uint64)
        frs = np.zeros(max(1, nr), dtype=np.uint32)
        frt = np.zeros(max(1, nr), dtype=np.uint16)
        self._check(self.L.colibri_flexgrams_fetch(self.h, fo.ctypes.data, fk.ctypes.data, fc.ctypes.data, fro.ctypes.data, frs.ctypes.data, frt.ctypes.data))
        return fo, fk[:kb], fc[:nf], (fro, frs[:nr], frt[:nr])

    # -- parity / measurement hooks --------------------------------------------------------------
    def hash_windows(self, n):
        out = np.zeros(max(1, self.positions()), dtype=np.uint64)
        self._check(self.L.colibri_hash_windows(self.h, n, out.ctypes.data))
        return out[: self.positions()]

    def hash_keys(self, keys):
        off = np.zeros(len(keys) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(k) for k in keys])
        blob = np.frombuffer(b"".join(keys) or b"\0", dtype=np.uint8)
        out = np.zeros(max(1, len(keys)), dtype=np.uint64)
        self._check(self.L.colibri_hash_keys(self.h, blob.ctypes.data, off.ctypes.data, len(keys), out.ctypes.data))
        return out[: len(keys)]

    def kernel_time(self, cls):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.L.colibri_kernel_time(self.h, cls, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def order2_records(self):
        """(records bi2_count_kernel read, admitted bigram windows counted in the dense head instead) of the last plain run"""
        rec, head = C.c_uint64(), C.c_uint64()
        self._check(self.L.colibri_order2_records(self.h, C.byref(rec), C.byref(head)))
        return rec.value, head.value


class HipShardEngine:
    """Per-rank engine of the sentence-sharded trainer (colibri_amd.dist.ShardedTrainer): thin marshalling over the
    colibri_shard_* entry points. Exchange buffers are torch tensors on this rank's device (int64 / int32 views of the
    u64 / u32 payloads); the collectives themselves are done by the trainer."""

    def __init__(self, ctx, torch, device):
        self.ctx, self.torch, self.device = ctx, torch, device
        self.L = ctx.L

    def local_tokens(self):
        return self.ctx.corpus_info()["tokens"]

    def begin(self, opt, world):
        self.world = world
        self.ctx._check(self.L.colibri_shard_begin(self.ctx.h, C.byref(opt), world))
        self.ctx.indexed = False  # exports of a sharded run go through export_local()
        self.indexed = bool(opt.indexed)
        self.doskipgrams = bool(opt.doskipgrams)
        self.use_aux = False

    # -- order 1 on class-indexed arrays (dense all-reduce instead of a key exchange) -----------------
    def uni_info(self):
        ok, mc = C.c_int(), C.c_uint64()
        self.ctx._check(self.L.colibri_shard_uni_info(self.ctx.h, C.byref(ok), C.byref(mc)))
        return bool(ok.value), int(mc.value)

    def uni_count(self, nclasses, rank):
        t = self.torch
        cnt = t.empty(nclasses, dtype=t.int32, device=self.device)
        mr = t.empty(nclasses, dtype=t.int32, device=self.device)
        self.ctx._check(self.L.colibri_shard_uni_count(self.ctx.h, C.c_void_p(cnt.data_ptr()), C.c_void_p(mr.data_ptr()), nclasses, rank))
        return cnt, mr

    def uni_apply(self, cnt, mr, nclasses, rank):
        f, k, e = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.ctx._check(self.L.colibri_shard_uni_apply(self.ctx.h, C.c_void_p(cnt.data_ptr()), C.c_void_p(mr.data_ptr()), nclasses, rank, C.byref(f), C.byref(k), C.byref(e)))
        return int(f.value), int(k.value), int(e.value)

    def count(self, n, mask=0, level=1):
        nc = C.c_uint64()
        per = np.zeros(self.world, dtype=np.uint64)
        self.ctx._check(self.L.colibri_shard_count(self.ctx.h, n, mask, level, C.byref(nc), per.ctypes.data))
        self.ncand = int(nc.value)
        # the distinct-source counts only travel in the skipgram passes of indexed models (colibri_shard_count: use_aux); every rank
        # derives this from the same options, so all ranks agree on whether the third buffer is exchanged
        if mask and self.doskipgrams:  # ... and only at the last level of the mask (the earlier ones intern pairs of ids: nothing is pruned there)
            parts, k = 0, 0
            while k < n:
                if (mask >> k) & 1:
                    k += 1
                    continue
                parts += 1
                while k < n and not (mask >> k) & 1:
                    k += 1
            self.use_aux = level == parts - 1
        else:
            self.use_aux = False
        return self.ncand, [int(x) for x in per]

    class _DeviceView:  # a torch tensor over device memory of the library (no copy): the CUDA array interface, which PyTorch-ROCm speaks too
        def __init__(self, ptr, n, typestr):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}

    _views_ok = True

    def send_buffers(self):
        t = self.torch
        if HipShardEngine._views_ok and self.ncand:
            try:
                k, cn, a = C.c_void_p(), C.c_void_p(), C.c_void_p()
                self.ctx._check(self.L.colibri_shard_send_view(self.ctx.h, C.byref(k), C.byref(cn), C.byref(a)))
                keys = t.as_tensor(HipShardEngine._DeviceView(k.value, self.ncand, "<i8"), device=self.device)
                cnts = t.as_tensor(HipShardEngine._DeviceView(cn.value, self.ncand, "<i4"), device=self.device)
                aux = t.as_tensor(HipShardEngine._DeviceView(a.value, self.ncand, "<i4"), device=self.device) if (self.use_aux and a.value) else None
                if self.use_aux and aux is None:
                    raise RuntimeError("no aux view")
                return keys, cnts, aux
            except Exception:  # noqa: BLE001 — a torch build without the interface: copy instead
                HipShardEngine._views_ok = False
        keys = t.empty(max(1, self.ncand), dtype=t.int64, device=self.device)
        cnts = t.empty(max(1, self.ncand), dtype=t.int32, device=self.device)
        aux = t.empty(max(1, self.ncand), dtype=t.int32, device=self.device) if self.use_aux else None
        self.ctx._check(self.L.colibri_shard_send(self.ctx.h, C.c_void_p(keys.data_ptr()), C.c_void_p(cnts.data_ptr()), C.c_void_p(aux.data_ptr()) if self.use_aux else None))
        return keys[: self.ncand], cnts[: self.ncand], (aux[: self.ncand] if self.use_aux else None)

    def merge(self, keys, cnts, aux, per_src):
        per = np.asarray(per_src, dtype=np.uint64)
        f, k = C.c_uint64(), C.c_uint64()
        self.nrecv = int(per.sum())
        self.ctx._check(self.L.colibri_shard_merge(self.ctx.h, C.c_void_p(keys.data_ptr()), C.c_void_p(cnts.data_ptr()), C.c_void_p(aux.data_ptr()) if aux is not None else None, per.ctypes.data,
                                                   C.byref(f), C.byref(k)))
        return int(f.value), int(k.value)

    def reply(self, gid_base):
        t = self.torch
        gid = t.empty(max(1, self.nrecv), dtype=t.int32, device=self.device)
        cnt = t.empty(max(1, self.nrecv), dtype=t.int32, device=self.device)
        self.ctx._check(self.L.colibri_shard_reply(self.ctx.h, gid_base, C.c_void_p(gid.data_ptr()), C.c_void_p(cnt.data_ptr())))
        return gid[: self.nrecv], cnt[: self.nrecv]

    def apply(self, gid, cnt):
        e, a = C.c_uint64(), C.c_uint64()
        self.ctx._check(self.L.colibri_shard_apply(self.ctx.h, C.c_void_p(gid.data_ptr()), C.c_void_p(cnt.data_ptr()), C.byref(e), C.byref(a)))
        return int(e.value), int(a.value)

    def finish(self, found_g, kept_g, tokens_g, maxn):
        f = np.zeros(MAX_ORDER, dtype=np.uint64)
        k = np.zeros(MAX_ORDER, dtype=np.uint64)
        f[: len(found_g)] = found_g
        k[: len(kept_g)] = kept_g
        st = Stats()
        self.ctx._check(self.L.colibri_shard_finish(self.ctx.h, f.ctypes.data, k.ctypes.data, tokens_g, maxn, C.byref(st)))
        self.ctx.stats = st
        return st

    def export_local(self):
        """This rank's share of the model: {"patterns": {gid: (key bytes, global count)}, "index": {gid: [(sentence, token)]} or None}."""
        key_off, key_bytes, counts, _ = self.ctx.export_arrays()
        gids = np.zeros(max(1, counts.size), dtype=np.uint32)
        self.ctx._check(self.L.colibri_shard_export_gids(self.ctx.h, gids.ctypes.data))
        kb, off = key_bytes.tobytes(), key_off.tolist()
        patterns = {int(g): (kb[off[j]: off[j + 1]], int(c)) for j, (g, c) in enumerate(zip(gids[: counts.size].tolist(), counts.tolist()))}
        index = None
        if self.indexed:
            ng, nr = C.c_uint64(), C.c_uint64()
            self.ctx._check(self.L.colibri_shard_index_sizes(self.ctx.h, C.byref(ng), C.byref(nr)))
            ug = np.zeros(max(1, ng.value), dtype=np.uint32)
            ro = np.zeros(ng.value + 1, dtype=np.uint64)
            rs = np.zeros(max(1, nr.value), dtype=np.uint32)
            rt = np.zeros(max(1, nr.value), dtype=np.uint16)
            self.ctx._check(self.L.colibri_shard_export_index(self.ctx.h, ug.ctypes.data, ro.ctypes.data, rs.ctypes.data, rt.ctypes.data))
            ro, rs, rt = ro.tolist(), rs.tolist(), rt.tolist()
            index = {int(g): list(zip(rs[ro[j]: ro[j + 1]], rt[ro[j]: ro[j + 1]])) for j, g in enumerate(ug[: ng.value].tolist())}
        return {"patterns": patterns, "index": index}


# ---- the multi-GPU trainer (include/colibri_sharded.h; host/src/sharded.cpp: C++ over the C ABI, RCCL linked directly) ---------------------------------
_shlib = None


def load_sharded():
    global _shlib
    if _shlib is None:
        path = os.environ.get("COLIBRI_SHARDED_LIB", SHARDED_LIB_PATH)  # (tests: lib/libcolibri_sharded_mock.so — the same driver over a CPU stand-in for the device layer)
        if "mock" not in os.path.basename(path):
            load()  # libcolibri_hip.so first (the trainer — and its test build with the fault hook, libcolibri_sharded_hooks.so — links it)
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        S = C.CDLL(path)
        S.colibri_sharded_unique_id.argtypes = [C.c_void_p]
        S.colibri_sharded_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        S.colibri_sharded_destroy.argtypes = [C.c_void_p]
        S.colibri_sharded_destroy.restype = None
        S.colibri_sharded_last_error.argtypes = [C.c_void_p]
        S.colibri_sharded_last_error.restype = C.c_char_p
        S.colibri_sharded_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32]
        S.colibri_sharded_upload_split.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
        S.colibri_sharded_set_protocol.argtypes = [C.c_void_p, C.c_int]
        S.colibri_sharded_train.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(Stats), C.POINTER(ShardedInfo)]
        S.colibri_sharded_kernel_time.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        S.colibri_sharded_result_sizes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        S.colibri_sharded_export_unindexed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        S.colibri_sharded_export_gids.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        S.colibri_sharded_index_sizes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        S.colibri_sharded_export_index.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _shlib = S
    return _shlib


def sharded_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    rc = load_sharded().colibri_sharded_unique_id(buf)
    if rc != 0:
        raise ColibriError(rc, "colibri_sharded_unique_id failed")
    return buf.raw


class ShardedTrainer:
    """`nlocal` ranks of a `world`-rank run held by this process: all of them (one host thread each), or one (one process per rank; `unique_id` from
    sharded_unique_id(), the same bytes in every process). devices: HIP ordinal per local rank (two ranks on one device exchange by device copies: tests)."""

    def __init__(self, world, nlocal=None, first_rank=0, devices=None, unique_id=None):
        self.S = load_sharded()
        self.world = world
        self.nlocal = world if nlocal is None else nlocal
        h = C.c_void_p()
        dev = (C.c_int * self.nlocal)(*devices) if devices is not None else None
        uid = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        rc = self.S.colibri_sharded_create(C.byref(h), world, self.nlocal, first_rank, dev, uid)
        if rc != 0:
            raise ColibriError(rc, "colibri_sharded_create failed (devices visible? RCCL?)")
        self.h = h
        self.stats = None
        self.info = None

    def close(self):
        if self.h:
            self.S.colibri_sharded_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise ColibriError(rc, (self.S.colibri_sharded_last_error(self.h) or b"").decode(errors="replace"))

    def upload(self, local_rank, payload, first_sentence=1):
        buf = np.frombuffer(payload, dtype=np.uint8)
        self._check(self.S.colibri_sharded_upload(self.h, local_rank, buf.ctypes.data if buf.size else None, buf.size, first_sentence))

    def upload_split(self, payload, first_sentence=1):
        buf = np.frombuffer(payload, dtype=np.uint8)
        self._check(self.S.colibri_sharded_upload_split(self.h, buf.ctypes.data if buf.size else None, buf.size, first_sentence))

    def set_protocol(self, protocol):
        """0: key-sharded counting where the run allows it (default); 1: always the candidate exchange"""
        self._check(self.S.colibri_sharded_set_protocol(self.h, protocol))

    def train(self, opt=None, **kw):
        opt = opt if opt is not None else Options.defaults(**kw)
        st, info = Stats(), ShardedInfo()
        self._check(self.S.colibri_sharded_train(self.h, C.byref(opt), C.byref(st), C.byref(info)))
        self.stats, self.info = st, info
        return st

    def kernel_time(self, cls, local_rank=0):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.S.colibri_sharded_kernel_time(self.h, local_rank, cls, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def export_arrays(self, local_rank):
        npat, kb = C.c_uint64(), C.c_uint64()
        self._check(self.S.colibri_sharded_result_sizes(self.h, local_rank, C.byref(npat), C.byref(kb)))
        key_off = np.zeros(npat.value + 1, dtype=np.uint64)
        key_bytes = np.zeros(max(1, kb.value), dtype=np.uint8)
        counts = np.zeros(max(1, npat.value), dtype=np.uint32)
        self._check(self.S.colibri_sharded_export_unindexed(self.h, local_rank, key_off.ctypes.data, key_bytes.ctypes.data, counts.ctypes.data))
        return key_off, key_bytes[: kb.value], counts[: npat.value]

    def export_local(self, local_rank):
        """one local rank's share of an indexed model, as HipShardEngine.export_local gives it: {"patterns": {global number: (key bytes, count)},
        "index": {global number: [(sentence, token)]}} — colibri_amd.dist.merge_exports over the ranks' shares (in rank order) is the model with its reference lists"""
        key_off, key_bytes, counts = self.export_arrays(local_rank)
        gids = np.zeros(max(1, counts.size), dtype=np.uint32)
        self._check(self.S.colibri_sharded_export_gids(self.h, local_rank, gids.ctypes.data))
        raw, off = key_bytes.tobytes(), key_off.tolist()
        patterns = {int(g): (raw[off[j]: off[j + 1]], int(c)) for j, (g, c) in enumerate(zip(gids[: counts.size].tolist(), counts.tolist()))}
        ng, nr = C.c_uint64(), C.c_uint64()
        self._check(self.S.colibri_sharded_index_sizes(self.h, local_rank, C.byref(ng), C.byref(nr)))
        ug, ro = np.zeros(max(1, ng.value), dtype=np.uint32), np.zeros(ng.value + 1, dtype=np.uint64)
        rs, rt = np.zeros(max(1, nr.value), dtype=np.uint32), np.zeros(max(1, nr.value), dtype=np.uint16)
        self._check(self.S.colibri_sharded_export_index(self.h, local_rank, ug.ctypes.data, ro.ctypes.data, rs.ctypes.data, rt.ctypes.data))
        ro, rs, rt = ro.tolist(), rs.tolist(), rt.tolist()
        return {"patterns": patterns, "index": {int(g): list(zip(rs[ro[j]: ro[j + 1]], rt[ro[j]: ro[j + 1]])) for j, g in enumerate(ug[: ng.value].tolist())}}

    def export_dict(self):
        """the union of the local ranks' shares: {key bytes: count} (every pattern is exported by exactly one rank: a duplicate raises)"""
        out = {}
        for r in range(self.nlocal):
            key_off, key_bytes, counts = self.export_arrays(r)
            raw = key_bytes.tobytes()
            for j in range(counts.size):
                k = raw[int(key_off[j]): int(key_off[j + 1])]
                if k in out:
                    raise ValueError(f"pattern {k.hex()} exported by more than one rank")
                out[k] = int(counts[j])
        return out
