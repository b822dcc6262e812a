// mock_device.cpp — TEST INFRASTRUCTURE: a CPU stand-in for everything host/src/sharded.cpp talks to, so that the product's own multi-GPU driver — its rank threads,
// rendezvous, agreement steps, routing of sizes and buffers, termination ("None found"), failure paths — runs under `pytest -m "not gpu"` with world 2 and 4 and no GPU:
//   * the HIP runtime calls it makes (memory, copies, streams; "device memory" is host memory);
//   * the RCCL entry points it calls, between the ranks of one process (below): ranks on distinct "devices" take the trainer's RCCL back end — groups of sends and
//     receives, all-reduces, all-gathers, communicators from ncclCommInitAll or from a unique id — ranks that share "device 0" its device-copies back end;
//   * the C ABI of a device context as far as the key-sharded run uses it (include/colibri_hip.h: colibri_create ... colibri_kshard_*), computed on the CPU in the
//     protocol's own terms — records to the owner of their key, a bit per record and a number per surviving record back, exports to the lowest rank holding an
//     occurrence. The formats inside the buffers are this file's own (the driver moves bytes and sizes, it never looks inside);
//   * the candidate exchange (colibri_shard_*: every other model kind — exhaustive skipgrams, indexed models, indexed skipgrams — and every run the key-sharded
//     protocol hands back) in the terms include/colibri_hip.h states for it: a pass (n, mask, level) counts its keys locally, the distinct candidates travel to the
//     owner of their key with their local counts (and, at an indexed skipgram's last level, the number of its distinct source n-grams this rank exports), the owner
//     sums, prunes, numbers the survivors globally and names the lowest contributing rank exporter, the replies come back in the order the records left.
// Built into lib/libcolibri_sharded_mock.so together with the UNCHANGED sharded.cpp (host/Makefile, target `mock`); never linked into the product.
// Reference semantics restated: PatternModel::train's order loop, look-back, add, prune (reference include/patternmodel.h:1078-1245); the masked forms of every
// admitted window (:1163-1171, computeskipgrams :1370-1527 as executed: the validity test reduces to the window's look-back); IndexedPatternModel::trainskipgrams
// (:2969-3010) with the derived pruneskipgrams (:3362-3383: distinct skip contents = distinct source n-grams).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <vector>

#include "colibri_hip.h"

// ---- HIP: host memory behind the device API ------------------------------------------------------------------------------------------------------------------------------
extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 8; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "mock HIP error"; }
}  // extern "C"

// ---- RCCL between the ranks of ONE process: the calls host/src/sharded.cpp makes, carried out on the CPU when the LAST rank of the communicator arrives -----------------------------
// (an all-gather, an all-reduce, or a group of sends and receives is complete when the call returns — a stream is a formality here). What it checks that copies between
// contexts cannot: every send meets a receive of the same size on the peer it names, in the order both sides enqueued them; every rank of a collective passes the same count,
// type and operator; a communicator that was aborted wakes and fails everyone inside it. ncclCommInitAll serves the one-thread-per-rank trainer, ncclGetUniqueId +
// ncclCommInitRank the one-trainer-per-rank form (bench.py --gpus N: here one trainer per Python thread). COLIBRI_NO_RCCL=1 (the trainer's own switch) still selects the copies.
namespace {
struct P2P { const char* send; char* recv; size_t bytes; int peer; };
struct Posted;
struct Coll { int kind = 0; const void* send = nullptr; void* recv = nullptr; size_t count = 0; int dtype = 0, op = 0; };  // kind 1: all-reduce, 2: all-gather
struct CommGroup {
    std::mutex               m;
    std::condition_variable  cv;
    int                      n = 0, joined = 0, arrived = 0;
    uint64_t                 generation = 0;
    bool                     aborted = false, mismatch = false;
    std::map<std::pair<int, int>, std::deque<Posted*>> queue;  // (source, destination) -> the sends posted and not yet taken, in order
    std::vector<Coll>        coll;               // [rank]
    std::vector<std::vector<unsigned char>> tmp; // [rank]: an all-reduce's result before anyone's (in-place) buffer is overwritten
    // all ranks meet; the last one to arrive runs `work` (under the lock: the others are parked). false: the communicator was aborted or the ranks disagreed
    template <class F>
    bool meet(F work) {
        std::unique_lock<std::mutex> l(m);
        if (aborted) return false;
        const uint64_t g = generation;
        if (++arrived == n) {
            work();
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else if (!cv.wait_for(l, std::chrono::seconds(60), [&] { return generation != g || aborted; })) {
            if (std::getenv("COLIBRI_MOCK_RCCL_DEBUG")) std::fprintf(stderr, "mock RCCL: a collective of %d ranks saw only %d arrive\n", n, arrived);
            aborted = true;  // (a peer never came: a test must fail, not hang)
            cv.notify_all();
        }
        return !aborted && !mismatch;
    }
};
std::mutex                                          g_idm;
std::map<std::string, std::shared_ptr<CommGroup>>   g_groups;  // unique id -> the communicator its ranks are joining
uint64_t                                            g_next_id = 1;
size_t nccl_size(ncclDataType_t t) { return t == ncclUint64 || t == ncclInt64 || t == ncclFloat64 ? 8 : (t == ncclUint8 || t == ncclInt8) ? 1 : (t == ncclFloat16 || t == ncclBfloat16) ? 2 : 4; }
}  // namespace
struct ncclComm {
    std::shared_ptr<CommGroup> g;
    int                        rank = 0;
};
namespace {
struct Posted { ncclComm* cm; P2P op; bool is_send, done; };
}  // namespace
namespace {
thread_local std::vector<Posted> t_ops;  // what this thread enqueued inside its open group
thread_local bool                t_open = false;
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::lock_guard<std::mutex> l(g_idm);
    std::memset(id, 0, sizeof *id);
    const uint64_t v = g_next_id++;
    std::memcpy(id->internal, &v, sizeof v);
    std::memcpy(id->internal + 8, "colibri-mock-rccl", 17);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::shared_ptr<CommGroup> g;
    {
        std::lock_guard<std::mutex> l(g_idm);
        auto& slot = g_groups[std::string(id.internal, sizeof id.internal)];
        if (!slot) {
            slot = std::make_shared<CommGroup>();
            slot->n = nranks;
            slot->coll.resize((size_t)nranks); slot->tmp.resize((size_t)nranks);
        }
        g = slot;
    }
    if (g->n != nranks) return ncclInvalidArgument;
    std::unique_lock<std::mutex> l(g->m);
    ++g->joined;
    g->cv.notify_all();
    if (!g->cv.wait_for(l, std::chrono::seconds(60), [&] { return g->joined >= g->n; })) return ncclSystemError;  // (the real call blocks until every rank has joined, too)
    *out = new ncclComm{g, rank};
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int*) {
    auto g = std::make_shared<CommGroup>();
    g->n = g->joined = ndev;
    g->coll.resize((size_t)ndev); g->tmp.resize((size_t)ndev);
    for (int r = 0; r < ndev; ++r) comms[r] = new ncclComm{g, r};
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t cm) {
    if (!cm) return ncclInvalidArgument;
    {
        std::lock_guard<std::mutex> l(cm->g->m);
        cm->g->aborted = true;
        cm->g->cv.notify_all();
    }
    delete cm;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t cm) { delete cm; return ncclSuccess; }
ncclResult_t ncclGroupStart() {
    if (t_open) return ncclInvalidUsage;
    t_open = true;
    t_ops.clear();
    return ncclSuccess;
}
static ncclResult_t enqueue(ncclComm_t cm, const P2P& op, bool is_send) {
    if (!cm || op.peer < 0 || op.peer >= cm->g->n) return ncclInvalidArgument;
    if (!t_open) return ncclInvalidUsage;  // (the trainer never sends outside a group: a lone ncclSend would block on its peer)
    t_ops.push_back({cm, op, is_send, false});
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t cm, hipStream_t) { return enqueue(cm, P2P{(const char*)buf, nullptr, count * nccl_size(t), peer}, true); }
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t cm, hipStream_t) { return enqueue(cm, P2P{nullptr, (char*)buf, count * nccl_size(t), peer}, false); }
// point-to-point semantics: the group is complete when each of its sends was taken by the peer's matching receive and each of its receives was filled — only the pairs
// that talk to each other wait for each other (a rank with an empty shard posts nothing and waits for nobody)
ncclResult_t ncclGroupEnd() {
    if (!t_open) return ncclInvalidUsage;
    t_open = false;
    if (t_ops.empty()) return ncclSuccess;
    std::shared_ptr<CommGroup> g = t_ops[0].cm->g;
    for (const Posted& o : t_ops)
        if (o.cm->g != g) return ncclInvalidUsage;  // (one communicator per group is all the trainer needs)
    std::unique_lock<std::mutex> l(g->m);
    static const bool trace = std::getenv("COLIBRI_MOCK_RCCL_TRACE") != nullptr;
    for (Posted& o : t_ops) {
        if (o.is_send) g->queue[{o.cm->rank, o.op.peer}].push_back(&o);
        if (trace) std::fprintf(stderr, "T rank %d posts %s %zu peer %d (%p)\n", o.cm->rank, o.is_send ? "send" : "recv", o.op.bytes, o.op.peer, (void*)&o);
    }
    g->cv.notify_all();
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(60);
    for (;;) {
        if (g->aborted || g->mismatch) break;
        bool all = true;
        for (Posted& o : t_ops) {
            if (o.done) continue;
            if (!o.is_send) {
                auto& q = g->queue[{o.op.peer, o.cm->rank}];
                if (!q.empty()) {
                    Posted* sd = q.front();
                    q.pop_front();
                    if (sd->op.bytes != o.op.bytes) {
                        if (std::getenv("COLIBRI_MOCK_RCCL_DEBUG")) std::fprintf(stderr, "mock RCCL: rank %d receives %zu bytes from rank %d, which sends %zu\n", o.cm->rank, o.op.bytes, o.op.peer, sd->op.bytes);
                        g->mismatch = true;
                        break;
                    }  // the k-th receive posted for a peer takes the k-th send that peer posted for this rank
                    std::memcpy(o.op.recv, sd->op.send, o.op.bytes);
                    sd->done = o.done = true;
                    if (trace) std::fprintf(stderr, "T rank %d takes %zu from %d (%p)\n", o.cm->rank, o.op.bytes, o.op.peer, (void*)sd);
                    g->cv.notify_all();
                    continue;
                }
            }
            all = false;
        }
        if (g->mismatch) { g->cv.notify_all(); break; }
        all = std::all_of(t_ops.begin(), t_ops.end(), [](const Posted& o) { return o.done; });  // (a send to itself is taken by a receive later in the same scan)
        if (all) break;
        if (g->cv.wait_until(l, deadline) == std::cv_status::timeout) {  // (a peer never came: a test must fail, not hang)
            if (std::getenv("COLIBRI_MOCK_RCCL_DEBUG"))
                for (const Posted& o : t_ops)
                    if (!o.done) std::fprintf(stderr, "mock RCCL: rank %d still waits to %s %zu bytes %s rank %d\n", o.cm->rank, o.is_send ? "send" : "receive", o.op.bytes, o.is_send ? "to" : "from", o.op.peer);
            g->aborted = true;
            g->cv.notify_all();
            break;
        }
    }
    if (trace) std::fprintf(stderr, "T rank %d leaves its group\n", t_ops[0].cm->rank);
    const bool ok = !g->aborted && !g->mismatch;
    if (!ok)  // nobody may keep a pointer into this thread's list
        for (auto& kv : g->queue)
            for (auto it = kv.second.begin(); it != kv.second.end();) it = (std::find_if(t_ops.begin(), t_ops.end(), [&](const Posted& o) { return &o == *it; }) != t_ops.end()) ? kv.second.erase(it) : it + 1;
    l.unlock();
    t_ops.clear();
    return ok ? ncclSuccess : ncclInternalError;
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t cm, hipStream_t) {
    if (!cm) return ncclInvalidArgument;
    if (t != ncclUint32 || (op != ncclSum && op != ncclMin)) return ncclInvalidArgument;  // (what the trainer reduces: u32 counts by SUM, ranks by MIN)
    const std::shared_ptr<CommGroup> keep = cm->g;  // (the communicator may be aborted — deleted — by a failing peer while this rank is parked)
    CommGroup&                       g    = *keep;
    { std::lock_guard<std::mutex> l(g.m); g.coll[(size_t)cm->rank] = Coll{1, send, recv, count, (int)t, (int)op}; }
    const bool ok = g.meet([&] {
        for (int r = 0; r < g.n; ++r)
            if (g.coll[(size_t)r].kind != 1 || g.coll[(size_t)r].count != count || g.coll[(size_t)r].op != (int)op) { g.mismatch = true; return; }
        std::vector<uint32_t> red((const uint32_t*)g.coll[0].send, (const uint32_t*)g.coll[0].send + count);
        for (int r = 1; r < g.n; ++r) {
            const uint32_t* o = (const uint32_t*)g.coll[(size_t)r].send;
            for (size_t j = 0; j < count; ++j) red[j] = op == ncclMin ? std::min(red[j], o[j]) : red[j] + o[j];
        }
        for (int r = 0; r < g.n; ++r) { std::memcpy(g.coll[(size_t)r].recv, red.data(), count * 4); g.coll[(size_t)r] = Coll(); }
    });
    return ok ? ncclSuccess : ncclInternalError;
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t cm, hipStream_t) {
    if (!cm) return ncclInvalidArgument;
    const std::shared_ptr<CommGroup> keep = cm->g;
    CommGroup&                       g    = *keep;
    const size_t                     nb   = count * nccl_size(t);
    { std::lock_guard<std::mutex> l(g.m); g.coll[(size_t)cm->rank] = Coll{2, send, recv, nb, (int)t, 0}; }
    const bool ok = g.meet([&] {
        for (int r = 0; r < g.n; ++r)
            if (g.coll[(size_t)r].kind != 2 || g.coll[(size_t)r].count != nb) { g.mismatch = true; return; }
        std::vector<unsigned char> all(nb * (size_t)g.n);
        for (int r = 0; r < g.n; ++r) std::memcpy(all.data() + nb * (size_t)r, g.coll[(size_t)r].send, nb);
        for (int r = 0; r < g.n; ++r) { std::memcpy(g.coll[(size_t)r].recv, all.data(), all.size()); g.coll[(size_t)r] = Coll(); }
    });
    return ok ? ncclSuccess : ncclInternalError;
}
const char* ncclGetErrorString(ncclResult_t e) { return e == ncclInternalError ? "mock RCCL: the communicator was aborted, a peer never arrived, or the ranks' calls do not match" : "mock RCCL error"; }
}

// ---- the device context, on the CPU ------------------------------------------------------------------------------------------------------------------------------------------
namespace {
struct Rec {  // what a source sends for one window (the mock's own record: 16 bytes)
    uint64_t key;
    uint32_t pad0, pad1;
};
struct Result {
    uint32_t pos, n, count, mask;  // mask: bit k set = token k of the window is a gap (a skipgram); 0: an n-gram
};
constexpr uint32_t INV = 0xFFFFFFFFu;
struct Own {  // one key on its owner: the sum over the ranks, the distinct source n-grams, the lowest rank that sent it, its global number
    uint32_t total, nsrc, minrank, gid;
};
struct ShardRun {  // the candidate exchange's state between its calls
    bool                               active = false;
    int                                world = 1, n = 0, level = 1;
    uint32_t                           thr = 2, thr_skip = 2, mask = 0, pthr = 2, minsrc = 0;
    bool                               final_level = true, use_aux = false;
    std::vector<std::vector<uint32_t>> ids;         // ids[n][position]: global number of the surviving n-gram that starts there, INV: none
    std::vector<uint32_t>              scratch[2];  // numbers of a skipgram's left part so far, per position
    std::vector<uint32_t>              mark;        // bit n: this rank exports the n-gram whose representative occurrence starts here
    std::vector<uint32_t>*             out = nullptr;
    std::vector<uint64_t>              keyof;                        // this pass's key per position, ~0: no candidate
    std::vector<uint64_t>              pkeys;                        // distinct candidates grouped by owner ...
    std::vector<uint32_t>              pcounts, paux, prep;          // ... their local counts, distinct-source counts, first positions
    std::vector<uint64_t>              rkeys;                        // owner side: the records as they arrived
    std::vector<uint32_t>              rsrc;
    std::map<uint64_t, Own>            owner;
    uint64_t                           admitted_n[COLIBRI_MAX_ORDER] = {0};
    std::vector<uint32_t>              res_gid;  // global number of every exported pattern (parallel to ctx->results)
    std::vector<std::pair<uint32_t, uint32_t>> pairs;  // indexed models: (global number, position) of every local occurrence of a surviving pattern
    // the local forward index once finished
    std::vector<uint32_t> ugid, ref_sentence;
    std::vector<uint64_t> uoff;
    std::vector<uint16_t> ref_token;
};
}  // namespace

struct colibri_ctx {
    std::string err;
    // corpus: one entry per position (a token, or a sentence delimiter with cls 0)
    std::vector<uint32_t> cls, bstart, blen;
    std::vector<uint8_t>  bytes;
    uint64_t              ntokens = 0, nsent = 0, maxclass = 0;
    colibri_options       opt{};
    int                   world = 1, rank = 0, n = 0;
    uint64_t              nclasses = 0;
    std::vector<uint32_t> cnt1, head;          // order 1's dense counts; a (zero) dense head, so that the driver's head all-reduce runs
    std::vector<uint64_t> cur, nxt;            // global number of the surviving (n-1)-gram at each position, ~0: none
    std::vector<Rec>      send, recv;          // records out (grouped by owner) / in (source after source)
    std::vector<uint32_t> sendpos, tab, tabr;  // position of every sent record; the (dummy) tables
    std::vector<uint32_t> sbase;               // first sent record of each owner's share
    std::vector<uint32_t> fb, fbr;             // feedback out / in
    std::vector<uint64_t> ex, exr;             // exports out / in
    std::vector<Result>   results;
    uint64_t              found[COLIBRI_MAX_ORDER] = {0}, kept[COLIBRI_MAX_ORDER] = {0}, admitted[COLIBRI_MAX_ORDER] = {0};
    colibri_stats         stats{};
    uint32_t              first_sentence = 1;
    uint64_t              kbase = 0;  // key-sharded run: the global number of the current order's survivor 0
    ShardRun              sh;
};

namespace {
int fail(colibri_ctx* c, int code, const char* msg) {
    if (c) c->err = msg;
    return code;
}
uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
}  // namespace

extern "C" {
int         colibri_abi_version(void) { return COLIBRI_ABI_VERSION; }
const char* colibri_last_error(const colibri_ctx* c) { return c ? c->err.c_str() : "null context"; }
int colibri_create(colibri_ctx** out, int) { *out = new colibri_ctx(); return COLIBRI_OK; }
void colibri_destroy(colibri_ctx* c) { delete c; }
void* colibri_stream(colibri_ctx*) { return nullptr; }
int colibri_kernel_time(const colibri_ctx*, int, double* ms, uint64_t* n) { if (ms) *ms = 0; if (n) *n = 0; return COLIBRI_OK; }

// .colibri.dat v2 payload: little-endian base-128, the high bit on every byte of a token but the last; 00 ends a sentence (reference src/classencoder.cpp:22-42, :550-600)
int colibri_upload_corpus(colibri_ctx* c, const uint8_t* p, uint64_t nbytes, uint32_t first_sentence) {
    c->first_sentence = first_sentence;
    c->cls.clear(); c->bstart.clear(); c->blen.clear();
    c->bytes.assign(p, p + nbytes);
    c->ntokens = c->nsent = c->maxclass = 0;
    uint64_t i = 0;
    while (i < nbytes) {
        uint64_t j = i;
        uint32_t v = 0, shift = 0;
        while (j < nbytes && (p[j] & 0x80)) { v |= (uint32_t)(p[j] & 0x7F) << shift; shift += 7; ++j; }
        if (j >= nbytes) break;  // (an unterminated tail: no token)
        v |= (uint32_t)p[j] << shift;
        c->cls.push_back(v); c->bstart.push_back((uint32_t)i); c->blen.push_back((uint32_t)(j + 1 - i));
        if (v == 0 && j == i) ++c->nsent; else { ++c->ntokens; c->maxclass = std::max<uint64_t>(c->maxclass, v); }
        i = j + 1;
    }
    return COLIBRI_OK;
}
int colibri_corpus_info(const colibri_ctx* c, uint64_t* ntokens, uint64_t* nsent, uint64_t* maxclass) {
    if (ntokens) *ntokens = c->ntokens;
    if (nsent) *nsent = c->nsent;
    if (maxclass) *maxclass = c->maxclass;
    return COLIBRI_OK;
}

int colibri_kshard_info(colibri_ctx* c, const colibri_options* o, int* eligible, uint64_t* maxclass, uint64_t* npos) {
    *eligible = !o->doskipgrams && !o->doskipgrams_exhaustive && !std::getenv("COLIBRI_NO_KSHARD");  // (plain and indexed models, as on the device)
    *maxclass = c->maxclass;
    *npos     = c->cls.size();
    return COLIBRI_OK;
}
int colibri_kshard_begin(colibri_ctx* c, const colibri_options* o, int world, int rank, uint64_t maxclass_global, uint64_t) {
    c->opt = *o; c->world = world; c->rank = rank; c->n = 0; c->nclasses = maxclass_global + 1;
    c->results.clear();
    c->sh    = ShardRun();  // (an indexed run names its patterns and keeps its forward index the way a candidate-exchange run does: global numbers)
    c->kbase = c->nclasses;  // a unigram's global number is its class id; order 2's survivors follow
    std::fill(std::begin(c->found), std::end(c->found), 0); std::fill(std::begin(c->kept), std::end(c->kept), 0); std::fill(std::begin(c->admitted), std::end(c->admitted), 0);
    return COLIBRI_OK;
}
int colibri_kshard_uni_count(colibri_ctx* c, void** cnt_dev, uint32_t* nclasses) {
    c->cnt1.assign(c->nclasses, 0);
    for (uint32_t v : c->cls) if (v) ++c->cnt1[v];
    *cnt_dev = c->cnt1.data(); *nclasses = (uint32_t)c->nclasses; c->n = 1;
    return COLIBRI_OK;
}
int colibri_kshard_uni_apply(colibri_ctx* c) {  // cnt1 holds the global counts now: the same survivors on every rank; rank 0 exports the unigrams
    const uint32_t thr = (uint32_t)c->opt.mintokens, wthr = std::max<uint32_t>(thr, (uint32_t)std::max(0, c->opt.mintokens_unigrams));
    c->cur.assign(c->cls.size(), ~0ull);
    std::vector<uint32_t> first(c->nclasses, ~0u);
    for (size_t i = 0; i < c->cls.size(); ++i) {
        const uint32_t v = c->cls[i];
        if (!v) continue;
        if (first[v] == ~0u) first[v] = (uint32_t)i;
        if (c->cnt1[v] >= wthr) c->cur[i] = v;  // (a longer window needs every word at the word threshold)
    }
    for (uint64_t v = 1; v < c->nclasses; ++v) {
        if (!c->cnt1[v]) continue;
        if (c->rank == 0) ++c->found[1];
        if (c->cnt1[v] >= thr && c->rank == 0) ++c->kept[1];
    }
    // a unigram is exported by the lowest rank that holds it — the mock keeps that simple: every rank remembers where it saw each class, rank 0 exports what it holds,
    // the others what rank 0 cannot name... which needs an exchange the protocol does not have. The product exports unigrams from their class ids (no position needed);
    // the mock does the same through a synthetic result whose key bytes are rebuilt from the class id.
    if (c->rank == 0)
        for (uint64_t v = 1; v < c->nclasses; ++v)
            if (c->cnt1[v] >= thr) { c->results.push_back({(uint32_t)v, 0u /* n = 0: a class id, not a position */, c->cnt1[v], 0u}); c->sh.res_gid.push_back((uint32_t)v); }
    for (size_t i = 0; i < c->cls.size(); ++i) {
        c->admitted[1] += c->cls[i] != 0;
        if (c->opt.indexed && c->cls[i] && c->cnt1[c->cls[i]] >= thr) c->sh.pairs.push_back({c->cls[i], (uint32_t)i});
    }
    return COLIBRI_OK;
}
int colibri_kshard_emit(colibri_ctx* c, int n, uint64_t, uint64_t, int, void** send_dev, void** tab_dev, uint32_t* tab_words, uint64_t* per_owner, uint32_t* recbytes, void** head_dev,
                        uint64_t* admitted) {
    if (n != c->n + 1) return fail(c, COLIBRI_ERR_STATE, "mock: colibri_kshard_emit out of order");
    c->n = n;
    std::vector<std::vector<Rec>>      by((size_t)c->world);
    std::vector<std::vector<uint32_t>> bp((size_t)c->world);
    uint64_t                           adm = 0;
    const size_t                       P = c->cls.size();
    for (size_t i = 0; i + (size_t)n <= P; ++i) {
        if (c->cur[i] == ~0ull || c->cur[i + 1] == ~0ull) continue;  // the look-back (reference :1139-1152): both (n-1)-grams survived (which also keeps the window inside its sentence)
        const uint64_t key = n == 2 ? ((uint64_t)c->cls[i] << 32 | c->cls[i + 1]) : (c->cur[i] << 24 | c->cls[i + n - 1]);  // exact: (number of the leading (n-1)-gram, last class)
        const int      d   = (int)(mix(key) % (uint64_t)c->world);
        by[(size_t)d].push_back({key, 0u, 0u});
        bp[(size_t)d].push_back((uint32_t)i);
        ++adm;
    }
    c->send.clear(); c->sendpos.clear(); c->sbase.assign((size_t)c->world + 1, 0);
    for (int d = 0; d < c->world; ++d) {
        c->sbase[(size_t)d] = (uint32_t)c->send.size();
        per_owner[d]        = by[(size_t)d].size();
        c->send.insert(c->send.end(), by[(size_t)d].begin(), by[(size_t)d].end());
        c->sendpos.insert(c->sendpos.end(), bp[(size_t)d].begin(), bp[(size_t)d].end());
    }
    c->sbase[(size_t)c->world] = (uint32_t)c->send.size();
    c->send.push_back({0, 0, 0});  // (never an empty buffer)
    c->tab.assign((size_t)c->world, 7u);
    c->admitted[n] = adm;
    *send_dev = c->send.data(); *tab_dev = c->tab.data(); *tab_words = 1; *recbytes = sizeof(Rec); *admitted = adm;
    if (n == 2) { c->head.assign(8192, 0u); for (int k = 4096; k < 8192; ++k) c->head[(size_t)k] = 0x7FFFFFFFu; *head_dev = c->head.data(); } else *head_dev = nullptr;
    return COLIBRI_OK;
}
int colibri_kshard_recv_buffers(colibri_ctx* c, uint64_t nrecords, void** recv_dev, void** tab_recv_dev) {
    c->recv.assign(nrecords + 1, Rec{0, 0, 0});
    c->tabr.assign((size_t)c->world, 0u);
    *recv_dev = c->recv.data(); *tab_recv_dev = c->tabr.data();
    return COLIBRI_OK;
}
int colibri_kshard_count(colibri_ctx* c, int n, const uint64_t* per_src, int more, void** fb_dev, uint64_t* fb_per_dst, uint32_t* fb_bytes, void** ex_dev, uint64_t* ex_per_dst,
                         uint64_t* kept_bins) {
    for (int s = 0; s < c->world; ++s)
        if (c->tabr[(size_t)s] != 7u) return fail(c, COLIBRI_ERR_STATE, "mock: a table did not arrive");  // (the driver moved every source's table row)
    struct Agg { uint32_t count, first; };
    std::map<uint64_t, Agg> m;  // the owner's add (reference :2059-2073): one entry per distinct key, its first record = lowest source rank, lowest index
    uint64_t                tot = 0;
    for (int s = 0; s < c->world; ++s) tot += per_src[s];
    for (uint64_t j = 0; j < tot; ++j) {
        auto it = m.find(c->recv[j].key);
        if (it == m.end()) m[c->recv[j].key] = {1u, (uint32_t)j}; else ++it->second.count;
    }
    const uint32_t thr = (uint32_t)c->opt.mintokens;
    std::map<uint64_t, uint32_t> number;  // prune (reference :2107-2128): the survivors, numbered densely
    std::vector<uint64_t>        rb((size_t)c->world + 1, 0);
    for (int s = 0; s < c->world; ++s) rb[(size_t)s + 1] = rb[(size_t)s] + per_src[s];
    std::vector<std::vector<uint64_t>> exd((size_t)c->world);
    for (auto& kv : m) {
        ++c->found[n];
        if (kv.second.count < thr) continue;
        number[kv.first] = (uint32_t)number.size();
        ++c->kept[n];
        int s = 0;
        while (kv.second.first >= rb[(size_t)s + 1]) ++s;
        exd[(size_t)s].push_back((uint64_t)(kv.second.first - rb[(size_t)s]) | ((uint64_t)kv.second.count << 32));
    }
    c->fb.clear(); c->ex.clear();
    for (int s = 0; s < c->world; ++s) {
        fb_per_dst[s] = 0;
        if (more || c->opt.indexed) {  // (an indexed model needs every survivor's number at its last order too: its references are keyed by it) per source, in the order it sent: one bit per record, then the numbers of the surviving records' keys
            const size_t          at = c->fb.size(), nw = (per_src[s] + 31) / 32;
            std::vector<uint32_t> codes;
            c->fb.resize(at + nw, 0u);
            for (uint64_t j = 0; j < per_src[s]; ++j) {
                auto it = number.find(c->recv[rb[(size_t)s] + j].key);
                if (it == number.end()) continue;
                c->fb[at + j / 32] |= 1u << (j % 32);
                codes.push_back(it->second);
            }
            c->fb.insert(c->fb.end(), codes.begin(), codes.end());
            fb_per_dst[s] = nw + codes.size();
        }
        ex_per_dst[s] = exd[(size_t)s].size();
        c->ex.insert(c->ex.end(), exd[(size_t)s].begin(), exd[(size_t)s].end());
    }
    c->fb.push_back(0); c->ex.push_back(0);
    *kept_bins = number.size(); *fb_dev = c->fb.data(); *fb_bytes = 4; *ex_dev = c->ex.data();
    return COLIBRI_OK;
}
int colibri_kshard_head_windows(const colibri_ctx*, uint64_t* windows) { *windows = 0; return COLIBRI_OK; }  // (the stand-in has no dense head: every window is a record)
int colibri_kshard_feedback_buffers(colibri_ctx* c, uint64_t nfb, uint64_t nex, void** fb_recv, void** ex_recv) {
    c->fbr.assign(nfb + 1, 0u); c->exr.assign(nex + 1, 0ull);
    *fb_recv = c->fbr.data(); *ex_recv = c->exr.data();
    return COLIBRI_OK;
}
int colibri_kshard_apply(colibri_ctx* c, int n, const uint64_t* fb_src, const uint64_t* ex_src, const uint64_t* kept_per_owner, int more, uint64_t* ids_global) {
    uint64_t gb = 0, off = 0, eo = 0;
    c->nxt.assign(c->cls.size(), ~0ull);
    for (int d = 0; d < c->world; ++d) {
        const uint32_t nd = c->sbase[(size_t)d + 1] - c->sbase[(size_t)d], nw = (nd + 31) / 32;
        if (more || c->opt.indexed) {
            if (fb_src[d] < nw) return fail(c, COLIBRI_ERR_STATE, "mock: short feedback");
            uint64_t ci = off + nw;
            for (uint32_t j = 0; j < nd; ++j)
                if (c->fbr[off + j / 32] >> (j % 32) & 1u) c->nxt[c->sendpos[c->sbase[(size_t)d] + j]] = gb + c->fbr[ci++];
            if (ci != off + fb_src[d]) return fail(c, COLIBRI_ERR_STATE, "mock: the feedback's numbers do not match its bits");
            off += fb_src[d];
        }
        for (uint64_t k = 0; k < ex_src[d]; ++k) {
            const uint64_t e = c->exr[eo + k];
            c->results.push_back({c->sendpos[c->sbase[(size_t)d] + (uint32_t)e], (uint32_t)n, (uint32_t)(e >> 32), 0u});
            if (c->opt.indexed) c->sh.res_gid.push_back((uint32_t)(c->kbase + c->nxt[c->results.back().pos]));  // (the owner's feedback numbered this window a moment ago)
        }
        eo += ex_src[d];
        gb += kept_per_owner[d];
    }
    if (gb >= (1ull << 40)) return fail(c, COLIBRI_ERR_OVERFLOW, "mock: too many survivors");
    if (c->opt.indexed)
        for (size_t i = 0; i < c->nxt.size(); ++i)
            if (c->nxt[i] != ~0ull) c->sh.pairs.push_back({(uint32_t)(c->kbase + c->nxt[i]), (uint32_t)i});
    c->kbase += gb;
    c->cur.swap(c->nxt);
    *ids_global = gb;
    return COLIBRI_OK;
}
}  // extern "C"
namespace {
// the local forward index: per global number (ascending) the occurrences in corpus order, as (sentence, token) — sentences numbered from the shard's first_sentence
void build_local_index(colibri_ctx* c) {
    ShardRun& sh = c->sh;
    std::sort(sh.pairs.begin(), sh.pairs.end());
    std::vector<uint32_t> sent(c->cls.size(), 0u), tok(c->cls.size(), 0u);
    uint32_t              sn = c->first_sentence, tn = 0;
    for (size_t i = 0; i < c->cls.size(); ++i) {
        sent[i] = sn; tok[i] = tn;
        if (c->cls[i] == 0 && c->blen[i] == 1) { ++sn; tn = 0; } else ++tn;
    }
    sh.ugid.clear(); sh.uoff.clear(); sh.ref_sentence.clear(); sh.ref_token.clear();
    for (size_t j = 0; j < sh.pairs.size(); ++j) {
        if (j == 0 || sh.pairs[j].first != sh.pairs[j - 1].first) { sh.ugid.push_back(sh.pairs[j].first); sh.uoff.push_back(j); }
        sh.ref_sentence.push_back(sent[sh.pairs[j].second]);
        sh.ref_token.push_back((uint16_t)tok[sh.pairs[j].second]);
    }
    sh.uoff.push_back(sh.pairs.size());
}
}  // namespace
extern "C" {
int colibri_kshard_local_stats(colibri_ctx* c, uint64_t* found, uint64_t* kept, uint64_t* admitted, uint32_t* syncs) {
    for (int n = 0; n < COLIBRI_MAX_ORDER; ++n) { found[n] = c->found[n]; kept[n] = c->kept[n]; admitted[n] = c->admitted[n]; }
    if (syncs) *syncs = 0;
    return COLIBRI_OK;
}
int colibri_kshard_finish(colibri_ctx* c, const uint64_t* found, const uint64_t* kept, const uint64_t* admitted, uint64_t tokens, int maxn, colibri_stats* out) {
    colibri_stats& s = c->stats;
    std::memset(&s, 0, sizeof s);
    s.totaltokens = tokens; s.nsentences = c->nsent; s.npatterns = c->results.size(); s.maxn = maxn; s.minn = maxn > 0 ? 1 : 0;
    for (int n = 1; n < COLIBRI_MAX_ORDER; ++n) {
        s.found[n] = n <= maxn ? found[n] : 0; s.kept[n] = n <= maxn ? kept[n] : 0; s.pruned[n] = s.found[n] - s.kept[n]; s.admitted[n] = n <= maxn ? admitted[n] : 0;
    }
    s.totaltypes = s.found[1];
    if (c->opt.indexed) { build_local_index(c); s.nrefs = c->sh.pairs.size(); }
    if (out) *out = s;
    return COLIBRI_OK;
}
static std::vector<uint8_t> mock_key(const colibri_ctx* c, const Result& r) {
    std::vector<uint8_t> k;
    if (r.n == 0) {  // a class id: its varint
        uint32_t v = r.pos;
        while (v >= 128) { k.push_back((uint8_t)(v & 0x7F) | 0x80); v >>= 7; }
        k.push_back((uint8_t)v);
        return k;
    }
    for (uint32_t t = 0; t < r.n; ++t) {
        if (r.mask >> t & 1u) k.push_back(3);  // a gapped token is the one byte 03 (reference src/pattern.cpp:886-908)
        else k.insert(k.end(), c->bytes.begin() + c->bstart[r.pos + t], c->bytes.begin() + c->bstart[r.pos + t] + c->blen[r.pos + t]);
    }
    return k;
}
int colibri_result_sizes(colibri_ctx* c, uint64_t* np, uint64_t* kb, uint64_t* nr) {
    uint64_t b = 0;
    for (const Result& r : c->results) b += mock_key(c, r).size();
    if (np) *np = c->results.size();
    if (kb) *kb = b;
    if (nr) *nr = 0;
    return COLIBRI_OK;
}
int colibri_export_unindexed(colibri_ctx* c, uint64_t* key_off, uint8_t* key_bytes, uint32_t* counts) {
    uint64_t b = 0;
    for (size_t j = 0; j < c->results.size(); ++j) {
        const auto k = mock_key(c, c->results[j]);
        key_off[j]   = b;
        std::memcpy(key_bytes + b, k.data(), k.size());
        b += k.size();
        counts[j] = c->results[j].count;
    }
    key_off[c->results.size()] = b;
    return COLIBRI_OK;
}

// ---- the candidate exchange (include/colibri_hip.h "colibri_shard_*"), on the CPU ------------------------------------------------------------------------------------------
namespace {
std::vector<std::pair<int, int>> parts_of(uint32_t mask, int n) {  // the contiguous non-gap stretches of a gap mask: (first token, tokens) (reference src/algorithms.cpp:33-54, complemented)
    std::vector<std::pair<int, int>> parts;
    for (int k = 0; k < n;) {
        if (mask >> k & 1u) { ++k; continue; }
        int e = k;
        while (e < n && !(mask >> e & 1u)) ++e;
        parts.push_back({k, e - k});
        k = e;
    }
    return parts;
}
uint32_t id_at(const std::vector<uint32_t>& v, size_t i) { return i < v.size() ? v[i] : INV; }
}  // namespace

int colibri_shard_begin(colibri_ctx* c, const colibri_options* o, int world) {
    if (!c || !o || world < 1 || world > 64) return COLIBRI_ERR_ARG;
    if (std::getenv("COLIBRI_MOCK_NO_CANDIDATES")) return fail(c, COLIBRI_ERR_UNSUPPORTED, "mock: the candidate exchange is switched off (COLIBRI_MOCK_NO_CANDIDATES)");
    c->opt = *o;
    if (c->opt.mintokens == -1) c->opt.mintokens = 2;  // (reference :883-888)
    if (c->opt.mintokens < 1) c->opt.mintokens = 1;
    if (c->opt.mintokens_skipgrams < c->opt.mintokens) c->opt.mintokens_skipgrams = c->opt.mintokens;
    ShardRun& sh = c->sh;
    sh           = ShardRun();
    sh.active    = true;
    sh.world     = world;
    sh.thr       = (uint32_t)c->opt.mintokens;
    sh.thr_skip  = (uint32_t)c->opt.mintokens_skipgrams;
    sh.mark.assign(c->cls.size() + 1, 0u);
    c->results.clear();
    return COLIBRI_OK;
}

// the local count of pass (n, mask, level); its distinct keys grouped by owner = mix(key) % world
int colibri_shard_count(colibri_ctx* c, int n, uint32_t mask, int level, uint64_t* ncandidates, uint64_t* per_owner) {
    if (!c || !ncandidates || !per_owner || n < 1 || n >= COLIBRI_MAX_ORDER) return COLIBRI_ERR_ARG;
    ShardRun& sh = c->sh;
    if (!sh.active) return fail(c, COLIBRI_ERR_STATE, "colibri_shard_begin first");
    const colibri_options& o = c->opt;
    const size_t           P = c->cls.size();
    if ((int)sh.ids.size() < n + 2) sh.ids.resize((size_t)n + 2);
    sh.n = n; sh.mask = mask; sh.level = level; sh.final_level = true; sh.use_aux = false; sh.pthr = sh.thr; sh.minsrc = 0;
    sh.keyof.assign(P, ~0ull);
    std::vector<std::pair<int, int>> parts;
    if (mask == 0) {
        sh.ids[(size_t)n].assign(P, INV);
        sh.out = &sh.ids[(size_t)n];
    } else {
        if (n < 3 || n > 31 || !(o.doskipgrams || o.doskipgrams_exhaustive)) return fail(c, COLIBRI_ERR_ARG, "skipgram pass needs 3 <= n <= 31 and a skipgram mode");
        parts = parts_of(mask, n);
        if (level < 1 || level >= (int)parts.size()) return fail(c, COLIBRI_ERR_ARG, "skipgram pass level out of range");
        sh.final_level = level + 1 == (int)parts.size();
        sh.out         = &sh.scratch[level & 1];
        if (!sh.final_level) sh.pthr = 1;  // an intermediate level only names pairs of numbers: everything survives, nothing is exported
        else if (o.doskipgrams) { sh.use_aux = true; sh.minsrc = o.minskiptypes > 1 ? (uint32_t)o.minskiptypes : 0u; }  // (reference :3000-3003, :3362-3383)
        else sh.pthr = o.minskiptypes > 1 ? sh.thr_skip : sh.thr;                                                       // (reference :1233-1243, :2167-2186)
    }
    struct Loc { uint32_t count, first, nsrc; };
    std::map<uint64_t, Loc> table;
    uint64_t                adm = 0;
    for (size_t i = 0; i < P; ++i) {
        uint64_t key;
        if (mask == 0 && n == 1) {
            if (!c->cls[i]) continue;
            key = c->cls[i];
        } else if (mask == 0) {  // the look-back (reference :1139-1152): both (n-1)-grams survived — which also keeps the window inside its sentence
            const uint32_t a = id_at(sh.ids[(size_t)n - 1], i), b = id_at(sh.ids[(size_t)n - 1], i + 1);
            if (a == INV || b == INV) continue;
            key = (uint64_t)a << 32 | b;
        } else {
            // indexed models: a masked form of every surviving n-gram (trainskipgrams); otherwise of every admitted window (:1163-1171)
            if (o.doskipgrams ? id_at(sh.ids[(size_t)n], i) == INV : (id_at(sh.ids[(size_t)n - 1], i) == INV || id_at(sh.ids[(size_t)n - 1], i + 1) == INV)) continue;
            const uint32_t l = level == 1 ? id_at(sh.ids[(size_t)parts[0].second], i + (size_t)parts[0].first) : id_at(sh.scratch[(level - 1) & 1], i);
            const uint32_t r = id_at(sh.ids[(size_t)parts[(size_t)level].second], i + (size_t)parts[(size_t)level].first);
            if (l == INV || r == INV) continue;
            key = (uint64_t)l << 32 | r;
        }
        ++adm;
        sh.keyof[i] = key;
        auto it     = table.find(key);
        if (it == table.end()) it = table.insert({key, Loc{0u, (uint32_t)i, 0u}}).first;
        ++it->second.count;
        if (sh.use_aux && (sh.mark[i] >> n & 1u)) ++it->second.nsrc;  // this occurrence is THE representative of an n-gram this rank exports: one more distinct filler
    }
    if (mask == 0) sh.admitted_n[n] = adm;
    if (mask != 0) sh.out->assign(P, INV);
    std::vector<std::vector<std::pair<uint64_t, Loc>>> by((size_t)sh.world);
    for (const auto& kv : table) by[(size_t)(mix(kv.first) % (uint64_t)sh.world)].push_back(kv);
    sh.pkeys.clear(); sh.pcounts.clear(); sh.paux.clear(); sh.prep.clear();
    for (int d = 0; d < sh.world; ++d) {
        per_owner[d] = by[(size_t)d].size();
        for (const auto& kv : by[(size_t)d]) { sh.pkeys.push_back(kv.first); sh.pcounts.push_back(kv.second.count); sh.paux.push_back(kv.second.nsrc); sh.prep.push_back(kv.second.first); }
    }
    *ncandidates = sh.pkeys.size();
    sh.pkeys.push_back(0); sh.pcounts.push_back(0); sh.paux.push_back(0);  // (never an empty buffer)
    return COLIBRI_OK;
}
int colibri_shard_send_view(colibri_ctx* c, void** keys, void** counts, void** aux) {
    if (!c || !keys || !counts) return COLIBRI_ERR_ARG;
    if (!c->sh.active) return fail(c, COLIBRI_ERR_STATE, "colibri_shard_begin first");
    *keys = c->sh.pkeys.data(); *counts = c->sh.pcounts.data();
    if (aux) *aux = c->sh.use_aux ? (void*)c->sh.paux.data() : nullptr;
    return COLIBRI_OK;
}
int colibri_shard_send(colibri_ctx* c, void* keys, void* counts, void* aux) {
    if (!c || !c->sh.active) return COLIBRI_ERR_STATE;
    const size_t m = c->sh.pkeys.size() - 1;
    if (m && (!keys || !counts)) return COLIBRI_ERR_ARG;
    if (m) { std::memcpy(keys, c->sh.pkeys.data(), 8 * m); std::memcpy(counts, c->sh.pcounts.data(), 4 * m); }
    if (m && aux) { if (c->sh.use_aux) std::memcpy(aux, c->sh.paux.data(), 4 * m); else std::memset(aux, 0, 4 * m); }
    return COLIBRI_OK;
}
// owner side: the records of every rank (source after source), summed per key; pruned by the pass's threshold and, for an indexed skipgram, its distinct fillers
int colibri_shard_merge(colibri_ctx* c, const void* keys, const void* counts, const void* aux, const uint64_t* per_src, uint64_t* found, uint64_t* kept) {
    if (!c || !per_src || !found || !kept) return COLIBRI_ERR_ARG;
    ShardRun& sh = c->sh;
    if (!sh.active) return fail(c, COLIBRI_ERR_STATE, "colibri_shard_begin first");
    uint64_t tot = 0;
    for (int s = 0; s < sh.world; ++s) tot += per_src[s];
    if (tot && (!keys || !counts)) return COLIBRI_ERR_ARG;
    if (sh.use_aux && tot && !aux) return fail(c, COLIBRI_ERR_ARG, "this pass needs the distinct-source counts (aux buffer)");
    sh.rkeys.assign((const uint64_t*)keys, (const uint64_t*)keys + tot);
    sh.rsrc.clear(); sh.owner.clear();
    for (int s = 0; s < sh.world; ++s) sh.rsrc.insert(sh.rsrc.end(), (size_t)per_src[s], (uint32_t)s);
    for (uint64_t j = 0; j < tot; ++j) {
        auto it = sh.owner.find(sh.rkeys[j]);
        if (it == sh.owner.end()) it = sh.owner.insert({sh.rkeys[j], Own{0u, 0u, INV, INV}}).first;
        else if (it->second.minrank == sh.rsrc[j]) return fail(c, COLIBRI_ERR_STATE, "mock: a rank sent one key twice");
        it->second.total += ((const uint32_t*)counts)[j];
        if (sh.use_aux) it->second.nsrc += ((const uint32_t*)aux)[j];
        it->second.minrank = std::min(it->second.minrank, sh.rsrc[j]);  // (sources arrive in rank order: the first one)
    }
    *found = sh.owner.size(); *kept = 0;
    for (const auto& kv : sh.owner) *kept += kv.second.total >= sh.pthr && kv.second.nsrc >= sh.minsrc;
    return COLIBRI_OK;
}
int colibri_shard_reply(colibri_ctx* c, uint32_t gid_base, void* reply_gid, void* reply_cnt) {
    if (!c) return COLIBRI_ERR_ARG;
    ShardRun& sh = c->sh;
    if (!sh.active) return fail(c, COLIBRI_ERR_STATE, "colibri_shard_begin first");
    if (!sh.rkeys.empty() && (!reply_gid || !reply_cnt)) return COLIBRI_ERR_ARG;
    uint32_t next = gid_base;
    for (auto& kv : sh.owner) kv.second.gid = (kv.second.total >= sh.pthr && kv.second.nsrc >= sh.minsrc) ? next++ : INV;
    for (size_t j = 0; j < sh.rkeys.size(); ++j) {
        const Own& e = sh.owner[sh.rkeys[j]];
        ((uint32_t*)reply_gid)[j] = e.gid == INV ? INV : (e.gid | (e.minrank == sh.rsrc[j] ? 0x80000000u : 0u));
        ((uint32_t*)reply_cnt)[j] = e.total;
    }
    return COLIBRI_OK;
}
// contributor side: the replies in the order the records left; the survivors' global numbers per position; the patterns this rank was named exporter of
int colibri_shard_apply(colibri_ctx* c, const void* reply_gid, const void* reply_cnt, uint64_t* exported, uint64_t* admitted) {
    if (!c) return COLIBRI_ERR_ARG;
    ShardRun& sh = c->sh;
    if (!sh.active || sh.n < 1) return fail(c, COLIBRI_ERR_STATE, "colibri_shard_apply out of order");
    const size_t m = sh.pkeys.size() - 1;
    if (m && (!reply_gid || !reply_cnt)) return COLIBRI_ERR_ARG;
    std::map<uint64_t, uint32_t> gmap;
    uint64_t                     k = 0;
    for (size_t j = 0; j < m; ++j) {
        const uint32_t g = ((const uint32_t*)reply_gid)[j];
        if (g == INV) continue;
        gmap[sh.pkeys[j]] = g & 0x7FFFFFFFu;
        if (!(g & 0x80000000u) || !sh.final_level) continue;
        c->results.push_back({sh.prep[j], (uint32_t)sh.n, ((const uint32_t*)reply_cnt)[j], sh.mask});
        sh.res_gid.push_back(g & 0x7FFFFFFFu);
        if (sh.mask == 0 && c->opt.doskipgrams && sh.n < 32) sh.mark[sh.prep[j]] |= 1u << sh.n;
        ++k;
    }
    for (size_t i = 0; i < sh.keyof.size(); ++i) {
        if (sh.keyof[i] == ~0ull) continue;
        auto it = gmap.find(sh.keyof[i]);
        if (it == gmap.end()) continue;
        (*sh.out)[i] = it->second;
        if (c->opt.indexed && sh.final_level) sh.pairs.push_back({it->second, (uint32_t)i});  // the forward index, keyed by GLOBAL number (reference :2789-2800)
    }
    if (exported) *exported = k;
    if (admitted) *admitted = sh.admitted_n[sh.n];
    return COLIBRI_OK;
}
// order 1 on class-indexed arrays (canonical encodings): local counts -> [all-reduce SUM / MIN by the caller] -> apply; a unigram's global number is its class id
int colibri_shard_uni_info(const colibri_ctx* c, int* eligible, uint64_t* maxclass) {
    if (!c || !eligible || !maxclass) return COLIBRI_ERR_ARG;
    *eligible = (std::getenv("COLIBRI_MOCK_NO_DENSE_UNIGRAMS") || c->maxclass >= (1u << 23)) ? 0 : 1;  // (wider ids: the keyed unigram pass — the product's bound is 2^28, a CPU test's is smaller)
    *maxclass = c->maxclass;
    return COLIBRI_OK;
}
int colibri_shard_uni_count(colibri_ctx* c, void* cnt, void* minrank, uint32_t nclasses, int rank) {
    if (!c || !cnt || !minrank || nclasses <= c->maxclass) return COLIBRI_ERR_ARG;
    if (!c->sh.active) return fail(c, COLIBRI_ERR_STATE, "colibri_shard_begin first");
    std::fill((uint32_t*)cnt, (uint32_t*)cnt + nclasses, 0u);
    std::fill((uint32_t*)minrank, (uint32_t*)minrank + nclasses, 0x7FFFFFFFu);
    for (uint32_t v : c->cls)
        if (v) { ++((uint32_t*)cnt)[v]; ((uint32_t*)minrank)[v] = (uint32_t)rank; }
    return COLIBRI_OK;
}
int colibri_shard_uni_apply(colibri_ctx* c, const void* cnt_g, const void* minrank_g, uint32_t nclasses, int rank, uint64_t* found, uint64_t* kept, uint64_t* exported) {
    if (!c || !cnt_g || !minrank_g || !found || !kept) return COLIBRI_ERR_ARG;
    ShardRun& sh = c->sh;
    if (!sh.active) return fail(c, COLIBRI_ERR_STATE, "colibri_shard_begin first");
    const uint32_t *cnt = (const uint32_t*)cnt_g, *mr = (const uint32_t*)minrank_g;
    const uint32_t  wthr = std::max<uint32_t>(sh.thr, (uint32_t)std::max(0, c->opt.mintokens_unigrams));  // (a longer window needs every word at the word threshold)
    if (sh.ids.size() < 3) sh.ids.resize(3);
    sh.ids[1].assign(c->cls.size(), INV);
    sh.n = 1; sh.mask = 0;
    *found = *kept = 0;
    uint64_t k = 0;
    for (uint32_t v = 1; v < nclasses; ++v) {
        if (!cnt[v]) continue;
        ++*found;
        if (cnt[v] < sh.thr) continue;
        ++*kept;
        if (mr[v] != (uint32_t)rank) continue;
        c->results.push_back({v, 0u /* a class id, not a position */, cnt[v], 0u});
        sh.res_gid.push_back(v);
        ++k;
    }
    for (size_t i = 0; i < c->cls.size(); ++i) {
        const uint32_t v = c->cls[i];
        if (!v) continue;
        ++sh.admitted_n[1];
        if (cnt[v] >= wthr) sh.ids[1][i] = v;
        if (cnt[v] >= sh.thr && c->opt.indexed) sh.pairs.push_back({v, (uint32_t)i});
    }
    if (exported) *exported = k;
    return COLIBRI_OK;
}
int colibri_shard_finish(colibri_ctx* c, const uint64_t* found, const uint64_t* kept, uint64_t tokens, int maxn, colibri_stats* out) {
    if (!c || !found || !kept) return COLIBRI_ERR_ARG;
    ShardRun& sh = c->sh;
    if (!sh.active) return fail(c, COLIBRI_ERR_STATE, "colibri_shard_begin first");
    colibri_stats& s = c->stats;
    std::memset(&s, 0, sizeof s);
    s.totaltokens = tokens; s.nsentences = c->nsent; s.npatterns = c->results.size(); s.maxn = maxn; s.minn = maxn > 0 ? 1 : 0;
    for (int n = 1; n < COLIBRI_MAX_ORDER; ++n) { s.found[n] = found[n]; s.kept[n] = kept[n]; s.pruned[n] = found[n] - kept[n]; s.admitted[n] = sh.admitted_n[n]; }
    s.totaltypes = s.found[1];
    build_local_index(c);
    s.nrefs   = c->opt.indexed ? sh.pairs.size() : 0;
    sh.active = false;
    if (out) *out = s;
    return COLIBRI_OK;
}
int colibri_shard_export_gids(colibri_ctx* c, uint32_t* gids) {
    if (!c || (!gids && !c->sh.res_gid.empty())) return COLIBRI_ERR_ARG;
    if (c->sh.res_gid.size() != c->results.size()) return fail(c, COLIBRI_ERR_STATE, "mock: no candidate-exchange run to name the patterns of");
    std::copy(c->sh.res_gid.begin(), c->sh.res_gid.end(), gids);
    return COLIBRI_OK;
}
int colibri_shard_index_sizes(const colibri_ctx* c, uint64_t* ngids, uint64_t* nrefs) {
    if (!c || !ngids || !nrefs) return COLIBRI_ERR_ARG;
    *ngids = c->sh.ugid.size(); *nrefs = c->sh.ref_sentence.size();
    return COLIBRI_OK;
}
int colibri_shard_export_index(colibri_ctx* c, uint32_t* gids, uint64_t* ref_off, uint32_t* ref_sentence, uint16_t* ref_token) {
    if (!c || !ref_off) return COLIBRI_ERR_ARG;
    const ShardRun& sh = c->sh;
    std::copy(sh.ugid.begin(), sh.ugid.end(), gids);
    if (sh.uoff.empty()) ref_off[0] = 0; else std::copy(sh.uoff.begin(), sh.uoff.end(), ref_off);
    std::copy(sh.ref_sentence.begin(), sh.ref_sentence.end(), ref_sentence);
    std::copy(sh.ref_token.begin(), sh.ref_token.end(), ref_token);
    return COLIBRI_OK;
}
// ---- single-device training and the queries on a resident model are not part of the mock --------------------------------------------------------------------------------------
static int mock_no_candidates(const colibri_ctx* c) {
    if (c) const_cast<colibri_ctx*>(c)->err = "mock: single-device training and resident-model queries are not mocked";
    return COLIBRI_ERR_UNSUPPORTED;
}
// what the C++ face's object file (colibri_host.o, linked for cut_sentences and friends) refers to beyond the sharded trainer: present, never reached in a mock run
int colibri_train(colibri_ctx* c, const colibri_options*, colibri_stats*) { return mock_no_candidates(c); }
int colibri_export_indexed(colibri_ctx* c, uint64_t*, uint8_t*, uint32_t*, uint64_t*, uint32_t*, uint16_t*) { return mock_no_candidates(c); }
int colibri_flexgrams(colibri_ctx* c, const uint64_t*, const uint8_t*, const uint64_t*, const uint32_t*, const uint16_t*, uint64_t, uint64_t*, uint64_t*, uint64_t*) { return mock_no_candidates(c); }
int colibri_flexgrams_fetch(colibri_ctx* c, uint64_t*, uint8_t*, uint32_t*, uint64_t*, uint32_t*, uint16_t*) { return mock_no_candidates(c); }
int colibri_flexgrams_resident(colibri_ctx* c, uint64_t*, uint64_t*, uint64_t*) { return mock_no_candidates(c); }
int colibri_set_constraint(colibri_ctx* c, const uint64_t*, const uint8_t*, uint64_t) { return mock_no_candidates(c); }
int colibri_set_continuation(colibri_ctx* c, const uint64_t*, const uint8_t*, uint64_t) { return mock_no_candidates(c); }
int colibri_set_filter(colibri_ctx* c, const uint64_t*, const uint8_t*, uint64_t) { return mock_no_candidates(c); }
}
