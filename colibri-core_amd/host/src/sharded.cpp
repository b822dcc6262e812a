// sharded.cpp — PatternModel::train across the GPUs of one node: the corpus cut into contiguous sentence ranges, one device context per rank, RCCL (linked
// directly) for the exchange steps. The reference is single-threaded (include/patternmodel.h:880-1345 is one loop over one map); what is distributed is that loop's
// only cross-shard dependency: the GLOBAL count of a candidate pattern before the prune of each order (:1195-1245). Two protocols over the C ABI:
//   key-sharded counting (include/colibri_hip.h "colibri_kshard_*", csrc/kshard.hpp) — the plain unindexed n-gram model, 1 / 2 / 4 / 8 ranks: order 1 is an
//     all-reduce of the dense per-class count array; at every higher order the RECORDS of the windows travel to the rank that owns their key, are counted there by
//     the single-device kernels against the exact global threshold, and only the survivors' feedback returns. Two host look-ups per order, everything else enqueued
//     on one stream per rank (the library's, which RCCL is handed too);
//   candidate exchange (include/colibri_hip.h "colibri_shard_*") — every other model kind (skipgrams, indexed models) and every other rank count: local count at
//     threshold 1, distinct candidates to their owner, global ids back.
// A trainer lives as long as its corpus shards should stay resident in HBM: colibri_sharded_* (include/colibri_sharded.h) is its C face — what bench.py --gpus N
// and the tests drive — and device_train_sharded() is PatternModel::train's use of it (colibri-patternmodeller --gpus N).
// Ranks and processes: `nlocal == world`: one host thread per rank in this process (ncclCommInitAll; or, when ranks are made to share a device — the one-GPU test
// boxes, COLIBRI_DEVICES=0,0 — device-to-device copies between the contexts instead of RCCL); `nlocal == 1`: one process per rank (ncclCommInitRank from an id the
// caller distributed, e.g. over torch.distributed), host values exchanged through RCCL as well.
// A rank that fails reports through the exchange every step begins with, so that no peer is left waiting in a collective; a failure in between aborts the
// communicators (ncclCommAbort) and the rendezvous.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "algorithms.h"
#include "colibri_hip.h"
#include "colibri_sharded.h"
#include "patternmodel.h"

namespace colibri_host {
namespace {

int g_gpus = 0;  // 0: not set (COLIBRI_GPUS or 1)

struct Aborted {};
struct AgreedFailure : std::runtime_error {  // every rank of the run throws it at the same step (RankDriver::agree)
    using std::runtime_error::runtime_error;
};

// barrier of the rank threads; abort() releases everyone (a rank that failed must not leave the others waiting)
class Rendezvous {
    std::mutex              m;
    std::condition_variable cv;
    int                     parties, waiting = 0;
    uint64_t                generation = 0;
    bool                    aborted    = false;

  public:
    explicit Rendezvous(int w) : parties(w) {}
    void wait() {
        std::unique_lock<std::mutex> l(m);
        if (aborted) throw Aborted();
        const uint64_t g = generation;
        if (++waiting == parties) {
            waiting = 0;
            ++generation;
            cv.notify_all();
            return;
        }
        cv.wait(l, [&] { return generation != g || aborted; });
        if (generation == g) throw Aborted();
    }
    void abort() {
        std::lock_guard<std::mutex> l(m);
        aborted = true;
        cv.notify_all();
    }
    void reset() {
        std::lock_guard<std::mutex> l(m);
        aborted = false;
        waiting = 0;
    }
};

struct DevMem {  // a device buffer that only grows (belongs to the device current when it was first reserved)
    void*  p = nullptr;
    size_t n = 0;
    void*  reserve(size_t bytes) {
        if (bytes > n) {
            if (p) (void)hipFree(p);
            p = nullptr;
            n = 0;
            const size_t want = std::max<size_t>(bytes + bytes / 4, 256);
            if (hipMalloc(&p, want) != hipSuccess) throw std::runtime_error("hipMalloc of an exchange buffer failed");
            n = want;
        }
        return p;
    }
    ~DevMem() {
        if (p) (void)hipFree(p);
    }
};

struct RankExport {
    std::vector<uint64_t>      key_off;
    std::vector<unsigned char> key_bytes;
    std::vector<uint32_t>      counts, gids;
    std::vector<uint32_t>      ugid, ref_sentence;  // local forward index, keyed by global id
    std::vector<uint64_t>      ref_off;
    std::vector<uint16_t>      ref_token;
};

#define HIPCHK(call)                                                                                                          \
    do {                                                                                                                      \
        const hipError_t e_ = (call);                                                                                         \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#call) + ": " + hipGetErrorString(e_));                    \
    } while (0)
#define NCCLCHK(call)                                                                                                         \
    do {                                                                                                                      \
        const ncclResult_t e_ = (call);                                                                                       \
        if (e_ != ncclSuccess) throw std::runtime_error(std::string(#call) + ": " + ncclGetErrorString(e_));                  \
    } while (0)

struct A2A {  // one all-to-all of a group: send_n[p] elements to rank p (in rank order in `send`), recv_n[p] from rank p (in rank order in `recv`)
    const void*                  send;
    const std::vector<uint64_t>* send_n;
    void*                        recv;
    const std::vector<uint64_t>* recv_n;
    size_t                       elem;
};
struct Reduce {  // in-place all-reduce of n u32
    void*  buf;
    size_t n;
    bool   minimum;
};

struct Shared {
    const int                          world, nlocal, first_rank;
    Rendezvous                         rv;
    bool                               use_rccl = false;
    std::vector<ncclComm_t>            comms;   // [nlocal]
    std::vector<int>                   device;  // [nlocal]
    // thread back ends: what the ranks show each other between two barriers
    std::vector<std::vector<uint64_t>>              ints;      // all-gather slots
    std::vector<std::vector<A2A>>                   ops;       // [rank]: the all-to-alls of the running group
    std::vector<std::vector<std::vector<uint32_t>>> hostred;   // [rank][reduction]: host copies (copies back end)
    std::mutex                                      errm;
    std::string                                     error;
    bool                                            poisoned = false;  // the communicators were aborted (a rank failed inside or beside a collective): no further run on this trainer
    // held shared by a rank while it hands its communicator to RCCL (between comm() and the return of the enqueueing calls); fail() asks for it exclusively, for a
    // bounded time, before it aborts that communicator: a peer that has just fetched its handle gets to finish its (microseconds of) enqueueing, a peer that is
    // stuck inside RCCL waiting for the failed rank is aborted all the same once the wait runs out (that is what the abort is for)
    std::unique_ptr<std::shared_timed_mutex[]>      enqueueing;
    Shared(int w, int nl, int first)
        : world(w), nlocal(nl), first_rank(first), rv(nl), comms((size_t)nl, nullptr), device((size_t)nl, 0), ints((size_t)w), ops((size_t)w), hostred((size_t)w),
          enqueueing(new std::shared_timed_mutex[(size_t)nl]) {}
    bool threads() const { return nlocal == world; }
    void fail(const std::string& what) {
        std::vector<std::pair<ncclComm_t, int>> mine;  // every communicator is aborted exactly once: the first failing rank takes them all (its peers' collectives then
        {                                              // return errors and those ranks come here too, to find nothing left)
            std::lock_guard<std::mutex> l(errm);
            if (error.empty()) error = what;
            if (use_rccl) {
                for (int k = 0; k < nlocal; ++k)
                    if (comms[(size_t)k]) mine.emplace_back(comms[(size_t)k], k), comms[(size_t)k] = nullptr;
                poisoned = true;
            }
        }
        rv.abort();
        for (const auto& [cm, k] : mine) {  // peers inside (or about to enter) a collective return with an error instead of waiting for this rank
            const bool quiet = enqueueing[(size_t)k].try_lock_for(std::chrono::milliseconds(200));  // (comm() hands out nothing any more: whoever holds this fetched its handle before)
            (void)ncclCommAbort(cm);
            if (quiet) enqueueing[(size_t)k].unlock();
        }
    }
};

class RankDriver {
    Shared&      sh;
    const int    li, rank, world, dev;  // local index, global rank
    colibri_ctx* c      = nullptr;
    hipStream_t  stream = nullptr;      // the candidate-exchange protocol's own stream (the key-sharded one runs on the library's)
    DevMem       rkeys, rcnts, raux, rgid, rtot, gid, tot, ucnt, umr, gather_dev;
    uint64_t     gid_total = 0;
    uint64_t*    gather_host = nullptr;  // pinned: host values of the process back end's all-gather
    uint64_t     bytes_a2a = 0, bytes_self = 0, bytes_reduce = 0;  // what this rank put into all-to-alls (of which to itself) and all-reduces in the running train()

    void chk(int rc, const char* what) {
        if (rc != COLIBRI_OK) throw std::runtime_error(std::string(what) + ": " + (c ? colibri_last_error(c) : "no context") + " (status " + std::to_string(rc) + ")");
    }
    ncclComm_t comm() const {
        std::lock_guard<std::mutex> l(sh.errm);
        const ncclComm_t            cm = sh.comms[(size_t)li];
        if (!cm) throw Aborted();  // (another rank failed and took the communicators down)
        return cm;
    }

    // ---- collectives ----------------------------------------------------------------------------------------------------
    // host values of every rank, [rank][k]. Threads: through shared memory; one process per rank: an RCCL all-gather on `s`
    std::vector<std::vector<uint64_t>> all_gather(const std::vector<uint64_t>& mine, hipStream_t s) {
        if (sh.threads()) {
            sh.ints[(size_t)rank] = mine;
            sh.rv.wait();
            std::vector<std::vector<uint64_t>> all = sh.ints;
            sh.rv.wait();
            return all;
        }
        const size_t k = mine.size();
        if (!gather_host) HIPCHK(hipHostMalloc((void**)&gather_host, sizeof(uint64_t) * 4096, hipHostMallocDefault));
        if (k * (size_t)(world + 1) > 4096) throw std::runtime_error("all_gather: too many host values");
        uint64_t* const d = (uint64_t*)gather_dev.reserve(sizeof(uint64_t) * 4096);
        std::memcpy(gather_host, mine.data(), k * sizeof(uint64_t));
        HIPCHK(hipMemcpyAsync(d, gather_host, k * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        {
            std::shared_lock<std::shared_timed_mutex> enq(sh.enqueueing[(size_t)li]);
            NCCLCHK(ncclAllGather(d, d + k, k, ncclUint64, comm(), s));
        }
        HIPCHK(hipMemcpyAsync(gather_host + k, d + k, k * (size_t)world * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::vector<std::vector<uint64_t>> all((size_t)world);
        for (int r = 0; r < world; ++r) all[(size_t)r].assign(gather_host + k + (size_t)r * k, gather_host + k + (size_t)(r + 1) * k);
        return all;
    }
    // a group of all-to-alls and all-reduces. RCCL: enqueued on `s` (one ncclGroup: all seven xGMI links of a GPU at once), nothing waits unless `sync`.
    // Copies (ranks sharing a device): the sources must be complete (every caller has waited for its stream), the copies are waited for, then a barrier.
    void exchange(const std::vector<A2A>& a2a, const std::vector<Reduce>& reds, hipStream_t s, bool sync) {
        for (const A2A& op : a2a)
            for (int p = 0; p < world; ++p) {
                bytes_a2a += (*op.send_n)[(size_t)p] * op.elem;
                if (p == rank) bytes_self += (*op.send_n)[(size_t)p] * op.elem;
            }
        for (const Reduce& r : reds) bytes_reduce += r.n * sizeof(uint32_t);
        if (sh.use_rccl) {
            std::shared_lock<std::shared_timed_mutex> enq(sh.enqueueing[(size_t)li]);  // (Shared::fail)
            if (!a2a.empty()) {
                NCCLCHK(ncclGroupStart());
                for (const A2A& op : a2a) {
                    uint64_t so = 0, ro = 0;
                    for (int p = 0; p < world; ++p) {
                        const uint64_t sn = (*op.send_n)[(size_t)p], rn = (*op.recv_n)[(size_t)p];
                        if (p == rank) {
                            // the rank's own share never meets a link: a device copy on the same stream (round 6; a send / receive pair to oneself runs inside RCCL's
                            // generic kernel on a few CUs — 0.30 ms for the 340 MB of a one-rank step's keys against 0.14 ms)
                            if (sn != rn) throw std::runtime_error("exchange: a rank's share for itself differs between its send and receive counts");
                            if (sn) HIPCHK(hipMemcpyAsync((char*)op.recv + ro * op.elem, (const char*)op.send + so * op.elem, sn * op.elem, hipMemcpyDeviceToDevice, s));
                        } else {
                            if (sn) NCCLCHK(ncclSend((const char*)op.send + so * op.elem, sn * op.elem, ncclUint8, p, comm(), s));
                            if (rn) NCCLCHK(ncclRecv((char*)op.recv + ro * op.elem, rn * op.elem, ncclUint8, p, comm(), s));
                        }
                        so += sn;
                        ro += rn;
                    }
                }
                NCCLCHK(ncclGroupEnd());
            }
            for (const Reduce& r : reds) NCCLCHK(ncclAllReduce(r.buf, r.buf, r.n, ncclUint32, r.minimum ? ncclMin : ncclSum, comm(), s));
            enq.unlock();
            if (sync) HIPCHK(hipStreamSynchronize(s));
            return;
        }
        sh.ops[(size_t)rank] = a2a;
        auto& mine = sh.hostred[(size_t)rank];
        mine.resize(reds.size());
        for (size_t k = 0; k < reds.size(); ++k) {
            mine[k].resize(reds[k].n);
            HIPCHK(hipMemcpyAsync(mine[k].data(), reds[k].buf, reds[k].n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipStreamSynchronize(s));
        sh.rv.wait();
        for (size_t k = 0; k < a2a.size(); ++k) {
            uint64_t ro = 0;
            for (int src = 0; src < world; ++src) {
                const A2A& theirs = sh.ops[(size_t)src][k];
                uint64_t   so = 0;
                for (int p = 0; p < rank; ++p) so += (*theirs.send_n)[(size_t)p];
                const uint64_t n = (*theirs.send_n)[(size_t)rank];
                if (n) {
                    const int sdev = sh.device[(size_t)src];
                    if (sdev == dev)
                        HIPCHK(hipMemcpyAsync((char*)a2a[k].recv + ro * a2a[k].elem, (const char*)theirs.send + so * theirs.elem, n * theirs.elem, hipMemcpyDeviceToDevice, s));
                    else
                        HIPCHK(hipMemcpyPeerAsync((char*)a2a[k].recv + ro * a2a[k].elem, dev, (const char*)theirs.send + so * theirs.elem, sdev, n * theirs.elem, s));
                }
                ro += n;
            }
        }
        for (size_t k = 0; k < reds.size(); ++k) {
            std::vector<uint32_t> red(sh.hostred[0][k]);
            for (int src = 1; src < world; ++src) {
                const auto& o = sh.hostred[(size_t)src][k];
                for (size_t j = 0; j < red.size(); ++j) red[j] = reds[k].minimum ? std::min(red[j], o[j]) : red[j] + o[j];
            }
            HIPCHK(hipMemcpy(reds[k].buf, red.data(), red.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        HIPCHK(hipStreamSynchronize(s));
        sh.rv.wait();  // nobody reuses a send buffer before everyone has read it
    }
    // every step that can fail on one rank alone begins with this: the ranks agree that all are fine, or all throw
    std::vector<std::vector<uint64_t>> agree(std::vector<uint64_t> mine, const std::string& err, const char* what, hipStream_t s) {
        mine.push_back(err.empty() ? 0 : 1);
        const auto all = all_gather(mine, s);
        std::string bad;
        for (int r = 0; r < world; ++r)
            if (all[(size_t)r].back()) bad += (bad.empty() ? "" : ", ") + std::to_string(r);
        if (!bad.empty()) throw AgreedFailure(std::string(what) + " failed on rank(s) " + bad + (err.empty() ? "" : ": " + err));
        return all;
    }
    // the same for steps that fail only when memory runs out and are FOLLOWED BY AN EXCHANGE into what they reserved. (Round 3 skipped the collective with one process
    // per rank and let the failing rank abort its communicator — which does not wake a peer already inside an intra-node RCCL kernel: the peers hung.)
    void agree_cheap(const std::string& err, const char* what, hipStream_t s) { agree({}, err, what, s); }

    // ---- key-sharded counting (colibri_kshard_*) ---------------------------------------------------------------------------------
    // false: not applicable to this run (some rank's corpus or the options are outside it): the caller takes the candidate exchange
    bool train_kshard(const colibri_options& o, colibri_stats& stats) {
        if (world > 8 || (world & (world - 1))) return false;
        hipStream_t const s = (hipStream_t)colibri_stream(c);
        int               ok = 0;
        uint64_t          maxclass = 0, npos = 0, tokens = 0;
        chk(colibri_kshard_info(c, &o, &ok, &maxclass, &npos), "colibri_kshard_info");
        chk(colibri_corpus_info(c, &tokens, nullptr, nullptr), "colibri_corpus_info");
        uint64_t tokens_g = 0, maxclass_g = 0, npos_g = 0;
        bool     all_ok = true;
        for (const auto& v : all_gather({(uint64_t)ok, maxclass, npos, tokens}, s)) {
            all_ok = all_ok && v[0] != 0;
            maxclass_g = std::max(maxclass_g, v[1]);
            npos_g     = std::max(npos_g, v[2]);
            tokens_g += v[3];
        }
        if (!all_ok || maxclass_g >= (1u << 21) || npos_g >= (1u << 28)) return false;
        std::string err;
        // COLIBRI_FAULT="<rank>:<step name>" (test builds only: -DCOLIBRI_TEST_HOOKS — tests/standin/lib/libcolibri_sharded_hooks.so and the mock; the shipped trainer does not read it):
        // that rank pretends the step failed — every rank must then leave the run together, with that message
#ifdef COLIBRI_TEST_HOOKS
        static const char* const fault = std::getenv("COLIBRI_FAULT");
#endif
        auto        step = [&](int rc, const char* what) {
            if (rc != COLIBRI_OK && err.empty()) err = std::string(what) + ": " + colibri_last_error(c) + " (status " + std::to_string(rc) + ")";
#ifdef COLIBRI_TEST_HOOKS
            if (fault && err.empty() && std::atoi(fault) == rank && std::strchr(fault, ':') && std::string(std::strchr(fault, ':') + 1) == what) err = std::string(what) + ": injected fault";
#endif
            return err.empty();
        };
        step(colibri_kshard_begin(c, &o, world, rank, maxclass_g, npos_g), "colibri_kshard_begin");
        agree({}, err, "key-sharded run: begin", s);
        const int maxlength = std::min<int>(o.maxlength, COLIBRI_MAX_ORDER - 1);
        // order 1: all-reduce of the dense per-class counts
        {
            void*    cnt = nullptr;
            uint32_t nclasses = 0;
            step(colibri_kshard_uni_count(c, &cnt, &nclasses), "colibri_kshard_uni_count");
            agree_cheap(err, "key-sharded run: order 1", s);
            exchange({}, {{cnt, nclasses, false}}, s, false);
            step(colibri_kshard_uni_apply(c), "colibri_kshard_uni_apply");  // (can fail on an allocation: its status travels with the next agreement, order 2's window scan)
        }
        int      maxn = tokens_g ? 1 : 0;
        uint64_t est = tokens_g, ids = 0;  // an upper bound of the next order's records over all ranks; the numbers the last order handed out
        for (int n = 2; n <= maxlength && tokens_g; ++n) {
            void *                send = nullptr, *tab = nullptr, *head = nullptr;
            std::vector<uint64_t> per_owner((size_t)world, 0), per_src((size_t)world, 0);
            uint32_t              recbytes = 0, tab_words = 0;
            uint64_t              admitted = 0;
            const int             more = n < maxlength;
            if (err.empty())  // (a failure of the order below's last step — nothing was exchanged since — travels with this agreement)
                step(colibri_kshard_emit(c, n, est, ids, more, &send, &tab, &tab_words, per_owner.data(), &recbytes, &head, &admitted), "colibri_kshard_emit");
            std::vector<uint64_t> mine = per_owner;
            mine.push_back(admitted);
            const auto everyone = agree(mine, err, "key-sharded run: window scan", s);
            uint64_t   admitted_g = 0, nrecv = 0;
            for (int r = 0; r < world; ++r) {
                admitted_g += everyone[(size_t)r][(size_t)world];
                per_src[(size_t)r] = everyone[(size_t)r][(size_t)rank];
                nrecv += per_src[(size_t)r];
            }
            if (admitted_g == 0) break;  // "None found" (patternmodel.h:1189-1194): no rank has a window of this order left
            maxn = n;
            void *recv = nullptr, *tab_recv = nullptr;
            step(colibri_kshard_recv_buffers(c, nrecv, &recv, &tab_recv), "colibri_kshard_recv_buffers");
            agree_cheap(err, "key-sharded run: receive buffers", s);
            std::vector<uint64_t> tabn((size_t)world, tab_words);
            std::vector<Reduce>   reds;
            if (head) {
                reds.push_back({head, 4096, false});
                reds.push_back({(uint32_t*)head + 4096, 4096, true});
            }
            exchange({{send, &per_owner, recv, &per_src, recbytes}, {tab, &tabn, tab_recv, &tabn, sizeof(uint32_t)}}, reds, s, false);
            void *                fb = nullptr, *ex = nullptr;
            std::vector<uint64_t> fb_dst((size_t)world, 0), ex_dst((size_t)world, 0), fb_src((size_t)world, 0), ex_src((size_t)world, 0), kept_by((size_t)world, 0);
            uint32_t              fb_bytes = 0;
            uint64_t              kept_bins = 0;
            step(colibri_kshard_count(c, n, per_src.data(), more, &fb, fb_dst.data(), &fb_bytes, &ex, ex_dst.data(), &kept_bins), "colibri_kshard_count");
            mine = fb_dst;
            mine.insert(mine.end(), ex_dst.begin(), ex_dst.end());
            mine.push_back(kept_bins);
            const auto back = agree(mine, err, "key-sharded run: count", s);
            uint64_t   nfb = 0, nex = 0, fb_all = 0, keys_all = 0;
            for (int r = 0; r < world; ++r) {
                fb_src[(size_t)r] = back[(size_t)r][(size_t)rank];
                ex_src[(size_t)r] = back[(size_t)r][(size_t)(world + rank)];
                kept_by[(size_t)r] = back[(size_t)r][(size_t)(2 * world)];
                nfb += fb_src[(size_t)r];
                nex += ex_src[(size_t)r];
                for (int q = 0; q < world; ++q) fb_all += back[(size_t)r][(size_t)q];
                for (int q = 0; q < world; ++q) keys_all += everyone[(size_t)r][(size_t)q];
            }
            void *fb_recv = nullptr, *ex_recv = nullptr;
            step(colibri_kshard_feedback_buffers(c, nfb, nex, &fb_recv, &ex_recv), "colibri_kshard_feedback_buffers");
            agree_cheap(err, "key-sharded run: feedback buffers", s);
            exchange({{fb, &fb_dst, fb_recv, &fb_src, fb_bytes}, {ex, &ex_dst, ex_recv, &ex_src, 8}}, {}, s, false);
            // the feedback is a bit per key and a number per surviving key's window: at most that many windows reach the next order (every rank computes the same bound)
            est = more ? std::min<uint64_t>(keys_all, fb_all) + 4096 * (uint64_t)world : 0;
            if (n == 2 && more) {  // order 2's head windows are on no owner's list: their exact number over all ranks (round 4 added a quarter of the corpus, a guess that a
                uint64_t hw = 0;   // small vocabulary breaks: fuller bins than the tables hold, and the whole run repeated on the candidate exchange)
                step(colibri_kshard_head_windows(c, &hw), "colibri_kshard_head_windows");
                est += hw;
            }
            step(colibri_kshard_apply(c, n, fb_src.data(), ex_src.data(), kept_by.data(), more, &ids), "colibri_kshard_apply");  // (its status: the next agreement's)
        }
        std::vector<uint64_t> mine(3 * COLIBRI_MAX_ORDER, 0), found_g(COLIBRI_MAX_ORDER, 0), kept_g(COLIBRI_MAX_ORDER, 0), adm_g(COLIBRI_MAX_ORDER, 0);
        uint32_t              syncs = 0;
        if (err.empty()) step(colibri_kshard_local_stats(c, mine.data(), mine.data() + COLIBRI_MAX_ORDER, mine.data() + 2 * COLIBRI_MAX_ORDER, &syncs), "colibri_kshard_local_stats");
        // (only the orders that ran travel: the process back end's gather buffer is small)
        std::vector<uint64_t> brief;
        for (int n = 1; n <= std::max(maxn, 1) && n < COLIBRI_MAX_ORDER; ++n) {
            brief.push_back(mine[(size_t)n]);
            brief.push_back(mine[(size_t)(COLIBRI_MAX_ORDER + n)]);
            brief.push_back(mine[(size_t)(2 * COLIBRI_MAX_ORDER + n)]);
        }
        const auto figs = agree(brief, err, "key-sharded run: statistics", s);
        for (int r = 0; r < world; ++r)
            for (int n = 1; n <= std::max(maxn, 1) && n < COLIBRI_MAX_ORDER; ++n) {
                found_g[(size_t)n] += figs[(size_t)r][(size_t)(3 * (n - 1))];
                kept_g[(size_t)n] += figs[(size_t)r][(size_t)(3 * (n - 1) + 1)];
                adm_g[(size_t)n] += figs[(size_t)r][(size_t)(3 * (n - 1) + 2)];
            }
        while (maxn > 0 && found_g[(size_t)maxn] == 0) --maxn;
        chk(colibri_kshard_finish(c, found_g.data(), kept_g.data(), adm_g.data(), tokens_g, maxn, &stats), "colibri_kshard_finish");
        last_syncs = syncs;
        return true;
    }

    // ---- candidate exchange (colibri_shard_*): one pass = local count -> exchange -> owner merge -> global ids back -------------------------------
    void all_to_all(const void* send, const std::vector<uint64_t>& send_n, void* recv, const std::vector<uint64_t>& recv_n, size_t elem) {
        exchange({{send, &send_n, recv, &recv_n, elem}}, {}, stream, true);
    }
    // COLIBRI_FAULT="<rank>:<step name>" in test builds (see train_kshard): that rank pretends the step failed
    void inject(std::string& err, const char* what) const {
#ifdef COLIBRI_TEST_HOOKS
        static const char* const fault = std::getenv("COLIBRI_FAULT");
        if (fault && err.empty() && std::atoi(fault) == rank && std::strchr(fault, ':') && std::string(std::strchr(fault, ':') + 1) == what) err = std::string(what) + ": injected fault";
#else
        (void)err, (void)what;
#endif
    }
    void pass(int n, uint32_t mask, int level, bool use_aux, uint64_t& found_all, uint64_t& kept_all) {
        uint64_t              ncand = 0;
        std::vector<uint64_t> per_owner((size_t)world, 0), per_src((size_t)world, 0);
        std::string           err;
        const int             rc0 = colibri_shard_count(c, n, mask, level, &ncand, per_owner.data());
        if (rc0 != COLIBRI_OK) err = std::string("colibri_shard_count: ") + colibri_last_error(c);
        inject(err, "colibri_shard_count");
        const auto sizes = agree(per_owner, err, "sharded pass: local count", stream);
        uint64_t   nrecv = 0;
        for (int r = 0; r < world; ++r) {
            per_src[(size_t)r] = sizes[(size_t)r][(size_t)rank];
            nrecv += per_src[(size_t)r];
        }
        const size_t ns = std::max<uint64_t>(ncand, 1), nr = std::max<uint64_t>(nrecv, 1);
        void *       skeys = nullptr, *scnts = nullptr, *saux = nullptr;  // the library's own partitioned buffers: sent from where they lie
        try {
            chk(colibri_shard_send_view(c, &skeys, &scnts, &saux), "colibri_shard_send_view");
            if (use_aux && saux == nullptr) throw std::runtime_error("the pass has no distinct-filler counts to send");
            rkeys.reserve(nr * 8), rcnts.reserve(nr * 4);
            if (use_aux) raux.reserve(nr * 4);
        } catch (const std::exception& e) {
            err = e.what();
        }
        inject(err, "colibri_shard_send_view");
        agree({}, err, "sharded pass: exchange buffers", stream);
        std::vector<A2A> ops{{skeys, &per_owner, rkeys.p, &per_src, 8}, {scnts, &per_owner, rcnts.p, &per_src, 4}};
        if (use_aux) ops.push_back({saux, &per_owner, raux.p, &per_src, 4});
        exchange(ops, {}, stream, true);
        uint64_t found = 0, kept = 0;
        const int rc1 = colibri_shard_merge(c, rkeys.p, rcnts.p, use_aux ? raux.p : nullptr, per_src.data(), &found, &kept);
        if (rc1 != COLIBRI_OK) err = std::string("colibri_shard_merge: ") + colibri_last_error(c);
        inject(err, "colibri_shard_merge");
        try {
            rgid.reserve(nr * 4), rtot.reserve(nr * 4), gid.reserve(ns * 4), tot.reserve(ns * 4);
        } catch (const std::exception& e) {
            if (err.empty()) err = e.what();
        }
        const auto everyone = agree({found, kept}, err, "sharded pass: owner merge", stream);
        found_all = kept_all = 0;
        uint64_t base = gid_total;
        for (int r = 0; r < world; ++r) {
            found_all += everyone[(size_t)r][0];
            kept_all += everyone[(size_t)r][1];
            if (r < rank) base += everyone[(size_t)r][1];
        }
        if (found_all == 0) return;  // nothing anywhere: every rank sees it at once (reference "None found", patternmodel.h:1189-1194)
        if (gid_total + kept_all >= (1ull << 31)) err = "more than 2^31 surviving patterns";
        if (err.empty() && colibri_shard_reply(c, (uint32_t)base, rgid.p, rtot.p) != COLIBRI_OK) err = std::string("colibri_shard_reply: ") + colibri_last_error(c);
        inject(err, "colibri_shard_reply");
        agree({}, err, "sharded pass: replies", stream);
        exchange({{rgid.p, &per_src, gid.p, &per_owner, 4}, {rtot.p, &per_src, tot.p, &per_owner, 4}}, {}, stream, true);
        uint64_t exported = 0, admitted = 0;
        {  // (a step no agreement follows: a rank that fails here aborts the run's rendezvous, which is what takes its peers out of the next one)
            std::string e;
            inject(e, "colibri_shard_apply");
            if (!e.empty()) throw std::runtime_error(e);
        }
        chk(colibri_shard_apply(c, gid.p, tot.p, &exported, &admitted), "colibri_shard_apply");
        gid_total += kept_all;
    }
    // order 1 without a key exchange (every rank's class encoding canonical): all-reduce of the dense per-class arrays
    bool unigrams_dense(uint64_t& found, uint64_t& kept) {
        int      ok = 0;
        uint64_t maxclass = 0;
        chk(colibri_shard_uni_info(c, &ok, &maxclass), "colibri_shard_uni_info");
        const auto everyone = all_gather({(uint64_t)ok, maxclass}, stream);
        uint64_t   nclasses = 0;
        for (const auto& v : everyone) {
            if (!v[0]) return false;
            nclasses = std::max(nclasses, v[1] + 1);
        }
        std::string err;
        try {
            ucnt.reserve(nclasses * 4), umr.reserve(nclasses * 4);
            chk(colibri_shard_uni_count(c, ucnt.p, umr.p, (uint32_t)nclasses, rank), "colibri_shard_uni_count");
        } catch (const std::exception& e) {
            err = e.what();
        }
        inject(err, "colibri_shard_uni_count");
        agree({}, err, "sharded run: order 1", stream);
        exchange({}, {{ucnt.p, (size_t)nclasses, false}, {umr.p, (size_t)nclasses, true}}, stream, true);
        uint64_t exported = 0;
        chk(colibri_shard_uni_apply(c, ucnt.p, umr.p, (uint32_t)nclasses, rank, &found, &kept, &exported), "colibri_shard_uni_apply");
        gid_total = std::max<uint64_t>(gid_total, nclasses);  // unigram ids are class ids: later passes number from nclasses on
        return true;
    }
    void skipgram_order(int n, const colibri_options& o, uint64_t& found_n, uint64_t& kept_n) {
        found_n = kept_n = 0;
        if (n > 31) throw std::runtime_error("skipgrams of patterns longer than 31 tokens do not exist (a gap mask has 32 bits; set MAXLENGTH)");
        for (uint32_t mask : compute_skip_configurations(n, o.maxskips)) {
            const int levels = (int)mask2vector(mask, n).size();  // gaps = parts - 1 = levels
            uint64_t  f = 0, k = 0;
            bool      complete = true;
            for (int level = 1; level <= levels; ++level) {
                pass(n, mask, level, o.doskipgrams != 0 && level == levels, f, k);  // the distinct-filler counts travel at the last level only
                if (f == 0) {
                    complete = false;
                    break;
                }
            }
            if (complete) {
                found_n += f;
                kept_n += k;
            }
        }
    }
    void train_candidates(colibri_options o, colibri_stats& stats) {
        gid_total = 0;
        std::string err;
        if (colibri_shard_begin(c, &o, world) != COLIBRI_OK) err = std::string("colibri_shard_begin: ") + colibri_last_error(c);
        inject(err, "colibri_shard_begin");
        agree({}, err, "sharded run: begin", stream);
        if (o.mintokens == -1) o.mintokens = 2;
        if (o.mintokens == 0) o.mintokens = 1;
        const int             maxlength = std::min<int>(o.maxlength, COLIBRI_MAX_ORDER - 1);
        std::vector<uint64_t> found_g(COLIBRI_MAX_ORDER, 0), kept_g(COLIBRI_MAX_ORDER, 0);
        uint64_t              tokens = 0, tokens_g = 0;
        chk(colibri_corpus_info(c, &tokens, nullptr, nullptr), "colibri_corpus_info");
        for (const auto& v : all_gather({tokens}, stream)) tokens_g += v[0];
        int maxn = 0;
        for (int n = 1; n <= maxlength; ++n) {
            uint64_t   found_all = 0, kept_all = 0;
            const bool dense = n == 1 && unigrams_dense(found_all, kept_all);
            if (n == 1 && !dense && o.mintokens_unigrams > std::max(1, o.mintokens))
                throw std::runtime_error("MINTOKENS_UNIGRAMS > MINTOKENS in a sharded run needs the class-indexed order 1 (canonical class encoding on every rank)");
            if (!dense) pass(n, 0, 1, false, found_all, kept_all);
            if (found_all == 0) break;
            maxn       = n;
            found_g[(size_t)n] = found_all;
            kept_g[(size_t)n]  = kept_all;
            if (o.doskipgrams_exhaustive && n >= 3) {  // every admissible window also counts its masked forms (patternmodel.h:1163-1171)
                uint64_t f = 0, k = 0;
                skipgram_order(n, o, f, k);
                found_g[(size_t)n] += f;
                kept_g[(size_t)n] += k;
            }
            if (kept_all == 0) break;  // nothing can be admitted at n + 1
        }
        if (o.doskipgrams && o.indexed) {  // IndexedPatternModel::trainskipgrams: from the surviving n-grams, n = 3.. (patternmodel.h:2969-3010)
            for (int n = 3; n <= std::min(maxlength, maxn); ++n) {
                uint64_t f = 0, k = 0;
                skipgram_order(n, o, f, k);
                found_g[(size_t)n] += f;
                kept_g[(size_t)n] += k;
                if (f == 0) break;
            }
        }
        chk(colibri_shard_finish(c, found_g.data(), kept_g.data(), tokens_g, maxn, &stats), "colibri_shard_finish");
        has_gids = true;
    }

  public:
    RankExport    out;
    colibri_stats stats{};
    bool          has_gids = false, uploaded = false, trained = false, took_kshard = false;
    uint32_t      last_syncs = 0;
    RankDriver(Shared& s, int local_index) : sh(s), li(local_index), rank(s.first_rank + local_index), world(s.world), dev(s.device[(size_t)local_index]) {}
    ~RankDriver() {
        (void)hipSetDevice(dev);
        if (c) colibri_destroy(c);
        if (stream) (void)hipStreamDestroy(stream);
        if (gather_host) (void)hipHostFree(gather_host);
    }
    int global_rank() const { return rank; }
    void open() {
        HIPCHK(hipSetDevice(dev));
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        chk(colibri_create(&c, dev), "colibri_create");
    }
    void upload(const unsigned char* payload, uint64_t nbytes, uint32_t first_sentence) {
        HIPCHK(hipSetDevice(dev));
        chk(colibri_upload_corpus(c, payload, nbytes, first_sentence), "colibri_upload_corpus");
        uploaded = true;
    }
    void train(const colibri_options& o, bool force_candidates) {
        HIPCHK(hipSetDevice(dev));
        if (!uploaded) throw std::runtime_error("no corpus uploaded");
        trained     = false;
        has_gids    = false;
        last_syncs  = 0;
        bytes_a2a = bytes_self = bytes_reduce = 0;
        took_kshard = false;
        if (!force_candidates) {
            try {
                took_kshard = train_kshard(o, stats);
            } catch (const AgreedFailure& e) {  // (every rank is here: e.g. a record region or a final bin overflowed on one of them) — the other protocol has its own fallbacks
                if (rank == 0) std::cerr << "key-sharded counting gave up (" << e.what() << "); repeating the run with the candidate exchange" << std::endl;
            }
        }
        if (!took_kshard) train_candidates(o, stats);
        else if (o.indexed) has_gids = true;  // (an indexed model's references stay with the ranks, keyed by the patterns' global numbers: the exporters name theirs)
        uint64_t ns = 0;
        chk(colibri_corpus_info(c, nullptr, &ns, nullptr), "colibri_corpus_info");
        stats.nsentences = ns;
        trained          = true;
    }
    void sizes(uint64_t* np, uint64_t* kb, uint64_t* nr) {
        HIPCHK(hipSetDevice(dev));
        chk(colibri_result_sizes(c, np, kb, nr), "colibri_result_sizes");
    }
    void traffic(uint64_t* a2a, uint64_t* self, uint64_t* reduce) const { *a2a = bytes_a2a, *self = bytes_self, *reduce = bytes_reduce; }
    void kernel_time(int cls, double* ms, uint64_t* launches) { chk(colibri_kernel_time(c, cls, ms, launches), "colibri_kernel_time"); }
    void export_unindexed(uint64_t* key_off, uint8_t* key_bytes, uint32_t* counts) {
        HIPCHK(hipSetDevice(dev));
        chk(colibri_export_unindexed(c, key_off, key_bytes, counts), "colibri_export_unindexed");
    }
    void export_gids(uint32_t* gids) {
        HIPCHK(hipSetDevice(dev));
        if (!trained || !has_gids) throw std::runtime_error("the last run left no global pattern numbers (an indexed model, or the candidate exchange, has them)");
        chk(colibri_shard_export_gids(c, gids), "colibri_shard_export_gids");
    }
    void index_sizes(uint64_t* ngids, uint64_t* nrefs) {
        HIPCHK(hipSetDevice(dev));
        if (!trained) throw std::runtime_error("no trained model");
        chk(colibri_shard_index_sizes(c, ngids, nrefs), "colibri_shard_index_sizes");
    }
    void export_index(uint32_t* gids, uint64_t* ref_off, uint32_t* ref_sentence, uint16_t* ref_token) {
        HIPCHK(hipSetDevice(dev));
        if (!trained) throw std::runtime_error("no trained model");
        chk(colibri_shard_export_index(c, gids, ref_off, ref_sentence, ref_token), "colibri_shard_export_index");
    }
    // this rank's share of the model into `out`
    void fetch(bool indexed) {
        HIPCHK(hipSetDevice(dev));
        uint64_t np = 0, kb = 0, nr = 0;
        chk(colibri_result_sizes(c, &np, &kb, &nr), "colibri_result_sizes");
        out = RankExport();
        out.key_off.assign(np + 1, 0);
        out.key_bytes.assign(kb + 1, 0);
        out.counts.assign(np + 1, 0);  // (never a null pointer: a rank of a small corpus may export nothing)
        chk(colibri_export_unindexed(c, out.key_off.data(), out.key_bytes.data(), out.counts.data()), "colibri_export_unindexed");
        out.counts.resize(np);
        if (has_gids) {
            out.gids.assign(np + 1, 0);
            chk(colibri_shard_export_gids(c, out.gids.data()), "colibri_shard_export_gids");
            out.gids.resize(np);
        }
        if (indexed) {
            uint64_t ng = 0, nrefs = 0;
            chk(colibri_shard_index_sizes(c, &ng, &nrefs), "colibri_shard_index_sizes");
            out.ugid.assign(ng + 1, 0);
            out.ref_off.assign(ng + 1, 0);
            out.ref_sentence.assign(nrefs + 1, 0);
            out.ref_token.assign(nrefs + 1, 0);
            chk(colibri_shard_export_index(c, out.ugid.data(), out.ref_off.data(), out.ref_sentence.data(), out.ref_token.data()), "colibri_shard_export_index");
            out.ugid.resize(ng);
        }
    }
};

// contiguous sentence ranges of about equal bytes: [(begin, end, first sentence)]
struct Cut {
    uint64_t begin, end;
    uint32_t first_sentence;
};
std::vector<Cut> cut_sentences(const unsigned char* p, uint64_t n, int world, uint32_t firstsentence) {
    std::vector<Cut> cuts;
    uint64_t         at = 0;
    uint32_t         sentences_before = 0;
    for (int r = 0; r < world; ++r) {
        uint64_t end = n;
        if (r + 1 < world) {
            end = std::max<uint64_t>(at, n * (uint64_t)(r + 1) / (uint64_t)world);
            // forward to the byte after the next sentence delimiter: a 00 whose predecessor ends a token (or starts the data)
            while (end < n && !(p[end] == 0 && (end == 0 || p[end - 1] < 128))) ++end;
            if (end < n) ++end;
        }
        cuts.push_back({at, end, firstsentence + sentences_before});
        for (uint64_t k = at; k < end; ++k)
            if (p[k] == 0 && (k == 0 || p[k - 1] < 128)) ++sentences_before;
        at = end;
    }
    return cuts;
}

}  // namespace

// ---- the trainer: contexts, communicators and the corpus shards live as long as it does ---------------------------------------------------------------
class ShardedTrainer {
  public:
    Shared                                   sh;
    std::vector<std::unique_ptr<RankDriver>> ranks;
    std::string                              error;
    double                                   last_ms = 0;
    bool                                     force_candidates = false;

    ShardedTrainer(int world, int nlocal, int first_rank) : sh(world, nlocal, first_rank) {}
    ~ShardedTrainer() {
        ranks.clear();
        if (sh.use_rccl)
            for (auto& cm : sh.comms)
                if (cm) (void)ncclCommDestroy(cm);
    }
    // devices: [nlocal] or NULL (local rank i -> device i; COLIBRI_DEVICES overrides). unique_id: 128 bytes from colibri_sharded_unique_id when nlocal < world
    void open(const int* devices, const void* unique_id) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device visible");
        for (int r = 0; r < sh.nlocal; ++r) sh.device[(size_t)r] = devices ? devices[r] : r;
        if (!devices)
            if (const char* e = std::getenv("COLIBRI_DEVICES")) {
                std::stringstream ss(e);
                std::string       tok;
                for (int r = 0; r < sh.nlocal && std::getline(ss, tok, ','); ++r) sh.device[(size_t)r] = std::atoi(tok.c_str());
            }
        bool distinct = true;
        for (int r = 0; r < sh.nlocal; ++r) {
            if (sh.device[(size_t)r] < 0 || sh.device[(size_t)r] >= ndev)
                throw std::runtime_error("rank " + std::to_string(sh.first_rank + r) + " asks for device " + std::to_string(sh.device[(size_t)r]) + ", " + std::to_string(ndev) +
                                         " visible (set COLIBRI_DEVICES)");
            for (int q = 0; q < r; ++q) distinct = distinct && sh.device[(size_t)q] != sh.device[(size_t)r];
        }
        if (!sh.threads()) {
            if (sh.nlocal != 1 || !unique_id) throw std::runtime_error("a trainer holds either every rank of the run or exactly one (with the run's unique id)");
            ncclUniqueId id;
            static_assert(sizeof(ncclUniqueId) == COLIBRI_SHARDED_ID_BYTES, "unique id size");
            std::memcpy(&id, unique_id, sizeof id);
            HIPCHK(hipSetDevice(sh.device[0]));
            NCCLCHK(ncclCommInitRank(&sh.comms[0], sh.world, id, sh.first_rank));
            sh.use_rccl = true;
        } else {
            sh.use_rccl = distinct && !std::getenv("COLIBRI_NO_RCCL");
            if (sh.use_rccl) {
                NCCLCHK(ncclCommInitAll(sh.comms.data(), sh.world, sh.device.data()));
            } else {
                for (int r = 0; r < sh.world; ++r)  // peer copies between distinct devices need peer access
                    for (int q = 0; q < sh.world; ++q)
                        if (sh.device[(size_t)q] != sh.device[(size_t)r] && hipSetDevice(sh.device[(size_t)r]) == hipSuccess) (void)hipDeviceEnablePeerAccess(sh.device[(size_t)q], 0);
            }
        }
        for (int r = 0; r < sh.nlocal; ++r) {
            ranks.emplace_back(new RankDriver(sh, r));
            ranks.back()->open();
        }
    }
    const char* exchange_name() const { return sh.use_rccl ? "RCCL" : "device copies between contexts"; }
    // one host thread per local rank runs `f(rank driver)`; the first failure is the trainer's error
    template <class F>
    bool run(F f) {
        if (sh.poisoned) {  // (colibri_sharded.h: the trainer survives between runs — unless a run tore its communicators down)
            error = "this trainer's communicators were aborted by an earlier failure (" + sh.error + "): create a new trainer";
            return false;
        }
        sh.rv.reset();
        sh.error.clear();
        std::vector<std::thread> threads;
        for (size_t r = 0; r < ranks.size(); ++r)
            threads.emplace_back([&, r] {
                try {
                    f(*ranks[r]);
                } catch (const Aborted&) {
                } catch (const std::exception& e) {
                    sh.fail("rank " + std::to_string(ranks[r]->global_rank()) + ": " + e.what());
                }
            });
        for (auto& t : threads) t.join();
        error = sh.error;
        return error.empty();
    }
    bool train(const colibri_options& o) {
        const auto t0 = std::chrono::steady_clock::now();
        const bool fc = force_candidates || std::getenv("COLIBRI_NO_KSHARD") != nullptr;
        const bool ok = run([&](RankDriver& rk) { rk.train(o, fc); });
        last_ms       = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return ok;
    }
};

void set_gpus(int n) { g_gpus = n; }
int  gpus() {
    if (g_gpus > 0) return g_gpus;
    const char* e = std::getenv("COLIBRI_GPUS");
    const int   v = e ? std::atoi(e) : 1;
    return v > 0 ? v : 1;
}

void device_train_sharded(const unsigned char* payload, uint64_t nbytes, const colibri_options& opt, uint32_t firstsentence, TrainResult& out, int world) {
    if (world < 1 || world > 64) {
        std::cerr << "ERROR: --gpus must be between 1 and 64" << std::endl;
        throw InternalError();
    }
    release_cached();  // (the ranks' contexts need their GPUs' memory: an idle plain-train context of this process would keep several GB of it — ADVICE r5)
    const auto     t0 = std::chrono::steady_clock::now();
    ShardedTrainer tr(world, world, 0);
    try {
        tr.open(nullptr, nullptr);
    } catch (const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        throw InternalError();
    }
    std::cerr << "Training sentence-sharded over " << world << " GPU" << (world > 1 ? "s" : "") << " (exchange: " << tr.exchange_name() << ")" << std::endl;
    const std::vector<Cut> cuts = cut_sentences(payload, nbytes, world, firstsentence);
    auto&                  ranks = tr.ranks;
    bool ok = tr.run([&](RankDriver& rk) {
        const Cut& ct = cuts[(size_t)rk.global_rank()];
        rk.upload(payload + ct.begin, ct.end - ct.begin, ct.first_sentence);
    });
    ok = ok && tr.train(opt);
    ok = ok && tr.run([&](RankDriver& rk) { rk.fetch(opt.indexed != 0); });
    if (!ok) {
        std::cerr << "ERROR: " << tr.error << std::endl;
        throw InternalError();
    }
    if (ranks[0]->took_kshard) std::cerr << "Counted key-sharded: every rank counts the keys it owns (" << ranks[0]->last_syncs << " host look-ups)" << std::endl;
    // ---- the union of the ranks' exports is the model ---------------------------------------------------------------------
    out.stats = ranks[0]->stats;  // global found / kept / totals are the same on every rank
    const bool have_gids = ranks[0]->has_gids;
    uint64_t   np = 0, kb = 0, gid_max = 0;
    for (auto& rk : ranks) {
        np += rk->out.counts.size();
        kb += rk->out.key_off.empty() ? 0 : rk->out.key_off.back();
        for (uint32_t g : rk->out.gids) gid_max = std::max<uint64_t>(gid_max, g);
    }
    out.key_off.assign(np + 1, 0);
    out.key_bytes.assign(kb + 1, 0);
    out.counts.assign(np, 0);
    std::vector<uint32_t> where(have_gids ? gid_max + 2 : 0, 0xFFFFFFFFu);  // global id -> pattern number
    uint64_t              j = 0, b = 0;
    out.stats.nsentences = 0;
    const bool sum_admitted = !ranks[0]->took_kshard;  // (the key-sharded run reports global figures already)
    if (sum_admitted)
        for (int n = 1; n < COLIBRI_MAX_ORDER; ++n) out.stats.admitted[n] = 0;
    for (int n = 1; n < COLIBRI_MAX_ORDER; ++n) out.stats.windows[n] = 0;
    for (auto& rk : ranks) {
        const RankExport& e = rk->out;
        for (size_t k = 0; k < e.counts.size(); ++k, ++j) {
            const uint64_t len = e.key_off[k + 1] - e.key_off[k];
            out.key_off[j]     = b;
            std::memcpy(out.key_bytes.data() + b, e.key_bytes.data() + e.key_off[k], len);
            b += len;
            out.counts[j] = e.counts[k];
            if (have_gids) {
                if (where[e.gids[k]] != 0xFFFFFFFFu) {
                    std::cerr << "ERROR: a pattern was exported by more than one rank" << std::endl;
                    throw InternalError();
                }
                where[e.gids[k]] = (uint32_t)j;
            }
        }
        out.stats.nsentences += rk->stats.nsentences;
        for (int n = 1; n < COLIBRI_MAX_ORDER; ++n) {
            out.stats.windows[n] += rk->stats.windows[n];
            if (sum_admitted) out.stats.admitted[n] += rk->stats.admitted[n];
        }
    }
    out.key_off[np]     = b;
    out.stats.npatterns = np;
    out.stats.keybytes  = b;
    out.stats.minn      = np ? 1 : 0;
    out.stats.nrefs     = 0;
    if (opt.indexed) {
        // a pattern's index = its runs in rank order (the ranks hold disjoint, ascending sentence ranges: the concatenation is sorted)
        out.ref_off.assign(np + 1, 0);
        for (auto& rk : ranks) {
            const RankExport& e = rk->out;
            for (size_t k = 0; k < e.ugid.size(); ++k) {
                const uint32_t g = e.ugid[k];
                if (g < where.size() && where[g] != 0xFFFFFFFFu) out.ref_off[where[g] + 1] += e.ref_off[k + 1] - e.ref_off[k];
            }
        }
        for (uint64_t k = 0; k < np; ++k) out.ref_off[k + 1] += out.ref_off[k];
        const uint64_t nrefs = out.ref_off[np];
        out.ref_sentence.assign(nrefs + 1, 0);
        out.ref_token.assign(nrefs + 1, 0);
        std::vector<uint64_t> cursor(out.ref_off.begin(), out.ref_off.end() - 1);
        for (auto& rk : ranks) {
            const RankExport& e = rk->out;
            for (size_t k = 0; k < e.ugid.size(); ++k) {
                const uint32_t g = e.ugid[k];
                if (g >= where.size() || where[g] == 0xFFFFFFFFu) continue;
                uint64_t& at = cursor[where[g]];
                for (uint64_t q = e.ref_off[k]; q < e.ref_off[k + 1]; ++q, ++at) {
                    out.ref_sentence[at] = e.ref_sentence[q];
                    out.ref_token[at]    = e.ref_token[q];
                }
            }
        }
        out.stats.nrefs = nrefs;
    }
    out.stats.train_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace colibri_host

// ---- C face (include/colibri_sharded.h) ------------------------------------------------------------------------------------------------------------------
struct colibri_sharded {
    colibri_host::ShardedTrainer tr;
    std::string                  err;
    colibri_sharded(int world, int nlocal, int first) : tr(world, nlocal, first) {}
};

extern "C" {

int colibri_sharded_unique_id(void* out) {
    if (!out) return COLIBRI_ERR_ARG;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return COLIBRI_ERR_HIP;
    std::memcpy(out, &id, sizeof id);
    return COLIBRI_OK;
}

int colibri_sharded_create(colibri_sharded** out, int world, int nlocal, int first_rank, const int* devices, const void* unique_id) {
    if (!out || world < 1 || world > 64 || nlocal < 1 || nlocal > world || first_rank < 0 || first_rank + nlocal > world) return COLIBRI_ERR_ARG;
    *out = nullptr;
    colibri_sharded* t = new (std::nothrow) colibri_sharded(world, nlocal, first_rank);
    if (!t) return COLIBRI_ERR_ARG;
    try {
        t->tr.open(devices, unique_id);
    } catch (const std::exception& e) {
        std::cerr << "colibri_sharded_create: " << e.what() << std::endl;
        delete t;
        return COLIBRI_ERR_HIP;
    }
    *out = t;
    return COLIBRI_OK;
}
void colibri_sharded_destroy(colibri_sharded* t) { delete t; }
const char* colibri_sharded_last_error(const colibri_sharded* t) { return t ? t->err.c_str() : "null trainer"; }

int colibri_sharded_upload(colibri_sharded* t, int local_rank, const uint8_t* payload, uint64_t nbytes, uint32_t first_sentence) {
    if (!t || local_rank < 0 || local_rank >= (int)t->tr.ranks.size()) return COLIBRI_ERR_ARG;
    try {
        t->tr.ranks[(size_t)local_rank]->upload(payload, nbytes, first_sentence);
    } catch (const std::exception& e) {
        t->err = e.what();
        return COLIBRI_ERR_HIP;
    }
    return COLIBRI_OK;
}
int colibri_sharded_upload_split(colibri_sharded* t, const uint8_t* payload, uint64_t nbytes, uint32_t first_sentence) {
    if (!t || (!payload && nbytes)) return COLIBRI_ERR_ARG;
    if (!t->tr.sh.threads()) {
        t->err = "colibri_sharded_upload_split needs a trainer that holds every rank";
        return COLIBRI_ERR_STATE;
    }
    const auto cuts = colibri_host::cut_sentences(payload, nbytes, t->tr.sh.world, first_sentence);
    const bool ok   = t->tr.run([&](colibri_host::RankDriver& rk) {
        const auto& ct = cuts[(size_t)rk.global_rank()];
        rk.upload(payload + ct.begin, ct.end - ct.begin, ct.first_sentence);
    });
    if (!ok) t->err = t->tr.error;
    return ok ? COLIBRI_OK : COLIBRI_ERR_HIP;
}
int colibri_sharded_set_protocol(colibri_sharded* t, int protocol) {
    if (!t || protocol < 0 || protocol > 1) return COLIBRI_ERR_ARG;
    t->tr.force_candidates = protocol == 1;
    return COLIBRI_OK;
}
int colibri_sharded_train(colibri_sharded* t, const colibri_options* opt, colibri_stats* stats, colibri_sharded_info* info) {
    if (!t || !opt) return COLIBRI_ERR_ARG;
    const bool ok = t->tr.train(*opt);
    if (!ok) {
        t->err = t->tr.error;
        return COLIBRI_ERR_HIP;
    }
    const auto& r0 = *t->tr.ranks[0];
    if (stats) {
        *stats = r0.stats;  // found / kept / tokens / types are global on every rank; patterns and sentences are this trainer's ranks'
        stats->npatterns = stats->nsentences = 0;
        for (auto& rk : t->tr.ranks) {
            stats->npatterns += rk->stats.npatterns;
            stats->nsentences += rk->stats.nsentences;
            if (&*rk != &r0)
                for (int n = 1; n < COLIBRI_MAX_ORDER; ++n) {
                    stats->windows[n] += rk->stats.windows[n];
                    if (!r0.took_kshard) stats->admitted[n] += rk->stats.admitted[n];  // (the key-sharded run reports the global figure on every rank)
                }
        }
        stats->train_ms = t->tr.last_ms;
    }
    if (info) {
        info->protocol     = r0.took_kshard ? 0 : 1;
        info->rccl         = t->tr.sh.use_rccl ? 1 : 0;
        info->host_lookups = r0.last_syncs;
        info->wall_ms      = t->tr.last_ms;
        r0.traffic(&info->alltoall_bytes, &info->alltoall_bytes_to_self, &info->allreduce_bytes);
    }
    return COLIBRI_OK;
}
int colibri_sharded_result_sizes(colibri_sharded* t, int local_rank, uint64_t* npatterns, uint64_t* keybytes) {
    if (!t || local_rank < 0 || local_rank >= (int)t->tr.ranks.size()) return COLIBRI_ERR_ARG;
    try {
        uint64_t nr = 0;
        t->tr.ranks[(size_t)local_rank]->sizes(npatterns, keybytes, &nr);
    } catch (const std::exception& e) {
        t->err = e.what();
        return COLIBRI_ERR_HIP;
    }
    return COLIBRI_OK;
}
int colibri_sharded_kernel_time(colibri_sharded* t, int local_rank, int kernel_class, double* total_ms, uint64_t* launches) {
    if (!t || local_rank < 0 || local_rank >= (int)t->tr.ranks.size()) return COLIBRI_ERR_ARG;
    try {
        t->tr.ranks[(size_t)local_rank]->kernel_time(kernel_class, total_ms, launches);
    } catch (const std::exception& e) {
        t->err = e.what();
        return COLIBRI_ERR_ARG;
    }
    return COLIBRI_OK;
}
int colibri_sharded_export_unindexed(colibri_sharded* t, int local_rank, uint64_t* key_off, uint8_t* key_bytes, uint32_t* counts) {
    if (!t || local_rank < 0 || local_rank >= (int)t->tr.ranks.size()) return COLIBRI_ERR_ARG;
    try {
        t->tr.ranks[(size_t)local_rank]->export_unindexed(key_off, key_bytes, counts);
    } catch (const std::exception& e) {
        t->err = e.what();
        return COLIBRI_ERR_HIP;
    }
    return COLIBRI_OK;
}
int colibri_sharded_export_gids(colibri_sharded* t, int local_rank, uint32_t* gids) {
    if (!t || !gids || local_rank < 0 || local_rank >= (int)t->tr.ranks.size()) return COLIBRI_ERR_ARG;
    try {
        t->tr.ranks[(size_t)local_rank]->export_gids(gids);
    } catch (const std::exception& e) {
        t->err = e.what();
        return COLIBRI_ERR_STATE;
    }
    return COLIBRI_OK;
}
int colibri_sharded_index_sizes(colibri_sharded* t, int local_rank, uint64_t* ngids, uint64_t* nrefs) {
    if (!t || !ngids || !nrefs || local_rank < 0 || local_rank >= (int)t->tr.ranks.size()) return COLIBRI_ERR_ARG;
    try {
        t->tr.ranks[(size_t)local_rank]->index_sizes(ngids, nrefs);
    } catch (const std::exception& e) {
        t->err = e.what();
        return COLIBRI_ERR_STATE;
    }
    return COLIBRI_OK;
}
int colibri_sharded_export_index(colibri_sharded* t, int local_rank, uint32_t* gids, uint64_t* ref_off, uint32_t* ref_sentence, uint16_t* ref_token) {
    if (!t || !gids || !ref_off || !ref_sentence || !ref_token || local_rank < 0 || local_rank >= (int)t->tr.ranks.size()) return COLIBRI_ERR_ARG;
    try {
        t->tr.ranks[(size_t)local_rank]->export_index(gids, ref_off, ref_sentence, ref_token);
    } catch (const std::exception& e) {
        t->err = e.what();
        return COLIBRI_ERR_STATE;
    }
    return COLIBRI_OK;
}

}  // extern "C"
