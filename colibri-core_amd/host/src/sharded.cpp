// sharded.cpp — PatternModel::train across the GPUs of one node, from the C++ face: one host thread and one device context per rank, the corpus cut into
// contiguous sentence ranges, RCCL (linked directly) for the exchange steps. The reference is single-threaded (include/patternmodel.h:880-1345 is one loop
// over one map); what is distributed is that loop's only cross-shard dependency: the GLOBAL count of a candidate pattern before the prune of each order.
// The protocol is the one of colibri_amd/dist.py (include/colibri_hip.h, "sentence-sharded multi-GPU training"), per pass:
//     colibri_shard_count -> all-to-all sizes -> colibri_shard_send -> all-to-all (key, count [, distinct fillers]) -> colibri_shard_merge ->
//     all-gather (found, kept) -> colibri_shard_reply -> all-to-all replies -> colibri_shard_apply
// and order 1 as an all-reduce (SUM / MIN) of the dense per-class arrays. Exchange back ends:
//     RCCL   every rank has a device of its own: ncclCommInitAll, ncclSend/ncclRecv groups (all seven xGMI links of a GPU at once), ncclAllReduce;
//     copies two ranks share a device (tests on a one-GPU box; COLIBRI_DEVICES=0,0): device-to-device copies out of the peers' send buffers between two
//            barriers, and the two dense arrays of order 1 reduced on the host.
// Afterwards every rank exports the patterns it was named exporter of and its local forward index; the union is the model.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "algorithms.h"
#include "colibri_hip.h"
#include "patternmodel.h"

namespace colibri_host {
namespace {

int g_gpus = 0;  // 0: not set (COLIBRI_GPUS or 1)

struct Aborted {};

// barrier of the rank threads; abort() releases everyone (a rank that failed must not leave the others waiting)
class Rendezvous {
    std::mutex              m;
    std::condition_variable cv;
    int                     world, waiting = 0;
    uint64_t                generation = 0;
    bool                    aborted    = false;

  public:
    explicit Rendezvous(int w) : world(w) {}
    void wait() {
        std::unique_lock<std::mutex> l(m);
        if (aborted) throw Aborted();
        const uint64_t g = generation;
        if (++waiting == world) {
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
    colibri_stats              stats{};
    std::vector<uint64_t>      key_off;
    std::vector<unsigned char> key_bytes;
    std::vector<uint32_t>      counts, gids;
    std::vector<uint32_t>      ugid, ref_sentence;  // local forward index, keyed by global id
    std::vector<uint64_t>      ref_off;
    std::vector<uint16_t>      ref_token;
};

struct Shared {
    int                                world;
    Rendezvous                         rv;
    bool                               use_rccl = false;
    std::vector<ncclComm_t>            comms;
    std::vector<int>                   device;
    std::vector<std::vector<uint64_t>> sizes;    // [src][dst]: elements src sends to dst in the running all-to-all
    std::vector<const void*>           sendptr;  // [src]: its send buffer
    std::vector<std::vector<uint64_t>> ints;     // all-gather slots
    std::vector<std::vector<uint32_t>> host_a, host_b;  // copies back end: the dense arrays of order 1
    std::mutex                         errm;
    std::string                        error;
    explicit Shared(int w) : world(w), rv(w), device(w, 0), sizes(w, std::vector<uint64_t>(w, 0)), sendptr(w, nullptr), ints(w), host_a(w), host_b(w) {}
    void fail(const std::string& what) {
        {
            std::lock_guard<std::mutex> l(errm);
            if (error.empty()) error = what;
        }
        rv.abort();
    }
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

class RankDriver {
    Shared&      sh;
    const int    rank, world, dev;
    colibri_ctx* c      = nullptr;
    hipStream_t  stream = nullptr;
    DevMem       rkeys, rcnts, raux, rgid, rtot, gid, tot, ucnt, umr;  // receive / reply buffers (the candidates are sent from the library's own buffers)
    uint64_t     gid_total = 0;

    void chk(int rc, const char* what) {
        if (rc != COLIBRI_OK) throw std::runtime_error(std::string(what) + ": " + (c ? colibri_last_error(c) : "no context") + " (status " + std::to_string(rc) + ")");
    }

    // ---- collectives ----------------------------------------------------------------------------------------------------
    std::vector<std::vector<uint64_t>> all_gather(const std::vector<uint64_t>& mine) {
        sh.ints[rank] = mine;
        sh.rv.wait();
        std::vector<std::vector<uint64_t>> all = sh.ints;
        sh.rv.wait();
        return all;
    }
    // per_owner[p] elements of `send` (partitioned by destination, in rank order) go to rank p; returns what arrives, concatenated in rank order
    void publish_sizes(const std::vector<uint64_t>& per_owner, std::vector<uint64_t>& per_src) {
        sh.sizes[rank] = per_owner;
        sh.rv.wait();
        per_src.resize(world);
        for (int s = 0; s < world; ++s) per_src[s] = sh.sizes[s][rank];
        sh.rv.wait();
    }
    void all_to_all(const void* send, const std::vector<uint64_t>& send_n, void* recv, const std::vector<uint64_t>& recv_n, size_t elem) {
        if (sh.use_rccl) {
            NCCLCHK(ncclGroupStart());
            uint64_t so = 0, ro = 0;
            for (int p = 0; p < world; ++p) {
                if (send_n[p]) NCCLCHK(ncclSend((const char*)send + so * elem, send_n[p] * elem, ncclUint8, p, sh.comms[rank], stream));
                if (recv_n[p]) NCCLCHK(ncclRecv((char*)recv + ro * elem, recv_n[p] * elem, ncclUint8, p, sh.comms[rank], stream));
                so += send_n[p];
                ro += recv_n[p];
            }
            NCCLCHK(ncclGroupEnd());
            HIPCHK(hipStreamSynchronize(stream));
            return;
        }
        // copies: every rank publishes its send buffer and its partition sizes, then pulls its share out of everyone's buffer
        sh.sendptr[rank] = send;
        sh.sizes[rank]   = send_n;
        sh.rv.wait();
        uint64_t ro = 0;
        for (int s = 0; s < world; ++s) {
            uint64_t so = 0;
            for (int p = 0; p < rank; ++p) so += sh.sizes[s][p];
            const uint64_t n = sh.sizes[s][rank];
            if (n) {
                if (sh.device[s] == dev)
                    HIPCHK(hipMemcpyAsync((char*)recv + ro * elem, (const char*)sh.sendptr[s] + so * elem, n * elem, hipMemcpyDeviceToDevice, stream));
                else
                    HIPCHK(hipMemcpyPeerAsync((char*)recv + ro * elem, dev, (const char*)sh.sendptr[s] + so * elem, sh.device[s], n * elem, stream));
            }
            ro += n;
        }
        HIPCHK(hipStreamSynchronize(stream));
        sh.rv.wait();  // nobody reuses a send buffer before everyone has read it
    }
    void all_reduce_u32(void* buf, size_t n, bool minimum, std::vector<std::vector<uint32_t>>& slots) {
        if (sh.use_rccl) {
            NCCLCHK(ncclAllReduce(buf, buf, n, ncclUint32, minimum ? ncclMin : ncclSum, sh.comms[rank], stream));
            HIPCHK(hipStreamSynchronize(stream));
            return;
        }
        slots[rank].resize(n);
        HIPCHK(hipMemcpy(slots[rank].data(), buf, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        sh.rv.wait();
        std::vector<uint32_t> red(slots[0]);
        for (int s = 1; s < world; ++s)
            for (size_t k = 0; k < n; ++k) red[k] = minimum ? std::min(red[k], slots[s][k]) : red[k] + slots[s][k];
        HIPCHK(hipMemcpy(buf, red.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
        sh.rv.wait();
    }

    // ---- one pass: local count -> exchange -> owner merge -> global ids back (dist.py: ShardedTrainer._pass) ----------------
    void pass(int n, uint32_t mask, int level, bool use_aux, uint64_t& found_all, uint64_t& kept_all) {
        uint64_t              ncand = 0;
        std::vector<uint64_t> per_owner(world, 0), per_src;
        chk(colibri_shard_count(c, n, mask, level, &ncand, per_owner.data()), "colibri_shard_count");
        publish_sizes(per_owner, per_src);
        uint64_t nrecv = 0;
        for (uint64_t v : per_src) nrecv += v;
        const size_t ns = std::max<uint64_t>(ncand, 1), nr = std::max<uint64_t>(nrecv, 1);
        void *skeys = nullptr, *scnts = nullptr, *saux = nullptr;  // the library's own partitioned buffers: sent from where they lie
        chk(colibri_shard_send_view(c, &skeys, &scnts, &saux), "colibri_shard_send_view");
        if (use_aux && saux == nullptr) throw std::runtime_error("the pass has no distinct-filler counts to send");
        rkeys.reserve(nr * 8), rcnts.reserve(nr * 4);
        all_to_all(skeys, per_owner, rkeys.p, per_src, 8);
        all_to_all(scnts, per_owner, rcnts.p, per_src, 4);
        if (use_aux) {
            raux.reserve(nr * 4);
            all_to_all(saux, per_owner, raux.p, per_src, 4);
        }
        uint64_t found = 0, kept = 0;
        chk(colibri_shard_merge(c, rkeys.p, rcnts.p, use_aux ? raux.p : nullptr, per_src.data(), &found, &kept), "colibri_shard_merge");
        const auto everyone = all_gather({found, kept});
        found_all = kept_all = 0;
        uint64_t base = gid_total;
        for (int r = 0; r < world; ++r) {
            found_all += everyone[r][0];
            kept_all += everyone[r][1];
            if (r < rank) base += everyone[r][1];
        }
        if (found_all == 0) return;  // nothing anywhere: every rank sees it at once (reference "None found", patternmodel.h:1189-1194)
        if (gid_total + kept_all >= (1ull << 31)) throw std::runtime_error("more than 2^31 surviving patterns");
        rgid.reserve(nr * 4), rtot.reserve(nr * 4), gid.reserve(ns * 4), tot.reserve(ns * 4);
        chk(colibri_shard_reply(c, (uint32_t)base, rgid.p, rtot.p), "colibri_shard_reply");
        all_to_all(rgid.p, per_src, gid.p, per_owner, 4);
        all_to_all(rtot.p, per_src, tot.p, per_owner, 4);
        uint64_t exported = 0, admitted = 0;
        chk(colibri_shard_apply(c, gid.p, tot.p, &exported, &admitted), "colibri_shard_apply");
        gid_total += kept_all;
    }
    // order 1 without a key exchange (every rank's class encoding canonical): all-reduce of the dense per-class arrays
    bool unigrams_dense(uint64_t& found, uint64_t& kept) {
        int      ok = 0;
        uint64_t maxclass = 0;
        chk(colibri_shard_uni_info(c, &ok, &maxclass), "colibri_shard_uni_info");
        const auto everyone = all_gather({(uint64_t)ok, maxclass});
        uint64_t   nclasses = 0;
        for (const auto& v : everyone) {
            if (!v[0]) return false;
            nclasses = std::max(nclasses, v[1] + 1);
        }
        ucnt.reserve(nclasses * 4), umr.reserve(nclasses * 4);
        chk(colibri_shard_uni_count(c, ucnt.p, umr.p, (uint32_t)nclasses, rank), "colibri_shard_uni_count");
        all_reduce_u32(ucnt.p, nclasses, false, sh.host_a);
        all_reduce_u32(umr.p, nclasses, true, sh.host_b);
        uint64_t exported = 0;
        chk(colibri_shard_uni_apply(c, ucnt.p, umr.p, (uint32_t)nclasses, rank, &found, &kept, &exported), "colibri_shard_uni_apply");
        gid_total = std::max<uint64_t>(gid_total, nclasses);  // unigram ids are class ids: later passes number from nclasses on
        return true;
    }
    void skipgram_order(int n, const colibri_options& o, uint64_t& found_n, uint64_t& kept_n) {
        found_n = kept_n = 0;
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

  public:
    RankExport out;
    RankDriver(Shared& s, int r) : sh(s), rank(r), world(s.world), dev(s.device[(size_t)r]) {}
    ~RankDriver() {
        if (c) colibri_destroy(c);
        if (stream) (void)hipStreamDestroy(stream);
    }
    void run(const unsigned char* payload, uint64_t nbytes, uint32_t first_sentence, colibri_options o) {
        HIPCHK(hipSetDevice(dev));
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        chk(colibri_create(&c, dev), "colibri_create");
        chk(colibri_upload_corpus(c, payload, nbytes, first_sentence), "colibri_upload_corpus");
        chk(colibri_shard_begin(c, &o, world), "colibri_shard_begin");
        if (o.mintokens == -1) o.mintokens = 2;
        if (o.mintokens == 0) o.mintokens = 1;
        const int             maxlength = std::min<int>(o.maxlength, COLIBRI_MAX_ORDER - 1);
        std::vector<uint64_t> found_g(COLIBRI_MAX_ORDER, 0), kept_g(COLIBRI_MAX_ORDER, 0);
        uint64_t              tokens = 0, tokens_g = 0;
        chk(colibri_corpus_info(c, &tokens, nullptr, nullptr), "colibri_corpus_info");
        for (const auto& v : all_gather({tokens})) tokens_g += v[0];
        int maxn = 0;
        for (int n = 1; n <= maxlength; ++n) {
            uint64_t found_all = 0, kept_all = 0;
            const bool dense = n == 1 && unigrams_dense(found_all, kept_all);
            if (n == 1 && !dense && o.mintokens_unigrams > std::max(1, o.mintokens))
                throw std::runtime_error("MINTOKENS_UNIGRAMS > MINTOKENS in a sharded run needs the class-indexed order 1 (canonical class encoding on every rank)");
            if (!dense) pass(n, 0, 1, false, found_all, kept_all);
            if (found_all == 0) break;
            maxn       = n;
            found_g[n] = found_all;
            kept_g[n]  = kept_all;
            if (o.doskipgrams_exhaustive && n >= 3) {  // every admissible window also counts its masked forms (patternmodel.h:1163-1171)
                uint64_t f = 0, k = 0;
                skipgram_order(n, o, f, k);
                found_g[n] += f;
                kept_g[n] += k;
            }
            if (kept_all == 0) break;  // nothing can be admitted at n + 1
        }
        if (o.doskipgrams && o.indexed) {  // IndexedPatternModel::trainskipgrams: from the surviving n-grams, n = 3.. (patternmodel.h:2969-3010)
            for (int n = 3; n <= std::min(maxlength, maxn); ++n) {
                uint64_t f = 0, k = 0;
                skipgram_order(n, o, f, k);
                found_g[n] += f;
                kept_g[n] += k;
                if (f == 0) break;
            }
        }
        chk(colibri_shard_finish(c, found_g.data(), kept_g.data(), tokens_g, maxn, &out.stats), "colibri_shard_finish");
        // this rank's share of the model
        uint64_t np = 0, kb = 0, nr = 0;
        chk(colibri_result_sizes(c, &np, &kb, &nr), "colibri_result_sizes");
        out.key_off.assign(np + 1, 0);
        out.key_bytes.assign(kb + 1, 0);
        out.counts.assign(np, 0);
        out.gids.assign(np + 1, 0);
        chk(colibri_export_unindexed(c, out.key_off.data(), out.key_bytes.data(), out.counts.data()), "colibri_export_unindexed");
        chk(colibri_shard_export_gids(c, out.gids.data()), "colibri_shard_export_gids");
        out.gids.resize(np);
        if (o.indexed) {
            uint64_t ng = 0, nrefs = 0;
            chk(colibri_shard_index_sizes(c, &ng, &nrefs), "colibri_shard_index_sizes");
            out.ugid.assign(ng + 1, 0);
            out.ref_off.assign(ng + 1, 0);
            out.ref_sentence.assign(nrefs + 1, 0);
            out.ref_token.assign(nrefs + 1, 0);
            chk(colibri_shard_export_index(c, out.ugid.data(), out.ref_off.data(), out.ref_sentence.data(), out.ref_token.data()), "colibri_shard_export_index");
            out.ugid.resize(ng);
        }
        out.stats.nsentences = 0;
        uint64_t ns = 0;
        chk(colibri_corpus_info(c, nullptr, &ns, nullptr), "colibri_corpus_info");
        out.stats.nsentences = ns;
        sh.rv.wait();  // the contexts (and their exchange buffers) go away together
    }
};

// contiguous sentence ranges of about equal bytes: [(begin, end, first sentence)] (dist.py: shard_payload)
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
    const auto t0 = std::chrono::steady_clock::now();
    Shared     sh(world);
    // rank -> device: 0..world-1, or the list in COLIBRI_DEVICES (a device may appear twice: the copies back end is used then)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        std::cerr << "ERROR: no HIP device visible" << std::endl;
        throw InternalError();
    }
    for (int r = 0; r < world; ++r) sh.device[(size_t)r] = r;
    if (const char* e = std::getenv("COLIBRI_DEVICES")) {
        std::stringstream ss(e);
        std::string       tok;
        for (int r = 0; r < world && std::getline(ss, tok, ','); ++r) sh.device[(size_t)r] = std::atoi(tok.c_str());
    }
    bool distinct = true;
    for (int r = 0; r < world; ++r) {
        if (sh.device[(size_t)r] < 0 || sh.device[(size_t)r] >= ndev) {
            std::cerr << "ERROR: rank " << r << " asks for device " << sh.device[(size_t)r] << ", " << ndev << " visible (set COLIBRI_DEVICES)" << std::endl;
            throw InternalError();
        }
        for (int q = 0; q < r; ++q) distinct = distinct && sh.device[(size_t)q] != sh.device[(size_t)r];
    }
    sh.use_rccl = distinct && !std::getenv("COLIBRI_NO_RCCL");
    if (sh.use_rccl) {
        sh.comms.resize((size_t)world);
        const ncclResult_t e = ncclCommInitAll(sh.comms.data(), world, sh.device.data());
        if (e != ncclSuccess) {
            std::cerr << "ERROR: ncclCommInitAll: " << ncclGetErrorString(e) << std::endl;
            throw InternalError();
        }
    } else {
        for (int r = 0; r < world; ++r)  // peer copies between distinct devices need peer access
            for (int q = 0; q < world; ++q)
                if (sh.device[(size_t)q] != sh.device[(size_t)r] && hipSetDevice(sh.device[(size_t)r]) == hipSuccess) (void)hipDeviceEnablePeerAccess(sh.device[(size_t)q], 0);
    }
    std::cerr << "Training sentence-sharded over " << world << " GPU" << (world > 1 ? "s" : "") << " (exchange: " << (sh.use_rccl ? "RCCL" : "device copies between contexts") << ")" << std::endl;
    const std::vector<Cut>                   cuts = cut_sentences(payload, nbytes, world, firstsentence);
    std::vector<std::unique_ptr<RankDriver>> ranks;
    for (int r = 0; r < world; ++r) ranks.emplace_back(new RankDriver(sh, r));
    std::vector<std::thread> threads;
    for (int r = 0; r < world; ++r)
        threads.emplace_back([&, r] {
            try {
                ranks[(size_t)r]->run(payload + cuts[(size_t)r].begin, cuts[(size_t)r].end - cuts[(size_t)r].begin, cuts[(size_t)r].first_sentence, opt);
            } catch (const Aborted&) {
            } catch (const std::exception& e) {
                sh.fail("rank " + std::to_string(r) + ": " + e.what());
            }
        });
    for (auto& t : threads) t.join();
    if (sh.use_rccl)
        for (auto& cm : sh.comms) (void)ncclCommDestroy(cm);
    if (!sh.error.empty()) {
        std::cerr << "ERROR: " << sh.error << std::endl;
        throw InternalError();
    }
    // ---- the union of the ranks' exports is the model ---------------------------------------------------------------------
    out.stats = ranks[0]->out.stats;  // global found / kept / totals are the same on every rank
    uint64_t np = 0, kb = 0, gid_max = 0;
    for (auto& rk : ranks) {
        np += rk->out.counts.size();
        kb += rk->out.key_off.empty() ? 0 : rk->out.key_off.back();
        for (uint32_t g : rk->out.gids) gid_max = std::max<uint64_t>(gid_max, g);
    }
    out.key_off.assign(np + 1, 0);
    out.key_bytes.assign(kb + 1, 0);
    out.counts.assign(np, 0);
    std::vector<uint32_t> where(gid_max + 2, 0xFFFFFFFFu);  // global id -> pattern number
    uint64_t              j = 0, b = 0;
    out.stats.nsentences = 0;
    for (int n = 1; n < COLIBRI_MAX_ORDER; ++n) out.stats.windows[n] = out.stats.admitted[n] = 0;
    for (auto& rk : ranks) {
        const RankExport& e = rk->out;
        for (size_t k = 0; k < e.counts.size(); ++k, ++j) {
            const uint64_t len = e.key_off[k + 1] - e.key_off[k];
            out.key_off[j]     = b;
            std::memcpy(out.key_bytes.data() + b, e.key_bytes.data() + e.key_off[k], len);
            b += len;
            out.counts[j] = e.counts[k];
            if (where[e.gids[k]] != 0xFFFFFFFFu) {
                std::cerr << "ERROR: a pattern was exported by more than one rank" << std::endl;
                throw InternalError();
            }
            where[e.gids[k]] = (uint32_t)j;
        }
        out.stats.nsentences += e.stats.nsentences;
        for (int n = 1; n < COLIBRI_MAX_ORDER; ++n) {
            out.stats.windows[n] += e.stats.windows[n];
            out.stats.admitted[n] += e.stats.admitted[n];
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
