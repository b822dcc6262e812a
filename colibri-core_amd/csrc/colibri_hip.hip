// colibri_hip.hip — host orchestration + the C ABI of libcolibri_hip.so (include/colibri_hip.h).
//
// The order loop of PatternModel::train (reference include/patternmodel.h:981-1270) is enqueued on one HIP
// stream with every per-order quantity (table capacity, candidate/survivor counts, result offsets, the
// "None found" termination) kept in a DevState record in HBM: kernels of order n+1 read what order n left
// there, so the host does not synchronise between orders. There is no CPU implementation behind this
// file: every entry point either runs the kernels in kernels.hpp or fails with a status.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <vector>

#include "colibri_hip.h"
#include "binned.hpp"
#include "bigram2.hpp"
#include "chain.hpp"
#include "kshard.hpp"
#include "kshard2.hpp"
#include "textenc.hpp"
#include "constrained.hpp"
#include "flexgrams.hpp"
#include "patternlist.hpp"
#include "kernels.hpp"

using namespace colibri;

namespace {

template <class T>
struct DevBuf {
    T*     p = nullptr;
    size_t n = 0;  // elements
};

struct EventPair {
    hipEvent_t a, b;
    int        cls;
};

}  // namespace

struct colibri_ctx {
    int         device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // corpus
    bool              have_corpus = false;
    bool              split_exact = false;  // sliced orders of this corpus take the exact split (a run of the direct one outgrew its room)
    uint64_t          nbytes      = 0;
    uint32_t          first_sentence = 1;
    uint32_t          npos = 0, ndelim = 0, nsent = 0, maxclass = 0, flags = 0;
    uint64_t          ntokens = 0;
    DevBuf<uint8_t>   bytes;
    DevBuf<uint32_t>  tokstart;
    DevBuf<uint32_t>  delimpos;
    DevBuf<uint32_t>  cls;              // class id per position (0 = delimiter)
    // the tokeniser's scratch, kept between uploads: five hipMalloc / hipFree pairs per upload were most of what a caller waited for beside the ~2 ms of kernels
    // (bench.py tokenise_ms_untimed, round 5: 11.6 ms)
    DevBuf<uint32_t>  tk_blockcnt, tk_total, tk_dcnt;
    DevBuf<CorpusInfo> tk_info;
    DevBuf<unsigned long long> tk_hist;
    DevBuf<PosBlock>  pos_blocks;       // per 64 positions: sentences before, start of the running sentence, delimiter bits (built at the first indexed run on a corpus)
    bool              pos_blocks_valid = false;
    DevBuf<uint32_t>  cnt1, rep1;       // order-1 fast path: count / representative position per class
    DevBuf<UniState>  unistate;         // ... its atomic-free variant: tail-bin sizes / offsets / cursors
    DevBuf<uint16_t>  uni_tail;         // ... and the tail tokens as 2-byte offsets inside their class-range bin
    DevBuf<uint32_t>  uni_rows, uni_surv; // ... per-block head histograms (rows), survivor bitmap of the classes
    DevBuf<uint32_t>  uni_resid;          // per-pass modes: result index per class
    std::vector<uint64_t> lenhist;  // sentence-length histogram (host copy)
    uint64_t              windows_n[COLIBRI_MAX_ORDER] = {0};  // W_n = n-token windows inside sentences, from the histogram (once per upload)

    // training state
    bool              trained = false;
    colibri_options   opt{};
    std::vector<DevBuf<uint32_t>> ids;  // plain mode: 2 ping-pong buffers; skipgram / indexed modes: one per order
    DevBuf<uint32_t>  scratch[2];       // per-position slot arrays of the skipgram passes
    DevBuf<uint32_t>  nsrc;             // per-slot distinct-source counter (indexed skipgrams)
    const uint32_t*   skl = nullptr;    // the list the skipgram passes of the current order walk (c->sklist, or the order's own active list), and its length
    const uint32_t*   skl_n = nullptr;
    DevBuf<uint32_t>  seglog;           // skipgram passes enqueued without read-backs: what each left (kernels.hpp: skip_pass_end_kernel)
    DevBuf<uint32_t>  skip_tmp;         // radix skipgram passes: the survivors of the distinct-fillers filter on their way back into the results
    DevBuf<unsigned long long> skip_off;
    DevBuf<unsigned long long> pairs[2];        // forward index: (result id << 32 | position) pairs, ping-pong for the radix sort (kept between runs: a release and a
                                                // new reservation of these GB-sized buffers per train() cost more than the kernels)
    DevBuf<uint32_t>  idx_cnt, sort_hist;       // ... per-block pair counts of a pass; per-block digit histograms of a sort pass
    DevBuf<unsigned long long> sort_off, sort_bsum;
    uint64_t          npairs = 0;
    DevBuf<unsigned long long> pair_chain;  // the pair counters (kernels.hpp: emit_write_kernel)
    int               pair_pass = 0;
    bool              pair_split = false;  // the pairs travel as two u32 arrays (id, sentence << tb | token) and the sort drops the id byte a pass has used (kernels.hpp: isort_*)
    uint32_t          pair_sb = 0, pair_tb = 0;  // packed pairs (id << (sb + tb) | sentence << tb | token): bits of the sentence / token fields; 0 / 0: id << 32 | position
    DevBuf<uint32_t>  ref_sentence;
    DevBuf<uint16_t>  ref_token;
    DevBuf<uint32_t>  hot_cnt;             // order 1 of an indexed model: occurrences of the hot unigrams per tile (kernels.hpp: emit_hot_*), [kHotIds][tiles]
    DevBuf<HotInfo>   hot_info;
    bool              hot_used = false;    // ... their references lie in ref_sentence / ref_token already: [hot_below, hot_below + hot_n)
    bool              lds_tested = false;   // lds_order_selftest_kernel has run on this context (pairs_begin)
    bool              hot_disorder = false, hot_off = false;  // a hot list came out of order, or a spot-checked row of the index sort (the ranks from LDS adds rest on the lanes' service
                                                             // order): this context matches every rank with ballots and sorts every reference from now on
    uint64_t          hot_below = 0, hot_n = 0;
    DevBuf<Slot>      table;
    DevBuf<Rec>       recs[2];          // binned path: record ping-pong
    DevBuf<uint32_t>  sklist, sklist_n; // skipgram passes: the positions that can take part in the current order
    DevBuf<uint32_t>  rep_of, ids_at;   // binned path: representative position per window; survivor id at representative positions
    DevBuf<uint8_t>   flags_at, flag2;  // flag mode of order 2 (KeyTrigramCls): survivor byte at representative positions / per position
    struct ConstraintSet {              // constrained training (constrained.hpp): the pattern set J and its lookup table
        DevBuf<uint8_t>            bytes;
        DevBuf<unsigned long long> off;
        DevBuf<CSlot>              table;
        DevBuf<uint32_t>           rem;    // tokens left in the sentence per position (built per corpus)
        DevBuf<uint32_t>           memb;   // pattern number of the window at each position, kProbeLengths lengths at a time
        uint32_t                   n = 0, cap = 0;
        bool                       rem_valid = false;
        bool                       closed = false;  // every pattern's prefix (without its last token) is a pattern too: the probe may stop at the first miss
        // colibri_set_continuation: the set is not a constraint but the model a continued run starts from (train(..., continued = true)): the orders it has
        // n-grams of are not counted again, their windows only get the patterns' numbers as survivor ids for the look-back of the next order
        bool                       continuation = false;
        // colibri_set_filter: the set is train()'s `filter` (patternmodel.h:1106-1133): a window is counted iff it contains one of the set's n-grams or is an
        // instance of one of its skipgrams; no look-back. shapes = the (length, gap mask) pairs of the set's skipgrams
        bool                       filter = false;
        std::vector<std::pair<int, uint32_t>> shapes;
        uint64_t                   orders[2]    = {0, 0};  // bit n: the set has n-grams of n tokens (n < 128)
        bool has_order(int n) const { return n >= 1 && n < 128 && ((orders[n >> 6] >> (n & 63)) & 1ull); }
    } cs;
    struct TextState {                  // class encoder (textenc.hpp): the uploaded text, its word table, the encoded stream
        DevBuf<uint8_t>            text, out;
        DevBuf<uint32_t>           slot_of, first, widx, wstart, wlen, wcount, cls, repeat, outlen, events, evcnt;
        uint32_t                   nevents = 0;
        DevBuf<unsigned long long> outoff, bsum, ntok;
        DevBuf<Slot>               table;
        DevBuf<DevState>           state;
        DevBuf<TextInfo>           info;
        TextInfo                   hinfo{};
        uint32_t                   n = 0, cap = 0, ndistinct = 0;
        uint64_t                   outbytes = 0, nwords = 0;
        int                        rules = -1;
        bool                       encoded = false;
    } tx;
    struct FlexState {                  // flexgrams from skipgrams (flexgrams.hpp): the result of the last colibri_flexgrams call
        DevBuf<uint8_t>            keys;
        DevBuf<unsigned long long> keyoff, refoff;
        DevBuf<uint32_t>           cnt, sentence;
        DevBuf<uint16_t>           token;
        uint64_t                   ngroups = 0, keybytes = 0, nrefs = 0;
        bool                       valid = false;
    } fx;
    struct Bigram2 {                    // second-generation order 2 (bigram2.hpp)
        DevBuf<Bi2State> state;
        DevBuf<uint32_t> boff, head_rows, wlist, wcnt, plist, bitmap, headsurv;
        DevBuf<uint8_t>  sid;                    // sliced orders: the key slice of the window at every position (first pass), read by the later passes
        DevBuf<uint32_t> wcode, pcode, headid;  // the modes that keep ids: (bin, rank) codes beside the positions, result index of every head bigram
        bool             disabled = false;  // set for the rerun after this path could not hold an order (region / bin / list overflow)
        DevBuf<uint32_t> wpre, btot;        // ... chain_bitmap_kernel's rank tables (per bitmap word / per bucket) when the forward index's pairs come straight from the lists
        bool             pairs2_direct = false;  // ... and order 2's from chain_ids_full_kernel's LDS parts (round 5)
        bool             pairs_direct = false;  // an indexed model on the chained engine: the pairs of the orders >= 3 come from chain_pairs_kernel, not from sweeps over ids per position
        DevBuf<uint32_t> steps;             // ... chain_steps_kernel: the step tables of the eight XCDs, then their lengths
        DevBuf<Bi2State> state2, state3;    // chain.hpp: orders >= 3 on this engine ping-pong between these two (odd orders: state2); order 2's stays in `state`
        hipStream_t      aux = nullptr;     // ... the hot bins' workgroup kernel runs beside the wave kernel
        hipEvent_t       ev_fork = nullptr, ev_join = nullptr, ev_fork2 = nullptr, ev_join2 = nullptr;
        bool             compact_pending = false;  // an order's survivors are being copied to the result list on the second stream (nothing on the path reads them before the export)
        bool             chain_disabled = false;  // set for the rerun after an order >= 3 did not fit the engine (key bits, a region, a bin)
        bool             attr_set = false, ids_attr_set = false;
    } b2;
    DevBuf<uint32_t>  alist[2], alist_n; // binned path: active-position lists (ping-pong) and their lengths [2]
    DevBuf<BinState>  binstate;
    bool              export_ready = false;  // keylen / keyoff / keybytes of the trained model are computed
    std::vector<uint8_t> ids_built;          // ids[n] holds THIS run's survivor ids (the id-keeping loop skips the orders nobody reads: a reader of a skipped order is refused, built_ids)
    bool              ids1_is_cls = false;   // this run keeps no per-position order-1 ids: one-token skipgram parts are named by their class ids (part_ids)
    int               last_mode = 0;    // 1 = global table, 2 = binned (what the last train() actually ran)
    int               profile_class = COLIBRI_K_COUNT;  // profile = 2: the one kernel class that is bracketed with events
    int               last_passes = 1;  // passes over key slices of the order-2 stage of that run
    int32_t           run_path = 0, run_fallback = 0, run_retries = 0;  // colibri_stats.path / fallback_reason / retries of the colibri_train call in progress
    struct Segment {
        uint32_t first, count;
        int      n;
        uint32_t mask;
    };
    std::vector<Segment> segments;      // result ranges: one per order (n-grams) and one per (order, gap mask) pass
    DevBuf<uint32_t>  res_rep, res_cnt;
    uint64_t          res_cap_used = 0;  // result capacity of the last run
    uint32_t          res_scale = 1;  // multiplier of the result capacity: raised (and the run repeated) when a corpus keeps more patterns per position than the usual bound
    DevBuf<DevState>  state;
    DevState          hstate{};
    colibri_stats     stats{};
    // export scratch
    DevBuf<uint32_t>           keylen;
    DevBuf<unsigned long long> keyoff, bsum;
    uint64_t                   keybytes = 0;

    // sentence-sharded multi-GPU state
    struct Shard {
        bool     active = false, final_level = true, use_aux = false, owner_aux = false;
        int      world = 1, n = 0, level = 0;
        uint32_t mask = 0, thr = 2, minsrc = 0;
        uint32_t ncand = 0, nrecv = 0, res_total = 0;
        uint32_t* out = nullptr;  // per-position output of the current pass (ids[n] or a scratch array)
        uint64_t exported_n[COLIBRI_MAX_ORDER] = {0}, admitted_n[COLIBRI_MAX_ORDER] = {0};
        uint32_t valid_n[COLIBRI_MAX_ORDER] = {0};
        DevBuf<uint32_t> taux, paux, onsrc;   // distinct-source counts: extracted / partitioned / owner-side sums
        DevBuf<uint32_t> res_gid, mark;       // global id of every exported pattern; per position: bit n = exported representative of an n-gram
        DevBuf<uint32_t> sorted_gid, ugid;    // forward index: sorted global ids of the pairs; distinct ids
        DevBuf<unsigned long long> uoff;      // first reference of each distinct id
        uint64_t         index_gids = 0;
        bool             radix = false, pass_radix = false, list_valid = false, pass_list = false;  // local counting of n-gram passes on the radix path (<= 128 M tokens per rank)
        uint32_t         nsparse = 0;                         // sparse candidate range of the current radix pass
        DevBuf<uint32_t> gid_of_sparse;
        DevBuf<unsigned long long> tkeys, pkeys;      // candidates: extracted, then partitioned by owner
        DevBuf<uint32_t>           tcounts, tslots, pcounts, pslots;
        DevBuf<uint32_t>           small;             // [0]=ncand, [1..64]=owner hist, [65..129]=owner offsets, [130..193]=cursors, [200..265]=src offsets
        DevBuf<Slot>               otable;            // owner-side table
        DevBuf<uint32_t>           ominrank, oslot;   // per owner slot: lowest contributing rank; per received record: owner slot
        DevBuf<DevState>           ostate;
        DevBuf<Rec>                orecs[2];          // owner-side merge on the radix path: records, bin bookkeeping, per received candidate
        DevBuf<BinState>           obin;              // the encoded survivor rank (+ export bit) and, at the exporter's candidate, the global count
        DevBuf<uint32_t>           oids_at, ocnt_at;
        bool                       merge_radix = false;
        bool                       uni_from_class = false;  // order 1 counted without representative positions
        uint32_t                   ocap = 0;
    } sh;

    // key-sharded multi-GPU state (kshard.hpp, kshard_api.inc)
    struct KShard {
        bool     active = false, first_gen = false, pending_uni = false, ran = false;  // ran: the context's trained model comes from a key-sharded run
        int      world = 1, rank = 0, n = 0, cur = 0;
        uint32_t w = 0, nclasses = 0, uni_shift = 0, clsbits = 0, posbits = 0, kbits = 0, owcap = 0, fin_total = 0, syncs = 0, head_windows = 0;
        DevBuf<KsSplitState> split;
        DevBuf<KsRouteState> rstate;  // [0] feedback, [1] exports
        DevBuf<KsStats>      stats;
        DevBuf<DevState>     ostate;  // the owner side's run state
        DevBuf<Bi2State>     obs;
        DevBuf<BinState>     obin;
        DevBuf<uint32_t>     slotbase, tab_recv, headg, oboff, owcnt, owlist, lcnt, loff, reply_at;
        DevBuf<uint32_t>     osp_rep, osp_cnt, ores_rep, ores_cnt;  // owner: sparse per-bin survivors, dense survivors of the order
        DevBuf<uint32_t>     fin_rep, fin_cnt;                      // this rank's share of the model
        DevBuf<unsigned char> sbuf, rbuf[2], fbs, exs, fbr, exr;    // records out / in (+ level-B output); feedback and exports out / in
        // second form (kshard2.hpp): the source's complete partition, the owner's per-record codes, the feedback's tiles
        DevBuf<Ks2State>  ks2;
        DevBuf<Ks2FbInfo> fbinfo;
        DevBuf<uint32_t>  tab, key4, posbuf, rowtot, code_at, tcnt, fpos, fcode, zero8k, small, cbhist;
        uint32_t          sbase[kKsWorld + 1] = {0};  // this rank's send order: first key of each owner's share
        uint32_t          bshift = 0;                 // the order's B-bin shift, agreed by all ranks
        bool              indexed = false;            // an indexed model: every order's surviving windows also become (global number, position) pairs of the local forward index
        uint32_t          gid_off = 0;                // ... the order's numbers are shifted by this (order 1: class ids; then the orders' numbers one after the other)
        DevBuf<uint32_t>  fin_gid;                    // ... and every pattern this rank exports has its global number
        Bi2State*         src_state = nullptr;        // the source side's Bi2State of the running order
        hipStream_t side = nullptr;   // the host's early looks at exchange sizes (kshard_api.inc: ks_peek_*)
        hipEvent_t  ev = nullptr;
        uint32_t*   pinned = nullptr;
    } ks;

    // profiling
    int                    profile = 0;    // 0 off, 1 every kernel class, 2 only the dominant-kernel classes
    std::vector<EventPair> events;
    std::vector<hipEvent_t> event_pool;  // events are recycled: creating/destroying ~60 of them per train() costs milliseconds
    double                 k_ms[COLIBRI_K_NCLASSES]{};
    uint64_t               k_launches[COLIBRI_K_NCLASSES]{};
};

namespace {

int fail(colibri_ctx* c, int code, const char* fmt, ...) {
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIP_TRY(ctx, call)                                                                                              \
    do {                                                                                                                \
        hipError_t e_ = (call);                                                                                         \
        if (e_ != hipSuccess) return fail((ctx), COLIBRI_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <class T>
int dev_alloc(colibri_ctx* c, DevBuf<T>& b, size_t n) {
    if (b.p && b.n >= n) return COLIBRI_OK;
    if (b.p) {
        (void)hipFree(b.p);
        b.p = nullptr;
        b.n = 0;
    }
    if (n == 0) n = 1;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&b.p), n * sizeof(T)));
    b.n = n;
    return COLIBRI_OK;
}
template <class T>
void dev_free(DevBuf<T>& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.n = 0;
}

// a temporary of one function: freed on every way out of it (the HIP_TRY early returns included)
template <class T>
struct ScopedBuf : DevBuf<T> {
    ScopedBuf() = default;
    ScopedBuf(const ScopedBuf&) = delete;
    ScopedBuf& operator=(const ScopedBuf&) = delete;
    ~ScopedBuf() { dev_free(static_cast<DevBuf<T>&>(*this)); }
};

inline uint32_t blocks_for(uint64_t n, uint32_t per) { return (uint32_t)((n + per - 1) / per); }
// grid for grid-stride streaming kernels: enough waves to fill 256 CUs x 8 blocks, no more
inline uint32_t stream_grid(uint64_t n) { return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + kBlock - 1) / kBlock, 256ull * 16)); }

struct Prof {
    colibri_ctx* c;
    int          cls;
    size_t       idx = (size_t)-1;
    Prof(colibri_ctx* c_, int cls_) : c(c_), cls(cls_) {
        if (!c->profile) return;
        if (c->profile == 2 && cls != c->profile_class) return;  // only the class that holds the dominant kernel of the path this run takes (one event pair per step)
        EventPair ev{};
        ev.cls = cls;
        auto take = [&](hipEvent_t& e) {
            if (!c->event_pool.empty()) {
                e = c->event_pool.back();
                c->event_pool.pop_back();
                return true;
            }
            return hipEventCreate(&e) == hipSuccess;
        };
        if (!take(ev.a) || !take(ev.b)) return;
        (void)hipEventRecord(ev.a, c->stream);
        c->events.push_back(ev);
        idx = c->events.size() - 1;
    }
    ~Prof() {
        if (idx != (size_t)-1) (void)hipEventRecord(c->events[idx].b, c->stream);
        static const bool dbg = getenv("COLIBRI_DEBUG_SYNC") != nullptr;  // (hunting a faulting kernel: every stage is waited for and named)
        if (dbg) {
            const hipError_t e = hipStreamSynchronize(c->stream);
            fprintf(stderr, "colibri: stage %d done (%s)\n", cls, hipGetErrorString(e));
            fflush(stderr);
        }
    }
};

void collect_events(colibri_ctx* c) {
    for (auto& ev : c->events) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
            c->k_ms[ev.cls] += ms;
            c->k_launches[ev.cls] += 1;
        }
        c->event_pool.push_back(ev.a);
        c->event_pool.push_back(ev.b);
    }
    c->events.clear();
}

// ---- corpus ingestion -----------------------------------------------------------------------------
int tokenise(colibri_ctx* c) {
    const uint64_t B = c->nbytes;
    // upper bound of positions = bytes; sized exactly after the count pass
    const uint32_t   nblk = std::max<uint32_t>(1, blocks_for(B, kTokBytesPerBlock));
    DevBuf<uint32_t>&           blockcnt = c->tk_blockcnt, &total = c->tk_total, &dcnt = c->tk_dcnt;  // (the context's: grown, never freed between uploads)
    DevBuf<CorpusInfo>&         info     = c->tk_info;
    DevBuf<unsigned long long>& hist     = c->tk_hist;
    int rc;
    if ((rc = dev_alloc(c, blockcnt, nblk + 1))) return rc;
    if ((rc = dev_alloc(c, total, 1))) return rc;
    if ((rc = dev_alloc(c, info, 1))) return rc;
    if ((rc = dev_alloc(c, hist, kLenHistBins))) return rc;
    {
        Prof p(c, COLIBRI_K_TOKENISE);
        hipLaunchKernelGGL(tokenise_count_kernel, dim3(nblk), dim3(kBlock), 0, c->stream, c->bytes.p, B, blockcnt.p);
        hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(kBlock), 0, c->stream, blockcnt.p, nblk, total.p);
    }
    uint32_t npos = 0;
    HIP_TRY(c, hipMemcpyAsync(&npos, total.p, sizeof npos, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->npos = npos;
    c->cs.rem_valid = false;  // per-position sentence remainders belong to the previous corpus
    c->pos_blocks_valid = false;
    if ((rc = dev_alloc(c, c->tokstart, (size_t)npos + 2)) || (rc = dev_alloc(c, c->cls, (size_t)npos + 128))) return rc;  // class ids are read as whole 16-byte vectors past the end (zeros)
    HIP_TRY(c, hipMemsetAsync(c->tokstart.p, 0, sizeof(uint32_t), c->stream));  // tokstart[0] = 0
    HIP_TRY(c, hipMemsetAsync(c->cls.p + npos, 0, sizeof(uint32_t) * 128, c->stream));
    HIP_TRY(c, hipMemsetAsync(info.p, 0, sizeof(CorpusInfo), c->stream));
    HIP_TRY(c, hipMemsetAsync(hist.p, 0, sizeof(unsigned long long) * kLenHistBins, c->stream));
    const uint32_t   pblk = std::max<uint32_t>(1, blocks_for(npos, kBlock));
    if ((rc = dev_alloc(c, dcnt, pblk + 1))) return rc;
    {
        Prof p(c, COLIBRI_K_TOKENISE);
        hipLaunchKernelGGL(tokenise_write_kernel, dim3(nblk), dim3(kBlock), 0, c->stream, c->bytes.p, B, blockcnt.p, c->tokstart.p);
        hipLaunchKernelGGL(position_info_kernel, dim3(pblk), dim3(kBlock), 0, c->stream, c->bytes.p, c->tokstart.p, npos, info.p, dcnt.p, c->cls.p);
        hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(kBlock), 0, c->stream, dcnt.p, pblk, total.p);
    }
    uint32_t   ndelim = 0;
    CorpusInfo hinfo{};
    HIP_TRY(c, hipMemcpyAsync(&ndelim, total.p, sizeof ndelim, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&hinfo, info.p, sizeof hinfo, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->ndelim   = ndelim;
    c->maxclass = hinfo.maxclass;
    c->flags    = hinfo.flags;
    c->ntokens  = (uint64_t)npos - ndelim;
    if ((rc = dev_alloc(c, c->delimpos, (size_t)ndelim + 1))) return rc;
    uint32_t last_delim = 0;
    {
        Prof p(c, COLIBRI_K_TOKENISE);
        hipLaunchKernelGGL(delimiter_write_kernel, dim3(pblk), dim3(kBlock), 0, c->stream, c->bytes.p, c->tokstart.p, npos, dcnt.p, c->delimpos.p);
        hipLaunchKernelGGL(sentence_length_kernel, dim3(stream_grid((uint64_t)ndelim + 1)), dim3(kBlock), 0, c->stream, c->delimpos.p, ndelim, npos, hist.p);
    }
    c->lenhist.assign(kLenHistBins, 0);
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "u64");
    HIP_TRY(c, hipMemcpyAsync(c->lenhist.data(), hist.p, sizeof(uint64_t) * kLenHistBins, hipMemcpyDeviceToHost, c->stream));
    if (ndelim) HIP_TRY(c, hipMemcpyAsync(&last_delim, c->delimpos.p + (ndelim - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    {  // W_n = sum_len hist[len] * (len - n + 1) for len >= n, via suffix sums: S0(n) = #sentences with len >= n, S1(n) = sum of their lengths
        uint64_t s0 = c->lenhist[65536], s1 = c->lenhist[65537];  // the sentences of 65536 tokens and more: their number and total length
        std::fill(std::begin(c->windows_n), std::end(c->windows_n), 0);
        for (size_t len = 65536; len-- > 1;) {
            s0 += c->lenhist[len];
            s1 += c->lenhist[len] * (uint64_t)len;
            if (len < COLIBRI_MAX_ORDER) c->windows_n[len] = s1 - s0 * (uint64_t)(len - 1);
        }
    }
    const uint32_t trailing = ndelim ? npos - (last_delim + 1) : npos;
    c->nsent                = ndelim + (trailing ? 1u : 0u);
    if (c->flags & kFlagTokenTooLong) return fail(c, COLIBRI_ERR_CORPUS, "corpus has a token longer than 8 bytes; not a valid class encoding for the accelerated path");
    if (c->flags & kFlagFlexClass)
        return fail(c, COLIBRI_ERR_CORPUS, "corpus contains the literal flexgram class {**} (04); the reference collapses these while counting, not accelerated");
    return COLIBRI_OK;
}

int ingest(colibri_ctx* c, const void* src, uint64_t nbytes, uint32_t first_sentence, hipMemcpyKind kind) {
    if (!c) return COLIBRI_ERR_ARG;
    if (!src && nbytes) return fail(c, COLIBRI_ERR_ARG, "payload is NULL");
    if (nbytes >= 0xFFFFFF00ull) return fail(c, COLIBRI_ERR_CORPUS, "corpus shard of %llu bytes exceeds the 4 GiB per-device limit (32-bit byte offsets); shard it", (unsigned long long)nbytes);
    HIP_TRY(c, hipSetDevice(c->device));
    c->have_corpus    = false;
    c->trained        = false;
    c->nbytes         = nbytes;
    c->first_sentence = first_sentence;
    const size_t padded = ((size_t)nbytes + 15) / 16 * 16 + 64;  // 16-byte tokenise loads + 15-byte hash tail over-read
    int          rc;
    if ((rc = dev_alloc(c, c->bytes, padded))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->bytes.p + nbytes, 0, padded - nbytes, c->stream));
    const auto u0 = std::chrono::steady_clock::now();
    if (nbytes) HIP_TRY(c, hipMemcpyAsync(c->bytes.p, src, nbytes, kind, c->stream));
    if (getenv("COLIBRI_HOST_TIMING")) {  // (with the C++ face's line of the same switch: the copy alone. Round 6 measured it for the 196 MB of 10^8 tokens: 3.4-3.9 ms
        // whenever the runtime pins the caller's pages and lets the copy engines read them (every upload of a Python caller; a C++ caller's first uploads), and 10-35 ms,
        // call by call, in `host_selftest bench` from the second train() on — there the copy engines are busy for < 1 ms of the call (rocprofv3 --memory-copy-trace): the
        // runtime copies through its own staging buffers with one host thread. It happens when the process' idle context is reused AND a look-up on the model (32 host
        // threads building the flat index) ran in between; a fresh context per call (COLIBRI_CTX_CACHE=0) or no look-up gives 3.5 ms again. Neither a pinned ring of the
        // context filled by eight host threads, nor registering the corpus buffer, nor 2 MB pages for it changed that (tools/notes/README.md))
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        fprintf(stderr, "COLIBRI_HOST_TIMING   upload: copy of %.1f MB %.2f ms\n", (double)nbytes / 1e6, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - u0).count());
    }
    if ((rc = tokenise(c))) return rc;
    c->have_corpus = true;
    c->split_exact = false;
    return COLIBRI_OK;
}

int check_options(colibri_ctx* c, colibri_options& o) {
    if (o.mintokens == -1) o.mintokens = 2;  // patternmodel.h:883-886
    if (o.mintokens == 0) o.mintokens = 1;
    if (o.mintokens_skipgrams < o.mintokens) o.mintokens_skipgrams = o.mintokens;  // :887-888
    if (o.maxlength < 1) return fail(c, COLIBRI_ERR_ARG, "MAXLENGTH must be >= 1");
    const bool constrained = c->cs.n != 0 && !c->cs.continuation && !c->cs.filter;  // a constraint set makes the run single-pass by definition: every threshold and minimum length is fine
    // a filtered run (patternmodel.h:1106-1133): every order counts the windows that match the filter, by their bytes, without look-back
    if (c->cs.n != 0 && c->cs.filter &&
        (o.doskipgrams || o.doskipgrams_exhaustive || o.dopatternperline || o.minlength > 1 || o.maxbackofflength < o.maxlength || o.mintokens_unigrams > o.mintokens || c->npos >= 0x7FFFFFF0u))
        return fail(c, COLIBRI_ERR_UNSUPPORTED, "training with a filter is on the accelerated path for MINLENGTH = 1, without skipgrams, back-off length, word threshold or pattern list");
    // a continued run (patternmodel.h:983-995): the orders the loaded model lacks are counted with the usual look-back, which asks the loaded patterns too
    if (c->cs.n != 0 && c->cs.continuation &&
        (o.mintokens < 2 || o.doskipgrams || o.doskipgrams_exhaustive || o.dopatternperline || o.minlength > 1 || o.maxbackofflength < o.maxlength || o.mintokens_unigrams > o.mintokens))
        return fail(c, COLIBRI_ERR_UNSUPPORTED, "continued training is on the accelerated path for MINTOKENS >= 2, MINLENGTH = 1, without skipgrams, back-off length, word threshold or pattern list");
    // skipgrams in a constrained run (patternmodel.h:941-956 turns DOSKIPGRAMS into the exhaustive kind there, for either model type): both flags mean the same
    // thing, and only a run at MINTOKENS = 1 ever computes any (:1163)
    if (constrained && (o.doskipgrams || o.doskipgrams_exhaustive) && (c->flags & kFlagSkipClass))
        return fail(c, COLIBRI_ERR_UNSUPPORTED, "corpus contains the literal skip class {*} (03): skipgrams are not accelerated for it");
    if (o.minlength < 1) o.minlength = 1;
    // MINTOKENS = 1: the reference counts all lengths in one pass without look-back (patternmodel.h:1069-1072); nothing is ever pruned, so the
    // order loop admits every window and yields the same model — with skipgrams too: every window of three or more tokens then counts all its
    // masked forms (goldens of the reference: tests/golden/*.ust1.*, *.ist1.*).
    if (o.minlength > 1 && !constrained) return fail(c, COLIBRI_ERR_UNSUPPORTED, "MINLENGTH>1 is not on the accelerated path");
    if (o.minlength > o.maxlength) return fail(c, COLIBRI_ERR_ARG, "MINLENGTH > MAXLENGTH");
    // MAXBACKOFFLENGTH < MAXLENGTH: above order MAXBACKOFFLENGTH + 1 the look-back consults the sub-patterns of MAXBACKOFFLENGTH tokens only (patternlist.hpp)
    if (o.maxbackofflength < o.maxlength && (o.maxbackofflength < 1 || o.mintokens < 2 || o.doskipgrams || o.doskipgrams_exhaustive || constrained || c->npos >= 0x7FFFFFF0u))
        return fail(c, COLIBRI_ERR_UNSUPPORTED, "MAXBACKOFFLENGTH < MAXLENGTH is on the accelerated path for MAXBACKOFFLENGTH >= 1, MINTOKENS >= 2, without skipgrams or a constraint set");
    if (o.mintokens_unigrams > o.mintokens && (constrained || o.mintokens < 2 || o.table_mode == 2))
        return fail(c, COLIBRI_ERR_UNSUPPORTED, "MINTOKENS_UNIGRAMS > MINTOKENS with a constraint set, MINTOKENS = 1 or table_mode 2 is not on the accelerated path");
    if (o.prunenonsubsumed || o.prunesubsumed) return fail(c, COLIBRI_ERR_UNSUPPORTED, "PRUNE(NON)SUBSUMED are post-hoc passes of the caller, not of colibri_train");
    // one pattern per line: the CLI's -L implies MINTOKENS = 1 and an unindexed model (src/patternmodeller.cpp:571-574, :677-678); with a higher
    // threshold the reference re-counts every line once per order until nothing new turns up — not reproduced
    if (o.dopatternperline && (o.mintokens != 1 || o.indexed || o.doskipgrams || o.doskipgrams_exhaustive || constrained || o.minlength > 1))
        return fail(c, COLIBRI_ERR_UNSUPPORTED, "DOPATTERNPERLINE is on the accelerated path with MINTOKENS = 1, MINLENGTH = 1, unindexed, without skipgrams or a constraint set");
    if (o.doskipgrams && o.doskipgrams_exhaustive && !constrained)
        return fail(c, COLIBRI_ERR_ARG, "Both DOSKIPGRAMS as well as DOSKIPGRAMS_EXHAUSTIVE are set, this shouldn't happen, choose one.");  // :958-963
    if (o.doskipgrams && !o.indexed && !constrained)
        return fail(c, COLIBRI_ERR_UNSUPPORTED, "Can not compute skipgrams on unindexed model (except exhaustively during train() )");  // reference patternmodel.h:1558
    if (o.doskipgrams_exhaustive && o.indexed && !constrained) return fail(c, COLIBRI_ERR_UNSUPPORTED, "exhaustive skipgrams on an indexed model are not on the accelerated path");
    if ((o.doskipgrams || o.doskipgrams_exhaustive) && (c->flags & kFlagSkipClass))
        return fail(c, COLIBRI_ERR_UNSUPPORTED, "corpus contains the literal skip class {*} (03): skipgram validity then follows the reference's extra checks, not accelerated");
    if (o.maxskips < 1) return fail(c, COLIBRI_ERR_ARG, "MAXSKIPS must be >= 1");
    return COLIBRI_OK;
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

int colibri_abi_version(void) { return COLIBRI_ABI_VERSION; }

int colibri_create(colibri_ctx** out, int device) {
    if (!out) return COLIBRI_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return COLIBRI_ERR_NODEVICE;
    if (device < 0 || device >= ndev) return COLIBRI_ERR_ARG;
    colibri_ctx* c = new (std::nothrow) colibri_ctx();
    if (!c) return COLIBRI_ERR_ARG;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return COLIBRI_ERR_HIP;
    }
    *out = c;
    return COLIBRI_OK;
}

void colibri_destroy(colibri_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    collect_events(c);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    c->event_pool.clear();
    dev_free(c->bytes);
    dev_free(c->tokstart);
    dev_free(c->delimpos);
    dev_free(c->pos_blocks);
    dev_free(c->cls);
    dev_free(c->tk_blockcnt);
    dev_free(c->tk_total);
    dev_free(c->tk_dcnt);
    dev_free(c->tk_info);
    dev_free(c->tk_hist);
    dev_free(c->cnt1);
    dev_free(c->rep1);
    dev_free(c->unistate);
    dev_free(c->uni_tail);
    dev_free(c->uni_rows);
    dev_free(c->uni_surv);
    dev_free(c->uni_resid);
    for (auto& b : c->ids) dev_free(b);
    dev_free(c->scratch[0]);
    dev_free(c->scratch[1]);
    dev_free(c->nsrc);
    dev_free(c->pair_chain);
    dev_free(c->skip_tmp);
    dev_free(c->seglog);
    dev_free(c->skip_off);
    dev_free(c->recs[0]);
    dev_free(c->recs[1]);
    dev_free(c->rep_of);
    dev_free(c->sklist);
    dev_free(c->sklist_n);
    dev_free(c->flags_at);
    dev_free(c->tx.text); dev_free(c->tx.out); dev_free(c->tx.slot_of); dev_free(c->tx.first); dev_free(c->tx.widx); dev_free(c->tx.wstart); dev_free(c->tx.wlen);
    dev_free(c->tx.wcount); dev_free(c->tx.cls); dev_free(c->tx.repeat); dev_free(c->tx.outlen); dev_free(c->tx.outoff); dev_free(c->tx.bsum); dev_free(c->tx.ntok);
    dev_free(c->cs.bytes); dev_free(c->cs.off); dev_free(c->cs.table); dev_free(c->cs.rem); dev_free(c->cs.memb);
    dev_free(c->fx.keys); dev_free(c->fx.keyoff); dev_free(c->fx.refoff); dev_free(c->fx.cnt); dev_free(c->fx.sentence); dev_free(c->fx.token);
    dev_free(c->tx.table); dev_free(c->tx.state); dev_free(c->tx.info); dev_free(c->tx.events); dev_free(c->tx.evcnt);
    dev_free(c->flag2);
    dev_free(c->b2.wcode); dev_free(c->b2.pcode); dev_free(c->b2.headid); dev_free(c->b2.sid);
    dev_free(c->b2.state2); dev_free(c->b2.state3); dev_free(c->b2.steps);
    dev_free(c->b2.state); dev_free(c->b2.boff); dev_free(c->b2.head_rows); dev_free(c->b2.wlist); dev_free(c->b2.wcnt); dev_free(c->b2.plist); dev_free(c->b2.bitmap); dev_free(c->b2.headsurv);
    dev_free(c->ids_at);
    dev_free(c->alist[0]);
    dev_free(c->alist[1]);
    dev_free(c->alist_n);
    dev_free(c->binstate);
    dev_free(c->pairs[0]);
    dev_free(c->pairs[1]);
    dev_free(c->idx_cnt); dev_free(c->sort_hist); dev_free(c->sort_off); dev_free(c->sort_bsum);
    dev_free(c->ref_sentence);
    dev_free(c->ref_token);
    dev_free(c->hot_cnt);
    dev_free(c->hot_info);
    dev_free(c->sh.tkeys);
    dev_free(c->sh.pkeys);
    dev_free(c->sh.tcounts);
    dev_free(c->sh.tslots);
    dev_free(c->sh.pcounts);
    dev_free(c->sh.pslots);
    dev_free(c->sh.small);
    dev_free(c->sh.otable);
    dev_free(c->sh.ominrank);
    dev_free(c->sh.oslot);
    dev_free(c->sh.ostate);
    dev_free(c->sh.orecs[0]);
    dev_free(c->sh.orecs[1]);
    dev_free(c->sh.obin);
    dev_free(c->sh.oids_at);
    dev_free(c->sh.ocnt_at);
    dev_free(c->sh.taux);
    dev_free(c->sh.paux);
    dev_free(c->sh.onsrc);
    dev_free(c->sh.res_gid);
    dev_free(c->sh.mark);
    dev_free(c->sh.sorted_gid);
    dev_free(c->sh.ugid);
    dev_free(c->sh.uoff);
    dev_free(c->sh.gid_of_sparse);
    {
        auto& k = c->ks;
        dev_free(k.split); dev_free(k.rstate); dev_free(k.stats); dev_free(k.ostate); dev_free(k.obs); dev_free(k.obin); dev_free(k.slotbase); dev_free(k.tab_recv); dev_free(k.headg);
        dev_free(k.oboff); dev_free(k.owcnt); dev_free(k.owlist); dev_free(k.lcnt); dev_free(k.loff); dev_free(k.reply_at); dev_free(k.osp_rep); dev_free(k.osp_cnt); dev_free(k.ores_rep);
        dev_free(k.ores_cnt); dev_free(k.fin_rep); dev_free(k.fin_cnt); dev_free(k.sbuf); dev_free(k.rbuf[0]); dev_free(k.rbuf[1]); dev_free(k.fbs); dev_free(k.exs); dev_free(k.fbr);
        dev_free(k.exr);
        dev_free(k.ks2); dev_free(k.fbinfo); dev_free(k.tab); dev_free(k.key4); dev_free(k.posbuf); dev_free(k.rowtot); dev_free(k.code_at); dev_free(k.tcnt); dev_free(k.fpos);
        dev_free(k.fcode); dev_free(k.zero8k); dev_free(k.small); dev_free(k.cbhist); dev_free(k.fin_gid);
        if (k.side) (void)hipStreamDestroy(k.side);
        if (k.ev) (void)hipEventDestroy(k.ev);
        if (k.pinned) (void)hipHostFree(k.pinned);
    }
    dev_free(c->table);
    dev_free(c->res_rep);
    dev_free(c->res_cnt);
    dev_free(c->state);
    dev_free(c->keylen);
    dev_free(c->keyoff);
    dev_free(c->bsum);
    if (c->b2.aux) {
        (void)hipStreamDestroy(c->b2.aux);
        (void)hipEventDestroy(c->b2.ev_fork);
        (void)hipEventDestroy(c->b2.ev_join);
        (void)hipEventDestroy(c->b2.ev_fork2);
        (void)hipEventDestroy(c->b2.ev_join2);
    }
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* colibri_last_error(const colibri_ctx* c) { return c ? c->err.c_str() : "null context"; }

int colibri_upload_corpus(colibri_ctx* c, const uint8_t* payload, uint64_t nbytes, uint32_t first_sentence) {
    return ingest(c, payload, nbytes, first_sentence, hipMemcpyHostToDevice);
}
int colibri_upload_corpus_device(colibri_ctx* c, const void* device_payload, uint64_t nbytes, uint32_t first_sentence) {
    return ingest(c, device_payload, nbytes, first_sentence, hipMemcpyDeviceToDevice);
}

int colibri_corpus_info(const colibri_ctx* c, uint64_t* ntokens, uint64_t* nsentences, uint64_t* maxclass) {
    if (!c) return COLIBRI_ERR_ARG;
    if (!c->have_corpus) return COLIBRI_ERR_STATE;
    if (ntokens) *ntokens = c->ntokens;
    if (nsentences) *nsentences = c->nsent;
    if (maxclass) *maxclass = c->maxclass;
    return COLIBRI_OK;
}

int colibri_last_mode(const colibri_ctx* c, int* passes) {
    if (!c) return 0;
    if (passes) *passes = c->last_passes;
    return c->trained ? c->last_mode : 0;
}

// what the dominant kernel of a plain run on the second-generation order 2 processed (the 64 x 64 most frequent class pairs are counted in the scan's dense LDS
// histogram and never become records): *records = 8-byte records bi2_count_kernel read in its (last) launch, *head_windows = admitted bigram windows counted in the head
int colibri_order2_records(colibri_ctx* c, uint64_t* records, uint64_t* head_windows) {
    if (!c || !records || !head_windows) return COLIBRI_ERR_ARG;
    if (!c->trained || c->last_mode != 2 || !c->b2.state.p || c->profile_class != COLIBRI_K_COUNT2) return fail(c, COLIBRI_ERR_STATE, "the last run did not count order 2 on the second-generation kernels");
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t nrec = 0;
    // (a key-sharded run: the records this rank counted as an OWNER; the head is global there and reported as 0)
    Bi2State* const bs = c->ks.ran ? c->ks.obs.p : c->b2.state.p;
    HIP_TRY(c, hipMemcpy(&nrec, &bs->nrec, sizeof nrec, hipMemcpyDeviceToHost));
    *records      = nrec;
    *head_windows = (!c->ks.ran && c->stats.admitted[2] >= nrec) ? c->stats.admitted[2] - nrec : 0;
    return COLIBRI_OK;
}

int colibri_positions(const colibri_ctx* c, uint64_t* npositions) {
    if (!c || !npositions) return COLIBRI_ERR_ARG;
    if (!c->have_corpus) return COLIBRI_ERR_STATE;
    *npositions = c->npos;
    return COLIBRI_OK;
}

}  // extern "C" (reopened after colibri_train's helpers)

// ---- helpers of colibri_train ---------------------------------------------------------------------------------
namespace {

constexpr int kMaxSkipgramTokens = 31;  // a gap mask is a uint32_t that never covers either end (reference include/pattern.h:368, src/algorithms.cpp:79-94)
struct TrainPlan {
    uint32_t npos, table_slots, res_cap, thr;
    uint32_t cnt_grid, tab_grid, pos_grid;
};

// gap masks of an n-token pattern: bit i = token i is a gap; never at either end; at most `maxskips` separate gaps when
// n - 2 >= maxskips (reference src/algorithms.cpp:79-94)
std::vector<uint32_t> gap_masks(int n, int maxskips) {
    std::vector<uint32_t> out;
    if (n < 3 || n > kMaxSkipgramTokens) return out;
    if (n - 2 < maxskips || n <= 16) {  // short patterns: the reference's own enumeration (all 2^(n-2) candidates, in its order)
        for (uint32_t i = 1; i < (1u << (n - 2)); ++i) {
            const uint32_t mask = i << 1;
            int            runs = 0, in = 0;
            for (int k = 0; k < n; ++k) {
                const int g = (mask >> k) & 1;
                runs += (g && !in);
                in = g;
            }
            if (n - 2 >= maxskips && runs > maxskips) continue;
            out.push_back(mask);
        }
        return out;
    }
    // long patterns (up to 31 tokens: a gap mask is a uint32_t, include/pattern.h:368): the same set built from its runs — 2^(n-2) candidates are 5 x 10^8 at n = 31,
    // the masks with at most `maxskips` gaps a few hundred thousand. Ascending, which is the reference's order.
    std::vector<int> b;  // run boundaries: gap runs [b0, b1), [b2, b3), ... over the inner tokens 1 .. n-2
    std::function<void(int, int)> rec = [&](int from, int runs_left) {
        if (!b.empty() && b.size() % 2 == 0) {
            uint32_t mask = 0;
            for (size_t q = 0; q < b.size(); q += 2)
                for (int k = b[q]; k < b[q + 1]; ++k) mask |= 1u << k;
            out.push_back(mask);
        }
        if (runs_left == 0) return;
        for (int s = from; s <= n - 2; ++s)
            for (int e = s + 1; e <= n - 1; ++e) {
                b.push_back(s);
                b.push_back(e);
                rec(e + 1, runs_left - 1);
                b.pop_back();
                b.pop_back();
            }
    };
    rec(1, maxskips);
    std::sort(out.begin(), out.end());
    return out;
}
// contiguous runs of non-gap tokens: (first token, length)
std::vector<std::pair<int, int>> mask_parts(uint32_t mask, int n) {
    std::vector<std::pair<int, int>> parts;
    int                              k = 0;
    while (k < n) {
        if ((mask >> k) & 1) {
            ++k;
            continue;
        }
        int e = k;
        while (e < n && !((mask >> e) & 1)) ++e;
        parts.push_back({k, e - k});
        k = e;
    }
    return parts;
}

int read_state(colibri_ctx* c) {
    HIP_TRY(c, hipMemcpyAsync(&c->hstate, c->state.p, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    if (c->hstate.overflow) return fail(c, COLIBRI_ERR_OVERFLOW, "device result buffer or table exhausted");
    return COLIBRI_OK;
}
int write_state(colibri_ctx* c) {
    HIP_TRY(c, hipMemcpyAsync(c->state.p, &c->hstate, sizeof(DevState), hipMemcpyHostToDevice, c->stream));
    return COLIBRI_OK;
}

template <class KeyFn>
void launch_count(colibri_ctx* c, const TrainPlan& pl, const KeyFn& fn, uint32_t* slot_of, int track, int cls, const uint32_t* list = nullptr, const uint32_t* nlist = nullptr) {
    Prof p(c, cls);
    hipLaunchKernelGGL((count_kernel<KeyFn>), dim3(pl.cnt_grid), dim3(kBlock), 0, c->stream, fn, slot_of, c->table.p, c->state.p, pl.npos, track, list, nlist);
}
void launch_clear(colibri_ctx* c, const TrainPlan& pl) {
    Prof p(c, COLIBRI_K_CLEAR);
    hipLaunchKernelGGL(clear_table_kernel, dim3(pl.tab_grid), dim3(kBlock), 0, c->stream, c->table.p, c->state.p);
}
void launch_prune(colibri_ctx* c, const TrainPlan& pl, uint32_t thr, const uint32_t* nsrc, uint32_t minsrc) {
    Prof p(c, COLIBRI_K_PRUNE);
    hipLaunchKernelGGL(prune_kernel, dim3(pl.tab_grid), dim3(kBlock), 0, c->stream, c->table.p, c->state.p, thr, c->res_rep.p, c->res_cnt.p, nsrc, minsrc, pl.res_cap);
}
void launch_resolve(colibri_ctx* c, const TrainPlan& pl, uint32_t* ids) {
    Prof p(c, COLIBRI_K_RESOLVE);
    hipLaunchKernelGGL(resolve_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, ids, c->table.p, c->state.p, pl.npos);
}

// ---- binned path: one order = emit -> scatter A -> hist2 -> scan -> scatter B -> per-bin LDS count -> resolve --------------
// `use_list`: iterate the active list built by the previous order's resolve instead of all positions (orders >= 3);
// `build_list`: make resolve build the list for the next order. Lists ping-pong: order n reads alist[n & 1], writes alist[(n+1) & 1].
struct BinnedIO {
    uint32_t*           sp_rep;
    uint32_t*           sp_cnt;
    unsigned long long* sp_key;  // only filled for sharded runs (the sparse arrays are then the local candidate list)
};
// class-range width of the atomic-free order 1 (kernels.hpp §2b): 256 ranges must cover every class and a range's counters must fit
// in LDS (<= 2^14 classes); 0 = not applicable (more than 4 M classes): the atomics kernel is used
constexpr uint32_t kUniHeadGrid = 512;
uint32_t uni_range_shift(const colibri_ctx* c) {
    uint32_t shift = 12;
    while (shift <= 14 && ((uint64_t)c->maxclass >> shift) >= (uint64_t)kUniBins) ++shift;
    return shift > 14 ? 0u : shift;
}
// room of a tail bin of the one-pass order 1 (kernels.hpp uni_onepass_kernel): 1.5 x the even share of ALL positions + slack, a multiple of 8 (16-byte loads)
// Round 6: a bin is kUniSub runs (one per block index mod kUniSub) — this is the room of ONE run; the array ends with a tile of slack (a tile that puts more tokens into
// the last run than the run has room for writes them all, from the run's start: ADVICE r5)
inline uint32_t uni_bin_cap(uint32_t npos) { return (uint32_t)((((uint64_t)npos / kUniBins / kUniSub) * 3 / 2 + 2048 + 7) & ~7ull); }
inline size_t   uni_tail_words(uint32_t npos) { return (size_t)kUniBins * kUniSub * uni_bin_cap(npos) + kUni1Tile + 8; }
int uni_alloc(colibri_ctx* c) {
    int rc;
    if ((rc = dev_alloc(c, c->unistate, 1)) || (rc = dev_alloc(c, c->uni_tail, uni_tail_words(c->npos))) || (rc = dev_alloc(c, c->uni_rows, (size_t)kUniHeadGrid * kUniHead)) ||
        (rc = dev_alloc(c, c->uni_surv, (size_t)c->maxclass / 32 + 4)))
        return rc;
    return COLIBRI_OK;
}
// dense per-class counts of the device's tokens into cnt (zeroed here), without a global atomic per token
int uni_count_partitioned(colibri_ctx* c, uint32_t shift, uint32_t* cnt, uint32_t nclasses) {
    HIP_TRY(c, hipMemsetAsync(cnt, 0, sizeof(uint32_t) * nclasses, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->unistate.p, 0, sizeof(UniState), c->stream));
    Prof p(c, COLIBRI_K_COUNT);
    static const bool two_pass = getenv("COLIBRI_UNI_TWO_PASS") != nullptr;  // (round 4's form, for comparison)
    if (!two_pass) {
        // COLIBRI_UNI_BIN_CAP (tests): a smaller room per tail bin, so that ordinary corpora take the overflow route (uni_tail_atomics_kernel) — the result is the same
        static const uint32_t cap_env = getenv("COLIBRI_UNI_BIN_CAP") ? (uint32_t)std::max(8l, atol(getenv("COLIBRI_UNI_BIN_CAP")) & ~7l) : 0u;
        const uint32_t cap = cap_env ? std::min(cap_env, uni_bin_cap(c->npos)) : uni_bin_cap(c->npos), nrows = (c->maxclass >> 12) + 1;
        // (measured, 10^8 tokens: this fused pass 0.43 ms for order 1; round 4's two passes 0.45; the head histogram on a second stream BESIDE a tail-only partition 0.48 —
        // the two kernels contend for the same LDS pipes)
        hipLaunchKernelGGL(uni_onepass_kernel<true>, dim3(kUniHeadGrid), dim3(kUni1Threads), 0, c->stream, c->cls.p, c->npos, cap, c->uni_rows.p, c->unistate.p, c->uni_tail.p, c->state.p);
        hipLaunchKernelGGL(uni_head_reduce_kernel, dim3(kUniHead / kBlock, 16), dim3(kBlock), 0, c->stream, c->uni_rows.p, kUniHeadGrid, cnt, nclasses, c->state.p);
        hipLaunchKernelGGL(uni_tail_count1_kernel, dim3(kUniBins * kUniSlices), dim3(kBlock), sizeof(uint32_t) * 16 * nrows, c->stream, c->uni_tail.p, c->unistate.p, cap, nrows, cnt,
                           nclasses, c->state.p);
        hipLaunchKernelGGL(uni_tail_atomics_kernel, dim3(2048), dim3(kBlock), 0, c->stream, c->cls.p, c->npos, c->unistate.p, cnt, c->state.p);
        return COLIBRI_OK;
    }
    hipLaunchKernelGGL(uni_head_kernel<true>, dim3(kUniHeadGrid), dim3(kBlock), 0, c->stream, c->cls.p, c->npos, shift, c->uni_rows.p, c->unistate.p, c->state.p);
    hipLaunchKernelGGL(uni_head_reduce_kernel, dim3(kUniHead / kBlock, 16), dim3(kBlock), 0, c->stream, c->uni_rows.p, kUniHeadGrid, cnt, nclasses, c->state.p);
    hipLaunchKernelGGL(uni_offsets_kernel, dim3(1), dim3(kBlock), 0, c->stream, c->unistate.p);
    hipLaunchKernelGGL(uni_partition_kernel, dim3(256 * 4), dim3(kBlock), 0, c->stream, c->cls.p, c->npos, shift, c->unistate.p, c->uni_tail.p, c->state.p);
    hipLaunchKernelGGL(uni_tail_count_kernel, dim3(kUniBins * kUniSlices), dim3(kBlock), sizeof(uint32_t) << shift, c->stream, c->uni_tail.p, c->unistate.p, shift, cnt, nclasses,
                       c->state.p);
    return COLIBRI_OK;
}

BinnedIO binned_planes(colibri_ctx* c, const TrainPlan& pl, bool with_keys) {
    // the sparse survivor arrays of an order live in recs[0] (free again after scatter B): u32 planes of npos entries, then u64 keys
    BinnedIO io{};
    io.sp_rep = reinterpret_cast<uint32_t*>(c->recs[0].p);
    io.sp_cnt = io.sp_rep + pl.npos;
    io.sp_key = with_keys ? reinterpret_cast<unsigned long long*>(io.sp_rep + 2 * (size_t)pl.npos) : nullptr;
    return io;
}

template <class KeyFn>
int binned_count_stage(colibri_ctx* c, const TrainPlan& pl, const KeyFn& fn, int n, bool use_list, uint32_t thr, bool with_keys, bool need_ids = true, bool flag_mode = false,
                       bool dense_code = false, uint32_t sbits = 0, uint32_t slice = 0, const uint32_t* list_other = nullptr, const uint32_t* nlist_other = nullptr) {
    const uint32_t  tiles    = blocks_for(pl.npos, kScatTile) + 1 + kASlots;  // level-B tiles: every slot may end in a partial one
    const uint32_t* list_in  = list_other ? list_other : c->alist[n & 1].p;  // list_other: a list of the caller's (the skipgram passes walk c->sklist)
    const uint32_t* nlist_in = list_other ? nlist_other : c->alist_n.p + (n & 1);
    HIP_TRY(c, hipMemsetAsync(c->binstate.p, 0, sizeof(BinState), c->stream));
    uint32_t* const ids_at   = (need_ids && !flag_mode) ? c->ids_at.p : nullptr;  // reset by the emit kernel at every record position; not needed when nobody resolves ids
    uint8_t* const  flags_at = (need_ids && flag_mode) ? c->flags_at.p : nullptr;  // flag mode: a survivor byte instead of a survivor id
    // records leave the emit kernel already partitioned by A bin into fixed-capacity (sub-)regions of recs[0]; level B moves them to recs[1]
    const uint32_t region = (uint32_t)(c->recs[0].n / kASlots);
    {
        Prof p(c, COLIBRI_K_EMIT);
        if (use_list)
            hipLaunchKernelGGL((bin_emit_kernel<KeyFn, true>), dim3(pl.cnt_grid), dim3(kBlock), 0, c->stream, fn, c->recs[0].p, region, c->rep_of.p, c->state.p, c->binstate.p, pl.npos,
                               list_in, nlist_in, ids_at, flags_at, sbits, slice);
        else
            hipLaunchKernelGGL((bin_emit_kernel<KeyFn, false>), dim3(pl.cnt_grid), dim3(kBlock), 0, c->stream, fn, c->recs[0].p, region, c->rep_of.p, c->state.p, c->binstate.p, pl.npos,
                               (const uint32_t*)nullptr, (const uint32_t*)nullptr, ids_at, flags_at, sbits, slice);
    }
    {
        Prof p(c, COLIBRI_K_SCATTER);
        hipLaunchKernelGGL(bin_offsets_kernel, dim3(1), dim3(kBlock), 0, c->stream, c->binstate.p, region);
        hipLaunchKernelGGL(bin_hist2_kernel, dim3(tiles + kBins), dim3(kBlock), 0, c->stream, c->recs[0].p, c->state.p, c->binstate.p);
        hipLaunchKernelGGL(bin_scan2_kernel, dim3(kBins), dim3(kBlock), 0, c->stream, c->binstate.p);
        hipLaunchKernelGGL(bin_scatter_kernel, dim3(tiles + kBins), dim3(kBlock), 0, c->stream, c->recs[0].p, c->recs[1].p, c->state.p, c->binstate.p);
    }
    const BinnedIO io = binned_planes(c, pl, with_keys);
    {
        static const int walk_mode = getenv("COLIBRI_BIN_WALK") ? atoi(getenv("COLIBRI_BIN_WALK")) : 0;  // (experiments: 1 = fixed shares, 2 = queues; default: by the largest bin)
        if (walk_mode) HIP_TRY(c, hipMemcpyAsync(&c->binstate.p->walk_mode, &walk_mode, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        Prof p(c, COLIBRI_K_BINCOUNT);
        hipLaunchKernelGGL(bin_count_kernel, dim3(256 * 12), dim3(kBlock), 0, c->stream, c->recs[1].p, c->state.p, c->binstate.p, thr, io.sp_rep, io.sp_cnt, io.sp_key, ids_at, flags_at, dense_code);
    }
    return COLIBRI_OK;
}

int binned_resolve_stage(colibri_ctx* c, const TrainPlan& pl, uint32_t* ids_out, int n, bool use_list, bool build_list, const uint32_t* remap, uint32_t remap_base,
                         bool prefill_ids = true, bool decode = false, uint32_t decode_base = 0, const uint32_t* list_other = nullptr, const uint32_t* nlist_other = nullptr) {
    const uint32_t* list_in   = list_other ? list_other : c->alist[n & 1].p;
    const uint32_t* nlist_in  = list_other ? nlist_other : c->alist_n.p + (n & 1);
    uint32_t*       list_out  = build_list ? c->alist[(n + 1) & 1].p : nullptr;
    uint32_t*       nlist_out = c->alist_n.p + ((n + 1) & 1);
    if (build_list || !list_other) HIP_TRY(c, hipMemsetAsync(nlist_out, 0, sizeof(uint32_t), c->stream));
    if (use_list && prefill_ids) HIP_TRY(c, hipMemsetAsync(ids_out, 0xFF, sizeof(uint32_t) * (size_t)pl.npos, c->stream));
    Prof p(c, COLIBRI_K_RESOLVE);
    if (use_list)
        hipLaunchKernelGGL((bin_resolve_kernel<true>), dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->rep_of.p, c->ids_at.p, ids_out, c->state.p, pl.npos, list_in, nlist_in, list_out,
                           nlist_out, remap, remap_base, decode ? (const BinState*)c->binstate.p : (const BinState*)nullptr, decode_base);
    else
        hipLaunchKernelGGL((bin_resolve_kernel<false>), dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->rep_of.p, c->ids_at.p, ids_out, c->state.p, pl.npos, (const uint32_t*)nullptr,
                           (const uint32_t*)nullptr, list_out, nlist_out, remap, remap_base, decode ? (const BinState*)c->binstate.p : (const BinState*)nullptr, decode_base);
    return COLIBRI_OK;
}

// ---- order 2, second generation (bigram2.hpp): class-keyed 8-byte records, dense head, per-slot level B, one wave per final bin, position
// lists -> bitmap -> the active list of order 3. `want_list`: order 3 follows. Everything is enqueued; nothing is read back.
#ifndef COLIBRI_BI2_WPC
#define COLIBRI_BI2_WPC 20  // (waves of the count kernel per CU: what its 16-bit-counter form keeps resident; the other forms — 9+ KB of LDS — run 17 of them at a time)
#endif
#ifndef COLIBRI_BI2_SUB
#define COLIBRI_BI2_SUB 4
#endif
// sub-regions per A bin. A final bin is one run per sub-region: fewer runs make the count kernel cheaper (its run look-up is ~half its instructions at 8), more make
// the emit kernel's cursor reservations and level B's blocks cheaper. Measured per 10^8-token step on one box: 8 / 4 / 2 / 1 sub-regions 4.54 / 4.44 / 4.47 / 4.56 ms
// (count 0.85 / .. / 0.76 / 0.75, level B 0.48 / .. / 0.59 / 0.56, emit 0.51 / .. / 0.51 / 0.63). The split of corpora beyond one pass and the multi-GPU source
// side keep eight (kBi2SubWide): their sub-regions double as source ranks / carry three position bits.
constexpr uint32_t kBi2Sub = COLIBRI_BI2_SUB, kBi2SubWide = 8, kBi2EmitGrid = 512, kBi2Waves = 256 * COLIBRI_BI2_WPC;
// records a pass of the radix path takes on (final bins of ~700-1500 records). COLIBRI_SLICE_POSITIONS (tests): a smaller number, so that small corpora
// exercise the sliced passes of the path for corpora beyond ~128 M tokens per device
// Round 4: ONE pass of the second-generation engine holds ~2 x 10^8 positions of the bench distribution (a final bin's LDS table takes ~2500 distinct keys; at 2.5 x 10^8
// tokens bins overflow: Bi2State.overflow 2), and one full pass beats two half-full ones (200 M tokens: 9.3 against 11.4 ms; the id-keeping kinds, which have no sliced
// form, 23.6 / 19.8 against 52 / 58 ms on the global table). A run whose bins do overflow in that range repeats with the old pass size (tl_small_passes), not on the
// first-generation kernels.
thread_local bool tl_small_passes = false;
inline uint64_t slice_env() {
    static const uint64_t env = [] {
        const char* e = getenv("COLIBRI_SLICE_POSITIONS");
        return (uint64_t)(e ? strtoull(e, nullptr, 10) : 0ull);
    }();
    return env;
}
// records per pass ONCE an order is sliced: never more than round 3's size, whatever the environment says (fuller passes overflow their bins; a probe with four passes of
// 2.6 x 10^8 records ended in a memory fault, not in an overflow flag)
inline uint64_t slice_positions() { return slice_env() ? std::min<uint64_t>(slice_env(), 110ull * 1000 * 1000) : 110ull * 1000 * 1000; }
// ... and what one pass takes alone: 2.15 x 10^8 with the count kernels' 1024-slot bin tables (and the chained orders), twice that with their 2048-slot form (order 2
// of corpora in between: bigram2_order's `wide`; their orders >= 3 run round 3's kernels — a chained record has no room for 29 position bits beside its key)
constexpr uint64_t kNarrowPassPositions = 215ull * 1000 * 1000, kWidePassPositions = 430ull * 1000 * 1000;
inline uint64_t single_pass_positions() { return slice_env() ? slice_env() : tl_small_passes ? 110ull * 1000 * 1000 : kWidePassPositions; }
inline bool retry_with_small_passes(uint64_t npos) { return !tl_small_passes && !slice_env() && npos > 110ull * 1000 * 1000; }
inline uint64_t big_corpus_tokens() { return slice_env() ? slice_env() : tl_small_passes ? 128ull * 1000 * 1000 : 400ull * 1000 * 1000; }
inline bool chain_fits(uint64_t npos) { return slice_env() ? true : npos <= kNarrowPassPositions; }  // (one pass on the 1024-slot tables, <= 28 position bits)
// Round 5, plain runs: the chained orders also between 2.15 and 4.3 x 10^8 positions — eight sub-regions, the 2048-slot count kernels, and records whose position lacks
// the three bits that equal their sub-region (kshard2.hpp's pdrop: 29 position bits beside a 37-bit key do not fit 64) — COLIBRI_NO_WIDE_CHAIN: round 3's orders >= 3 there
inline bool chain_wide(uint64_t npos) {
    static const bool force = getenv("COLIBRI_FORCE_WIDE_CHAIN") != nullptr;  // (tests: small corpora through the wide form of the chained orders; same model)
    return force || (!slice_env() && npos > kNarrowPassPositions);
}
inline bool chain_fits_plain(uint64_t npos) { return chain_fits(npos) || (npos <= kWidePassPositions && !getenv("COLIBRI_NO_WIDE_CHAIN")); }
inline uint32_t slice_bits(uint64_t records) {  // passes needed for that many records, as a power of two (at most 64)
    if (records <= single_pass_positions()) return 0;
    // once an order is sliced, fuller passes are cheaper (the per-bin cost of the count kernels is mostly fixed): up to 14/11 of the single-pass size each
    uint32_t s = 1;
    while (s < 6 && ((slice_positions() * 14 / 11) << s) < records) ++s;
    return s;
}
struct Bigram2Plan {
    uint32_t nslots, region, pshift, nbuckets, wcap, wextra, clsbits, posbits, sbits;
    size_t   listn;  // entries of plist / pcode when the head windows' lists are behind the shards' (chain.hpp)
    Bi2Lists pl;
};
Bigram2Plan bigram2_plan(const colibri_ctx* c, uint32_t npos, uint32_t nsub = kBi2Sub) {
    Bigram2Plan b{};
    b.nslots = kBins * nsub;
    // records are 8 bytes: recs[0] (level-A output) and recs[1] (level-B output) hold twice their Rec capacity
    b.region = (uint32_t)(std::min<uint64_t>(2ull * c->recs[0].n, 2ull * c->recs[1].n) / b.nslots);
    // a pass over one slice of a big corpus fills a fraction of that: keep its slots close together (a bin's eight runs then lie ~MBs, not ~GBs, apart)
    if (slice_bits(npos)) b.region = (uint32_t)std::min<uint64_t>(b.region, (slice_positions() * 5 / 2) / b.nslots + 8192);
    b.pshift = 12;
    while (((uint64_t)npos >> b.pshift) > (uint64_t)kBi2Buckets - 1) ++b.pshift;
    b.nbuckets  = std::max<uint32_t>(1u, (uint32_t)(((uint64_t)npos + (1u << b.pshift) - 1) >> b.pshift));
    b.pl.pshift = b.pshift;
    b.pl.pcap   = ((1u << b.pshift) / 4 + 4096 + 3) & ~3u;  // a bucket's entries spread evenly over the 8 shards: twice the expected worst case
    b.pl.hbase  = kBi2Shards * kBi2Buckets * b.pl.pcap;     // (chain.hpp) the head windows' lists lie behind the shards'
    b.listn     = (size_t)b.pl.hbase + ((size_t)b.nbuckets << b.pshift) + 64;
    // a wave's list: twice its even share of the ~0.6 npos surviving windows — of the waves that are RESIDENT at a time: the 2048-slot count kernels of corpora beyond
    // 2.15 x 10^8 positions hold 17 KB of LDS each, nine per CU, and the first 2304 of the launched waves draw every bin (a run that lost windows this way falls back)
    const uint32_t resident = (!slice_env() && npos > kNarrowPassPositions) ? std::min(kBi2Waves, 256u * 9u) : kBi2Waves;
    b.wcap      = (uint32_t)(((uint64_t)npos * 6 / 10 / resident) * 2 + 4096);
    b.wextra    = npos / b.wcap + kBi2Waves + 64;  // the list pool of bi2_count_big_kernel: every position once, one partly filled list per WAVE of its grid (round 4 counted its
                                                   // blocks: 375 M tokens with ~10^3 hot trigram bins ran the pool dry — Bi2State.overflow 3 — once the lists grew)
    b.clsbits   = 1;
    while ((1ull << b.clsbits) <= (uint64_t)c->maxclass) ++b.clsbits;
    b.posbits = 1;
    while ((1ull << b.posbits) < (uint64_t)npos + 1) ++b.posbits;
    b.posbits = std::max(b.posbits, 27u);
    b.sbits   = slice_bits(npos);
    return b;
}
// the record must hold the mix bits below the A bin next to the position
bool bigram2_fits(const colibri_ctx* c, uint32_t npos) {
    const Bigram2Plan b = bigram2_plan(c, npos);
    return npos < (1u << kBi2MaxPosBits) && 2 * b.clsbits - b.sbits - 8 + b.posbits <= 64;
}
int bigram2_alloc(colibri_ctx* c, uint32_t npos, bool chain = false) {
    const Bigram2Plan b = bigram2_plan(c, npos);
    int               rc;
    if ((rc = dev_alloc(c, c->b2.state, 1)) || (rc = dev_alloc(c, c->b2.boff, (size_t)kBins * kBi2SubWide * (kBi2BBins + 1))) ||
        (rc = dev_alloc(c, c->b2.head_rows, (size_t)kBi2EmitGrid * 2 * kBi2HeadN)) || (rc = dev_alloc(c, c->b2.wlist, (size_t)(kBi2Waves + b.wextra) * b.wcap)) ||
        (rc = dev_alloc(c, c->b2.wcnt, (size_t)kBi2Waves + b.wextra + 1)) || (rc = dev_alloc(c, c->b2.plist, chain ? b.listn : (size_t)kBi2Shards * kBi2Buckets * b.pl.pcap + 64)) ||
        (rc = dev_alloc(c, c->b2.bitmap, (size_t)npos / 32 + 24)) || (rc = dev_alloc(c, c->b2.headsurv, kBi2HeadN / 32)))
        return rc;
    if (chain && c->b2.aux == nullptr) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->b2.aux, hipStreamNonBlocking));  // (round 6 measured the highest stream priority here, for the hot bins' kernel: no difference)
        HIP_TRY(c, hipEventCreateWithFlags(&c->b2.ev_fork, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->b2.ev_join, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->b2.ev_fork2, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->b2.ev_join2, hipEventDisableTiming));
    }
    if (chain && ((rc = dev_alloc(c, c->b2.steps, 2 * (size_t)kChXcds * chain_steps_cap(b.pl) + kChXcds)) || (rc = dev_alloc(c, c->b2.state2, 1)) || (rc = dev_alloc(c, c->b2.state3, 1)) || (rc = dev_alloc(c, c->b2.wcode, (size_t)(kBi2Waves + b.wextra) * b.wcap)) ||
                  (rc = dev_alloc(c, c->b2.pcode, b.listn)) || (rc = dev_alloc(c, c->b2.headid, kBi2HeadN))))
        return rc;
    if (b.sbits) {
        if ((rc = dev_alloc(c, c->b2.sid, (size_t)npos + 64))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->b2.sid.p + (npos & ~15u), 0xFF, 64, c->stream));  // the 16-byte loads of the last positions read "no window" beyond the corpus
    }
    if (!c->b2.attr_set) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)bi2_bitmap_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)1 << kBi2MaxPosBits) / kBi2Buckets / 8)));
        HIP_TRY(c, hipFuncSetAttribute((const void*)chain_bitmap_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)1 << kBi2MaxPosBits) / kBi2Buckets / 8)));
        c->b2.attr_set = true;
    }
    return COLIBRI_OK;
}
// ids_out (the modes that keep every order's ids; one pass only): also the RESULT index of the bigram at every position, kInvalid where it did not survive
// The copy of an order's survivors into the dense result list (bi2_compact_kernel) is read by nobody before the export: in a chained run it leaves the path and runs on
// the second stream while the main one sorts the pairs and builds the bitmap (0.06 + 0.03 + ... ms per step). It reads the sparse arrays in recs[0], which the NEXT
// order's emit kernel overwrites: chain_compact_join before that, and before the host looks at the results.
int chain_compact_fork(colibri_ctx* c, const BinnedIO& io, const Bi2State* bs, const TrainPlan& pl) {
    HIP_TRY(c, hipEventRecord(c->b2.ev_fork2, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->b2.aux, c->b2.ev_fork2, 0));
    hipLaunchKernelGGL(bi2_compact_kernel, dim3(1025), dim3(kBlock), 0, c->b2.aux, (const uint32_t*)io.sp_rep, (const uint32_t*)io.sp_cnt, (const DevState*)c->state.p, bs, c->res_rep.p,
                       c->res_cnt.p, pl.res_cap, true);
    HIP_TRY(c, hipEventRecord(c->b2.ev_join2, c->b2.aux));
    c->b2.compact_pending = true;
    return COLIBRI_OK;
}
int chain_compact_join(colibri_ctx* c) {
    if (!c->b2.compact_pending) return COLIBRI_OK;
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->b2.ev_join2, 0));
    c->b2.compact_pending = false;
    return COLIBRI_OK;
}
// Grids of the kernels that walk chain_steps_kernel's tables: block b works for XCD b % 8 and emits into sub-region b % nsub, so a grid must be a multiple of both —
// a tuning override that is not would make blocks repeat other blocks' steps (duplicated records). The environment value is rounded up to the next multiple of 8 (which
// kBi2Sub and kBi2SubWide divide); 0 or garbage falls back to the default.
inline uint32_t chain_grid(const char* env_name, uint32_t dflt) {
    const char* e = getenv(env_name);
    const long  v = e ? atol(e) : 0;
    if (v <= 0 || v > 65536) return dflt;
    return (uint32_t)((v + 7) / 8 * 8);
}
static_assert(8 % COLIBRI_BI2_SUB == 0, "chain_grid rounds to multiples of 8");
// chain_emit_kernel's measurement switches (bit 0 / 1: skip the bitmap / class gather — WRONG models, right clocks; bits 8+: the hot-bin limit) exist in experimental
// builds only (-DCOLIBRI_DEBUG_KNOBS): a stray environment variable must not be able to change a product run's result.
inline uint32_t chain_dbg() {
#ifdef COLIBRI_DEBUG_KNOBS
    static const uint32_t v = getenv("COLIBRI_CH_DBG") ? (uint32_t)atoi(getenv("COLIBRI_CH_DBG")) : 0u;
    return v;
#else
    return 0u;
#endif
}
// result index per position from the (position, dense number) pairs an order left in the position lists (chain_ids_kernel), `ids` pre-filled with kInvalid
// bucket windows of up to 2^18 positions are built part by part in LDS and written as whole lines (chain_ids_full_kernel); larger ones (corpora beyond 2.7 x 10^8
// positions) and COLIBRI_IDS_SCATTER: round 4's scatter into the pre-filled array
inline bool chain_ids_full_applies(const Bigram2Plan& b) {
    static const bool scatter = getenv("COLIBRI_IDS_SCATTER") != nullptr;
    return !scatter && b.pshift <= 18;
}
// pairs: also the forward index's pairs of the order, from the same LDS parts (indexed chained runs, order 2; the bitmap kernel has left wpre / btot); ids may then be null
int chain_ids(colibri_ctx* c, const Bigram2Plan& b, const Bi2State* bs, uint32_t* ids, const uint32_t* headid, bool pairs = false) {
    if (chain_ids_full_applies(b)) {
        ChainPairsOut po{};
        if (pairs) {
            po.wpre   = c->b2.wpre.p;
            po.btot   = c->b2.btot.p;
            po.blocks = reinterpret_cast<const uint4*>(c->pos_blocks.p);
            po.chain  = c->pair_chain.p;
            po.pairs  = c->pairs[0].p;
            po.pcap   = c->pairs[0].n;
            po.pay    = c->pair_split ? reinterpret_cast<uint32_t*>(c->pairs[0].p) + po.pcap : (uint32_t*)nullptr;
            po.which  = c->pair_pass;
            po.sb     = c->pair_sb;
            po.tb     = c->pair_tb;
        }
        static const uint32_t plog_min = getenv("COLIBRI_IDS_PLOG") ? (uint32_t)atoi(getenv("COLIBRI_IDS_PLOG")) : 2u;  // (parts per bucket, log2; measured at 10^8 tokens, indexed model: 2 / 3 / 4 -> 9.30 / 9.62 / 10.13 ms — the parts' re-reads of the lists reach HBM)
        const uint32_t plog = std::min(b.pshift, std::max(std::min(plog_min, 6u), b.pshift > 15 ? b.pshift - 15 : 0u));
        if (!c->b2.ids_attr_set) {
            HIP_TRY(c, hipFuncSetAttribute((const void*)chain_ids_full_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            c->b2.ids_attr_set = true;
        }
        hipLaunchKernelGGL(chain_ids_full_kernel, dim3(((b.nbuckets + kChXcds - 1) / kChXcds) * (kChXcds << plog)), dim3(kBi2Threads), sizeof(uint32_t) << (b.pshift - plog), c->stream,
                           (const uint32_t*)c->b2.plist.p, (const uint32_t*)c->b2.pcode.p, b.pl, b.nbuckets, plog, c->npos, bs, (const DevState*)c->state.p, ids, headid, po);
        if (pairs) {
            hipLaunchKernelGGL(pairs_advance_kernel, dim3(1), dim3(1), 0, c->stream, c->pair_chain.p, c->pair_pass, (const uint32_t*)&c->state.p->valid, po.pcap);
            c->pair_pass ^= 1;
        }
        return COLIBRI_OK;
    }
    if (pairs || ids == nullptr) return fail(c, COLIBRI_ERR_STATE, "chain_ids: pairs need the LDS form");
    HIP_TRY(c, hipMemsetAsync(ids, 0xFF, sizeof(uint32_t) * (size_t)c->npos, c->stream));
    const uint32_t cap = chain_steps_cap(b.pl);
    hipLaunchKernelGGL(chain_steps_kernel, dim3(kChXcds), dim3(kBi2Threads), 0, c->stream, bs, b.pl, b.nbuckets, reinterpret_cast<uint2*>(c->b2.steps.p), cap,
                       c->b2.steps.p + 2 * (size_t)kChXcds * cap, (const DevState*)c->state.p);
    static const uint32_t ids_grid = chain_grid("COLIBRI_IDS_GRID", 1024u);
    hipLaunchKernelGGL(chain_ids_kernel, dim3(ids_grid), dim3(kChThreads), 0, c->stream, (const uint32_t*)c->b2.plist.p, (const uint32_t*)c->b2.pcode.p,
                       reinterpret_cast<const uint2*>(c->b2.steps.p), cap, (const uint32_t*)(c->b2.steps.p + 2 * (size_t)kChXcds * cap), bs, (const DevState*)c->state.p, ids, headid);
    return COLIBRI_OK;
}
// the forward index's pairs of the order whose lists, bitmap and rank tables are in place (chain_pairs_kernel); the pair counters advance by the order's valid positions
void chain_pairs(colibri_ctx* c, const Bigram2Plan& b, const Bi2State* bs, const uint32_t* headid) {
    Prof           p(c, COLIBRI_K_INDEX);
    const uint32_t cap  = chain_steps_cap(b.pl);
    const uint64_t pcap = c->pairs[0].n;
    hipLaunchKernelGGL(chain_steps_kernel, dim3(kChXcds), dim3(kBi2Threads), 0, c->stream, bs, b.pl, b.nbuckets, reinterpret_cast<uint2*>(c->b2.steps.p), cap,
                       c->b2.steps.p + 2 * (size_t)kChXcds * cap, (const DevState*)c->state.p);
    static const uint32_t grid = chain_grid("COLIBRI_IDS_GRID", 1024u);
    hipLaunchKernelGGL(chain_pairs_kernel, dim3(grid), dim3(kChThreads), 0, c->stream, (const uint32_t*)c->b2.plist.p, (const uint32_t*)c->b2.pcode.p,
                       reinterpret_cast<const uint2*>(c->b2.steps.p), cap, (const uint32_t*)(c->b2.steps.p + 2 * (size_t)kChXcds * cap), bs, (const DevState*)c->state.p,
                       (const uint32_t*)c->b2.bitmap.p, (const uint32_t*)c->b2.wpre.p, (const uint32_t*)c->b2.btot.p, b.nbuckets, b.pshift,
                       reinterpret_cast<const uint4*>(c->pos_blocks.p), (const unsigned long long*)c->pair_chain.p, c->pair_pass, pcap, c->pairs[0].p,
                       c->pair_split ? reinterpret_cast<uint32_t*>(c->pairs[0].p) + pcap : (uint32_t*)nullptr, c->pair_sb, c->pair_tb, headid);
    hipLaunchKernelGGL(pairs_advance_kernel, dim3(1), dim3(1), 0, c->stream, c->pair_chain.p, c->pair_pass, (const uint32_t*)&c->state.p->valid, pcap);
    c->pair_pass ^= 1;
}
// chain (chain.hpp: order 3 runs on this engine too): instead of the bitmap -> list of order 3, the (position, code) pairs of the surviving windows sorted into position
// buckets and the bitmap with the head survivors in it — what chain_emit_kernel walks
int bigram2_order(colibri_ctx* c, const TrainPlan& pl, bool want_list, uint32_t* ids_out = nullptr, bool chain = false) {
    const uint32_t    npos = pl.npos, nsurv = c->maxclass / 32 + 1;
    const Bigram2Plan b    = bigram2_plan(c, npos);
    if (ids_out != nullptr) {
        int rc;
        if (b.sbits != 0 || !want_list) return fail(c, COLIBRI_ERR_STATE, "bigram2_order: ids need the single-pass form with lists");
        if ((rc = dev_alloc(c, c->b2.wcode, (size_t)(kBi2Waves + b.wextra) * b.wcap)) || (rc = dev_alloc(c, c->b2.pcode, (size_t)kBi2Shards * kBi2Buckets * b.pl.pcap + 64)) ||
            (rc = dev_alloc(c, c->b2.headid, kBi2HeadN)) || (rc = dev_alloc(c, c->b2.steps, 2 * (size_t)kChXcds * chain_steps_cap(b.pl) + kChXcds)))
            return rc;
    }
    if (chain && b.sbits != 0) return fail(c, COLIBRI_ERR_STATE, "bigram2_order: the chained orders need the single-pass form");
    const bool        with_codes = ids_out != nullptr || (chain && want_list);
    Bi2State* const   bs   = c->b2.state.p;
    auto* const       recsA = reinterpret_cast<unsigned long long*>(c->recs[0].p);
    auto* const       recsB = reinterpret_cast<unsigned long long*>(c->recs[1].p);
    uint32_t* const   nlist = c->alist_n.p + 1;  // order 3 reads alist[3 & 1]
    // a chained run's order 2 starts from ONE launch that clears its state and the lists' counts (as every later order does); the other forms keep their fills
    const bool one_reset = chain && b.sbits == 0;
    if (one_reset) {
        hipLaunchKernelGGL(chain_reset_kernel, dim3(256), dim3(kBlock), 0, c->stream, bs, c->b2.wcnt.p, kBi2Waves + b.wextra + 1);
    } else {
        HIP_TRY(c, hipMemsetAsync(nlist, 0, sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->b2.wcnt.p, 0, sizeof(uint32_t) * ((size_t)kBi2Waves + b.wextra + 1), c->stream));  // (the last word: the pool's cursor between the passes of a sliced order)
    }
    uint32_t* const   keep = c->b2.wcnt.p + kBi2Waves + b.wextra;
    const BinnedIO io = binned_planes(c, pl, false);  // the sparse survivor arrays live in recs[0], free again after level B
    for (uint32_t slice = 0; slice < (1u << b.sbits); ++slice) {  // one pass per slice of the keys (one pass unless the corpus exceeds ~110 M positions)
        if (!one_reset) HIP_TRY(c, hipMemsetAsync(bs, 0, sizeof(Bi2State), c->stream));
        {
            Prof p(c, COLIBRI_K_EMIT2);
            if (slice == 0 || b.sbits < 2 || getenv("COLIBRI_SLICED_EMIT_OFF"))  // (two slices: a step of 16 384 positions would fill the queue)
                hipLaunchKernelGGL(bi2_emit_kernel, dim3(kBi2EmitGrid), dim3(kBi2Threads), 0, c->stream, c->cls.p, c->uni_surv.p, nsurv, npos, b.clsbits, b.sbits, slice, b.posbits, recsA,
                                   b.region, kBi2Sub, bs, c->state.p, c->b2.head_rows.p, b.sbits ? c->b2.sid.p : (uint8_t*)nullptr, 0u, (chain && want_list) ? c->b2.plist.p : (uint32_t*)nullptr,
                                   (chain && want_list) ? c->b2.pcode.p : (uint32_t*)nullptr, b.pl);
            else  // the first pass left every window's slice in sid: the later ones only touch their own windows
                hipLaunchKernelGGL(bi2_emit_sliced_kernel, dim3(kBi2EmitGrid), dim3(kBi2Threads), 0, c->stream, c->cls.p, (const uint8_t*)c->b2.sid.p, npos, b.clsbits, b.sbits, slice, b.posbits,
                                   std::max(1u, std::min(8u, 1u << b.sbits) / 4u), recsA, b.region, kBi2Sub, bs, c->state.p);
            if (slice == 0)
                hipLaunchKernelGGL(bi2_head_reduce_kernel, dim3(kBi2HeadN / kBlock, kBi2HeadSplit), dim3(kBlock), 0, c->stream, c->b2.head_rows.p, kBi2EmitGrid, bs, c->state.p);
            // (records per A bin, scan, B-bin shift: the emit kernel's last block, bi2_offsets_tail)
        }
        {
            Prof p(c, COLIBRI_K_LEVELB2);
            hipLaunchKernelGGL(bi2_levelB_kernel<false>, dim3(b.nslots), dim3(kBi2Threads), 0, c->stream, recsA, recsB, b.region, bs, c->b2.boff.p, c->state.p);
            hipLaunchKernelGGL(bi2_binoff_kernel, dim3(kBins), dim3(kBi2BBins), 0, c->stream, bs, c->b2.boff.p, kBi2Sub, c->state.p);
        }
        {
            Prof p(c, COLIBRI_K_COUNT2);
            if (b.sbits) hipLaunchKernelGGL(bi2_chunk_cursor_kernel, dim3(1), dim3(1), 0, c->stream, bs, keep, true);
            const bool wide = b.sbits == 0 && !slice_env() && npos > kNarrowPassPositions;  // one pass over more records than the 1024-slot tables hold: the 2048-slot form
            if (wide)
                hipLaunchKernelGGL((bi2_count_big_kernel<(int)kBi2Sub, false, false, 2048>), dim3(kBi2Waves * kWave / kBi2BigThreads), dim3(kBi2BigThreads), 0, c->stream, recsB, b.region,
                                   c->b2.boff.p, bs, c->state.p, pl.thr, io.sp_rep, io.sp_cnt, c->b2.wlist.p, c->b2.wcnt.p, b.wcap, want_list, with_codes ? c->b2.wcode.p : (uint32_t*)nullptr,
                                   (const uint32_t*)nullptr, kBi2Waves, b.wextra);
            else
            hipLaunchKernelGGL((bi2_count_big_kernel<(int)kBi2Sub>), dim3(kBi2Waves * kWave / kBi2BigThreads), dim3(kBi2BigThreads), 0, c->stream, recsB, b.region, c->b2.boff.p, bs, c->state.p,
                               pl.thr, io.sp_rep, io.sp_cnt, c->b2.wlist.p, c->b2.wcnt.p, b.wcap, want_list, with_codes ? c->b2.wcode.p : (uint32_t*)nullptr, (const uint32_t*)nullptr,
                               kBi2Waves, b.wextra);
            if (b.sbits) hipLaunchKernelGGL(bi2_chunk_cursor_kernel, dim3(1), dim3(1), 0, c->stream, bs, keep, false);
            if (wide)
                hipLaunchKernelGGL((bi2_count_kernel<(int)kBi2Sub, false, kBi2WRows, false, 2048>), dim3(kBi2Waves), dim3(kWave), 0, c->stream, recsB, b.region, c->b2.boff.p, bs, c->state.p,
                                   pl.thr, io.sp_rep, io.sp_cnt, c->b2.wlist.p, c->b2.wcnt.p, b.wcap, want_list, with_codes ? c->b2.wcode.p : (uint32_t*)nullptr, (const uint32_t*)nullptr, true);
            else
            hipLaunchKernelGGL((bi2_count_kernel<(int)kBi2Sub>), dim3(kBi2Waves), dim3(kWave), 0, c->stream, recsB, b.region, c->b2.boff.p, bs, c->state.p, pl.thr, io.sp_rep, io.sp_cnt,
                               c->b2.wlist.p, c->b2.wcnt.p, b.wcap, want_list, with_codes ? c->b2.wcode.p : (uint32_t*)nullptr, (const uint32_t*)nullptr, true);
        }
        {
            Prof p(c, COLIBRI_K_PRUNE);
            hipLaunchKernelGGL(bi2_kept_finish_kernel, dim3(kBins + 1), dim3(kBi2BBins), 0, c->stream, c->state.p, bs, pl.thr, pl.res_cap, slice == 0 ? c->b2.headsurv.p : (uint32_t*)nullptr, 4u);
            if (chain) {
                int rcf;
                if ((rcf = chain_compact_fork(c, io, bs, pl))) return rcf;
            } else {
                hipLaunchKernelGGL(bi2_compact_kernel, dim3(1025), dim3(kBlock), 0, c->stream, io.sp_rep, io.sp_cnt, c->state.p, bs, c->res_rep.p, c->res_cnt.p, pl.res_cap);
            }
        }
    }
    if (!want_list) return COLIBRI_OK;
    if (!chain) HIP_TRY(c, hipMemsetAsync(c->b2.bitmap.p + npos / 32, 0, sizeof(uint32_t) * 16, c->stream));  // words beyond the corpus read as zero (chain_bitmap_kernel's last block clears them itself)
    {
        Prof p(c, COLIBRI_K_LISTS2);
        hipLaunchKernelGGL(bi2_pospart_kernel, dim3(512), dim3(kBi2Threads), 0, c->stream, c->b2.wlist.p, c->b2.wcnt.p, kBi2Waves + b.wextra, b.wcap, bs, c->state.p, c->b2.plist.p, b.pl,
                           with_codes ? (const uint32_t*)c->b2.wcode.p : (const uint32_t*)nullptr, with_codes ? c->b2.pcode.p : (uint32_t*)nullptr, 0u, /*dense=*/chain || ids_out != nullptr);
        if (chain) {  // who of the head pairs survived; the bitmap of all listed positions (and st->valid). The pairs stay where they are: chain_order(3) walks them
            hipLaunchKernelGGL(bi2_headids_kernel, dim3(1), dim3(kBlock), 0, c->stream, (const Bi2State*)bs, (const DevState*)c->state.p, c->b2.headid.p);
            const bool pairs2 = c->b2.pairs2_direct;  // (an indexed chained run: order 2's forward-index pairs leave with the ids' LDS parts)
            hipLaunchKernelGGL(chain_bitmap_kernel, dim3(b.nbuckets), dim3(kBi2BmThreads), ((size_t)1 << b.pshift) / 8, c->stream, npos, (const Bi2State*)bs, (const uint32_t*)c->b2.plist.p, b.pl,
                               c->state.p, c->b2.bitmap.p, (const uint32_t*)c->b2.pcode.p, (const uint32_t*)c->b2.headid.p, pairs2 ? c->b2.wpre.p : (uint32_t*)nullptr,
                               pairs2 ? c->b2.btot.p : (uint32_t*)nullptr);
            if (ids_out != nullptr || pairs2) {  // (the id-keeping modes on the chained engine: the head windows are in the lists)
                int rci;
                if ((rci = chain_ids(c, b, bs, ids_out, (const uint32_t*)c->b2.headid.p, pairs2))) return rci;
            }
            return COLIBRI_OK;
        }
        if (ids_out != nullptr) {
            // the ids' scatter in the step order of chain_emit_kernel (an XCD's blocks fill ~3 bucket windows at a time: whole lines leave L2), not one block per bucket
            int rci;
            if ((rci = chain_ids(c, b, bs, ids_out, nullptr))) return rci;
            hipLaunchKernelGGL(bi2_headids_kernel, dim3(1), dim3(kBlock), 0, c->stream, bs, c->state.p, c->b2.headid.p);
        }
        hipLaunchKernelGGL(bi2_bitmap_kernel, dim3(b.nbuckets), dim3(kBi2BmThreads), ((size_t)1 << b.pshift) / 8, c->stream, npos, bs, c->b2.plist.p, b.pl, c->state.p, c->b2.bitmap.p);
        hipLaunchKernelGGL(bi2_list3_kernel, dim3(2048), dim3(kBlock), 0, c->stream, c->cls.p, c->uni_surv.p, npos, c->b2.headsurv.p, c->b2.bitmap.p, c->state.p, c->alist[1].p, nlist, ids_out,
                           ids_out != nullptr ? (const uint32_t*)c->b2.headid.p : (const uint32_t*)nullptr);
    }
    return COLIBRI_OK;
}

// Order n >= 3 on the same engine (chain.hpp): the pairs order n - 1 left -> records -> level B -> one wave per final bin -> survivors; the pairs for order n + 1.
// Everything is enqueued; nothing is read back. Bi2State ping-pongs: order n's in state (n even) / state2 (n odd), order n - 1's is read for the list lengths,
// the per-bin dense offsets and the result base (order 2's own state is kept: colibri_order2_records reads it after the run).
int chain_order(colibri_ctx* c, const TrainPlan& pl, int n, bool want_next, uint32_t* ids_out = nullptr /* the id-keeping modes: the result index of the n-gram at every position */) {
    const uint32_t        npos = pl.npos;
    const bool            wide = chain_wide(npos);  // (eight sub-regions, 2048-slot tables, three position bits dropped: see chain_wide)
    const uint32_t        nsub = wide ? kBi2SubWide : kBi2Sub, pdrop = wide ? 3u : 0u;
    const Bigram2Plan     b    = bigram2_plan(c, npos, nsub);
    Bi2State* const       bs   = (n & 1) ? c->b2.state2.p : c->b2.state3.p;
    const Bi2State* const prev = n == 3 ? c->b2.state.p : (n & 1) ? c->b2.state3.p : c->b2.state2.p;
    auto* const           recsA = reinterpret_cast<unsigned long long*>(c->recs[0].p);
    auto* const           recsB = reinterpret_cast<unsigned long long*>(c->recs[1].p);
    const BinnedIO        io    = binned_planes(c, pl, false);
    int rcj;
    if ((rcj = chain_compact_join(c))) return rcj;  // (the emit kernel below overwrites the sparse arrays the order before is still being copied from)
    {
        Prof p(c, COLIBRI_K_EMIT);
        static const uint32_t grid = chain_grid("COLIBRI_CH_GRID", 768u);  // (three resident blocks per CU)
        const uint32_t cap = chain_steps_cap(b.pl);
        hipLaunchKernelGGL(chain_begin_kernel, dim3(kChXcds + kChResetBlocks), dim3(kBi2Threads), 0, c->stream, prev, b.pl, b.nbuckets, reinterpret_cast<uint2*>(c->b2.steps.p), cap,
                           c->b2.steps.p + 2 * (size_t)kChXcds * cap, (const DevState*)c->state.p, bs, c->b2.wcnt.p, kBi2Waves + b.wextra + 1);
        hipLaunchKernelGGL(chain_emit_kernel, dim3(grid), dim3(kChThreads), 0, c->stream, (const uint32_t*)c->cls.p, npos, (uint32_t)n, b.clsbits, b.posbits - pdrop, prev,
                           (const uint32_t*)c->b2.plist.p, (const uint32_t*)c->b2.pcode.p, b.pl, reinterpret_cast<const uint2*>(c->b2.steps.p), cap,
                           (const uint32_t*)(c->b2.steps.p + 2 * (size_t)kChXcds * cap),
                           (const uint32_t*)c->b2.bitmap.p, recsA, b.region, nsub, bs, c->state.p, (const uint32_t*)c->b2.headid.p, chain_dbg(), 0u, pdrop);
        // (records per A bin, scan, B-bin shift: the emit kernel's last block, bi2_offsets_tail)
    }
    {
        Prof p(c, COLIBRI_K_LEVELB2);
        hipLaunchKernelGGL(bi2_levelB_kernel<false>, dim3(b.nslots), dim3(kBi2Threads), 0, c->stream, recsA, recsB, b.region, bs, c->b2.boff.p, c->state.p);
        hipLaunchKernelGGL(bi2_binoff_kernel, dim3(kBins), dim3(kBi2BBins), 0, c->stream, bs, c->b2.boff.p, nsub, c->state.p);
    }
    {
        // the hot bins (a workgroup each: tens of thousands of windows of one frequent n-gram) run beside the wave kernel, on a second stream: they touch other bins, other
        // position lists (the pool behind the waves' own) and share only atomically updated counters. One after the other the few hot bins cost ~0.1 ms at order 3
        Prof p(c, COLIBRI_K_BINCOUNT);
        HIP_TRY(c, hipEventRecord(c->b2.ev_fork, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->b2.aux, c->b2.ev_fork, 0));
        if (wide)
            hipLaunchKernelGGL((bi2_count_big_kernel<(int)kBi2SubWide, false, false, 2048, true>), dim3(kBi2Waves * kWave / kBi2BigThreads), dim3(kBi2BigThreads), 0, c->b2.aux, recsB, b.region,
                               c->b2.boff.p, bs, c->state.p, pl.thr, io.sp_rep, io.sp_cnt, c->b2.wlist.p, c->b2.wcnt.p, b.wcap, want_next, want_next ? c->b2.wcode.p : (uint32_t*)nullptr,
                               (const uint32_t*)nullptr, kBi2Waves, b.wextra);
        else
            hipLaunchKernelGGL((bi2_count_big_kernel<(int)kBi2Sub>), dim3(kBi2Waves * kWave / kBi2BigThreads), dim3(kBi2BigThreads), 0, c->b2.aux, recsB, b.region, c->b2.boff.p, bs, c->state.p, pl.thr,
                               io.sp_rep, io.sp_cnt, c->b2.wlist.p, c->b2.wcnt.p, b.wcap, want_next, want_next ? c->b2.wcode.p : (uint32_t*)nullptr, (const uint32_t*)nullptr, kBi2Waves, b.wextra);
        HIP_TRY(c, hipEventRecord(c->b2.ev_join, c->b2.aux));
        if (wide)
            hipLaunchKernelGGL((bi2_count_kernel<(int)kBi2SubWide, false, kBi2WRows, false, 2048, true>), dim3(kBi2Waves), dim3(kWave), 0, c->stream, recsB, b.region, c->b2.boff.p, bs, c->state.p,
                               pl.thr, io.sp_rep, io.sp_cnt, c->b2.wlist.p, c->b2.wcnt.p, b.wcap, want_next, want_next ? c->b2.wcode.p : (uint32_t*)nullptr, (const uint32_t*)nullptr, true);
        else
            hipLaunchKernelGGL((bi2_count_kernel<(int)kBi2Sub>), dim3(kBi2Waves), dim3(kWave), 0, c->stream, recsB, b.region, c->b2.boff.p, bs, c->state.p, pl.thr, io.sp_rep, io.sp_cnt,
                               c->b2.wlist.p, c->b2.wcnt.p, b.wcap, want_next, want_next ? c->b2.wcode.p : (uint32_t*)nullptr, (const uint32_t*)nullptr, true);
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->b2.ev_join, 0));
    }
    {
        Prof p(c, COLIBRI_K_PRUNE);
        hipLaunchKernelGGL(bi2_kept_finish_kernel, dim3(kBins + 1), dim3(kBi2BBins), 0, c->stream, c->state.p, bs, pl.thr, pl.res_cap, (uint32_t*)nullptr, 16u);
        int rcf;
        if ((rcf = chain_compact_fork(c, io, bs, pl))) return rcf;
    }
    if (!want_next) return COLIBRI_OK;
    {
        Prof p(c, COLIBRI_K_LISTS2);
        hipLaunchKernelGGL(bi2_pospart_kernel, dim3(512), dim3(kBi2Threads), 0, c->stream, c->b2.wlist.p, c->b2.wcnt.p, kBi2Waves + b.wextra, b.wcap, bs, c->state.p, c->b2.plist.p, b.pl,
                           (const uint32_t*)c->b2.wcode.p, c->b2.pcode.p, 0u, /*dense=*/true);
        hipLaunchKernelGGL(chain_bitmap_kernel, dim3(b.nbuckets), dim3(kBi2BmThreads), ((size_t)1 << b.pshift) / 8, c->stream, npos, (const Bi2State*)bs, (const uint32_t*)c->b2.plist.p, b.pl,
                           c->state.p, c->b2.bitmap.p, (const uint32_t*)nullptr, (const uint32_t*)nullptr, c->b2.pairs_direct ? c->b2.wpre.p : (uint32_t*)nullptr,
                           c->b2.pairs_direct ? c->b2.btot.p : (uint32_t*)nullptr);
    }
    if (c->b2.pairs_direct) chain_pairs(c, b, bs, nullptr);
    if (ids_out != nullptr) {
        Prof p(c, COLIBRI_K_RESOLVE);
        int rci;
        if ((rci = chain_ids(c, b, bs, ids_out, nullptr))) return rci;
    }
    return COLIBRI_OK;
}

// One two-part skipgram pass of order n on the same engine (skip_emit_kernel), enqueued: reset -> records -> level B -> count -> survivors -> the run's log
// (skip_pass_begin / _end_kernel). Bi2State: the one chain_order(n + 1) will reset anyway (order n - 1's; order 2's own state stays). No lists, no ids: the passes of
// an unindexed model leave patterns and counts only.
int skip_pass_chain(colibri_ctx* c, const TrainPlan& pl, int n, uint32_t mask, const uint32_t* left, uint32_t offl, bool l_is_cls, const uint32_t* right, uint32_t offr, bool r_is_cls,
                    uint32_t thr, uint32_t* seglog) {
    const Bigram2Plan b    = bigram2_plan(c, pl.npos);
    Bi2State* const   bs   = (n & 1) ? c->b2.state3.p : c->b2.state2.p;
    auto* const       recsA = reinterpret_cast<unsigned long long*>(c->recs[0].p);
    auto* const       recsB = reinterpret_cast<unsigned long long*>(c->recs[1].p);
    const BinnedIO    io    = binned_planes(c, pl, false);
    Prof              p(c, COLIBRI_K_SKIPGRAM);
    hipLaunchKernelGGL(skip_pass_begin_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p);
    hipLaunchKernelGGL(chain_reset_kernel, dim3(256), dim3(kBlock), 0, c->stream, bs, c->b2.wcnt.p, 0u);
    if (l_is_cls && r_is_cls) {  // both parts one token: the dense head of order 2 (frames of two frequent words)
        hipLaunchKernelGGL(skip_emit_kernel<true>, dim3(kBi2EmitGrid), dim3(kChThreads), 0, c->stream, (const uint32_t*)c->skl, (const uint32_t*)c->skl_n, left, offl, right, offr, 1u, 1u,
                           b.clsbits, b.posbits, recsA, b.region, kBi2Sub, bs, c->state.p, c->b2.head_rows.p);
        hipLaunchKernelGGL(bi2_head_reduce_kernel, dim3(kBi2HeadN / kBlock, kBi2HeadSplit), dim3(kBlock), 0, c->stream, c->b2.head_rows.p, kBi2EmitGrid, bs, c->state.p);
    } else {
        static const uint32_t grid = chain_grid("COLIBRI_CH_GRID", 768u);
        hipLaunchKernelGGL(skip_emit_kernel<false>, dim3(grid), dim3(kChThreads), 0, c->stream, (const uint32_t*)c->skl, (const uint32_t*)c->skl_n, left, offl, right, offr,
                           l_is_cls ? 1u : 0u, r_is_cls ? 1u : 0u, b.clsbits, b.posbits, recsA, b.region, kBi2Sub, bs, c->state.p, (uint32_t*)nullptr);
    }
    // (records per A bin, scan, B-bin shift: the emit kernel's last block, bi2_offsets_tail)
    hipLaunchKernelGGL(bi2_levelB_kernel<false>, dim3(b.nslots), dim3(kBi2Threads), 0, c->stream, recsA, recsB, b.region, bs, c->b2.boff.p, c->state.p);
    hipLaunchKernelGGL(bi2_binoff_kernel, dim3(kBins), dim3(kBi2BBins), 0, c->stream, bs, c->b2.boff.p, kBi2Sub, c->state.p);
    // (the hot bins' workgroups beside the wave kernel on the second stream, as in chain_order: 0.05 ms per pass of n = 4)
    HIP_TRY(c, hipEventRecord(c->b2.ev_fork, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->b2.aux, c->b2.ev_fork, 0));
    hipLaunchKernelGGL((bi2_count_big_kernel<(int)kBi2Sub>), dim3(kBi2Waves * kWave / kBi2BigThreads), dim3(kBi2BigThreads), 0, c->b2.aux, recsB, b.region, c->b2.boff.p, bs, c->state.p, thr,
                       io.sp_rep, io.sp_cnt, c->b2.wlist.p, c->b2.wcnt.p, b.wcap, false, (uint32_t*)nullptr, (const uint32_t*)nullptr, kBi2Waves, b.wextra);
    HIP_TRY(c, hipEventRecord(c->b2.ev_join, c->b2.aux));
    hipLaunchKernelGGL((bi2_count_kernel<(int)kBi2Sub>), dim3(kBi2Waves), dim3(kWave), 0, c->stream, recsB, b.region, c->b2.boff.p, bs, c->state.p, thr, io.sp_rep, io.sp_cnt, c->b2.wlist.p,
                       c->b2.wcnt.p, b.wcap, false, (uint32_t*)nullptr, (const uint32_t*)nullptr, true);
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->b2.ev_join, 0));
    hipLaunchKernelGGL(bi2_kept_finish_kernel, dim3(kBins + 1), dim3(kBi2BBins), 0, c->stream, c->state.p, bs, thr, pl.res_cap, (uint32_t*)nullptr, 16u);
    hipLaunchKernelGGL(bi2_compact_kernel, dim3(1025), dim3(kBlock), 0, c->stream, (const uint32_t*)io.sp_rep, (const uint32_t*)io.sp_cnt, (const DevState*)c->state.p, (const Bi2State*)bs,
                       c->res_rep.p, c->res_cnt.p, pl.res_cap, false);
    hipLaunchKernelGGL(skip_pass_end_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p, seglog, (uint32_t)n, mask);
    return COLIBRI_OK;
}

// Order 2 of a corpus beyond one pass (more than ~110 M records: 10^9 tokens on one device), round 3: the windows are scanned ONCE and their records cut ONCE into 2^s key
// slices (ks_split_*: the split of the multi-GPU protocol with the slices as "owners" and the sub-regions of the one source as "source ranks"); every slice then
// runs level B and the count on its own dense segment. Rounds 1-2 re-scanned the corpus per slice (bi2_emit_sliced_kernel: 1.4 ms per pass on 10^9 positions,
// 8 passes) and counted bins of ~810 records on the count kernel's streaming path. Plain mode only (no ids).
constexpr uint32_t kSplitMaxBits = 3;
bool bigram2_split_fits(const colibri_ctx* c, uint32_t npos) {
    const Bigram2Plan b = bigram2_plan(c, npos);
    return b.sbits >= 1 && b.sbits <= kSplitMaxBits && std::max(2u * b.clsbits, 17u + b.sbits) - 8 + b.posbits <= 64 && !getenv("COLIBRI_RESCAN_SLICES");
}
int bigram2_order_split(colibri_ctx* c, const TrainPlan& pl, bool want_list) {
    const uint32_t    npos = pl.npos, nsurv = c->maxclass / 32 + 1;
    const Bigram2Plan b    = bigram2_plan(c, npos, kBi2SubWide);
    const uint32_t    s    = b.sbits, V = 1u << s;
    auto&             ks   = c->ks;
    int               rc;
    if ((rc = dev_alloc(c, ks.split, 1)) || (rc = dev_alloc(c, ks.obs, 1)) || (rc = dev_alloc(c, ks.slotbase, kKsSlots)) || (rc = dev_alloc(c, ks.oboff, (size_t)kKsSlots * (kBi2BBins + 1))) ||
        (rc = dev_alloc(c, ks.lcnt, 64)))
        return rc;
    const uint32_t K       = std::max(2u * b.clsbits, 17u + s);
    const uint32_t region  = (uint32_t)(2ull * c->recs[0].n / b.nslots);  // every record of the order at once: the emit kernel's regions are not cut down to a slice
    // the count kernel's position lists: a pool of chunks with room for every position plus one partly filled chunk per wave (of either count kernel) and slice
    if ((rc = dev_alloc(c, c->b2.wlist, (size_t)npos + ((size_t)V * 2 * kBi2Waves + 64) * kBi2Chunk))) return rc;
    const uint32_t nchunks = (uint32_t)(c->b2.wlist.n / kBi2Chunk);
    if ((rc = dev_alloc(c, c->b2.wcnt, std::max<size_t>(kBi2Waves, nchunks)))) return rc;
    Bi2State* const sbs   = c->b2.state.p;
    Bi2State* const obs   = ks.obs.p;
    auto* const     recsA = reinterpret_cast<unsigned long long*>(c->recs[0].p);
    // the slices' segments and the level-B output of one slice. Exact split (histogram, scan, move): dense segments in the first half of recs[1], level B into the second.
    // Direct split (the default): every (slot, slice) run has room for `cap` records anywhere in recs[1], level B writes into recs[0] behind the two sparse planes
    const bool      direct = !c->split_exact && !getenv("COLIBRI_SPLIT_EXACT");
    auto* const     seg    = reinterpret_cast<unsigned long long*>(c->recs[1].p);
    // (level B keeps the slot layout of its input: its output needs the extent of the segments, not the size of a slice)
    const uint64_t  extent = std::min<uint64_t>(2ull * c->recs[1].n, 2ull * c->recs[0].n - npos - 2);
    const uint32_t  cap    = (uint32_t)std::min<uint64_t>(extent / ((uint64_t)kKsSlots * kKsWorld), 0xFFFFFFFFull / ((uint64_t)kKsSlots * kKsWorld));
    auto* const     segB   = direct ? reinterpret_cast<unsigned long long*>(c->recs[0].p) + ((size_t)npos + 2) : seg + c->recs[1].n;
    const uint32_t  roomB  = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, direct ? extent : (uint64_t)c->recs[1].n);
    uint32_t* const nlist = c->alist_n.p + 1;
    uint32_t* const keep  = ks.lcnt.p;  // the position-list pool's cursor between slices
    HIP_TRY(c, hipMemsetAsync(nlist, 0, sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->b2.wcnt.p, 0, sizeof(uint32_t) * nchunks, c->stream));
    HIP_TRY(c, hipMemsetAsync(keep, 0, sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(sbs, 0, sizeof(Bi2State), c->stream));
    HIP_TRY(c, hipMemsetAsync(ks.split.p, 0, sizeof(KsSplitState), c->stream));
    {
        Prof p(c, COLIBRI_K_EMIT2);
        hipLaunchKernelGGL(bi2_emit_kernel, dim3(kBi2EmitGrid), dim3(kBi2Threads), 0, c->stream, c->cls.p, c->uni_surv.p, nsurv, npos, b.clsbits, 0u, 0u, b.posbits, recsA, region, kBi2SubWide, sbs,
                           c->state.p, c->b2.head_rows.p, (uint8_t*)nullptr, K);
        hipLaunchKernelGGL(bi2_head_reduce_kernel, dim3(kBi2HeadN / kBlock, kBi2HeadSplit), dim3(kBlock), 0, c->stream, c->b2.head_rows.p, kBi2EmitGrid, sbs, c->state.p);
    }
    {
        Prof           p(c, COLIBRI_K_SCATTER);
        const KsSplit8 sp{s, b.posbits + K - 8 - s, b.posbits, 0u};
        if (direct) {
            hipLaunchKernelGGL((ks_split_direct_kernel<unsigned long long, KsSplit8, 4>), dim3(kKsSlots), dim3(kKsThreads), 0, c->stream, (const unsigned long long*)recsA, region,
                               (const uint32_t*)sbs->curA, sp, cap, ks.split.p, seg);
        } else {
            hipLaunchKernelGGL((ks_split_hist_kernel<unsigned long long, KsSplit8>), dim3(kKsSlots), dim3(kKsThreads), 0, c->stream, (const unsigned long long*)recsA, region,
                               (const uint32_t*)sbs->curA, sp, ks.split.p);
            hipLaunchKernelGGL(ks_split_scan_kernel, dim3(1), dim3(kKsThreads), 0, c->stream, ks.split.p, s, V, (const DevState*)c->state.p);
            hipLaunchKernelGGL((ks_split_move_kernel<unsigned long long, KsSplit8, 4>), dim3(kKsSlots), dim3(kKsThreads), 0, c->stream, (const unsigned long long*)recsA, region,
                               (const uint32_t*)sbs->curA, sp, (const KsSplitState*)ks.split.p, seg);
            hipLaunchKernelGGL(ks_split_flag_kernel, dim3(1), dim3(1), 0, c->stream, (const KsSplitState*)ks.split.p, (const BinState*)nullptr, c->state.p);
        }
    }
    const BinnedIO io = binned_planes(c, pl, false);  // the sparse survivor arrays live in recs[0]: the emit kernel's regions are free once the split has moved them
    for (uint32_t v = 0; v < V; ++v) {
        HIP_TRY(c, hipMemsetAsync(obs, 0, sizeof(Bi2State), c->stream));
        if (v == 0)  // the dense head was counted by the scan: it belongs to the first slice's figures
            HIP_TRY(c, hipMemcpyAsync(obs->headcnt, sbs->headcnt, 2 * sizeof(uint32_t) * kBi2HeadN, hipMemcpyDeviceToDevice, c->stream));
        {
            Prof p(c, COLIBRI_K_LEVELB2);
            hipLaunchKernelGGL(ks_local_init2_kernel, dim3(1), dim3(kKsThreads), 0, c->stream, obs, ks.slotbase.p, (const KsSplitState*)ks.split.p, v, s, K - s, b.posbits + s, (const uint32_t*)keep,
                               roomB, c->state.p);
            hipLaunchKernelGGL(bi2_offsets_kernel, dim3(1), dim3(kBlock), 0, c->stream, obs, 0xFFFFFFFFu, kBi2SubWide, (const DevState*)c->state.p);
            hipLaunchKernelGGL(bi2_levelB_kernel<false>, dim3(kKsSlots), dim3(kBi2Threads), 0, c->stream, (const unsigned long long*)seg, segB, 0xFFFFFFFFu, (const Bi2State*)obs, ks.oboff.p,
                               (const DevState*)c->state.p, (const uint32_t*)ks.slotbase.p);
            hipLaunchKernelGGL(bi2_binoff_kernel, dim3(kBins), dim3(kBi2BBins), 0, c->stream, obs, (const uint32_t*)ks.oboff.p, kBi2SubWide, (const DevState*)c->state.p);
        }
        {
            Prof p(c, COLIBRI_K_COUNT2);
            hipLaunchKernelGGL((bi2_count_big_kernel<(int)kBi2SubWide, true>), dim3(kBi2Waves * kWave / kBi2BigThreads), dim3(kBi2BigThreads), 0, c->stream, (const unsigned long long*)segB, 0u,
                               (const uint32_t*)ks.oboff.p, obs, c->state.p, pl.thr, io.sp_rep, io.sp_cnt, c->b2.wlist.p, c->b2.wcnt.p, nchunks, want_list, (uint32_t*)nullptr,
                               (const uint32_t*)ks.slotbase.p);
            hipLaunchKernelGGL((bi2_count_kernel<(int)kBi2SubWide, true, 16>), dim3(kBi2Waves), dim3(kWave), 0, c->stream, (const unsigned long long*)segB, 0u, (const uint32_t*)ks.oboff.p, obs,
                               c->state.p, pl.thr, io.sp_rep, io.sp_cnt, c->b2.wlist.p, c->b2.wcnt.p, nchunks, want_list, (uint32_t*)nullptr, (const uint32_t*)ks.slotbase.p, true);
            hipLaunchKernelGGL(ks_keep_chunk_kernel, dim3(1), dim3(1), 0, c->stream, (const Bi2State*)obs, keep);
        }
        {
            Prof p(c, COLIBRI_K_PRUNE);
            hipLaunchKernelGGL(bi2_kept_scan_kernel, dim3(kBins), dim3(kBi2BBins), 0, c->stream, obs, c->state.p);
            hipLaunchKernelGGL(bi2_finish_kernel, dim3(1), dim3(kBlock), 0, c->stream, c->state.p, obs, pl.thr, pl.res_cap, v == 0 ? c->b2.headsurv.p : (uint32_t*)nullptr);
            hipLaunchKernelGGL(bi2_compact_kernel, dim3(1025), dim3(kBlock), 0, c->stream, io.sp_rep, io.sp_cnt, c->state.p, obs, c->res_rep.p, c->res_cnt.p, pl.res_cap);
        }
    }
    if (!want_list) return COLIBRI_OK;
    HIP_TRY(c, hipMemsetAsync(c->b2.bitmap.p + npos / 32, 0, sizeof(uint32_t) * 16, c->stream));
    {
        Prof p(c, COLIBRI_K_LISTS2);
        hipLaunchKernelGGL(bi2_pospart_kernel, dim3(512), dim3(kBi2Threads), 0, c->stream, c->b2.wlist.p, c->b2.wcnt.p, nchunks, kBi2Chunk, sbs, c->state.p, c->b2.plist.p, b.pl,
                           (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u);
        hipLaunchKernelGGL(bi2_bitmap_kernel, dim3(b.nbuckets), dim3(kBi2BmThreads), ((size_t)1 << b.pshift) / 8, c->stream, npos, sbs, c->b2.plist.p, b.pl, c->state.p, c->b2.bitmap.p);
        hipLaunchKernelGGL(bi2_list3_kernel, dim3(2048), dim3(kBlock), 0, c->stream, c->cls.p, c->uni_surv.p, npos, c->b2.headsurv.p, c->b2.bitmap.p, c->state.p, c->alist[1].p, nlist,
                           (uint32_t*)nullptr, (const uint32_t*)nullptr);
    }
    return COLIBRI_OK;
}

// sbits: the order is counted in 2^sbits passes over disjoint slices of its keys (corpora beyond ~128 M tokens per device: a final bin must fit its LDS table);
// the passes append their survivors, the ids are resolved once at the end.
template <class KeyFn>
int binned_order(colibri_ctx* c, const TrainPlan& pl, const KeyFn& fn, uint32_t* ids_out, int n, bool use_list, bool need_ids, bool flag_mode = false, bool prefill_ids = false,
                 uint32_t sbits = 0) {
    int rc;
    const BinnedIO io = binned_planes(c, pl, false);
    for (uint32_t slice = 0; slice < (1u << sbits); ++slice) {
        if ((rc = binned_count_stage(c, pl, fn, n, use_list, pl.thr, false, need_ids, flag_mode, false, sbits, slice))) return rc;
        Prof p(c, COLIBRI_K_PRUNE);
        hipLaunchKernelGGL(bin_kept_scan_kernel, dim3(kBins), dim3(kBlock), 0, c->stream, c->state.p, c->binstate.p, pl.res_cap, sbits != 0);
        hipLaunchKernelGGL(compact_bins_kernel, dim3(1024), dim3(kBlock), 0, c->stream, io.sp_rep, io.sp_cnt, c->state.p, c->binstate.p, c->res_rep.p, c->res_cnt.p, pl.res_cap,
                           use_list ? (const uint32_t*)c->alist[n & 1].p : (const uint32_t*)nullptr);
        hipLaunchKernelGGL(bin_advance_prepare_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p, c->binstate.p, sbits != 0);
    }
    if (!need_ids) return COLIBRI_OK;
    if (flag_mode) {  // all-positions order whose successor builds its keys from class ids: a byte per position and the active list
        uint32_t* nlist_out = c->alist_n.p + ((n + 1) & 1);
        HIP_TRY(c, hipMemsetAsync(nlist_out, 0, sizeof(uint32_t), c->stream));
        Prof p(c, COLIBRI_K_RESOLVE);
        hipLaunchKernelGGL(bin_resolve_flags_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->rep_of.p, c->flags_at.p, c->flag2.p, c->state.p, pl.npos, c->alist[(n + 1) & 1].p,
                           nlist_out);
        return COLIBRI_OK;
    }
    // the only later reader is the next order's emit kernel (ids at i and i+1 for i on the new active list): no fill of ids_out when this order's list
    // covers every position whose (n-1)-gram survived. (prefill_ids: the list holds only the admissible windows — a survivor at i then says nothing
    // about i + 1 being listed, so the unlisted positions must read as "no survivor")
    return binned_resolve_stage(c, pl, ids_out, n, use_list, /*build_list=*/n >= 2, nullptr, 0u, prefill_ids);
}

// An order >= 3 of a corpus beyond one pass, round 3: the listed windows are turned into records ONCE and the records cut ONCE into the 2^s key slices (ks_split_*,
// as bigram2_order_split does for order 2); every slice then runs level B and the count on its own dense segment of recs[1]. Rounds 1-2 walked the list and
// hashed every window once per slice (3.8 ms per walk of the 520 M order-3 windows of a 10^9-token corpus, four walks).
template <class KeyFn>
int binned_order_split(colibri_ctx* c, const TrainPlan& pl, const KeyFn& fn, uint32_t* ids_out, int n, bool need_ids, bool prefill_ids, uint32_t s) {
    auto& ks = c->ks;
    int   rc;
    if ((rc = dev_alloc(c, ks.split, 1))) return rc;
    const uint32_t  V      = 1u << s;
    const uint32_t  tiles  = blocks_for(pl.npos, kScatTile) + 1 + kASlots;
    const uint32_t  region = (uint32_t)(c->recs[0].n / kASlots);
    const uint32_t* list_in  = c->alist[n & 1].p;
    const uint32_t* nlist_in = c->alist_n.p + (n & 1);
    uint32_t* const ids_at   = need_ids ? c->ids_at.p : nullptr;
    const BinnedIO  io       = binned_planes(c, pl, false);
    Rec* const      seg      = c->recs[1].p;                      // the slices' segments (direct split, the default: room for `cap` records per (slot, slice) run; exact: dense)
    Rec* const      R2       = c->recs[0].p + (pl.npos / 2 + 1);  // level-B output of one slice: recs[0] behind the two sparse planes (free once the split has moved the records)
    const bool      direct   = !c->split_exact && !getenv("COLIBRI_SPLIT_EXACT");
    const uint32_t  cap      = (uint32_t)std::min<uint64_t>(c->recs[1].n / ((uint64_t)kKsSlots * kKsWorld), 0xFFFFFFFFull / ((uint64_t)kKsSlots * kKsWorld));
    const uint32_t  room2    = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, c->recs[0].n - (pl.npos / 2 + 1));
    HIP_TRY(c, hipMemsetAsync(c->binstate.p, 0, sizeof(BinState), c->stream));
    HIP_TRY(c, hipMemsetAsync(ks.split.p, 0, sizeof(KsSplitState), c->stream));
    {
        Prof p(c, COLIBRI_K_EMIT);
        hipLaunchKernelGGL((bin_emit_kernel<KeyFn, true>), dim3(pl.cnt_grid), dim3(kBlock), 0, c->stream, fn, c->recs[0].p, region, c->rep_of.p, c->state.p, c->binstate.p, pl.npos, list_in, nlist_in,
                           ids_at, (uint8_t*)nullptr, 0u, 0u);
    }
    {
        Prof            p(c, COLIBRI_K_SCATTER);
        const KsSplit16 sp{s, 0u};
        const uint4*    recs4 = reinterpret_cast<const uint4*>(c->recs[0].p);
        if (direct) {
            hipLaunchKernelGGL((ks_split_direct_kernel<uint4, KsSplit16, 2>), dim3(kKsSlots), dim3(kKsThreads), 0, c->stream, recs4, region, (const uint32_t*)c->binstate.p->curA, sp, cap, ks.split.p,
                               reinterpret_cast<uint4*>(seg));
        } else {
            hipLaunchKernelGGL((ks_split_hist_kernel<uint4, KsSplit16>), dim3(kKsSlots), dim3(kKsThreads), 0, c->stream, recs4, region, (const uint32_t*)c->binstate.p->curA, sp, ks.split.p);
            hipLaunchKernelGGL(ks_split_scan_kernel, dim3(1), dim3(kKsThreads), 0, c->stream, ks.split.p, s, V, (const DevState*)c->state.p);
            hipLaunchKernelGGL((ks_split_move_kernel<uint4, KsSplit16, 2>), dim3(kKsSlots), dim3(kKsThreads), 0, c->stream, recs4, region, (const uint32_t*)c->binstate.p->curA, sp,
                               (const KsSplitState*)ks.split.p, reinterpret_cast<uint4*>(seg));
            hipLaunchKernelGGL(ks_split_flag_kernel, dim3(1), dim3(1), 0, c->stream, (const KsSplitState*)ks.split.p, (const BinState*)c->binstate.p, c->state.p);
        }
    }
    for (uint32_t v = 0; v < V; ++v) {
        HIP_TRY(c, hipMemsetAsync(c->binstate.p, 0, sizeof(BinState), c->stream));
        {
            Prof p(c, COLIBRI_K_SCATTER);
            hipLaunchKernelGGL(ks_local_init_kernel, dim3(1), dim3(kBlock), 0, c->stream, c->binstate.p, (const KsSplitState*)ks.split.p, v, s, room2, c->state.p);
            hipLaunchKernelGGL(bin_hist2_kernel, dim3(tiles + kBins), dim3(kBlock), 0, c->stream, (const Rec*)seg, (const DevState*)c->state.p, c->binstate.p);
            hipLaunchKernelGGL(bin_scan2_kernel, dim3(kBins), dim3(kBlock), 0, c->stream, c->binstate.p);
            hipLaunchKernelGGL(bin_scatter_kernel, dim3(tiles + kBins), dim3(kBlock), 0, c->stream, (const Rec*)seg, R2, (const DevState*)c->state.p, c->binstate.p);
        }
        {
            Prof p(c, COLIBRI_K_BINCOUNT);
            hipLaunchKernelGGL(bin_count_kernel, dim3(256 * 12), dim3(kBlock), 0, c->stream, (const Rec*)R2, c->state.p, c->binstate.p, pl.thr, io.sp_rep, io.sp_cnt, io.sp_key, ids_at,
                               (uint8_t*)nullptr, false);
        }
        Prof p(c, COLIBRI_K_PRUNE);
        hipLaunchKernelGGL(bin_kept_scan_kernel, dim3(kBins), dim3(kBlock), 0, c->stream, c->state.p, c->binstate.p, pl.res_cap, true);
        hipLaunchKernelGGL(compact_bins_kernel, dim3(1024), dim3(kBlock), 0, c->stream, io.sp_rep, io.sp_cnt, c->state.p, c->binstate.p, c->res_rep.p, c->res_cnt.p, pl.res_cap,
                           (const uint32_t*)c->alist[n & 1].p);
        hipLaunchKernelGGL(bin_advance_prepare_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p, c->binstate.p, true);
    }
    if (!need_ids) return COLIBRI_OK;
    return binned_resolve_stage(c, pl, ids_out, n, true, /*build_list=*/true, nullptr, 0u, prefill_ids);
}
inline bool binned_split_fits(uint32_t sbits) { return sbits >= 1 && sbits <= kSplitMaxBits && !getenv("COLIBRI_RESCAN_SLICES"); }

// One (order, gap mask) skipgram pass. Exact identity of a skipgram = the survivor ids of its contiguous parts, paired
// left to right: level 1 interns (part1, part2) into slot numbers, level j pairs those with part j+1; the last level counts.
// gate/gate2 select the windows that take part (exhaustive: both (n-1)-grams survived; indexed: the n-gram survived).
// the positions whose `gate` entry is valid -> c->sklist / c->sklist_n: the skipgram passes of an order walk this list (at orders 4 and 5 a
// few percent of the positions) instead of the whole corpus
int build_skip_list(colibri_ctx* c, const TrainPlan& pl, const uint32_t* gate) {
    const uint32_t ntiles = std::max<uint32_t>(1, blocks_for(pl.npos, kPairTile));
    int            rc;
    if ((rc = dev_alloc(c, c->sklist, (size_t)pl.npos + 1)) || (rc = dev_alloc(c, c->sklist_n, 1)) || (rc = dev_alloc(c, c->idx_cnt, (size_t)ntiles + 2))) return rc;
    Prof p(c, COLIBRI_K_SKIPGRAM);
    // in position order (count per tile, short scan, write): the references of an indexed model's skipgram passes are emitted entry by entry of this list
    hipLaunchKernelGGL(emit_count_kernel, dim3(ntiles), dim3(kPairThreads), 0, c->stream, gate, pl.npos, c->idx_cnt.p, (const DevState*)nullptr);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(kPairThreads), 0, c->stream, c->idx_cnt.p, ntiles, c->idx_cnt.p + ntiles);
    hipLaunchKernelGGL(list_write_kernel, dim3(ntiles), dim3(kPairThreads), 0, c->stream, gate, pl.npos, (const uint32_t*)c->idx_cnt.p, c->sklist.p, c->sklist_n.p);
    c->skl   = c->sklist.p;
    c->skl_n = c->sklist_n.p;
    return COLIBRI_OK;
}

// ids[n] of the running train, or nullptr (and the context's message) when this run did not build them: the id-keeping loop leaves out what its readers, listed at
// want_ids, do not ask for, and the buffers keep an earlier run's ids — a reader that list does not know fails here instead of counting garbage
inline const uint32_t* built_ids(colibri_ctx* c, int n) {
    if (n >= 0 && (size_t)n < c->ids_built.size() && c->ids_built[(size_t)n]) return c->ids[(size_t)n].p;
    (void)fail(c, COLIBRI_ERR_STATE, "internal: the survivor ids of order %d were not built in this run (colibri_train: want_ids)", n);
    return nullptr;
}
// identity of a skipgram part of `len` tokens at a position: the survivor id of that n-gram — or, for one-token parts of a run that keeps no order-1 ids
// (c->ids1_is_cls: unindexed, class-keyed order 2), the token's class id, which names a surviving word just as well
inline const uint32_t* part_ids(const colibri_ctx* c, int len) { return (len == 1 && c->ids1_is_cls) ? (const uint32_t*)c->cls.p : (const uint32_t*)c->ids[(size_t)len].p; }

int skipgram_pass(colibri_ctx* c, const TrainPlan& pl, int n, uint32_t mask, const uint32_t* gate, const uint32_t* gate2, uint32_t participants, uint32_t thr, bool count_sources,
                  uint32_t minsrc, uint32_t* found_out, uint32_t* kept_out, int* final_scratch) {
    const std::vector<std::pair<int, int>> parts = mask_parts(mask, n);
    const uint32_t cap = (uint32_t)std::min<uint64_t>(pl.table_slots, (uint64_t)participants + (participants >> 1) + 1024u);
    const uint32_t* left = part_ids(c, parts[0].second);
    uint32_t        offl = (uint32_t)parts[0].first;
    uint32_t*       out  = nullptr;
    int             rc;
    for (size_t j = 1; j < parts.size(); ++j) {
        const bool last = j + 1 == parts.size();
        c->hstate.cap   = cap;
        c->hstate.found = c->hstate.kept = 0;
        if ((rc = write_state(c))) return rc;
        launch_clear(c, pl);
        out = c->scratch[j & 1].p;
        HIP_TRY(c, hipMemsetAsync(out, 0xFF, sizeof(uint32_t) * (size_t)pl.npos, c->stream));  // only the listed positions are written
        KeyPair fn{gate, gate2, left, offl, part_ids(c, parts[j].second), (uint32_t)parts[j].first};
        launch_count(c, pl, fn, out, last ? 2 : 0, COLIBRI_K_SKIPGRAM, c->skl, c->skl_n);
        left = out;
        offl = 0;
    }
    const uint32_t* nsrc = nullptr;
    if (count_sources) {
        HIP_TRY(c, hipMemsetAsync(c->nsrc.p, 0, sizeof(uint32_t) * cap, c->stream));
        Prof p(c, COLIBRI_K_SKIPGRAM);
        hipLaunchKernelGGL(skip_sources_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, gate, c->res_rep.p, out, c->nsrc.p, c->state.p, pl.npos);
        nsrc = c->nsrc.p;
    }
    launch_prune(c, pl, thr, nsrc, minsrc);
    if ((rc = read_state(c))) return rc;
    *found_out = c->hstate.found;
    *kept_out  = c->hstate.kept;
    if (final_scratch) *final_scratch = (int)((parts.size() - 1) & 1);
    return COLIBRI_OK;
}

int emit_pairs(colibri_ctx* c, const TrainPlan& pl, const uint32_t* ids, bool ensure, const uint32_t* surv = nullptr, const uint32_t* resid = nullptr, bool hot1 = false);
int emit_pairs_list(colibri_ctx* c, uint32_t bound, const uint32_t* ids);
// MINSKIPTYPES of an indexed model: a skipgram needs that many distinct fillers = distinct surviving n-grams (results [src_first, src_first + src_count)) whose
// representative window it masks. `ids`: the RESULT index of every window's skipgram (the k1 survivors of the count threshold sit at res_total..); the ones
// with too few fillers leave the results again and the per-position indices follow (over `list`, or over every position when there is none).
int scan_u32(colibri_ctx* c, const uint32_t* in, uint32_t n, unsigned long long* out, unsigned long long* total);
int filter_by_fillers(colibri_ctx* c, const TrainPlan& pl, uint32_t* ids, uint32_t res_total, uint32_t k1, uint32_t minsrc, uint32_t src_first, uint32_t src_count, const uint32_t* list,
                      const uint32_t* nlist, uint32_t* kept_out) {
    int rc;
    if ((rc = dev_alloc(c, c->nsrc, (size_t)k1 + 1)) || (rc = dev_alloc(c, c->skip_off, (size_t)k1 + 1)) || (rc = dev_alloc(c, c->skip_tmp, 2 * (size_t)k1 + 2))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->nsrc.p, 0, sizeof(uint32_t) * k1, c->stream));
    unsigned long long k2 = 0;
    {
        Prof p(c, COLIBRI_K_SKIPGRAM);
        if (src_count)
            hipLaunchKernelGGL(skip_sources_results_kernel, dim3(stream_grid(src_count)), dim3(kBlock), 0, c->stream, c->res_rep.p, src_first, src_count, ids, res_total, c->nsrc.p);
        hipLaunchKernelGGL(skip_keep_flags_kernel, dim3(stream_grid(k1)), dim3(kBlock), 0, c->stream, c->nsrc.p, k1, minsrc);
    }
    if ((rc = scan_u32(c, c->nsrc.p, k1, c->skip_off.p, &k2))) return rc;
    if (k2 != k1) {
        Prof p(c, COLIBRI_K_SKIPGRAM);
        hipLaunchKernelGGL(skip_filter_gather_kernel, dim3(stream_grid(k1)), dim3(kBlock), 0, c->stream, c->nsrc.p, c->skip_off.p, k1, c->res_rep.p, c->res_cnt.p, res_total, c->skip_tmp.p);
        if (k2) hipLaunchKernelGGL(skip_filter_store_kernel, dim3(stream_grid(k2)), dim3(kBlock), 0, c->stream, c->skip_tmp.p, k1, (uint32_t)k2, c->res_rep.p, c->res_cnt.p, res_total);
        hipLaunchKernelGGL(skip_remap_ids_kernel, dim3(stream_grid(pl.npos / 4 + 1)), dim3(kBlock), 0, c->stream, list, nlist, pl.npos, ids, c->nsrc.p, c->skip_off.p, res_total);
    }
    *kept_out      = (uint32_t)k2;
    c->hstate.kept = (uint32_t)k2;
    return COLIBRI_OK;
}

// ... enqueued (no look at the pass from the host): k1 and the pass's first result are read on the device, `bound` >= k1
int filter_by_fillers_dev(colibri_ctx* c, const TrainPlan& pl, uint32_t* ids, uint32_t bound, uint32_t minsrc, uint32_t src_first, uint32_t src_count, const uint32_t* list,
                          const uint32_t* nlist) {
    int rc;
    (void)pl;
    if ((rc = dev_alloc(c, c->nsrc, (size_t)bound + 1)) || (rc = dev_alloc(c, c->skip_off, (size_t)bound + 1)) || (rc = dev_alloc(c, c->skip_tmp, 2 * (size_t)bound + 2))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->nsrc.p, 0, sizeof(uint32_t) * bound, c->stream));
    {
        Prof p(c, COLIBRI_K_SKIPGRAM);
        if (src_count)
            hipLaunchKernelGGL(skip_sources_results_dev_kernel, dim3(stream_grid(src_count)), dim3(kBlock), 0, c->stream, c->res_rep.p, src_first, src_count, ids, (const DevState*)c->state.p,
                               c->nsrc.p, bound);
        hipLaunchKernelGGL(skip_keep_flags_dev_kernel, dim3(stream_grid(bound)), dim3(kBlock), 0, c->stream, c->nsrc.p, (const DevState*)c->state.p, minsrc, bound);
    }
    if ((rc = scan_u32(c, c->nsrc.p, bound, c->skip_off.p, nullptr))) return rc;
    const unsigned long long* total = c->bsum.p + std::max<uint32_t>(1, blocks_for(bound, kBlock * 4));
    {
        Prof p(c, COLIBRI_K_SKIPGRAM);
        hipLaunchKernelGGL(skip_filter_gather_dev_kernel, dim3(stream_grid(bound)), dim3(kBlock), 0, c->stream, c->nsrc.p, c->skip_off.p, c->state.p, c->res_rep.p, c->res_cnt.p, c->skip_tmp.p, bound);
        hipLaunchKernelGGL(skip_filter_store_dev_kernel, dim3(stream_grid(bound)), dim3(kBlock), 0, c->stream, c->skip_tmp.p, bound, total, c->res_rep.p, c->res_cnt.p, (const DevState*)c->state.p);
        hipLaunchKernelGGL(skip_remap_ids_dev_kernel, dim3(stream_grid(pl.npos / 4 + 1)), dim3(kBlock), 0, c->stream, list, nlist, ids, c->nsrc.p, c->skip_off.p, (const DevState*)c->state.p, bound);
        hipLaunchKernelGGL(skip_set_kept_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p, total);
    }
    return COLIBRI_OK;
}

// The same pass on the radix path (emit -> level B -> per-bin LDS count -> resolve, over c->sklist): no global atomics. A level that is not the last
// interns every pair (threshold 1) and hands dense ids to the next one; the last level appends its survivors to the results and, when `ids_out` is
// given (indexed models), leaves every window's RESULT index there. kRerunOnTable: a bin outgrew its LDS table (the caller re-runs on the global table).
constexpr int kRerunOnTable = 1000;
// minsrc > 0 (indexed models, MINSKIPTYPES): a skipgram also needs that many distinct fillers = distinct surviving n-grams [src_first, src_first + src_count) of the
// results
// seglog (unindexed passes only: nothing of the pass is needed on the host before the order's other passes have run): no read-back — the pass is reset, logged and
// added to the run's state on the device (skip_pass_begin / skip_pass_end); *found_out / *kept_out stay untouched, the caller reads the log after the order.
int skipgram_pass_radix(colibri_ctx* c, const TrainPlan& pl, int n, uint32_t mask, const uint32_t* gate, const uint32_t* gate2, uint32_t thr, uint32_t res_total, uint32_t* found_out,
                        uint32_t* kept_out, uint32_t** ids_out, uint32_t minsrc = 0, uint32_t src_first = 0, uint32_t src_count = 0, uint32_t* seglog = nullptr,
                        uint32_t kept_bound = 0 /* logged passes with a filler filter: what the host knows the pass cannot keep more than */) {
    const std::vector<std::pair<int, int>> parts = mask_parts(mask, n);
    if (seglog != nullptr && minsrc > 1 && (ids_out == nullptr || kept_bound == 0)) return fail(c, COLIBRI_ERR_STATE, "skipgram_pass_radix: a logged pass with a filler filter needs ids and a bound");
    const uint32_t* left = part_ids(c, parts[0].second);
    uint32_t        offl = (uint32_t)parts[0].first;
    int             rc;
    for (size_t j = 1; j < parts.size(); ++j) {
        const bool last     = j + 1 == parts.size();
        const bool need_ids = !last || ids_out != nullptr;
        if (seglog != nullptr) {
            hipLaunchKernelGGL(skip_pass_begin_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p);
        } else {
            c->hstate.found = c->hstate.kept = c->hstate.admitted = c->hstate.valid = 0;
            c->hstate.radix_overflow = 0;
            if ((rc = write_state(c))) return rc;
        }
        uint32_t* const out = c->scratch[j & 1].p;
        KeyPair fn{gate, gate2, left, offl, part_ids(c, parts[j].second), (uint32_t)parts[j].first};
        if ((rc = binned_count_stage(c, pl, fn, n, true, last ? thr : 1u, false, need_ids, false, /*dense_code=*/true, 0, 0, c->skl, c->skl_n))) return rc;
        const BinnedIO io = binned_planes(c, pl, false);
        {
            Prof p(c, COLIBRI_K_PRUNE);
            hipLaunchKernelGGL(bin_kept_scan_kernel, dim3(kBins), dim3(kBlock), 0, c->stream, c->state.p, c->binstate.p, last ? pl.res_cap : 0xFFFFFFFFu);
            if (last)
                hipLaunchKernelGGL(compact_bins_kernel, dim3(1024), dim3(kBlock), 0, c->stream, io.sp_rep, io.sp_cnt, c->state.p, c->binstate.p, c->res_rep.p, c->res_cnt.p, pl.res_cap,
                                   (const uint32_t*)c->skl);
            hipLaunchKernelGGL(bin_advance_prepare_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p, c->binstate.p);
        }
        // (no fill of `out`: every listed position is written, valid or not, and every reader — the next level's keys, the filler filter, the emission of the
        // references — walks the same list)
        if (need_ids && (rc = binned_resolve_stage(c, pl, out, n, true, false, nullptr, 0u, /*prefill_ids=*/false, /*decode=*/true,
                                                   last ? (seglog != nullptr ? kDecodeBaseOnDevice : res_total) : 0u, c->skl, c->skl_n)))
            return rc;
        if (last && ids_out) *ids_out = out;
        left = out;
        offl = 0;
    }
    if (seglog != nullptr) {
        if (minsrc > 1 && (rc = filter_by_fillers_dev(c, pl, *ids_out, kept_bound, minsrc, src_first, src_count, c->skl, c->skl_n))) return rc;
        hipLaunchKernelGGL(skip_pass_end_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p, seglog, (uint32_t)n, mask);
        return COLIBRI_OK;
    }
    if ((rc = read_state(c))) return rc;
    if (c->hstate.radix_overflow) return kRerunOnTable;
    *found_out = c->hstate.found;
    *kept_out  = c->hstate.kept;
    const uint32_t k1 = c->hstate.kept;
    if (minsrc > 1 && k1 && ids_out) {
        uint32_t k2 = k1;
        if ((rc = filter_by_fillers(c, pl, *ids_out, res_total, k1, minsrc, src_first, src_count, c->skl, c->skl_n, &k2))) return rc;
        *kept_out = k2;
    }
    return COLIBRI_OK;
}

// One (length, gap mask) skipgram pass of a CONSTRAINED run (reference include/patternmodel.h:1163-1171 -> computeskipgrams :1410-1411): the masked form of a
// member window counts iff the constraint set holds it; its identity is then its pattern number there (constraint_probe_masked_kernel + KeyMember), counted
// on the radix path or on the table like the n-gram passes of the run. `gate`: the pattern numbers of the unmasked windows of this length.
// Pruning is the BASE pruneskipgrams (:2167-2186) for either model type — train() calls it with an unsigned threshold, which the indexed model's
// pruneskipgrams(int, int, int) (:3362) does not override —: nothing when MINSKIPTYPES <= 1, else the occurrence threshold MINTOKENS_SKIPGRAMS
// (checked against the reference: tests/golden/constrained.js_*).
int constrained_skipgram_pass(colibri_ctx* c, const TrainPlan& pl, const colibri_options& o, int n, uint32_t mask, const uint32_t* gate, bool radix, uint32_t res_total,
                              uint32_t* found_out, uint32_t* kept_out) {
    int rc;
    if ((rc = dev_alloc(c, c->scratch[0], (size_t)pl.npos + 1)) || (rc = dev_alloc(c, c->scratch[1], (size_t)pl.npos + 1))) return rc;
    uint32_t* const memb = c->scratch[0].p;
    uint32_t* const ids  = c->scratch[1].p;
    {
        Prof p(c, COLIBRI_K_SKIPGRAM);
        hipLaunchKernelGGL(constraint_probe_masked_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->bytes.p, c->tokstart.p, gate, c->cs.table.p, c->cs.cap, c->cs.bytes.p, c->cs.off.p,
                           pl.npos, n, mask, memb);
    }
    const uint32_t thr = o.minskiptypes > 1 ? (uint32_t)std::max(1, o.mintokens_skipgrams) : 1u;
    c->hstate.found = c->hstate.kept = c->hstate.admitted = c->hstate.valid = 0;
    c->hstate.radix_overflow = 0;
    if ((rc = write_state(c))) return rc;
    const KeyMember km{memb};
    if (radix) {
        if ((rc = binned_count_stage(c, pl, km, n, false, thr, false, true, false, /*dense_code=*/true))) return rc;
        const BinnedIO io = binned_planes(c, pl, false);
        {
            Prof p(c, COLIBRI_K_PRUNE);
            hipLaunchKernelGGL(bin_kept_scan_kernel, dim3(kBins), dim3(kBlock), 0, c->stream, c->state.p, c->binstate.p, pl.res_cap);
            hipLaunchKernelGGL(compact_bins_kernel, dim3(1024), dim3(kBlock), 0, c->stream, io.sp_rep, io.sp_cnt, c->state.p, c->binstate.p, c->res_rep.p, c->res_cnt.p, pl.res_cap,
                               (const uint32_t*)nullptr);
            hipLaunchKernelGGL(bin_advance_prepare_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p, c->binstate.p);
        }
        if ((rc = binned_resolve_stage(c, pl, ids, n, false, false, nullptr, 0u, /*prefill_ids=*/true, /*decode=*/true, res_total))) return rc;
    } else {
        launch_clear(c, pl);
        launch_count(c, pl, km, ids, 3, COLIBRI_K_SKIPGRAM);
        launch_prune(c, pl, thr, nullptr, 0);
        launch_resolve(c, pl, ids);
    }
    if ((rc = read_state(c))) return rc;
    if (c->hstate.radix_overflow) return kRerunOnTable;
    *found_out = c->hstate.found;
    *kept_out  = c->hstate.kept;
    if (o.indexed && *kept_out && (rc = emit_pairs(c, pl, ids, false))) return rc;
    return COLIBRI_OK;
}

// key byte lengths + offsets on the device, so that result_sizes can answer and export is a gather
int prepare_export(colibri_ctx* c) {
    int            rc;
    const uint32_t R = c->hstate.res_total;
    c->keybytes      = 0;
    if ((rc = dev_alloc(c, c->keylen, (size_t)R + 1))) return rc;
    if ((rc = dev_alloc(c, c->keyoff, (size_t)R + 1))) return rc;
    if (R) {
        Prof p(c, COLIBRI_K_EXPORT);
        for (const auto& sg : c->segments)
            hipLaunchKernelGGL(export_len_kernel, dim3(blocks_for(sg.count, kBlock)), dim3(kBlock), 0, c->stream, c->tokstart.p, c->res_rep.p, sg.first, sg.count, sg.n, sg.mask, c->keylen.p);
        const uint32_t nb = blocks_for(R, kBlock * 4);
        if ((rc = dev_alloc(c, c->bsum, (size_t)nb + 1))) return rc;
        hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(kBlock), 0, c->stream, c->keylen.p, R, c->bsum.p);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, c->stream, c->bsum.p, nb, c->bsum.p + nb);
        hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(kBlock), 0, c->stream, c->keylen.p, R, c->bsum.p, c->keyoff.p);
        unsigned long long total = 0;
        HIP_TRY(c, hipMemcpyAsync(&total, c->bsum.p + nb, sizeof total, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipGetLastError());
        c->keybytes = total;
    }
    return COLIBRI_OK;
}

// grow a pair buffer keeping its first `keep` elements
template <class T>
int grow_keep(colibri_ctx* c, DevBuf<T>& b, uint64_t need, uint64_t keep) {
    if (b.p && b.n >= need) return COLIBRI_OK;
    DevBuf<T> nb;
    int              rc;
    if ((rc = dev_alloc(c, nb, (size_t)std::max<uint64_t>(need, b.n * 2)))) return rc;
    if (b.p && keep) HIP_TRY(c, hipMemcpyAsync(nb.p, b.p, keep * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    dev_free(b);
    b = nb;
    return COLIBRI_OK;
}

// append (result id, position) for every position of `ids` that carries a result id, in position order; nothing is read back — the number of
// pairs lives on the device (c->pair_chain) until pairs_count() fetches it. Pairs beyond the buffer's room are counted, not written; `ensure` (the
// sharded levels, which cannot be run again as a whole): wait, and when the room was short grow the buffer and repeat the pass.
int pairs_begin(colibri_ctx* c, uint32_t npos) {
    int rc;
    if ((rc = dev_alloc(c, c->pair_chain, (size_t)kChainWords + 1))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->pair_chain.p, 0, sizeof(unsigned long long) * kChainWords, c->stream));
    c->pair_pass = 0;
    c->npairs    = 0;
    c->hot_used  = false;
    c->hot_below = c->hot_n = 0;
    if (!c->lds_tested) {  // once per context: are ranks from returning LDS adds the ranks? (kernels.hpp: lds_order_selftest_kernel); otherwise every rank by ballots
        uint32_t bad = 0;
        HIP_TRY(c, hipMemsetAsync(c->pair_chain.p + kChainWords, 0, sizeof(unsigned long long), c->stream));
        hipLaunchKernelGGL(lds_order_selftest_kernel, dim3(64), dim3(kS64Threads), 0, c->stream, reinterpret_cast<uint32_t*>(c->pair_chain.p + kChainWords));
        HIP_TRY(c, hipMemcpyAsync(&bad, c->pair_chain.p + kChainWords, sizeof bad, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipGetLastError());
#ifdef COLIBRI_TEST_HOOKS  // (tests/standin/lib/libcolibri_hip_hooks.so only)
        if (getenv("COLIBRI_FAULT_LDS_ORDER") && !strcmp(getenv("COLIBRI_FAULT_LDS_ORDER"), "selftest")) bad = 1;
#endif
        if (bad) c->hot_off = true;
        c->lds_tested = true;
    }
    c->pair_split = false;  // (the caller's to set: colibri_train_once does when the pairs are packed; the sharded runs keep whole pairs — they cut the sorted references by id)
    if (c->pairs[0].n < 2ull * npos && (rc = dev_alloc(c, c->pairs[0], (size_t)(2ull * npos) + 1))) return rc;  // the usual model: ~1.6 pairs per position at n <= 5
    // position -> (sentence, token) table of the corpus (once per upload)
    if (!c->pos_blocks_valid) {
        if ((rc = dev_alloc(c, c->pos_blocks, (size_t)c->npos / 64 + 2))) return rc;
        Prof p(c, COLIBRI_K_INDEX);
        hipLaunchKernelGGL(position_blocks_kernel, dim3(stream_grid((uint64_t)c->npos / 16 + 1)), dim3(kBlock), 0, c->stream, c->cls.p, c->delimpos.p, c->ndelim, c->npos, c->pos_blocks.p);
        c->pos_blocks_valid = true;
    }
    // packed pairs: a 31-bit id, the sentence (counted from the shard's first) and the token offset (u16 in the reference, datatypes.h:36) in one word —
    // possible whenever sentence count and longest sentence leave the room, i.e. always in practice; else id << 32 | position and the look-up after the sort
    uint64_t longest = 0;
    for (size_t len = c->lenhist.size(); len-- > 0;)
        if (c->lenhist[len]) {
            longest = len;
            break;
        }
    uint32_t sb = 1, tb = 1;
    while ((1ull << sb) < (uint64_t)c->nsent + 2) ++sb;
    while (tb < 16 && (1ull << tb) < longest + 1) ++tb;
    if (longest >= 65536) tb = 16;
    if (sb + tb <= 33 && !getenv("COLIBRI_UNPACKED_PAIRS")) {
        c->pair_sb = sb;
        c->pair_tb = tb;
    } else {
        c->pair_sb = c->pair_tb = 0;
    }
    return COLIBRI_OK;
}
// pairs so far; *overflowed: the buffer was too small for them
int pairs_count(colibri_ctx* c, uint64_t* n, bool* overflowed) {
    unsigned long long h[kChainWords] = {0, 0, 0, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(h, c->pair_chain.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    *n          = h[c->pair_pass];
    *overflowed = h[2] != 0;
    c->hot_n     = c->hot_used ? h[kChainHot] : 0;  // (references that never were pairs: emit_hot_write_kernel)
    c->hot_below = c->hot_used ? h[kChainBelow] : 0;
    c->hot_disorder = c->hot_used && h[kChainDisorder] != 0;
    return COLIBRI_OK;
}
// surv / resid (order 1 of an indexed model whose ids[1] nobody else reads): `ids` is the class per position; a class's survivor bit and result index stand in for the id
// hot1 (order 1 of an enqueued indexed run with split pairs; c->uni_resid holds the classes' result indices): the references of the most frequent unigrams go straight to
// their final places and never become pairs (kernels.hpp: emit_hot_*)
int emit_pairs(colibri_ctx* c, const TrainPlan& pl, const uint32_t* ids, bool ensure, const uint32_t* surv, const uint32_t* resid, bool hot1) {
    const uint32_t ntiles = std::max<uint32_t>(1, blocks_for(pl.npos, kPairTile));
    int            rc0;
    if ((rc0 = dev_alloc(c, c->idx_cnt, (size_t)ntiles + 2))) return rc0;
    uint32_t* const cnt = c->idx_cnt.p;
    if (hot1 && c->pair_split && c->pair_sb != 0 && !ensure && !c->hot_used && !c->hot_off && !getenv("COLIBRI_NO_HOT_REFS")) {
        const uint64_t cap = c->pairs[0].n;
        // the reference arrays exist from here on (finalize_index only ever asks for less): every pair's room plus the hot references'
        if ((rc0 = dev_alloc(c, c->ref_sentence, (size_t)cap + pl.npos + 2)) || (rc0 = dev_alloc(c, c->ref_token, (size_t)cap + pl.npos + 2)) ||
            (rc0 = dev_alloc(c, c->hot_cnt, (size_t)kHotIds * ntiles)) || (rc0 = dev_alloc(c, c->hot_info, 1)))
            return rc0;
        Prof p(c, COLIBRI_K_INDEX);
        hipLaunchKernelGGL(hot_setup_kernel, dim3(1), dim3(kPairThreads), 0, c->stream, (const uint32_t*)c->uni_resid.p, c->maxclass + 1, (const uint32_t*)c->res_cnt.p,
                           (const DevState*)c->state.p, c->hot_info.p, /*closed=*/surv == nullptr ? 1u : 0u);  // (the class form is emitted before the order's figures are closed)
        hipLaunchKernelGGL(emit_hot_count_kernel, dim3(ntiles), dim3(kPairThreads), 0, c->stream, ids, pl.npos, cnt, (const DevState*)c->state.p, surv, resid,
                           surv != nullptr ? &c->state.p->valid : (uint32_t*)nullptr, (const HotInfo*)c->hot_info.p, c->hot_cnt.p, ntiles);
        hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(kPairThreads), 0, c->stream, cnt, ntiles, cnt + ntiles);
        hipLaunchKernelGGL(hot_scan_kernel, dim3(kHotIds), dim3(kPairThreads), 0, c->stream, c->hot_cnt.p, ntiles, c->hot_info.p);
        hipLaunchKernelGGL(hot_base_kernel, dim3(1), dim3(kHotIds), 0, c->stream, c->hot_info.p, c->pair_chain.p);
        hipLaunchKernelGGL(pairs_advance_kernel, dim3(1), dim3(1), 0, c->stream, c->pair_chain.p, c->pair_pass, cnt + ntiles, cap);
        hipLaunchKernelGGL(emit_hot_write_kernel, dim3(8 * ((ntiles + 7) / 8)), dim3(kPairThreads), 0, c->stream, ids, pl.npos, (const uint32_t*)cnt,
                           c->pair_chain.p, c->pair_pass, cap, reinterpret_cast<uint32_t*>(c->pairs[0].p), reinterpret_cast<uint32_t*>(c->pairs[0].p) + cap,
                           (const PosBlock*)c->pos_blocks.p, c->pair_tb, surv, resid, (const HotInfo*)c->hot_info.p, (const uint32_t*)c->hot_cnt.p, ntiles, c->first_sentence,
                           c->ref_sentence.p, c->ref_token.p, (const DevState*)c->state.p);
        c->pair_pass ^= 1;
        c->hot_used = true;
        return COLIBRI_OK;
    }
    for (;;) {
        {
            Prof p(c, COLIBRI_K_INDEX);
            const uint64_t cap = c->pairs[0].n;
            hipLaunchKernelGGL(emit_count_kernel, dim3(ntiles), dim3(kPairThreads), 0, c->stream, ids, pl.npos, cnt, (const DevState*)c->state.p, surv,
                               surv != nullptr ? &c->state.p->valid : (uint32_t*)nullptr);
            hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(kPairThreads), 0, c->stream, cnt, ntiles, cnt + ntiles);
            hipLaunchKernelGGL(pairs_advance_kernel, dim3(1), dim3(1), 0, c->stream, c->pair_chain.p, c->pair_pass, cnt + ntiles, cap);
            const bool packed = c->pair_sb != 0;
            hipLaunchKernelGGL(emit_write_kernel, dim3(ntiles), dim3(kPairThreads), 0, c->stream, ids, pl.npos, cnt, c->pair_chain.p, c->pair_pass, cap, c->pairs[0].p,
                               packed ? (const PosBlock*)c->pos_blocks.p : (const PosBlock*)nullptr, c->pair_sb, c->pair_tb,
                               c->pair_split ? reinterpret_cast<uint32_t*>(c->pairs[0].p) + cap : (uint32_t*)nullptr, resid);
        }
        c->pair_pass ^= 1;
        if (!ensure) return COLIBRI_OK;
        if (c->pair_split) return fail(c, COLIBRI_ERR_STATE, "emit_pairs: split pairs do not grow in place");
        uint64_t n = 0;
        bool     over = false;
        int      rc;
        if ((rc = pairs_count(c, &n, &over))) return rc;
        if (!over) {
            c->npairs = n;
            return COLIBRI_OK;
        }
        if ((rc = grow_keep(c, c->pairs[0], n, c->npairs))) return rc;  // the pairs of the passes before this one stay; the pass itself runs again
        HIP_TRY(c, hipMemsetAsync(c->pair_chain.p + 2, 0, sizeof(unsigned long long), c->stream));
        c->pair_pass ^= 1;
    }
}

// the same over the entries of the order's skip list (c->skl, at most `bound` of them): the occurrences of the skipgrams a pass kept (kernels.hpp: emit_*_list_kernel)
int emit_pairs_list(colibri_ctx* c, uint32_t bound, const uint32_t* ids) {
    const uint32_t ntiles = std::max<uint32_t>(1, blocks_for(bound, kPairTile));
    int            rc;
    if ((rc = dev_alloc(c, c->idx_cnt, (size_t)ntiles + 2))) return rc;
    uint32_t* const cnt = c->idx_cnt.p;
    Prof            p(c, COLIBRI_K_INDEX);
    const uint64_t  cap = c->pairs[0].n;
    hipLaunchKernelGGL(emit_count_list_kernel, dim3(ntiles), dim3(kPairThreads), 0, c->stream, ids, (const uint32_t*)c->skl, (const uint32_t*)c->skl_n, cnt, c->state.p);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(kPairThreads), 0, c->stream, cnt, ntiles, cnt + ntiles);
    hipLaunchKernelGGL(pairs_advance_kernel, dim3(1), dim3(1), 0, c->stream, c->pair_chain.p, c->pair_pass, cnt + ntiles, cap);
    const bool packed = c->pair_sb != 0;
    hipLaunchKernelGGL(emit_write_list_kernel, dim3(ntiles), dim3(kPairThreads), 0, c->stream, ids, (const uint32_t*)c->skl, (const uint32_t*)c->skl_n, (const uint32_t*)cnt, c->pair_chain.p,
                       c->pair_pass, cap, c->pairs[0].p, packed ? (const PosBlock*)c->pos_blocks.p : (const PosBlock*)nullptr, c->pair_sb, c->pair_tb,
                       c->pair_split ? reinterpret_cast<uint32_t*>(c->pairs[0].p) + cap : (uint32_t*)nullptr);
    c->pair_pass ^= 1;
    return COLIBRI_OK;
}

// two-level exclusive scan of n u32 values into u64 offsets (out[0..n-1]); *total (optional) = their sum, read back after a sync
int scan_u32(colibri_ctx* c, const uint32_t* in, uint32_t n, unsigned long long* out, unsigned long long* total) {
    const uint32_t nb = std::max<uint32_t>(1, blocks_for(n, kBlock * 4));
    int            rc;
    if ((rc = dev_alloc(c, c->bsum, (size_t)nb + 1))) return rc;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(kBlock), 0, c->stream, in, n, c->bsum.p);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, c->stream, c->bsum.p, nb, c->bsum.p + nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(kBlock), 0, c->stream, in, n, c->bsum.p, out);
    if (total) {
        HIP_TRY(c, hipMemcpyAsync(total, c->bsum.p + nb, sizeof *total, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipGetLastError());
    }
    return COLIBRI_OK;
}

// stable LSD radix sort of (key, value) pairs by the low `bits` bits of the key, 8 bits per pass, ping-pong between key[0/1], val[0/1];
// `cur` names the buffers holding the input and, on return, the output
int radix_sort_pairs(colibri_ctx* c, uint32_t* const key[2], uint32_t* const val[2], uint64_t n, int bits, int& cur) {
    if (!n) return COLIBRI_OK;
    const uint32_t             nblocks = (uint32_t)((n + kSortTile - 1) / kSortTile);
    const uint32_t             nh      = 256u * nblocks;
    const uint32_t             nb      = blocks_for(nh, kBlock * 4);
    int                        rc;
    if ((rc = dev_alloc(c, c->sort_hist, nh)) || (rc = dev_alloc(c, c->sort_off, nh)) || (rc = dev_alloc(c, c->sort_bsum, (size_t)nb + 1))) return rc;
    DevBuf<uint32_t>&           ghist = c->sort_hist;
    DevBuf<unsigned long long>& goff = c->sort_off, &bsum = c->sort_bsum;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(sort_hist_kernel, dim3(nblocks), dim3(kBlock), 0, c->stream, key[cur], n, shift, nblocks, ghist.p);
        hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(kBlock), 0, c->stream, ghist.p, nh, bsum.p);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, c->stream, bsum.p, nb, bsum.p + nb);
        hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(kBlock), 0, c->stream, ghist.p, nh, bsum.p, goff.p);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblocks), dim3(kBlock), 0, c->stream, key[cur], val[cur], n, shift, nblocks, goff.p, key[cur ^ 1], val[cur ^ 1]);
        cur ^= 1;
    }
    return COLIBRI_OK;  // (the scratch belongs to the context: nothing to wait for)
}
inline int bits_for(uint64_t nvalues) {
    int bits = 1;
    while (bits < 32 && (1ull << bits) < nvalues) ++bits;
    return bits;
}

// group the pairs by result id (stable LSD radix sort) and turn positions into (sentence, token)
constexpr int kRerunPairs = 1001;  // the pair buffer was too small (it has been enlarged): the run again
constexpr int kRerunRanks = 1002;  // a checked rank disagreed (c->hot_off is set): the run again, every rank matched with ballots
int finalize_index(colibri_ctx* c, uint32_t nresults, bool keep_sorted_ids = false) {
    int      rc;
    uint64_t n = 0;
    bool     over = false;
    if ((rc = pairs_count(c, &n, &over))) return rc;
    if (over) {
        dev_free(c->pairs[0]);
        if ((rc = dev_alloc(c, c->pairs[0], (size_t)(n + n / 8) + 1))) return rc;
        return kRerunPairs;
    }
#ifdef COLIBRI_TEST_HOOKS  // (tests/standin/lib/libcolibri_hip_hooks.so only: the shipped library does not read it) the first attempt pretends a hot run came out of order
    if (c->hot_used && !c->hot_off && getenv("COLIBRI_FAULT_LDS_ORDER") && !strcmp(getenv("COLIBRI_FAULT_LDS_ORDER"), "hot")) c->hot_disorder = true;
#endif
    if (c->hot_disorder) {  // (never seen: emit_hot_write_kernel's ranks rest on the order in which the LDS serves the lanes of one instruction)
        c->hot_off = true;
        return kRerunRanks;
    }
    c->npairs = n + c->hot_n;  // (the hot unigrams' references were never pairs: they lie where they belong already)
    if (c->hot_used && (c->ref_sentence.n < c->npairs + 1 || c->ref_token.n < c->npairs + 1)) return fail(c, COLIBRI_ERR_STATE, "finalize_index: the reference arrays are smaller than the model's references");
    if ((rc = dev_alloc(c, c->ref_sentence, (size_t)c->npairs + 1)) || (rc = dev_alloc(c, c->ref_token, (size_t)c->npairs + 1))) return rc;
    if (!n) return COLIBRI_OK;
    if ((rc = dev_alloc(c, c->pairs[1], (size_t)n))) return rc;
    int cur = 0;
    {
        Prof p(c, COLIBRI_K_INDEX);
        // stable LSD passes over the id (high word), 8 bits each
        const uint32_t nblocks = (uint32_t)((n + kS64Tile - 1) / kS64Tile);
        const uint32_t nh = 256u * nblocks, nb = blocks_for(nh, kBlock * 4);
        if ((rc = dev_alloc(c, c->sort_hist, nh)) || (rc = dev_alloc(c, c->sort_off, nh)) || (rc = dev_alloc(c, c->sort_bsum, (size_t)nb + 1))) return rc;
        if (keep_sorted_ids && (rc = dev_alloc(c, c->sh.sorted_gid, (size_t)n))) return rc;  // sharded mode: the caller still needs the (sorted) global ids to cut the references into runs
        const int  nbits  = bits_for(nresults);
        const bool packed = c->pair_sb != 0;
        const int  idshift = packed ? (int)(c->pair_sb + c->pair_tb) : 32;
        if (c->pair_split) {
            if (keep_sorted_ids) return fail(c, COLIBRI_ERR_STATE, "finalize_index: split pairs keep no ids");
            const uint32_t nblocks = (uint32_t)((n + (uint64_t)kITile * kISuper - 1) / ((uint64_t)kITile * kISuper));  // (one table column per block of kISuper tiles)
            const uint32_t nh = 256u * nblocks, nb = blocks_for(nh, kBlock * 4);
            // buffers: pairs[0] = { id u32[cap], reference u32[cap] } as emitted; a pass writes { reference u32[n], id rest TOUT[n] } into the other buffer
            const uint64_t  cap0 = c->pairs[0].n;
            const uint32_t  exact_ranks = (c->hot_off || getenv("COLIBRI_EXACT_RANKS")) ? 1u : 0u;  // (kernels.hpp: isort_scatter_kernel)
            const void*     dig  = c->pairs[0].p;
            const uint32_t* pay  = reinterpret_cast<const uint32_t*>(c->pairs[0].p) + cap0;
            int             in_bytes = 4;
            for (int shift = 0; shift < nbits; shift += 8) {
                const bool last = shift + 8 >= nbits;
                const int  rem  = nbits - shift - 8, out_bytes = rem <= 8 ? 1 : rem <= 16 ? 2 : 4;
                uint32_t* const opay = reinterpret_cast<uint32_t*>(c->pairs[cur ^ 1].p);
                void* const     odig = opay + n;
                if (in_bytes == 4)
                    hipLaunchKernelGGL(isort_hist_kernel<uint32_t>, dim3(nblocks), dim3(kS64Threads), 0, c->stream, (const uint32_t*)dig, n, nblocks, c->sort_hist.p);
                else if (in_bytes == 2)
                    hipLaunchKernelGGL(isort_hist_kernel<uint16_t>, dim3(nblocks), dim3(kS64Threads), 0, c->stream, (const uint16_t*)dig, n, nblocks, c->sort_hist.p);
                else
                    hipLaunchKernelGGL(isort_hist_kernel<uint8_t>, dim3(nblocks), dim3(kS64Threads), 0, c->stream, (const uint8_t*)dig, n, nblocks, c->sort_hist.p);
                hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(kBlock), 0, c->stream, c->sort_hist.p, nh, c->sort_bsum.p);
                hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, c->stream, c->sort_bsum.p, nb, c->sort_bsum.p + nb);
                hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(kBlock), 0, c->stream, c->sort_hist.p, nh, c->sort_bsum.p, c->sort_off.p);
#define ISORT(TIN, TOUT, FIN)                                                                                                                                                   \
    hipLaunchKernelGGL((isort_scatter_kernel<TIN, TOUT, FIN>), dim3(nblocks), dim3(kS64Threads), 0, c->stream, (const TIN*)dig, pay, n, nblocks, c->sort_off.p, opay, (TOUT*)odig, \
                       c->first_sentence, c->ref_sentence.p, c->ref_token.p, c->pair_tb, c->hot_below, c->hot_n, exact_ranks, c->pair_chain.p)
                if (last) {
                    if (in_bytes == 4) ISORT(uint32_t, uint8_t, true);
                    else if (in_bytes == 2) ISORT(uint16_t, uint8_t, true);
                    else ISORT(uint8_t, uint8_t, true);
                } else if (in_bytes == 4) {
                    if (out_bytes == 4) ISORT(uint32_t, uint32_t, false);
                    else if (out_bytes == 2) ISORT(uint32_t, uint16_t, false);
                    else ISORT(uint32_t, uint8_t, false);
                } else {
                    ISORT(uint16_t, uint8_t, false);  // (u16 in: 9..16 bits left, 1..8 after this pass)
                }
#undef ISORT
                cur ^= 1;
                dig      = odig;
                pay      = opay;
                in_bytes = out_bytes;
            }
        } else
        for (int shift = 0; shift < nbits; shift += 8) {
            const bool last = shift + 8 >= nbits;  // the last pass writes (sentence, token) [and the ids] instead of pairs
            hipLaunchKernelGGL(sort64_hist_kernel, dim3(nblocks), dim3(kS64Threads), 0, c->stream, c->pairs[cur].p, n, idshift + shift, nblocks, c->sort_hist.p);
            hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(kBlock), 0, c->stream, c->sort_hist.p, nh, c->sort_bsum.p);
            hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, c->stream, c->sort_bsum.p, nb, c->sort_bsum.p + nb);
            hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(kBlock), 0, c->stream, c->sort_hist.p, nh, c->sort_bsum.p, c->sort_off.p);
            if (!last)
                hipLaunchKernelGGL((sort64_scatter_kernel<false>), dim3(nblocks), dim3(kS64Threads), 0, c->stream, c->pairs[cur].p, n, idshift + shift, nblocks, c->sort_off.p,
                                   c->pairs[cur ^ 1].p, (const PosBlock*)nullptr, 0u, (uint32_t*)nullptr, (uint16_t*)nullptr, (uint32_t*)nullptr, 0u, 0u);
            else
                hipLaunchKernelGGL((sort64_scatter_kernel<true>), dim3(nblocks), dim3(kS64Threads), 0, c->stream, c->pairs[cur].p, n, idshift + shift, nblocks, c->sort_off.p,
                                   (unsigned long long*)nullptr, packed ? (const PosBlock*)nullptr : (const PosBlock*)c->pos_blocks.p, c->first_sentence, c->ref_sentence.p,
                                   c->ref_token.p, keep_sorted_ids ? c->sh.sorted_gid.p : (uint32_t*)nullptr, c->pair_sb, c->pair_tb);
            cur ^= 1;
        }
    }
    unsigned long long disorder = 0;
    if (c->pair_split) HIP_TRY(c, hipMemcpyAsync(&disorder, c->pair_chain.p + kChainDisorder, sizeof disorder, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
#ifdef COLIBRI_TEST_HOOKS
    if (c->pair_split && !c->hot_off && getenv("COLIBRI_FAULT_LDS_ORDER") && !strcmp(getenv("COLIBRI_FAULT_LDS_ORDER"), "sort")) disorder = 1;
#endif
    if (disorder) {  // (never seen) a spot-checked row of the sort disagreed with its ballots: again, with every rank matched
        c->hot_off = true;
        return kRerunRanks;
    }
    return COLIBRI_OK;
}

// DOPATTERNPERLINE (patternlist.hpp): one grouping pass over the lines per line length that occurs
int train_pattern_list(colibri_ctx* c, const colibri_options& o, colibri_stats* stats_out) {
    const auto     t0 = std::chrono::steady_clock::now();
    int            rc;
    colibri_stats& s = c->stats;
    std::memset(&s, 0, sizeof s);
    c->segments.clear();
    c->npairs           = 0;
    c->hstate           = DevState{};
    const uint32_t npos = c->npos, nlines = c->ndelim;
    if (npos) {
        uint32_t last = 0;
        if (nlines) HIP_TRY(c, hipMemcpy(&last, c->delimpos.p + (nlines - 1), sizeof last, hipMemcpyDeviceToHost));
        if (!nlines || last + 1 != npos) return fail(c, COLIBRI_ERR_CORPUS, "pattern list: the last line lacks its end-of-line marker");
    }
    const uint32_t res_cap = nlines + 1;
    if ((rc = dev_alloc(c, c->res_rep, res_cap)) || (rc = dev_alloc(c, c->res_cnt, res_cap)) || (rc = dev_alloc(c, c->state, 1))) return rc;
    uint32_t res_total = 0;
    if (nlines) {
        DevBuf<uint32_t>           line_pos, line_ntok, flen, slot_of, isrep;
        DevBuf<unsigned long long> line_off, unit, rank;
        DevBuf<FSlot>              table;
        DevBuf<FlexInfo>           info;
        auto                       cleanup = [&]() {
            dev_free(line_pos); dev_free(line_ntok); dev_free(flen); dev_free(slot_of); dev_free(isrep); dev_free(line_off); dev_free(unit); dev_free(rank); dev_free(table); dev_free(info);
        };
        struct Guard {
            decltype(cleanup)& f;
            ~Guard() { f(); }
        } guard{cleanup};
        if ((rc = dev_alloc(c, line_pos, nlines)) || (rc = dev_alloc(c, line_ntok, nlines)) || (rc = dev_alloc(c, flen, (size_t)nlines + 1)) || (rc = dev_alloc(c, slot_of, nlines)) ||
            (rc = dev_alloc(c, isrep, (size_t)nlines + 1)) || (rc = dev_alloc(c, line_off, nlines)) || (rc = dev_alloc(c, unit, (size_t)nlines + 1)) ||
            (rc = dev_alloc(c, rank, (size_t)nlines + 1)) || (rc = dev_alloc(c, info, 1)))
            return rc;
        uint64_t most = 0;  // the most lines any one length has: sizes the table
        for (size_t n = 1; n < c->lenhist.size() && n <= (size_t)o.maxlength; ++n) most = std::max<uint64_t>(most, c->lenhist[n]);
        const uint32_t cap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, most + (most >> 1) + 1024);
        if ((rc = dev_alloc(c, table, cap))) return rc;
        {
            Prof p(c, COLIBRI_K_COUNT);
            hipLaunchKernelGGL(ppl_lines_kernel, dim3(stream_grid((uint64_t)nlines + 1)), dim3(kBlock), 0, c->stream, c->delimpos.p, nlines, c->tokstart.p, line_pos.p, line_ntok.p, line_off.p,
                               unit.p);
        }
        for (int n = 1; n <= o.maxlength && n < COLIBRI_MAX_ORDER && (size_t)n < c->lenhist.size(); ++n) {
            if (c->lenhist[(size_t)n] == 0) continue;
            unsigned long long groups = 0;
            bool               grouped = false;
            for (int attempt = 0; attempt < 4 && !grouped; ++attempt) {
                const uint64_t seed = 0xBB67AE8584CAA73Bull + 0x9E3779B97F4A7C15ull * (uint64_t)attempt;
                HIP_TRY(c, hipMemsetAsync(info.p, 0, sizeof(FlexInfo), c->stream));
                {
                    Prof p(c, COLIBRI_K_COUNT);
                    hipLaunchKernelGGL(ppl_select_kernel, dim3(stream_grid(nlines)), dim3(kBlock), 0, c->stream, line_pos.p, line_ntok.p, nlines, c->tokstart.p, (uint32_t)n, flen.p);
                    hipLaunchKernelGGL(flex_clear_kernel, dim3(stream_grid(cap)), dim3(kBlock), 0, c->stream, table.p, cap);
                    hipLaunchKernelGGL(flex_insert_kernel, dim3(stream_grid(nlines)), dim3(kBlock), 0, c->stream, c->bytes.p, line_off.p, flen.p, unit.p, nlines, seed, table.p, cap, slot_of.p);
                    hipLaunchKernelGGL(flex_verify_kernel, dim3(stream_grid(nlines)), dim3(kBlock), 0, c->stream, c->bytes.p, line_off.p, flen.p, nlines, table.p, slot_of.p, isrep.p, info.p);
                }
                FlexInfo got{};
                HIP_TRY(c, hipMemcpyAsync(&got, info.p, sizeof got, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                HIP_TRY(c, hipGetLastError());
                grouped = !got.collision;
            }
            if (!grouped) return fail(c, COLIBRI_ERR_OVERFLOW, "pattern list: hash collisions under four seeds");
            if ((rc = scan_u32(c, isrep.p, nlines, rank.p, &groups))) return rc;
            const uint32_t k = (uint32_t)groups;
            {
                Prof p(c, COLIBRI_K_PRUNE);
                hipLaunchKernelGGL(ppl_results_kernel, dim3(stream_grid(nlines)), dim3(kBlock), 0, c->stream, isrep.p, rank.p, table.p, slot_of.p, line_pos.p, nlines, res_total, res_cap,
                                   c->res_rep.p, c->res_cnt.p);
            }
            s.found[n] = s.kept[n] = k;
            s.admitted[n]          = c->lenhist[(size_t)n];
            if (k) {
                c->segments.push_back({res_total, k, n, 0u});
                if (!s.minn) s.minn = n;
                s.maxn = n;
            }
            res_total += k;
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipGetLastError());
    }
    c->hstate.res_total = res_total;
    collect_events(c);
    s.totaltokens = c->ntokens;  // every token of the corpus, the lines that are too long included (patternmodel.h:1047-1048 comes before :1056)
    s.totaltypes  = s.kept[1];   // the distinct one-token lines (totalwordtypesingroup(NGRAM, 1) at :1201-1207)
    s.nsentences  = c->nsent;
    s.npatterns   = res_total;
    s.train_ms    = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->trained    = true;
    c->keybytes   = 0;
    c->last_mode  = 0;
    c->export_ready = false;  // key lengths / offsets: computed when the results are first asked for (colibri_result_sizes)
    if (stats_out) *stats_out = s;
    return COLIBRI_OK;
}

}  // namespace

static int colibri_train_once(colibri_ctx* c, const colibri_options* opt_in, colibri_stats* stats_out);
// a run is about to be repeated (exactly) on another engine or with more room: what colibri_stats.fallback_reason / retries report
static inline void note_retry(colibri_ctx* c, int reason) {
    if (getenv("COLIBRI_DEBUG_OVERFLOW")) fprintf(stderr, "colibri: the run is repeated (COLIBRI_FALLBACK_* %d)\n", reason);
    if (c->run_retries++ == 0) c->run_fallback = reason;
}
extern "C" int colibri_train(colibri_ctx* c, const colibri_options* opt_in, colibri_stats* stats_out) {
    if (!c || !opt_in) return COLIBRI_ERR_ARG;
    c->run_path = c->run_fallback = c->run_retries = 0;
    auto report = [&](int rc_) {
        if (stats_out && rc_ == COLIBRI_OK) {
            stats_out->path            = c->run_path;
            stats_out->fallback_reason = c->run_fallback;
            stats_out->retries         = c->run_retries;
            stats_out->reserved_       = 0;
        }
        return rc_;
    };
    for (;;) {
        const int rc = colibri_train_once(c, opt_in, stats_out);
#ifdef BI2_PROF  // (experimental builds only) where bi2_count_kernel's waves spent their cycles: sections of process_bin, summed over waves and launches of this call
        {
            unsigned long long h[16] = {0}, z[16] = {0};
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(colibri::bi2_prof), sizeof h) == hipSuccess) {
                unsigned long long t = 0;
                for (int k = 0; k < 12; ++k) t += h[k];
                fprintf(stderr, "BI2_PROF");
                for (int k = 0; k < 12; ++k) fprintf(stderr, " s%d=%.1f%%", k, t ? 100.0 * (double)h[k] / (double)t : 0.0);
                fprintf(stderr, " total=%llu\n", t);
                (void)hipMemcpyToSymbol(HIP_SYMBOL(colibri::bi2_prof), z, sizeof z);
            }
        }
#endif
#ifdef COLIBRI_KPROF  // (experimental builds only) phase clocks of the block-structured kernels (device_common.hpp): share of each section, summed over blocks and launches of this call
        {
            static const char* const names[8] = {"bi2_emit", "levelB", "uni_onepass", "pospart", "chain_emit", "hot_write", "k6", "k7"};
            unsigned long long h[8][12], z[8][12];
            memset(z, 0, sizeof z);
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(colibri::kprof), sizeof h) == hipSuccess) {
                for (int k = 0; k < 8; ++k) {
                    unsigned long long t = 0;
                    for (int s = 0; s < 12; ++s) t += h[k][s];
                    if (!t) continue;
                    fprintf(stderr, "KPROF %-12s", names[k]);
                    for (int s = 0; s < 12; ++s)
                        if (h[k][s]) fprintf(stderr, " s%d=%.1f%%", s, 100.0 * (double)h[k][s] / (double)t);
                    fprintf(stderr, " total=%.1fM\n", (double)t / 1e6);
                }
                (void)hipMemcpyToSymbol(HIP_SYMBOL(colibri::kprof), z, sizeof z);
            }
        }
#endif
        // a result buffer ran out (a corpus that keeps unusually many patterns per position: duplicated text): more room, again
        // (one result per position and order — with skipgrams one per gap mask as well: up to ~6 x 10^5 masks for a window of 31 tokens)
        const uint64_t per_window = (opt_in->doskipgrams || opt_in->doskipgrams_exhaustive) ? 700000ull : 1ull;
        const uint64_t bound = (uint64_t)std::max(1, std::min<int>(opt_in->maxlength, COLIBRI_MAX_ORDER - 1)) * ((uint64_t)c->npos + 1) * per_window;
        if (rc != COLIBRI_ERR_OVERFLOW || !c->hstate.overflow || c->res_cap_used >= std::min<uint64_t>(0x7FFFFFF0ull, bound)) {
            c->res_scale = 1;
            return report(rc);
        }
        note_retry(c, COLIBRI_FALLBACK_RESULTS);
        c->res_scale *= 4;
    }
}
static int colibri_train_once(colibri_ctx* c, const colibri_options* opt_in, colibri_stats* stats_out) {
    if (!c || !opt_in) return COLIBRI_ERR_ARG;
    if (!c->have_corpus) return fail(c, COLIBRI_ERR_STATE, "no corpus uploaded");
    colibri_options o = *opt_in;
    int             rc;
    if ((rc = check_options(c, o))) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    c->opt     = o;
    c->trained = false;
    c->ks.ran  = false;
    c->profile = o.profile;
    collect_events(c);
    std::fill(std::begin(c->k_ms), std::end(c->k_ms), 0.0);
    std::fill(std::begin(c->k_launches), std::end(c->k_launches), 0);
    c->segments.clear();
    c->npairs = 0;
    c->b2.pairs_direct = false;
    c->b2.pairs2_direct = false;
    if (c->b2.compact_pending) {  // (a run that ended early left a copy on the second stream)
        (void)hipStreamSynchronize(c->b2.aux);
        c->b2.compact_pending = false;
    }
    if (o.dopatternperline) return train_pattern_list(c, o, stats_out);

    const uint32_t npos   = c->npos;
    const bool     constrained = c->cs.n != 0 && !c->cs.continuation && !c->cs.filter;  // train(..., constrainbymodel): one membership-filtered pass per length, no look-back (constrained.hpp)
    const bool     filtered    = c->cs.n != 0 && c->cs.filter;         // train(..., filter): only the windows that match the set are counted, without look-back
    const bool     continued   = c->cs.n != 0 && c->cs.continuation;   // train(..., continued = true): the set is the model the run starts from
    const int      backoff = (o.maxbackofflength >= 1 && o.maxbackofflength + 1 < std::min<int>(o.maxlength, COLIBRI_MAX_ORDER - 1)) ? o.maxbackofflength : 0;  // orders above backoff + 1 differ
    const bool     synced = o.indexed || o.doskipgrams || o.doskipgrams_exhaustive || constrained || backoff || continued || filtered;  // these modes keep every order's ids and talk to the host per order
    // order 1 counted per class id when the encoding is canonical (class id <-> token bytes is then a bijection) and the class
    // space is small enough for a dense array; table_mode 1 / 2 force the generic table / radix implementations (tests)
    const bool uni_direct = !synced && o.table_mode == 0 && !(c->flags & kFlagNonCanonical) && c->maxclass < (1u << 28);
    if (uni_direct && ((rc = dev_alloc(c, c->cnt1, (size_t)c->maxclass + 2)) || (rc = dev_alloc(c, c->rep1, (size_t)c->maxclass + 2)))) return rc;
    const uint32_t uni_shift = (uni_direct && !synced) ? uni_range_shift(c) : 0u;  // 0: more than 4 M classes, the atomics kernel stays
    if (uni_shift && (rc = uni_alloc(c))) return rc;
    // second-generation order 2 (bigram2.hpp): class-keyed, positions in up to 30 bits; a run that it could not hold (c->b2.disabled, set below) repeats on
    // the first-generation kernels
    const bool bi2_ok = uni_direct && !synced && uni_shift != 0 && c->maxclass < (1u << 21) && o.maxlength >= 2 && bigram2_fits(c, npos) && !c->b2.disabled;
    // radix-partition + LDS count for the plain n-gram path: every final bin must fit its LDS table. Up to ~128 M tokens per device one pass per order does;
    // beyond, an order is counted in passes over slices of its keys (bigram2_order / binned_order: `big`) — which needs the class-keyed orders 2 and 3;
    // otherwise (or on request) the global open-addressed table
    const bool big = c->ntokens > big_corpus_tokens();
    bool binned = !synced && (o.table_mode == 2 || (o.table_mode == 0 && (!big || bi2_ok)));
    // three class ids in one key: order 3 is keyed by classes, order 2 leaves survivor bytes instead of ids (KeyTrigramCls)
    const bool tri_cls = binned && uni_direct && !synced && c->maxclass < (1u << 21) && o.maxlength >= 3;
    const bool bi_cls = tri_cls && uni_shift != 0;  // ... and order 2 is keyed by classes + the order-1 survivor bitmap: no per-position order-1 ids at all
    const bool bi2 = binned && bi2_ok;
    // orders >= 3 on the same engine (chain.hpp): one pass per order (corpora a single pass holds), the plain run
    const bool chain = bi2 && !big && bigram2_plan(c, npos).sbits == 0 && chain_fits_plain(npos) && o.maxlength >= 3 && !c->b2.chain_disabled && !getenv("COLIBRI_NO_CHAIN");
    if (tri_cls && !bi2 && ((rc = dev_alloc(c, c->flags_at, (size_t)npos + 1)) || (rc = dev_alloc(c, c->flag2, (size_t)npos + 4)))) return rc;
    // ---- HBM layout (sized once; nothing is allocated inside the unsynced order loop) ----------------
    // table: an order admits at most `npos` windows -> 1.5x slots; results: every survivor has >= 2 occurrences
    const uint64_t table_slots64 = (uint64_t)npos + (npos >> 1) + 2048;
    if (table_slots64 >= 0x7FFFFFFFull) return fail(c, COLIBRI_ERR_CORPUS, "corpus shard too large for one device table");
    TrainPlan pl{};
    pl.npos        = npos;
    pl.table_slots = (uint32_t)table_slots64;
    // results: every survivor has >= MINTOKENS occurrences; at MINTOKENS = 1 every window may be its own pattern (exhaustion is reported, never silent)
    uint64_t per_pos = o.mintokens < 2 ? (uint64_t)std::min(o.maxlength, 8) : (synced ? 4u : 2u);  // results per corpus position the run can produce
    if (o.mintokens < 2 && (o.doskipgrams || o.doskipgrams_exhaustive))  // threshold 1 keeps every masked form of every window as well
        for (int n = 3; n <= std::min(o.maxlength, 13); ++n) per_pos += gap_masks(n, o.maxskips).size();  // (longer orders: the run repeats with more room if they turn up)
    // the usual bound (a survivor has >= MINTOKENS occurrences, and few orders keep many) is not a bound for repetitive corpora — every distinct sentence
    // of L tokens occurring twice keeps L (L + 1) / 2 patterns per pair —: exhaustion is reported by the kernels and colibri_train repeats the run with
    // res_scale x 4 (up to one result per position and order)
    pl.res_cap     = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, (uint64_t)npos * per_pos * c->res_scale + 1024);
    pl.thr         = (uint32_t)o.mintokens;
    c->res_cap_used = pl.res_cap;
    const uint32_t wthr = o.mintokens_unigrams > o.mintokens ? (uint32_t)o.mintokens_unigrams : 0u;  // secondary word threshold (-W), 0 = none
    constexpr uint32_t kCountLdsBytes    = kCountTile * 16u + kCountLSlot * 4u + 64u;  // keyL + cntL + slotL + winL
    constexpr uint32_t kCountBlocksPerCU = (160u * 1024u / kCountLdsBytes) < 8u ? (160u * 1024u / kCountLdsBytes) : 8u;
    pl.cnt_grid = std::max<uint32_t>(1, std::min<uint32_t>(blocks_for(npos, kCountTile), 256u * kCountBlocksPerCU));  // persistent blocks: all resident
    pl.tab_grid = stream_grid(pl.table_slots);
    pl.pos_grid = stream_grid(npos);
    const int maxlength = std::min<int>(o.maxlength, COLIBRI_MAX_ORDER - 1);
    if (o.indexed && (rc = pairs_begin(c, npos))) return rc;
    c->pair_split = o.indexed && c->pair_sb != 0 && c->pair_sb + c->pair_tb <= 32 && !getenv("COLIBRI_WHOLE_PAIRS");

    if (c->ids.size() < 2) c->ids.resize(2);
    if ((rc = dev_alloc(c, c->ids[0], (size_t)npos + 1))) return rc;
    if ((rc = dev_alloc(c, c->ids[1], (size_t)npos + 1))) return rc;
    // the per-pass modes count their n-gram passes of order >= 2 on the radix path too (result indices as ids, see bin_count's dense codes)
    const bool radix_synced = synced && !constrained && o.table_mode == 0 && c->ntokens <= big_corpus_tokens();
    c->profile_class = bi2 ? COLIBRI_K_COUNT2 : (binned || radix_synced) ? COLIBRI_K_BINCOUNT : COLIBRI_K_COUNT;
    // ... and so do the passes of a constrained run: a member window's key is its pattern number in the constraint set, counted in LDS like any other key
    const bool radix_constrained = constrained && o.table_mode == 0 && c->ntokens <= 128ull * 1000 * 1000;
    if (binned || radix_synced || radix_constrained) {
        // recs[0]: 256 fixed-capacity A-bin regions (25 % slack over a uniform split + one scatter tile each); recs[1]: exact
        if ((rc = dev_alloc(c, c->recs[0], ((size_t)npos + (npos >> 2)) / kBins * kBins + (size_t)kBins * kScatTile * 4)) ||
            (rc = dev_alloc(c, c->recs[1], std::max<size_t>((size_t)npos + 1, (size_t)kBins * kScatTile * 4))))  // (floors: a small corpus with one hot bigram still fits its slot of the order-2 records)
            return rc;
        if ((rc = dev_alloc(c, c->rep_of, (size_t)npos + 1)) || (rc = dev_alloc(c, c->ids_at, (size_t)npos + 1)) || (rc = dev_alloc(c, c->binstate, 1))) return rc;
        if ((rc = dev_alloc(c, c->alist[0], (size_t)npos + 1)) || (rc = dev_alloc(c, c->alist[1], (size_t)npos + 1)) || (rc = dev_alloc(c, c->alist_n, 2))) return rc;
        if (bi2 && (rc = bigram2_alloc(c, npos, chain))) return rc;
    }
    // the modes that keep every order's ids run order 2 on the second-generation kernels as well, which then also leave the result index of the bigram at
    // every position (chain_ids_kernel); one pass only, class-keyed, no word threshold (its cut of the order-1 ids comes after their references are emitted)
    const bool bi2_synced = radix_synced && !continued && !filtered && !backoff && wthr == 0 && o.table_mode == 0 && !(c->flags & kFlagNonCanonical) && uni_range_shift(c) != 0 &&
                            c->maxclass < (1u << 21) && o.maxlength >= 2 && bigram2_fits(c, npos) && bigram2_plan(c, npos).sbits == 0 && !c->b2.disabled;
    // ... and, since round 4, their orders >= 3 on the chained engine (chain.hpp) like the plain run's: an order's (position, dense number) pairs become its ids per position
    // (the wide form of the chained orders serves indexed models too; the skipgram passes on the chained engine — skip_pass_chain — have no wide form)
    const bool chain_synced = bi2_synced && (chain_fits(npos) || (chain_fits_plain(npos) && !o.doskipgrams && !o.doskipgrams_exhaustive)) && o.maxlength >= 3 && !c->b2.chain_disabled && !getenv("COLIBRI_NO_CHAIN") && !getenv("COLIBRI_NO_CHAIN_IDS");
    if (bi2_synced && (rc = bigram2_alloc(c, npos, chain_synced))) return rc;
    if (!binned && (rc = dev_alloc(c, c->table, pl.table_slots))) return rc;  // the plain radix run needs no table (a bin overflow re-runs with table_mode = 1)
    if ((rc = dev_alloc(c, c->res_rep, pl.res_cap))) return rc;
    if ((rc = dev_alloc(c, c->res_cnt, pl.res_cap))) return rc;
    if ((rc = dev_alloc(c, c->state, 1))) return rc;
    if (o.doskipgrams || o.doskipgrams_exhaustive) {
        if ((rc = dev_alloc(c, c->scratch[0], (size_t)npos + 1))) return rc;
        if ((rc = dev_alloc(c, c->scratch[1], (size_t)npos + 1))) return rc;
    }
    if (o.doskipgrams && (rc = dev_alloc(c, c->nsrc, pl.table_slots))) return rc;

    // order 1: distinct unigrams <= distinct class ids when the encoding is canonical
    DevState init{};
    uint64_t cap1 = (uint64_t)c->ntokens + (c->ntokens >> 1) + 1024;
    if (!(c->flags & kFlagNonCanonical)) cap1 = std::min<uint64_t>(cap1, 2ull * ((uint64_t)c->maxclass + 1) + 1024);
    init.cap = (uint32_t)std::min<uint64_t>(cap1, pl.table_slots);
    if (c->ntokens == 0) init.done = 1;  // empty corpus: "None found" at n = 1
    c->hstate = init;
    if ((rc = write_state(c))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));

    colibri_stats& s = c->stats;
    std::memset(&s, 0, sizeof s);
    const auto t0 = std::chrono::steady_clock::now();

    if (!synced) {
        // ---------- the headline path: all orders enqueued back to back, no host round trip per order ----------
        int      cur = 0;  // ids[cur] = survivor ids of order n-1; ids[cur^1] receives order n
        uint32_t sbits_next = 0;  // passes (as a power of two) of the next first-generation order: more than one only for corpora beyond ~128 M tokens
        for (int n = 1; n <= maxlength; ++n) {
            uint32_t* id_prev = c->ids[cur].p;
            uint32_t* id_cur  = c->ids[cur ^ 1].p;
            if (n == 1 && uni_direct) {
                // order 1 on the class-indexed count array (kernels.hpp §2b): no hashing, no table, LDS histogram for the Zipf head
                const uint32_t nclasses = c->maxclass + 1;
                if (uni_shift) {
                    // no per-token global atomics: head histogram in LDS, tail partitioned into 256 class ranges and counted per range in LDS
                    if ((rc = uni_count_partitioned(c, uni_shift, c->cnt1.p, nclasses))) return rc;
                } else {
                    HIP_TRY(c, hipMemsetAsync(c->cnt1.p, 0, sizeof(uint32_t) * nclasses, c->stream));
                    Prof p(c, COLIBRI_K_COUNT);
                    hipLaunchKernelGGL(uni_count_kernel, dim3(512), dim3(kBlock), 0, c->stream, c->cls.p, npos, c->cnt1.p, c->rep1.p, c->state.p);
                }
                {
                    Prof p(c, COLIBRI_K_PRUNE);
                    hipLaunchKernelGGL(uni_finish_kernel, dim3(stream_grid(nclasses)), dim3(kBlock), 0, c->stream, c->cnt1.p, uni_shift ? (const uint32_t*)nullptr : c->rep1.p, nclasses,
                                       pl.thr, c->state.p, c->res_rep.p, c->res_cnt.p, pl.res_cap, uni_shift ? reinterpret_cast<uint16_t*>(c->uni_surv.p) : (uint16_t*)nullptr, bi_cls || bi2,
                                       (uint32_t*)nullptr, wthr);
                }
                if (!bi_cls && !bi2) {  // with class-keyed orders 2 and 3 nothing reads order-1 ids per position
                    Prof p(c, COLIBRI_K_RESOLVE);
                    if (uni_shift)
                        hipLaunchKernelGGL(uni_ids_bitmap_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->cls.p, c->uni_surv.p, (nclasses + 31) / 32, id_cur, c->state.p, npos);
                    else
                        hipLaunchKernelGGL(uni_ids_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->cls.p, c->cnt1.p, std::max(pl.thr, wthr), id_cur, c->state.p, npos);
                }
            } else if (binned) {
                // orders 1-2 scan every position (almost all are admissible); from order 3 on only the positions that still
                // carry a survivor id are visited (the active list the previous order's resolve left behind)
                if (n == 1)
                    rc = binned_order(c, pl, KeyUnigram{c->bytes.p, c->tokstart.p}, id_cur, n, false, n < maxlength);
                else if (n == 2 && chain)
                    rc = bigram2_order(c, pl, /*want_list=*/n < maxlength, nullptr, /*chain=*/true);
                else if (n >= 3 && chain)
                    rc = chain_order(c, pl, n, /*want_next=*/n < maxlength);
                else if (n == 2 && bi2)
                    rc = bigram2_split_fits(c, npos) ? bigram2_order_split(c, pl, /*want_list=*/n < maxlength) : bigram2_order(c, pl, /*want_list=*/n < maxlength);
                else if (n == 3 && bi2)  // over the list bigram2 left: every listed window is admissible
                    rc = binned_split_fits(sbits_next) ? binned_order_split(c, pl, KeyTrigramClsListed{c->cls.p}, id_cur, n, n < maxlength, /*prefill_ids=*/true, sbits_next)
                                                       : binned_order(c, pl, KeyTrigramClsListed{c->cls.p}, id_cur, n, true, n < maxlength, false, /*prefill_ids=*/true, sbits_next);
                else if (n == 2 && bi_cls)
                    rc = binned_order(c, pl, KeyBigramCls{c->cls.p, c->uni_surv.p}, id_cur, n, false, true, /*flag_mode=*/true);
                else if (n == 2 && tri_cls)
                    rc = binned_order(c, pl, KeyNgram{id_prev, n}, id_cur, n, false, true, /*flag_mode=*/true);  // order 3 will not read order-2 ids
                else if (n == 3 && tri_cls)
                    rc = binned_order(c, pl, KeyTrigramCls{c->cls.p, c->flag2.p}, id_cur, n, true, n < maxlength);
                else if (n >= 3 && binned_split_fits(sbits_next))
                    rc = binned_order_split(c, pl, KeyNgram{id_prev, n}, id_cur, n, n < maxlength, false, sbits_next);
                else
                    rc = binned_order(c, pl, KeyNgram{id_prev, n}, id_cur, n, n >= 3, n < maxlength, false, false, sbits_next);
                if (rc) return rc;
            } else {
                launch_clear(c, pl);
                if (n == 1)
                    launch_count(c, pl, KeyUnigram{c->bytes.p, c->tokstart.p}, id_cur, 3, COLIBRI_K_COUNT);
                else
                    launch_count(c, pl, KeyNgram{id_prev, n}, id_cur, 3, COLIBRI_K_COUNT);
                launch_prune(c, pl, pl.thr, nullptr, 0);
                launch_resolve(c, pl, id_cur);
                if (n == 1 && wthr > pl.thr) hipLaunchKernelGGL(ids_min_count_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, id_cur, c->res_cnt.p, wthr, npos);
            }
            hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p, n, pl.table_slots);
            cur ^= 1;
            if (big && binned && n >= 2 && n < maxlength) {  // a big corpus: how many passes the next order needs follows from how many positions still carry a survivor
                uint32_t valid = 0;
                // (after the second-generation order 2 the next order's records are exactly the entries of the list it left: windows whose two bigrams both survived)
                const uint32_t* src = (n == 2 && bi2) ? (const uint32_t*)(c->alist_n.p + 1) : (const uint32_t*)&c->state.p->s_valid[n];
                HIP_TRY(c, hipMemcpyAsync(&valid, src, sizeof valid, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                sbits_next = slice_bits(valid);
            }
            // peek at the termination flag only every 8 orders (MAXLENGTH defaults to 100)
            if ((n % 8) == 0 && n < maxlength) {
                uint32_t done = 0;
                HIP_TRY(c, hipMemcpyAsync(&done, &c->state.p->done, sizeof done, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                if (done) break;
            }
        }
        if ((rc = chain_compact_join(c)) || (rc = read_state(c))) return rc;
        if (binned && c->hstate.radix_overflow == 8 && !c->split_exact) {  // a run of the direct split of a sliced order outgrew its room (keys far from uniform): the exact split
            c->split_exact = true;  // (for this corpus: reset by the next upload)
            note_retry(c, COLIBRI_FALLBACK_SPLIT);
            return colibri_train_once(c, &o, stats_out);
        }
        if (binned && c->hstate.radix_overflow == 16) {  // an order >= 3 did not fit the second-generation engine (key bits, a region, a bin): those orders on the first-generation kernels
            if (getenv("COLIBRI_DEBUG_OVERFLOW")) {  // (which part: Bi2State.overflow 1 a record region / the key bits, 2 a final bin's table, 3 a position list)
                uint32_t w2 = 0, w3 = 0;
                if (c->b2.state2.p) (void)hipMemcpy(&w2, &c->b2.state2.p->overflow, sizeof w2, hipMemcpyDeviceToHost);
                if (c->b2.state3.p) (void)hipMemcpy(&w3, &c->b2.state3.p->overflow, sizeof w3, hipMemcpyDeviceToHost);
                fprintf(stderr, "colibri: a chained order gave up (Bi2State.overflow of the odd / even orders now: %u / %u; first: overflow %u, key bits %u, position bits %u, bshift %u); "
                                "repeating with orders >= 3 on the first-generation kernels\n", w2, w3, c->hstate.pad[0] & 255u, (c->hstate.pad[0] >> 8) & 255u, (c->hstate.pad[0] >> 16) & 255u,
                        c->hstate.pad[0] >> 24);
            }
            c->b2.chain_disabled = true;
            note_retry(c, COLIBRI_FALLBACK_CHAIN);
            const int rc2        = colibri_train_once(c, &o, stats_out);
            c->b2.chain_disabled = false;
            return rc2;
        }
        if (binned && c->hstate.radix_overflow == 4) {  // the second-generation order 2 could not hold this corpus (a hot bigram outside the dense head): first-generation kernels
            note_retry(c, COLIBRI_FALLBACK_ORDER2);
            if (getenv("COLIBRI_DEBUG_OVERFLOW")) {  // (which part of it gave up: 1 a record region, 2 a final bin's table, 3 a position list; 0: the position buckets or a split)
                uint32_t why = 0;
                (void)hipMemcpy(&why, &c->b2.state.p->overflow, sizeof why, hipMemcpyDeviceToHost);
                fprintf(stderr, "colibri: second-generation order 2 gave up (Bi2State.overflow = %u); repeating %s\n", why,
                        retry_with_small_passes(npos) ? "with round 3's pass size" : "on the first-generation kernels");
            }
            if (retry_with_small_passes(npos)) {  // (a corpus beyond the old pass size: its bins get the old load back before anything slower is tried)
                tl_small_passes = true;
                const int rc2   = colibri_train_once(c, &o, stats_out);
                tl_small_passes = false;
                return rc2;
            }
            c->b2.disabled = true;
            const int rc2  = colibri_train_once(c, &o, stats_out);
            c->b2.disabled = false;
            return rc2;
        }
        if (binned && c->hstate.radix_overflow) {  // a final bin outgrew its LDS table (hash skew): run again on the global table — loud, exact, rare
            if (o.table_mode == 2)
                return fail(c, COLIBRI_ERR_OVERFLOW, "the radix path overflowed (%s; region %llu records, %u positions; table_mode = 2 forbids the global-table rerun)",
                            c->hstate.radix_overflow == 1 ? "an A-bin region" : c->hstate.radix_overflow == 2 ? "a final bin outgrew its LDS table" : "survivor id range",
                            (unsigned long long)(c->recs[0].n / kASlots), npos);
            colibri_options again = o;
            again.table_mode      = 1;
            note_retry(c, (int)c->hstate.radix_overflow <= 3 ? (int)c->hstate.radix_overflow : COLIBRI_FALLBACK_BIN);
            return colibri_train_once(c, &again, stats_out);
        }
        c->last_mode   = binned ? 2 : 1;
        c->last_passes = bi2 ? (1 << bigram2_plan(c, npos).sbits) : 1;
        c->run_path    = !binned ? COLIBRI_PATH_TABLE
                                 : (COLIBRI_PATH_RADIX | (bi2 ? COLIBRI_PATH_BI2 : 0) | (chain ? COLIBRI_PATH_CHAIN : 0) | ((chain && chain_wide(npos)) ? COLIBRI_PATH_WIDE : 0) |
                                    ((c->last_passes > 1 || big) ? COLIBRI_PATH_SLICED : 0));
        s.maxn = (int32_t)c->hstate.maxn;
        for (int n = 1; n < COLIBRI_MAX_ORDER; ++n) {
            s.found[n]    = c->hstate.s_found[n];
            s.kept[n]     = c->hstate.s_kept[n];
            s.admitted[n] = c->hstate.s_admitted[n];
            if (n <= s.maxn && s.kept[n])
                c->segments.push_back({c->hstate.res_off[n], c->hstate.res_off[n + 1] - c->hstate.res_off[n], n, (n == 1 && uni_shift) ? kMaskFromClass : 0u});
        }
    } else {
        // ---------- skipgram / indexed modes: one host round trip per pass (sizes, lazily grown per-order id arrays) ----------
        if ((int)c->ids.size() < maxlength + 2) c->ids.resize(maxlength + 2);
        c->ids_built.assign(c->ids.size(), 0);
        c->ids1_is_cls        = false;
        bool       list_valid = false;  // the active list of the previous radix pass exists
        const bool uni_synced = !constrained && o.table_mode == 0 && !(c->flags & kFlagNonCanonical) && c->maxclass < (1u << 28);
        std::vector<uint32_t> valid_n(maxlength + 2, 0), adm_n(maxlength + 2, 0), ngram_first(maxlength + 2, 0), ngram_kept(maxlength + 2, 0);
        uint32_t              res_total = 0;
        const uint32_t        thr_skip  = o.minskiptypes > 1 ? (uint32_t)o.mintokens_skipgrams : pl.thr;  // base pruneskipgrams is a no-op when MINSKIPTYPES <= 1 (patternmodel.h:2167-2186)
        if (constrained || continued || filtered) {
            if (!c->cs.rem_valid) {
                if ((rc = dev_alloc(c, c->cs.rem, (size_t)npos + 1))) return rc;
                hipLaunchKernelGGL(sentence_rem_kernel, dim3(stream_grid(npos)), dim3(kBlock), 0, c->stream, c->delimpos.p, c->ndelim, npos, c->cs.rem.p);
                c->cs.rem_valid = true;
            }
            // a length's distinct keys are patterns of J: the table never needs more slots than that
            c->hstate.cap = (uint32_t)std::min<uint64_t>(pl.table_slots, (uint64_t)c->cs.n + (c->cs.n >> 1) + 1024u);
            if ((rc = write_state(c))) return rc;
            if ((rc = dev_alloc(c, c->cs.memb, (size_t)(kProbeLengths + 1) * ((size_t)npos + 1)))) return rc;  // + one array: who was alive after the previous block of lengths
        }
        int probed_from = 0, probed_to = -1;  // window lengths whose membership arrays are current
        // back-off passes (MAXBACKOFFLENGTH): per-position scratch of the byte-grouping kernels
        DevBuf<uint32_t>           bo_run, bo_flen, bo_slot, bo_isrep, bo_keep;
        DevBuf<unsigned long long> bo_off, bo_unit, bo_rank;
        DevBuf<FSlot>              bo_table;
        DevBuf<FlexInfo>           bo_info;
        DevBuf<uint8_t>            flt_cont[2];            // filtered runs: "the window at i contains a filter n-gram", this length / the one below
        DevBuf<uint32_t>           flt_exists, flt_memb;   // ... gate and result of the probes for the filter's skipgram shapes
        auto                       bo_cleanup = [&]() {
            dev_free(flt_cont[0]); dev_free(flt_cont[1]); dev_free(flt_exists); dev_free(flt_memb);
            dev_free(bo_run); dev_free(bo_flen); dev_free(bo_slot); dev_free(bo_isrep); dev_free(bo_keep); dev_free(bo_off); dev_free(bo_unit); dev_free(bo_rank); dev_free(bo_table); dev_free(bo_info);
        };
        struct BoGuard {
            decltype(bo_cleanup)& f;
            ~BoGuard() { f(); }
        } bo_guard{bo_cleanup};
        bool bo_runs_valid = false;
        // ---- the benchmark's id-keeping modes (indexed model, exhaustive skipgrams; class-keyed second-generation order 2, no rarer option) with the order loop
        // ENQUEUED, as the plain mode's is: every per-order quantity lives in DevState (idm_ngram_end / idm_order_end / skip_pass_end), the host looks once, after
        // the last order (north star: "no host round-trip per iteration"; reference loop: include/patternmodel.h:981-1270, skipgram call site :1163-1171)
        bool enq = bi2_synced && !getenv("COLIBRI_SYNCED_LOOP");
        if (enq && o.doskipgrams_exhaustive) {  // the skipgram passes of ALL orders share one device log: a run that may reach orders with thousands of masks keeps the per-order loop
            size_t total = 0;
            for (int n = 3; n <= maxlength && total <= kSegLogCap; ++n) total += n > 16 ? (size_t)kSegLogCap + 1 : gap_masks(n, o.maxskips).size();
            enq = total <= kSegLogCap;
        }
        if (enq) {
            c->ids1_is_cls = !o.indexed;
            // an indexed model on the chained engine: the pairs of the orders >= 3 come straight from the orders' position lists (chain_pairs_kernel: a rank per listed
            // position instead of a fill and two sweeps over npos ids, which cost those sparse orders more than their counting). Order 2 keeps the sweeps: with 5 x 10^7 pairs
            // the ranks' gathers cost what the sweeps do (measured: 2.2 ms with three gathers per pair, ~1 ms at best). Without skipgram passes nobody reads the ids of the
            // orders >= 3 then, and they are not built
            c->b2.pairs_direct = chain_synced && o.indexed && c->pair_sb != 0 && !getenv("COLIBRI_NO_DIRECT_PAIRS");
            c->b2.pairs2_direct = c->b2.pairs_direct && chain_ids_full_applies(bigram2_plan(c, npos)) && !getenv("COLIBRI_NO_DIRECT_PAIRS2");  // order 2's too (round 5)
            // Who reads ids[n] (n >= 2) of a chained run: trainskipgrams' lists and keys (every order), emit_pairs (order 2; the higher orders unless their pairs come
            // from the lists), the exhaustive passes' part ids (parts have at most maxlength - 2 tokens; their gate is the list itself: chain_alist_kernel's windows ARE
            // the admitted ones). Nobody else: an order's fill + scatter (0.06 + 0.03..0.6 ms) is skipped where nobody does
            auto want_ids = [&](int n) {
                if (!chain_synced || o.doskipgrams || getenv("COLIBRI_ALL_IDS")) return true;
                if (o.indexed && ((n == 2 && !c->b2.pairs2_direct) || !c->b2.pairs_direct)) return true;
                return o.doskipgrams_exhaustive && n <= maxlength - 2;
            };
            // ... and order 1's come from the class ids (survivor bit, result index per class): nobody reads ids[1] then (order 2 is keyed by classes)
            const bool uni_pairs_direct = o.indexed && !o.doskipgrams && !o.doskipgrams_exhaustive && maxlength >= 2 && !getenv("COLIBRI_NO_DIRECT_PAIRS");
            if (c->b2.pairs_direct && ((rc = dev_alloc(c, c->b2.wpre, (size_t)npos / 32 + 64)) || (rc = dev_alloc(c, c->b2.btot, kBi2Buckets)))) return rc;
            const uint32_t nclasses = c->maxclass + 1;
            if ((rc = dev_alloc(c, c->cnt1, (size_t)nclasses + 1)) || (rc = dev_alloc(c, c->rep1, (size_t)nclasses + 1)) || (rc = dev_alloc(c, c->uni_resid, (size_t)nclasses + 1)) ||
                (rc = uni_alloc(c)))
                return rc;
            for (int n = 1; n <= maxlength; ++n)
                if ((rc = dev_alloc(c, c->ids[n], (size_t)npos + 1))) return rc;
            if (o.doskipgrams_exhaustive) {
                if ((rc = dev_alloc(c, c->seglog, 4 + 5 * (size_t)kSegLogCap))) return rc;
                HIP_TRY(c, hipMemsetAsync(c->seglog.p, 0, sizeof(uint32_t) * 4, c->stream));
            }
            size_t nlogged = 0;
            int    nmax = 0;  // orders enqueued
            for (int n = 1; n <= maxlength; ++n) {
                nmax = n;
                if (n == 1) {
                    if ((rc = uni_count_partitioned(c, uni_range_shift(c), c->cnt1.p, nclasses))) return rc;
                    {
                        Prof p(c, COLIBRI_K_PRUNE);
                        hipLaunchKernelGGL(uni_finish_kernel, dim3(stream_grid(nclasses)), dim3(kBlock), 0, c->stream, c->cnt1.p, (const uint32_t*)nullptr, nclasses, pl.thr, c->state.p,
                                           c->res_rep.p, c->res_cnt.p, pl.res_cap, reinterpret_cast<uint16_t*>(c->uni_surv.p), /*count_valid=*/!o.indexed, c->uni_resid.p);
                    }
                    if (!c->ids1_is_cls && !uni_pairs_direct) {
                        Prof p(c, COLIBRI_K_RESOLVE);
                        hipLaunchKernelGGL(uni_resid_ids_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->cls.p, c->uni_resid.p, c->ids[1].p, c->state.p, npos);
                        c->ids_built[1] = 1;
                    }
                } else if (n == 2) {
                    if ((rc = bigram2_order(c, pl, /*want_list=*/true, want_ids(2) ? c->ids[2].p : (uint32_t*)nullptr, /*chain=*/chain_synced))) return rc;
                    c->ids_built[2] = want_ids(2);
                } else if (chain_synced) {
                    if (o.doskipgrams_exhaustive) {  // the windows this order admits, for its skipgram passes: from the bitmap of order n - 1, which this order's own replaces
                        HIP_TRY(c, hipMemsetAsync(c->alist_n.p + (n & 1), 0, sizeof(uint32_t), c->stream));
                        hipLaunchKernelGGL(chain_alist_kernel, dim3(1024), dim3(kBlock), 0, c->stream, (const uint32_t*)c->b2.bitmap.p, npos, (const DevState*)c->state.p, c->alist[n & 1].p,
                                           c->alist_n.p + (n & 1));
                    }
                    if ((rc = chain_order(c, pl, n, /*want_next=*/true, want_ids(n) ? c->ids[n].p : (uint32_t*)nullptr))) return rc;
                    c->ids_built[(size_t)n] = want_ids(n);
                    if (o.doskipgrams_exhaustive && (rc = chain_compact_join(c))) return rc;  // (the skipgram passes count in the buffers the order's survivors are being copied from)
                } else {
                    if ((rc = binned_count_stage(c, pl, KeyNgram{c->ids[n - 1].p, n}, n, true, pl.thr, false, true, false, /*dense_code=*/true))) return rc;
                    const BinnedIO io = binned_planes(c, pl, false);
                    {
                        Prof p(c, COLIBRI_K_PRUNE);
                        hipLaunchKernelGGL(bin_kept_scan_kernel, dim3(kBins), dim3(kBlock), 0, c->stream, c->state.p, c->binstate.p, pl.res_cap);
                        hipLaunchKernelGGL(compact_bins_kernel, dim3(1024), dim3(kBlock), 0, c->stream, io.sp_rep, io.sp_cnt, c->state.p, c->binstate.p, c->res_rep.p, c->res_cnt.p, pl.res_cap,
                                           (const uint32_t*)c->alist[n & 1].p);
                        hipLaunchKernelGGL(bin_advance_prepare_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p, c->binstate.p);
                    }
                    if ((rc = binned_resolve_stage(c, pl, c->ids[n].p, n, true, true, nullptr, 0u, /*prefill_ids=*/true, /*decode=*/true, kDecodeBaseOnDevice))) return rc;
                    c->ids_built[(size_t)n] = 1;
                }
                if (n == 1 && uni_pairs_direct) {  // (before the order's figures are closed: the emission counts the positions with a surviving unigram, as uni_resid_ids_kernel does)
                    if ((rc = emit_pairs(c, pl, c->cls.p, false, c->uni_surv.p, c->uni_resid.p, /*hot1=*/true))) return rc;
                }
                hipLaunchKernelGGL(idm_ngram_end_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p, n);
                if (o.indexed && !(c->b2.pairs_direct && (n >= 3 || (n == 2 && c->b2.pairs2_direct))) && !(n == 1 && uni_pairs_direct)) {
                    const uint32_t* const idn = built_ids(c, n);
                    if (!idn) return COLIBRI_ERR_STATE;
                    if ((rc = emit_pairs(c, pl, idn, false, nullptr, nullptr, /*hot1=*/n == 1 && !c->ids1_is_cls))) return rc;  // (ids[1] holds uni_resid's result indices then)
                }
                if (o.doskipgrams_exhaustive && n >= 3) {  // patternmodel.h:1163-1171 -> computeskipgrams :1370-1527, for every admissible window: the order's own active list
                    if (n > kMaxSkipgramTokens) return fail(c, COLIBRI_ERR_UNSUPPORTED, "skipgrams of patterns longer than 31 tokens do not exist (a gap mask has 32 bits; set MAXLENGTH)");
                    c->skl   = c->alist[n & 1].p;
                    c->skl_n = c->alist_n.p + (n & 1);
                    const std::vector<uint32_t> masks = gap_masks(n, o.maxskips);
                    for (uint32_t mask : masks) {
                        uint32_t f = 0, k = 0;
                        const uint32_t* const gate = chain_synced && !getenv("COLIBRI_ALL_IDS") ? (const uint32_t*)nullptr : built_ids(c, n - 1);
                        if (!gate && !(chain_synced && !getenv("COLIBRI_ALL_IDS"))) return COLIBRI_ERR_STATE;
                        const std::vector<std::pair<int, int>> parts = mask_parts(mask, n);
                        // two parts, at least one of them a single token (class id, ~20 bits; two result indices of ~24 bits do not fit a record beside the position)
                        const bool one_token_part = parts.size() == 2 && c->ids1_is_cls && (parts[0].second == 1 || parts[1].second == 1);
                        if (chain_synced && one_token_part && !o.indexed && !getenv("COLIBRI_ALL_IDS") && !getenv("COLIBRI_NO_SKIP_CHAIN")) {  // ... on the chained engine
                            for (const auto& part : parts)
                                if (!(part.second == 1 && c->ids1_is_cls) && !built_ids(c, part.second)) return COLIBRI_ERR_STATE;
                            if ((rc = skip_pass_chain(c, pl, n, mask, part_ids(c, parts[0].second), (uint32_t)parts[0].first, parts[0].second == 1 && c->ids1_is_cls,
                                                      part_ids(c, parts[1].second), (uint32_t)parts[1].first, parts[1].second == 1 && c->ids1_is_cls, thr_skip, c->seglog.p)))
                                return rc;
                            continue;
                        }
                        if ((rc = skipgram_pass_radix(c, pl, n, mask, gate, gate, thr_skip, 0u, &f, &k, nullptr, 0, 0, 0, c->seglog.p))) return rc;
                    }
                    nlogged += masks.size();
                }
                hipLaunchKernelGGL(idm_order_end_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p, n);
                if ((n % 8) == 0 && n < maxlength) {  // peek at the termination flag only every 8 orders (MAXLENGTH defaults to 100)
                    uint32_t done = 0;
                    HIP_TRY(c, hipMemcpyAsync(&done, &c->state.p->done, sizeof done, hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    if (done) break;
                }
            }
            std::vector<uint32_t> log(4 + 5 * nlogged, 0);
            if (nlogged) HIP_TRY(c, hipMemcpyAsync(log.data(), c->seglog.p, sizeof(uint32_t) * log.size(), hipMemcpyDeviceToHost, c->stream));
            if ((rc = chain_compact_join(c)) || (rc = read_state(c))) return rc;
            if (c->hstate.radix_overflow == 16) {  // an order >= 3 did not fit the chained engine (key bits, a region, a bin): the run again with round 3's orders >= 3
                note_retry(c, COLIBRI_FALLBACK_CHAIN);
                c->b2.chain_disabled = true;
                const int rc2        = colibri_train_once(c, &o, stats_out);
                c->b2.chain_disabled = false;
                return rc2;
            }
            if (c->hstate.radix_overflow == 4) {  // the second-generation order 2 could not hold this corpus: again, on the first-generation kernels
                note_retry(c, COLIBRI_FALLBACK_ORDER2);
                if (retry_with_small_passes(npos)) {  // (a corpus beyond the old pass size: its bins get the old load back before anything slower is tried)
                    tl_small_passes = true;
                    const int rc2   = colibri_train_once(c, &o, stats_out);
                    tl_small_passes = false;
                    return rc2;
                }
                c->b2.disabled = true;
                const int rc2  = colibri_train_once(c, &o, stats_out);
                c->b2.disabled = false;
                return rc2;
            }
            if (c->hstate.radix_overflow) {  // a bin outgrew its LDS table: the whole run again on the table path (loud, exact, rare)
                colibri_options again = o;
                again.table_mode      = 1;
                note_retry(c, (c->hstate.radix_overflow >= 1 && c->hstate.radix_overflow <= 3) ? (int)c->hstate.radix_overflow : COLIBRI_FALLBACK_BIN);
                return colibri_train_once(c, &again, stats_out);
            }
            s.maxn = (int32_t)c->hstate.maxn;
            for (int n = 1; n <= std::min<int>(s.maxn, maxlength); ++n) {
                s.found[n]     = c->hstate.s_found[n];
                s.kept[n]      = c->hstate.s_kept[n];
                s.admitted[n]  = c->hstate.s_admitted[n];
                adm_n[n]       = c->hstate.s_admitted[n];
                valid_n[n]     = c->hstate.s_valid[n];
                ngram_first[n] = c->hstate.res_off[n];
                ngram_kept[n]  = c->hstate.s_kept[n];
                if (s.kept[n]) c->segments.push_back({c->hstate.res_off[n], c->hstate.s_kept[n], n, n == 1 ? kMaskFromClass : 0u});
                for (size_t e = 0; e < nlogged && e < log[0]; ++e) {  // the order's skipgram passes, in the order they ran: their result ranges follow the n-grams'
                    const uint32_t* x = log.data() + 4 + 5 * e;
                    if ((int)x[2] != n) continue;
                    s.found[n] += x[4];
                    s.kept[n] += x[1];
                    if (x[1]) c->segments.push_back({x[0], x[1], (int)x[2], x[3]});
                }
            }
            if (s.maxn < maxlength && s.maxn + 1 < COLIBRI_MAX_ORDER) s.admitted[s.maxn + 1] = c->hstate.s_admitted[s.maxn + 1];  // (the order that found nothing still counted its windows: 0)
            res_total = c->hstate.res_total;
            (void)nmax;
            if (c->ntokens) c->hstate.done = 0;  // (the passes that follow — skipgrams of an indexed model — write the host's copy of the state back: the loop's end is not theirs)
        }
        for (int n = constrained ? std::max(1, o.minlength) : 1; n <= maxlength && !c->hstate.done && !enq; ++n) {
            if ((rc = dev_alloc(c, c->ids[n], (size_t)npos + 1))) return rc;
            c->ids_built[(size_t)n] = 1;  // (the per-order loop builds every order's ids)
            if (continued && c->cs.has_order(n)) {
                // "Skipping n-grams, already in model" (patternmodel.h:983-995): nothing is counted; the windows that ARE patterns of the loaded model get
                // the pattern's number as their survivor id, which is all the look-back of the next order asks for (:1139-1152: this->has(subngram))
                Prof p(c, COLIBRI_K_COUNT);
                hipLaunchKernelGGL(constraint_probe_kernel<false>, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->bytes.p, c->tokstart.p, c->cs.rem.p, c->cs.table.p, c->cs.cap,
                                   c->cs.bytes.p, c->cs.off.p, npos, n, 1, c->ids[n].p, (size_t)npos + 1, (const uint32_t*)nullptr);
                list_valid = false;
                valid_n[n] = 1;  // (not counted: the order above simply finds no candidate when there is none)
                continue;
            }
            bool       listed_order = false;  // this order walked the active list (alist[n & 1])
            const bool radix_pass = (radix_constrained || (radix_synced && n >= 2)) && !(backoff && n > backoff + 1);
            if (constrained && n > probed_to) {  // which pattern of the constraint set is the window at each position, for the next lengths
                probed_from = n;
                probed_to   = std::min(maxlength, n + kProbeLengths - 1);
                Prof p(c, COLIBRI_K_COUNT);
                // a prefix-closed set probed from length 1 on stops at the first miss; the last length of a block of eight tells the next block who is still alive
                const bool early = c->cs.closed && (std::max(1, o.minlength) == 1);
                if (early && probed_from > 1)
                    HIP_TRY(c, hipMemcpyAsync(c->cs.memb.p + (size_t)kProbeLengths * ((size_t)npos + 1), c->cs.memb.p + (size_t)(kProbeLengths - 1) * ((size_t)npos + 1),
                                              sizeof(uint32_t) * (size_t)npos, hipMemcpyDeviceToDevice, c->stream));
                if (early)
                    hipLaunchKernelGGL(constraint_probe_kernel<true>, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->bytes.p, c->tokstart.p, c->cs.rem.p, c->cs.table.p, c->cs.cap,
                                       c->cs.bytes.p, c->cs.off.p, npos, probed_from, probed_to - probed_from + 1, c->cs.memb.p, (size_t)npos + 1,
                                       probed_from > 1 ? (const uint32_t*)(c->cs.memb.p + (size_t)kProbeLengths * ((size_t)npos + 1)) : (const uint32_t*)nullptr);
                else
                    hipLaunchKernelGGL(constraint_probe_kernel<false>, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->bytes.p, c->tokstart.p, c->cs.rem.p, c->cs.table.p, c->cs.cap,
                                       c->cs.bytes.p, c->cs.off.p, npos, probed_from, probed_to - probed_from + 1, c->cs.memb.p, (size_t)npos + 1, (const uint32_t*)nullptr);
            }
            const KeyMember member{c->cs.memb.p + (size_t)(constrained ? n - probed_from : 0) * ((size_t)npos + 1)};
            if (!(n == 1 && uni_synced) && !radix_pass) launch_clear(c, pl);  // only the table passes need the table cleared
            const bool backoff_pass = (backoff && n > backoff + 1) || filtered;
            if (backoff_pass) {
                // back-off: every window whose sub-patterns of `backoff` tokens all survived is a candidate; filtered run: every window that matches the filter
                // is (no look-back, patternmodel.h:1106-1137). Either way a candidate's identity is its bytes (patternlist.hpp)
                if (!bo_runs_valid) {
                    if ((rc = dev_alloc(c, bo_run, (size_t)npos + 1)) || (rc = dev_alloc(c, bo_flen, (size_t)npos + 1)) || (rc = dev_alloc(c, bo_slot, (size_t)npos + 1)) ||
                        (rc = dev_alloc(c, bo_isrep, (size_t)npos + 1)) || (rc = dev_alloc(c, bo_keep, (size_t)npos + 1)) || (rc = dev_alloc(c, bo_off, (size_t)npos + 1)) ||
                        (rc = dev_alloc(c, bo_unit, (size_t)npos + 1)) || (rc = dev_alloc(c, bo_rank, (size_t)npos + 1)) || (rc = dev_alloc(c, bo_info, 1)))
                        return rc;
                    if (!filtered) {
                        Prof p(c, COLIBRI_K_COUNT);
                        hipLaunchKernelGGL(backoff_runs_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->ids[backoff].p, npos, bo_run.p);
                    }
                    bo_runs_valid = true;
                }
                if (filtered) {  // bo_run[i] = 1 where the window of n tokens at i matches the filter
                    if ((rc = dev_alloc(c, flt_cont[0], (size_t)npos + 1)) || (rc = dev_alloc(c, flt_cont[1], (size_t)npos + 1)) || (rc = dev_alloc(c, flt_exists, (size_t)npos + 1)) ||
                        (rc = dev_alloc(c, flt_memb, (size_t)npos + 1)))
                        return rc;
                    Prof p(c, COLIBRI_K_COUNT);
                    const bool have_n = c->cs.has_order(n);
                    if (have_n)
                        hipLaunchKernelGGL(constraint_probe_kernel<false>, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->bytes.p, c->tokstart.p, c->cs.rem.p, c->cs.table.p, c->cs.cap,
                                           c->cs.bytes.p, c->cs.off.p, npos, n, 1, c->cs.memb.p, (size_t)npos + 1, (const uint32_t*)nullptr);
                    hipLaunchKernelGGL(filter_contains_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->cs.rem.p, have_n ? (const uint32_t*)c->cs.memb.p : (const uint32_t*)nullptr,
                                       n > 1 ? (const uint8_t*)flt_cont[(n - 1) & 1].p : (const uint8_t*)nullptr, npos, (uint32_t)n, flt_cont[n & 1].p, bo_run.p, flt_exists.p);
                    for (const auto& shape : c->cs.shapes) {
                        if (shape.first != n) continue;
                        hipLaunchKernelGGL(constraint_probe_masked_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->bytes.p, c->tokstart.p, flt_exists.p, c->cs.table.p, c->cs.cap,
                                           c->cs.bytes.p, c->cs.off.p, npos, n, shape.second, flt_memb.p);
                        hipLaunchKernelGGL(filter_or_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, flt_memb.p, npos, bo_run.p);
                    }
                }
                {
                    Prof p(c, COLIBRI_K_COUNT);
                    hipLaunchKernelGGL(backoff_select_kernel, dim3(stream_grid((uint64_t)npos + 1)), dim3(kBlock), 0, c->stream, bo_run.p, c->tokstart.p, npos, (uint32_t)n,
                                       filtered ? 1u : (uint32_t)(n - backoff + 1), bo_flen.p, bo_off.p, bo_unit.p, c->state.p);
                }
                if ((rc = read_state(c))) return rc;
                const uint32_t cand = c->hstate.admitted;
                uint32_t       kept_here = 0;
                if (cand) {
                    const uint32_t cap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, (uint64_t)cand + (cand >> 1) + 1024);
                    if ((rc = dev_alloc(c, bo_table, cap))) return rc;
                    bool grouped = false;
                    for (int attempt = 0; attempt < 4 && !grouped; ++attempt) {
                        const uint64_t seed = 0x3C6EF372FE94F82Bull + 0x9E3779B97F4A7C15ull * (uint64_t)attempt;
                        HIP_TRY(c, hipMemsetAsync(bo_info.p, 0, sizeof(FlexInfo), c->stream));
                        {
                            Prof p(c, COLIBRI_K_COUNT);
                            hipLaunchKernelGGL(flex_clear_kernel, dim3(stream_grid(cap)), dim3(kBlock), 0, c->stream, bo_table.p, cap);
                            hipLaunchKernelGGL(flex_insert_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->bytes.p, bo_off.p, bo_flen.p, bo_unit.p, npos, seed, bo_table.p, cap, bo_slot.p);
                            hipLaunchKernelGGL(flex_verify_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->bytes.p, bo_off.p, bo_flen.p, npos, bo_table.p, bo_slot.p, bo_isrep.p, bo_info.p);
                        }
                        FlexInfo got{};
                        HIP_TRY(c, hipMemcpyAsync(&got, bo_info.p, sizeof got, hipMemcpyDeviceToHost, c->stream));
                        HIP_TRY(c, hipStreamSynchronize(c->stream));
                        HIP_TRY(c, hipGetLastError());
                        grouped = !got.collision;
                    }
                    if (!grouped) return fail(c, COLIBRI_ERR_OVERFLOW, "back-off pass: hash collisions under four seeds");
                    {
                        Prof p(c, COLIBRI_K_PRUNE);
                        hipLaunchKernelGGL(backoff_keep_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, bo_isrep.p, bo_table.p, bo_slot.p, npos, pl.thr, bo_keep.p, c->state.p);
                    }
                    unsigned long long kk = 0;
                    if ((rc = scan_u32(c, bo_keep.p, npos, bo_rank.p, &kk))) return rc;
                    kept_here = (uint32_t)kk;
                    Prof p(c, COLIBRI_K_PRUNE);
                    hipLaunchKernelGGL(backoff_results_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, bo_keep.p, bo_rank.p, bo_table.p, bo_slot.p, npos, res_total, pl.res_cap,
                                       c->res_rep.p, c->res_cnt.p, c->state.p);
                    hipLaunchKernelGGL(backoff_ids_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, bo_flen.p, bo_table.p, bo_slot.p, npos, c->ids[n].p, c->state.p);
                } else {
                    HIP_TRY(c, hipMemsetAsync(c->ids[n].p, 0xFF, sizeof(uint32_t) * (size_t)npos, c->stream));
                }
                HIP_TRY(c, hipMemcpyAsync(&c->state.p->kept, &kept_here, sizeof kept_here, hipMemcpyHostToDevice, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                list_valid = false;
            } else if (n == 1 && uni_synced) {
                // order 1 on the class-indexed count array (as in the plain mode): no hashing, no table; the survivor id of a unigram is its
                // RESULT index here, read per position through a class -> result table
                const uint32_t nclasses = c->maxclass + 1;
                const uint32_t shift    = uni_range_shift(c);
                if ((rc = dev_alloc(c, c->cnt1, (size_t)nclasses + 1)) || (rc = dev_alloc(c, c->rep1, (size_t)nclasses + 1)) || (rc = dev_alloc(c, c->uni_resid, (size_t)nclasses + 1))) return rc;
                if (shift) {
                    if ((rc = uni_alloc(c)) || (rc = uni_count_partitioned(c, shift, c->cnt1.p, nclasses))) return rc;
                } else {
                    HIP_TRY(c, hipMemsetAsync(c->cnt1.p, 0, sizeof(uint32_t) * nclasses, c->stream));
                    Prof p(c, COLIBRI_K_COUNT);
                    hipLaunchKernelGGL(uni_count_kernel, dim3(512), dim3(kBlock), 0, c->stream, c->cls.p, npos, c->cnt1.p, c->rep1.p, c->state.p);
                }
                {
                    Prof p(c, COLIBRI_K_PRUNE);
                    hipLaunchKernelGGL(uni_finish_kernel, dim3(stream_grid(nclasses)), dim3(kBlock), 0, c->stream, c->cnt1.p, (const uint32_t*)nullptr, nclasses, pl.thr, c->state.p,
                                       c->res_rep.p, c->res_cnt.p, pl.res_cap, bi2_synced ? reinterpret_cast<uint16_t*>(c->uni_surv.p) : (uint16_t*)nullptr,
                                       /*count_valid=*/bi2_synced && !o.indexed /* no id pass follows then, see below */, c->uni_resid.p);
                }
                // per-position order-1 ids (result indices): read by the forward index and by a first-generation order 2; the skipgram passes of an unindexed run
                // name their one-token parts by class id instead (part_ids)
                c->ids1_is_cls = bi2_synced && !o.indexed;
                if (!c->ids1_is_cls) {
                    Prof p(c, COLIBRI_K_RESOLVE);
                    hipLaunchKernelGGL(uni_resid_ids_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->cls.p, c->uni_resid.p, c->ids[n].p, c->state.p, npos);
                }
            } else if (n == 2 && bi2_synced) {
                // order 2 on the second-generation kernels (bigram2.hpp), which also leave the bigrams' result indices per position and the active list of order 3
                c->hstate.radix_overflow = 0;
                if ((rc = write_state(c))) return rc;
                if ((rc = bigram2_order(c, pl, /*want_list=*/true, c->ids[n].p))) return rc;
                list_valid = true;
            } else if (radix_pass) {
                // n-gram pass on the radix path: emit -> level B -> per-bin LDS count; survivors leave as (bin, rank) codes that the resolve turns
                // into result indices (= the ids the skipgram passes and the forward index work with); from order 3 on only the active list is walked
                // (a constrained pass has no look-back, hence no list: every position is asked whether its window is a member)
                const bool use_list = !constrained && n >= 3 && list_valid;
                listed_order        = use_list;
                c->hstate.radix_overflow = 0;
                if ((rc = write_state(c))) return rc;
                if (constrained)
                    rc = binned_count_stage(c, pl, member, n, false, pl.thr, false, true, false, /*dense_code=*/true);
                else
                    rc = binned_count_stage(c, pl, KeyNgram{c->ids[n - 1].p, n}, n, use_list, pl.thr, false, true, false, /*dense_code=*/true);
                if (rc) return rc;
                const BinnedIO io = binned_planes(c, pl, false);
                {
                    Prof p(c, COLIBRI_K_PRUNE);
                    hipLaunchKernelGGL(bin_kept_scan_kernel, dim3(kBins), dim3(kBlock), 0, c->stream, c->state.p, c->binstate.p, pl.res_cap);
                    hipLaunchKernelGGL(compact_bins_kernel, dim3(1024), dim3(kBlock), 0, c->stream, io.sp_rep, io.sp_cnt, c->state.p, c->binstate.p, c->res_rep.p, c->res_cnt.p, pl.res_cap,
                                       use_list ? (const uint32_t*)c->alist[n & 1].p : (const uint32_t*)nullptr);
                    hipLaunchKernelGGL(bin_advance_prepare_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p, c->binstate.p);
                }
                if ((rc = binned_resolve_stage(c, pl, c->ids[n].p, n, use_list, !constrained, nullptr, 0u, /*prefill_ids=*/true, /*decode=*/true, res_total))) return rc;
                list_valid = !constrained;
            } else {
                if (constrained)
                    launch_count(c, pl, member, c->ids[n].p, 3, COLIBRI_K_COUNT);
                else if (n == 1)
                    launch_count(c, pl, KeyUnigram{c->bytes.p, c->tokstart.p}, c->ids[n].p, 3, COLIBRI_K_COUNT);
                else
                    launch_count(c, pl, KeyNgram{c->ids[n - 1].p, n}, c->ids[n].p, 3, COLIBRI_K_COUNT);
                launch_prune(c, pl, pl.thr, nullptr, 0);
                launch_resolve(c, pl, c->ids[n].p);
            }
            if ((rc = read_state(c))) return rc;
            if (bi2_synced && c->hstate.radix_overflow == 4) {  // the second-generation order 2 could not hold this corpus: again, on the first-generation kernels
                note_retry(c, COLIBRI_FALLBACK_ORDER2);
                if (retry_with_small_passes(npos)) {  // (a corpus beyond the old pass size: its bins get the old load back before anything slower is tried)
                    tl_small_passes = true;
                    const int rc2   = colibri_train_once(c, &o, stats_out);
                    tl_small_passes = false;
                    return rc2;
                }
                c->b2.disabled = true;
                const int rc2  = colibri_train_once(c, &o, stats_out);
                c->b2.disabled = false;
                return rc2;
            }
            if ((radix_synced || radix_constrained) && c->hstate.radix_overflow) {  // a bin outgrew its LDS table: the whole run again on the table path (loud, exact, rare)
                colibri_options again = o;
                again.table_mode      = 1;
                note_retry(c, (c->hstate.radix_overflow >= 1 && c->hstate.radix_overflow <= 3) ? (int)c->hstate.radix_overflow : COLIBRI_FALLBACK_BIN);
                return colibri_train_once(c, &again, stats_out);
            }
            const uint32_t found = c->hstate.found, kept = c->hstate.kept;
            adm_n[n]   = c->hstate.admitted;
            valid_n[n] = c->hstate.valid;
            s.admitted[n] = adm_n[n];
            if (found == 0 && !constrained && !continued && !(filtered && pl.thr == 1)) break;  // "None found" (patternmodel.h:1189-1194: `if (!continued) break`); a constrained run is one pass over all lengths
            if (found) s.maxn = n;
            s.found[n] = found;
            s.kept[n]  = kept;
            if (kept) c->segments.push_back({res_total, kept, n, (n == 1 && uni_synced && !backoff_pass) ? kMaskFromClass : 0u});
            ngram_first[n] = res_total;
            ngram_kept[n]  = kept;
            res_total += kept;
            c->hstate.res_total = res_total;
            if (o.indexed && kept && (rc = emit_pairs(c, pl, c->ids[n].p, false))) return rc;  // occurrences of the surviving n-grams (as many as positions with an id)
            // secondary word threshold: the unigrams below it keep their place (and references) in the model, but take no part in longer patterns
            if (n == 1 && wthr > pl.thr) hipLaunchKernelGGL(ids_min_count_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->ids[n].p, c->res_cnt.p, wthr, npos);
            if (constrained && (o.doskipgrams || o.doskipgrams_exhaustive) && pl.thr == 1 && n >= 3) {
                // skipgrams of a constrained run: the reference reaches computeskipgrams only in its single pass at MINTOKENS = 1 (:1163: `(n >= 3) || (MINTOKENS == 1)`
                // with n == 1 in that pass); with a higher threshold a constrained run has no skipgrams at all, and neither has this one
                if (n > kMaskedMaxTokens) return fail(c, COLIBRI_ERR_UNSUPPORTED, "skipgrams of patterns longer than 13 tokens are not on the accelerated path (set MAXLENGTH)");
                for (uint32_t mask : gap_masks(n, o.maxskips)) {
                    uint32_t f = 0, k = 0;
                    rc = constrained_skipgram_pass(c, pl, o, n, mask, member.memb, radix_constrained, res_total, &f, &k);
                    if (rc == kRerunOnTable) {
                        colibri_options again = o;
                        again.table_mode      = 1;
                        note_retry(c, (c->hstate.radix_overflow >= 1 && c->hstate.radix_overflow <= 3) ? (int)c->hstate.radix_overflow : COLIBRI_FALLBACK_BIN);
                        return colibri_train_once(c, &again, stats_out);
                    }
                    if (rc) return rc;
                    s.found[n] += f;
                    s.kept[n] += k;
                    if (k) c->segments.push_back({res_total, k, n, mask});
                    res_total += k;
                    c->hstate.res_total = res_total;
                }
            } else if (!constrained && o.doskipgrams_exhaustive && n >= 3) {  // patternmodel.h:1163-1171 -> computeskipgrams :1370-1527, for every admissible window
                if (n > kMaxSkipgramTokens) return fail(c, COLIBRI_ERR_UNSUPPORTED, "skipgrams of patterns longer than 31 tokens do not exist (a gap mask has 32 bits; set MAXLENGTH)");
                if (radix_synced && listed_order) {  // the order's own active list IS the list of positions whose (n-1)-gram survived
                    c->skl   = c->alist[n & 1].p;
                    c->skl_n = c->alist_n.p + (n & 1);
                } else if (!built_ids(c, n - 1))
                    return COLIBRI_ERR_STATE;
                else if ((rc = build_skip_list(c, pl, c->ids[n - 1].p)))
                    return rc;
                const std::vector<uint32_t> masks = gap_masks(n, o.maxskips);
                const bool logged = radix_synced && masks.size() <= kSegLogCap;  // the order's passes are enqueued without a read-back each; one look at the log afterwards
                if (logged) {
                    if ((rc = dev_alloc(c, c->seglog, 4 + 5 * (size_t)kSegLogCap))) return rc;
                    HIP_TRY(c, hipMemsetAsync(c->seglog.p, 0, sizeof(uint32_t) * 4, c->stream));
                    c->hstate.radix_overflow = 0;
                    if ((rc = write_state(c))) return rc;  // (res_total of the orders so far; the passes advance it on the device)
                }
                for (uint32_t mask : masks) {
                    uint32_t f = 0, k = 0;
                    if (radix_synced)
                        rc = skipgram_pass_radix(c, pl, n, mask, c->ids[n - 1].p, c->ids[n - 1].p, thr_skip, res_total, &f, &k, nullptr, 0, 0, 0, logged ? c->seglog.p : (uint32_t*)nullptr);
                    else
                        rc = skipgram_pass(c, pl, n, mask, c->ids[n - 1].p, c->ids[n - 1].p, adm_n[n], thr_skip, false, 0, &f, &k, nullptr);
                    if (rc == kRerunOnTable) {
                        colibri_options again = o;
                        again.table_mode      = 1;
                        note_retry(c, (c->hstate.radix_overflow >= 1 && c->hstate.radix_overflow <= 3) ? (int)c->hstate.radix_overflow : COLIBRI_FALLBACK_BIN);
                        return colibri_train_once(c, &again, stats_out);
                    }
                    if (rc) return rc;
                    if (logged) continue;
                    s.found[n] += f;
                    s.kept[n] += k;
                    if (k) c->segments.push_back({res_total, k, n, mask});
                    res_total += k;
                    c->hstate.res_total = res_total;
                }
                if (logged && !masks.empty()) {
                    std::vector<uint32_t> log(4 + 5 * masks.size());
                    HIP_TRY(c, hipMemcpyAsync(log.data(), c->seglog.p, sizeof(uint32_t) * log.size(), hipMemcpyDeviceToHost, c->stream));
                    if ((rc = read_state(c))) return rc;
                    if (c->hstate.radix_overflow) {
                        colibri_options again = o;
                        again.table_mode      = 1;
                        note_retry(c, (c->hstate.radix_overflow >= 1 && c->hstate.radix_overflow <= 3) ? (int)c->hstate.radix_overflow : COLIBRI_FALLBACK_BIN);
                        return colibri_train_once(c, &again, stats_out);
                    }
                    for (size_t e = 0; e < masks.size() && e < log[0]; ++e) {
                        const uint32_t* x = log.data() + 4 + 5 * e;
                        s.found[n] += x[4];
                        s.kept[n] += x[1];
                        if (x[1]) c->segments.push_back({x[0], x[1], (int)x[2], x[3]});
                    }
                    res_total           = c->hstate.res_total;
                    c->hstate.res_total = res_total;
                }
            }
            // next order
            if (!constrained) c->hstate.cap = (uint32_t)std::min<uint64_t>(pl.table_slots, (uint64_t)valid_n[n] + (valid_n[n] >> 1) + 1024u);
            c->hstate.found = c->hstate.kept = c->hstate.admitted = c->hstate.valid = 0;
            if ((rc = write_state(c))) return rc;
            if (valid_n[n] == 0 && !constrained && !continued && !filtered && !(backoff && n >= backoff + 1)) break;  // nothing can be admitted at n + 1 (a back-off order asks the order-b survivors instead)
        }
        // ... enqueued, when the n-gram orders were (round 4): no look at a pass from the host — its counters, the filler filter's sizes and "None found" live on the device,
        // one log per run (skip_pass_end_kernel) —; each pass used to cost two read-backs (0.9 ms of the indexed + skipgrams step in stream drains)
        bool skips_logged = false;
        if (o.doskipgrams && !constrained && radix_synced && enq && o.indexed && !getenv("COLIBRI_SYNCED_SKIPS")) {
            size_t total = 0;
            for (int n = 3; n <= std::min<int>(maxlength, s.maxn) && total <= kSegLogCap; ++n) total += n > 16 ? (size_t)kSegLogCap + 1 : gap_masks(n, o.maxskips).size();
            skips_logged = total <= kSegLogCap && std::min<int>(maxlength, s.maxn) <= kMaxSkipgramTokens;
        }
        if (skips_logged && std::min<int>(maxlength, s.maxn) >= 3) {
            if ((rc = dev_alloc(c, c->seglog, 4 + 5 * (size_t)kSegLogCap))) return rc;
            HIP_TRY(c, hipMemsetAsync(c->seglog.p, 0, sizeof(uint32_t) * 4, c->stream));
            c->hstate.found = c->hstate.kept = c->hstate.admitted = c->hstate.valid = 0;
            c->hstate.radix_overflow = 0;
            c->hstate.res_total      = res_total;
            if ((rc = write_state(c))) return rc;
            size_t nlogged = 0;
            for (int n = 3; n <= std::min<int>(maxlength, s.maxn); ++n) {
                if (!built_ids(c, n)) return COLIBRI_ERR_STATE;
                if ((rc = build_skip_list(c, pl, c->ids[n].p))) return rc;
                const std::vector<uint32_t> masks = gap_masks(n, o.maxskips);
                const uint32_t minsrc = o.minskiptypes > 1 ? (uint32_t)o.minskiptypes : 0u;
                const uint32_t bound  = valid_n[n] / std::max(1u, pl.thr) + 1;  // a kept skipgram has >= MINTOKENS of the list's windows
                for (uint32_t mask : masks) {
                    uint32_t  f = 0, k = 0;
                    uint32_t* ids = nullptr;
                    if ((rc = skipgram_pass_radix(c, pl, n, mask, c->ids[n].p, nullptr, pl.thr, 0u, &f, &k, &ids, minsrc, ngram_first[n], ngram_kept[n], c->seglog.p, bound))) return rc;
                    if ((rc = emit_pairs_list(c, valid_n[n], ids))) return rc;  // (a pass that kept nothing left no valid id)
                }
                hipLaunchKernelGGL(skip_order_end_kernel, dim3(1), dim3(1), 0, c->stream, c->state.p, (const uint32_t*)c->seglog.p, (uint32_t)nlogged, (uint32_t)masks.size());
                nlogged += masks.size();
            }
            std::vector<uint32_t> log(4 + 5 * nlogged, 0);
            if (nlogged) HIP_TRY(c, hipMemcpyAsync(log.data(), c->seglog.p, sizeof(uint32_t) * log.size(), hipMemcpyDeviceToHost, c->stream));
            if ((rc = read_state(c))) return rc;
            if (c->hstate.radix_overflow) {
                colibri_options again = o;
                again.table_mode      = 1;
                note_retry(c, (c->hstate.radix_overflow >= 1 && c->hstate.radix_overflow <= 3) ? (int)c->hstate.radix_overflow : COLIBRI_FALLBACK_BIN);
                return colibri_train_once(c, &again, stats_out);
            }
            for (size_t e = 0; e < nlogged && e < log[0]; ++e) {
                const uint32_t* x = log.data() + 4 + 5 * e;
                s.found[x[2]] += x[4];
                s.kept[x[2]] += x[1];
                if (x[1]) c->segments.push_back({x[0], x[1], (int)x[2], x[3]});
            }
            res_total = c->hstate.res_total;
        }
        if (o.doskipgrams && !constrained && !skips_logged) {  // IndexedPatternModel::trainskipgrams (patternmodel.h:2969-3010): from the SURVIVING n-grams, n = 3..
            for (int n = 3; n <= std::min<int>(maxlength, s.maxn); ++n) {
                if (n > kMaxSkipgramTokens) return fail(c, COLIBRI_ERR_UNSUPPORTED, "skipgrams of patterns longer than 31 tokens do not exist (a gap mask has 32 bits; set MAXLENGTH)");
                uint32_t found_n = 0;
                if (!built_ids(c, n)) return COLIBRI_ERR_STATE;
                if ((rc = build_skip_list(c, pl, c->ids[n].p))) return rc;
                for (uint32_t mask : gap_masks(n, o.maxskips)) {
                    uint32_t f = 0, k = 0;
                    int fs = 0;
                    if (radix_synced) {
                        uint32_t* ids = nullptr;
                        rc = skipgram_pass_radix(c, pl, n, mask, c->ids[n].p, nullptr, pl.thr, res_total, &f, &k, &ids, o.minskiptypes > 1 ? (uint32_t)o.minskiptypes : 0u, ngram_first[n],
                                                 ngram_kept[n]);
                        if (rc == kRerunOnTable) {
                            colibri_options again = o;
                            again.table_mode      = 1;
                            note_retry(c, (c->hstate.radix_overflow >= 1 && c->hstate.radix_overflow <= 3) ? (int)c->hstate.radix_overflow : COLIBRI_FALLBACK_BIN);
                            return colibri_train_once(c, &again, stats_out);
                        }
                        if (rc) return rc;
                        if (k && (rc = emit_pairs_list(c, valid_n[n], ids))) return rc;  // occurrences of the kept skipgrams of this pass -> forward index, over the order's list
                    } else {
                        if ((rc = skipgram_pass(c, pl, n, mask, c->ids[n].p, nullptr, valid_n[n], pl.thr, true, o.minskiptypes > 1 ? (uint32_t)o.minskiptypes : 0u, &f, &k, &fs))) return rc;
                        if (k) {  // occurrences of the kept skipgrams of this pass -> forward index
                            hipLaunchKernelGGL(skip_result_ids_kernel, dim3(pl.pos_grid), dim3(kBlock), 0, c->stream, c->scratch[fs].p, c->table.p, c->scratch[fs ^ 1].p, npos);
                            if ((rc = emit_pairs(c, pl, c->scratch[fs ^ 1].p, false))) return rc;
                        }
                    }
                    found_n += f;
                    s.found[n] += f;
                    s.kept[n] += k;
                    if (k) c->segments.push_back({res_total, k, n, mask});
                    res_total += k;
                    c->hstate.res_total = res_total;
                }
                if (!found_n) break;  // " None found" (:2992-2994)
            }
        }
        c->hstate.res_total = res_total;
        c->last_mode        = (radix_synced || radix_constrained) ? 2 : 1;
        c->last_passes      = 1;
        c->run_path         = !(radix_synced || radix_constrained) ? COLIBRI_PATH_TABLE
                                                                    : (COLIBRI_PATH_RADIX | (bi2_synced ? COLIBRI_PATH_BI2 : 0) | (chain_synced ? COLIBRI_PATH_CHAIN : 0) |
                                                                       ((chain_synced && chain_wide(npos)) ? COLIBRI_PATH_WIDE : 0));
        if (!enq) c->run_path |= COLIBRI_PATH_PER_PASS;
        if (o.indexed && (rc = finalize_index(c, res_total))) {
            if (rc == kRerunPairs || rc == kRerunRanks) {  // (a model with more than two references per position; a rank that failed its check)
                note_retry(c, rc == kRerunPairs ? COLIBRI_FALLBACK_PAIRS : COLIBRI_FALLBACK_LDS_ORDER);
                return colibri_train_once(c, opt_in, stats_out);
            }
            return rc;
        }
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    collect_events(c);

    s.totaltokens = c->ntokens;
    s.nsentences  = c->nsent;
    s.npatterns   = c->hstate.res_total;
    s.minn        = s.npatterns ? 1 : 0;
    s.train_ms    = ms;
    for (int n = 1; n < COLIBRI_MAX_ORDER; ++n) {
        s.pruned[n] = s.found[n] - s.kept[n];
        s.windows[n] = (n <= o.maxlength) ? c->windows_n[n] : 0;
    }
    s.totaltypes = constrained ? 0 : s.found[1];  // distinct unigrams before pruning (patternmodel.h:1199-1201); a constrained run leaves it unset (:1197: constrainbymodel != NULL)
    c->trained   = true;
    c->keybytes  = 0;
    c->export_ready = false;  // key lengths / offsets: computed when the results are first asked for (colibri_result_sizes), not part of counting
    s.nrefs    = o.indexed ? c->npairs : 0;
    if (stats_out) *stats_out = s;
    return COLIBRI_OK;
}

extern "C" {

// key byte lengths and offsets of the results, once per trained model
static int ensure_export(colibri_ctx* c) {
    if (c->export_ready) return COLIBRI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = prepare_export(c);
    if (rc) return rc;
    collect_events(c);
    c->stats.keybytes = c->keybytes;
    c->export_ready   = true;
    return COLIBRI_OK;
}

int colibri_result_sizes(colibri_ctx* c, uint64_t* npatterns, uint64_t* keybytes, uint64_t* nrefs) {
    if (!c) return COLIBRI_ERR_ARG;
    if (!c->trained) return COLIBRI_ERR_STATE;
    int rc0 = ensure_export(c);
    if (rc0) return rc0;
    if (npatterns) *npatterns = c->hstate.res_total;
    if (keybytes) *keybytes = c->keybytes;
    if (nrefs) *nrefs = c->opt.indexed ? c->npairs : 0;
    return COLIBRI_OK;
}

int colibri_export_unindexed(colibri_ctx* c, uint64_t* key_off, uint8_t* key_bytes, uint32_t* counts) {
    if (!c || !key_off || !counts) return COLIBRI_ERR_ARG;
    if (!c->trained) return fail(c, COLIBRI_ERR_STATE, "export before train");
    int rc0 = ensure_export(c);
    if (rc0) return rc0;
    if (!key_bytes && c->keybytes) return COLIBRI_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t R = c->hstate.res_total;
    key_off[R]       = c->keybytes;
    if (!R) return COLIBRI_OK;
    DevBuf<uint8_t> out;
    int             rc;
    if ((rc = dev_alloc(c, out, (size_t)c->keybytes + 1))) return rc;
    {
        Prof p(c, COLIBRI_K_EXPORT);
        for (const auto& sg : c->segments)
            hipLaunchKernelGGL(export_bytes_kernel, dim3(blocks_for(sg.count, kBlock)), dim3(kBlock), 0, c->stream, c->bytes.p, c->tokstart.p, c->res_rep.p, c->keylen.p, c->keyoff.p,
                               sg.first, sg.count, sg.n, sg.mask, out.p);
    }
    HIP_TRY(c, hipMemcpyAsync(key_bytes, out.p, c->keybytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(key_off, c->keyoff.p, sizeof(uint64_t) * R, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(counts, c->res_cnt.p, sizeof(uint32_t) * R, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    dev_free(out);
    collect_events(c);
    return COLIBRI_OK;
}

int colibri_export_indexed(colibri_ctx* c, uint64_t* key_off, uint8_t* key_bytes, uint32_t* counts, uint64_t* ref_off, uint32_t* ref_sentence, uint16_t* ref_token) {
    if (!c || !ref_off || (c->npairs && (!ref_sentence || !ref_token))) return COLIBRI_ERR_ARG;
    if (!c->trained || !c->opt.indexed) return fail(c, COLIBRI_ERR_STATE, "export_indexed needs a trained indexed model");
    int rc = colibri_export_unindexed(c, key_off, key_bytes, counts);
    if (rc) return rc;
    const uint32_t R = c->hstate.res_total;
    ref_off[R]       = c->npairs;
    if (!R) return COLIBRI_OK;
    // ref_off = exclusive scan of the counts (an indexed model's count IS its number of references)
    const uint32_t             nb = blocks_for(R, kBlock * 4);
    DevBuf<unsigned long long> off;
    if ((rc = dev_alloc(c, off, (size_t)R + 1)) || (rc = dev_alloc(c, c->bsum, (size_t)nb + 1))) return rc;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(kBlock), 0, c->stream, c->res_cnt.p, R, c->bsum.p);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, c->stream, c->bsum.p, nb, c->bsum.p + nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(kBlock), 0, c->stream, c->res_cnt.p, R, c->bsum.p, off.p);
    HIP_TRY(c, hipMemcpyAsync(ref_off, off.p, sizeof(uint64_t) * R, hipMemcpyDeviceToHost, c->stream));
    if (c->npairs) {
        HIP_TRY(c, hipMemcpyAsync(ref_sentence, c->ref_sentence.p, sizeof(uint32_t) * c->npairs, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(ref_token, c->ref_token.p, sizeof(uint16_t) * c->npairs, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    dev_free(off);
    return COLIBRI_OK;
}

int colibri_hash_windows(colibri_ctx* c, int n, uint64_t* out_host) {
    if (!c || !out_host || n < 1) return COLIBRI_ERR_ARG;
    if (!c->have_corpus) return fail(c, COLIBRI_ERR_STATE, "no corpus uploaded");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->npos) return COLIBRI_OK;
    DevBuf<uint64_t> out;
    int              rc;
    if ((rc = dev_alloc(c, out, c->npos))) return rc;
    hipLaunchKernelGGL(hash_windows_kernel, dim3(blocks_for(c->npos, kBlock)), dim3(kBlock), 0, c->stream, c->bytes.p, c->tokstart.p, c->npos, n, out.p);
    HIP_TRY(c, hipMemcpyAsync(out_host, out.p, sizeof(uint64_t) * c->npos, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    dev_free(out);
    return COLIBRI_OK;
}

int colibri_hash_keys(colibri_ctx* c, const uint8_t* bytes, const uint64_t* off, uint64_t nkeys, uint64_t* out_host) {
    if (!c || !off || !out_host || (!bytes && nkeys && off[nkeys])) return COLIBRI_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!nkeys) return COLIBRI_OK;
    for (uint64_t j = 0; j < nkeys; ++j)
        if (off[j + 1] - off[j] >= 192) return fail(c, COLIBRI_ERR_UNSUPPORTED, "keys of 192 bytes or more take SpookyHash's long path, which no pattern on this path reaches");
    DevBuf<uint8_t>            dbytes;
    DevBuf<unsigned long long> doff;
    DevBuf<uint64_t>           dout;
    int                        rc;
    const uint64_t             total = off[nkeys];
    if ((rc = dev_alloc(c, dbytes, (size_t)total + 64))) return rc;
    if ((rc = dev_alloc(c, doff, (size_t)nkeys + 1))) return rc;
    if ((rc = dev_alloc(c, dout, (size_t)nkeys))) return rc;
    HIP_TRY(c, hipMemsetAsync(dbytes.p, 0, total + 64, c->stream));
    if (total) HIP_TRY(c, hipMemcpyAsync(dbytes.p, bytes, total, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(doff.p, off, sizeof(uint64_t) * (nkeys + 1), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(hash_keys_kernel, dim3(blocks_for(nkeys, kBlock)), dim3(kBlock), 0, c->stream, dbytes.p, doff.p, nkeys, dout.p);
    HIP_TRY(c, hipMemcpyAsync(out_host, dout.p, sizeof(uint64_t) * nkeys, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    dev_free(dbytes);
    dev_free(doff);
    dev_free(dout);
    return COLIBRI_OK;
}

int colibri_kernel_time(const colibri_ctx* c, int cls, double* total_ms, uint64_t* launches) {
    if (!c || cls < 0 || cls >= COLIBRI_K_NCLASSES) return COLIBRI_ERR_ARG;
    if (total_ms) *total_ms = c->k_ms[cls];
    if (launches) *launches = c->k_launches[cls];
    return COLIBRI_OK;
}

}  // extern "C"

#include "shard_api.inc"  // colibri_shard_*: opens extern "C"
#include "kshard_api.inc" // colibri_kshard_*
#include "text_api.inc"   // colibri_set_constraint, colibri_text_*
#include "flex_api.inc"   // colibri_flexgrams, colibri_flexgrams_fetch

}  // extern "C"
