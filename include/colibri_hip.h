/* include/colibri_hip.h — the C ABI of libcolibri_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary of the hot path: the reference has no FFI layer of its
 * own, its path sits behind C++ template methods; these entry points are what a binding for that path
 * would call. The C++ face in colibri-core_amd/host/ (PatternModelOptions / PatternModel<uint32_t> /
 * IndexedPatternModel<> / IndexedCorpus with the reference's names and signatures) is implemented on
 * top of exactly these functions. Plain pointers and sizes only; no exceptions, no STL, no torch types.
 *
 * Every function returns COLIBRI_OK (0) or a negative status; colibri_last_error() gives the message the
 * C++ face prints before throwing InternalError (reference include/common.h:41-44). There is no CPU
 * fallback anywhere behind this interface: without a usable HIP device colibri_create fails.
 *
 * One context per host thread (the reference is not thread-safe either, include/classencoder.h:193-207).
 */
#ifndef COLIBRI_HIP_H
#define COLIBRI_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COLIBRI_ABI_VERSION 4 /* 4: colibri_stats says which engines counted the run and why it was repeated, if it was (path, fallback_reason, retries: round 6). 3: colibri_kshard_emit / _count / _apply carry 4-byte keys, bit + number feedback (round 4). 2: colibri_train leaves stats.keybytes at 0 (colibri_result_sizes computes it), kernel classes 11..14, colibri_kshard_*, colibri_stream */
#define COLIBRI_MAX_ORDER 128 /* per-order statistics are kept for n < 128; MAXLENGTH defaults to 100 in the reference */

enum {
    COLIBRI_OK              = 0,
    COLIBRI_ERR_ARG         = -1,  /* NULL / out-of-range argument                                         */
    COLIBRI_ERR_HIP         = -2,  /* a HIP runtime call failed (message has the call and hipGetErrorString) */
    COLIBRI_ERR_NODEVICE    = -3,  /* no gfx950-class device visible                                       */
    COLIBRI_ERR_UNSUPPORTED = -4,  /* options outside the accelerated subset            */
    COLIBRI_ERR_CORPUS      = -5,  /* corpus not usable: > 4 GiB per device, token > 8 bytes, flexgram class  */
    COLIBRI_ERR_STATE       = -6,  /* call order: no corpus uploaded / not trained yet                     */
    COLIBRI_ERR_OVERFLOW    = -7   /* a device table or result buffer was exhausted                        */
};

typedef struct colibri_ctx colibri_ctx;

/* POD mirror of the PatternModelOptions fields PatternModel::train reads on this path
 * (reference include/patternmodel.h:103-213; defaults :153-180). */
typedef struct colibri_options {
    /* Every combination colibri_train does not run is refused with COLIBRI_ERR_UNSUPPORTED / _ARG and a message (check_options in
     * csrc/colibri_hip.hip is the authority; the comments below follow it). "constraint set" = colibri_set_constraint is in force. */
    int32_t mintokens;              /* MINTOKENS: -1 -> 2, 0 -> 1 (patternmodel.h:883-886); any value >= 1 (1: every window is kept)    */
    int32_t maxlength;              /* MAXLENGTH (default 100), >= 1; per-order statistics are kept below COLIBRI_MAX_ORDER              */
    int32_t minlength;              /* MINLENGTH (default 1): > 1 only with a constraint set (the C++ face drops the short patterns of an
                                       unconstrained run itself, after the run)                                                        */
    int32_t maxbackofflength;       /* MAXBACKOFFLENGTH: >= maxlength (no effect), or 1 <= b < maxlength with mintokens >= 2, without
                                       skipgrams or a constraint set: orders above b + 1 look back at the b-token sub-patterns only     */
    int32_t mintokens_unigrams;     /* MINTOKENS_UNIGRAMS (-W): > mintokens = secondary word threshold; needs mintokens >= 2, no
                                       constraint set, table_mode != 2                                                                 */
    int32_t mintokens_skipgrams;    /* MINTOKENS_SKIPGRAMS (raised to mintokens when lower, :887-888)                                   */
    int32_t minskiptypes;           /* MINSKIPTYPES (default 2): distinct fillers a skipgram of an indexed model needs                  */
    int32_t maxskips;               /* MAXSKIPS (default 3), >= 1                                                                       */
    int32_t doskipgrams;            /* DOSKIPGRAMS: indexed models only (the reference's rule, :1558); not with a constraint set        */
    int32_t doskipgrams_exhaustive; /* DOSKIPGRAMS_EXHAUSTIVE: unindexed models only; not together with doskipgrams (:958-963); not with
                                       a constraint set. Either kind: patterns of more than 13 tokens stop the run when they turn up     */
    int32_t dopatternperline;       /* DOPATTERNPERLINE (-L): every line is one pattern; needs mintokens = 1, minlength = 1, unindexed,
                                       no skipgrams, no constraint set                                                                 */
    int32_t prunenonsubsumed;       /* PRUNENONSUBSUMED: must be 0 here (a post-hoc pass over the finished model: the C++ face does it) */
    int32_t prunesubsumed;          /* PRUNESUBSUMED: must be 0 here (same)                                                             */
    int32_t indexed;                /* 0: PatternModel<uint32_t> (model type 10), 1: IndexedPatternModel<> (20)                         */
    int32_t profile;                /* 1: bracket every kernel class with HIP events (colibri_kernel_time); 2: only the class that holds the
                                       dominant kernel of the path the run takes (K_COUNT2 / K_BINCOUNT / K_COUNT): 2 events per step   */
    int32_t table_mode;             /* 0: automatic; 1: force the global open-addressed table; 2: force radix-partition + LDS count     */
} colibri_options;

/* What train() reports: the numbers the reference keeps in the model (totaltokens/totaltypes/maxn/minn,
 * patternmodel.h:550-556) and prints per order on stderr (:1195-1245). */
typedef struct colibri_stats {
    uint64_t totaltokens;                  /* sum of sentence lengths (patternmodel.h:1047-1048)            */
    uint64_t totaltypes;                   /* distinct unigrams BEFORE pruning (:1199-1201)                  */
    uint64_t npatterns;                    /* size of the final model                                        */
    uint64_t keybytes;                     /* sum of key byte lengths over the final model: 0 in what colibri_train returns (serialised sizes
                                              are not part of counting); colibri_result_sizes computes and reports it                       */
    uint64_t nrefs;                        /* indexed: total index entries                                   */
    uint64_t nsentences;                   /* sentences incl. empty ones                                     */
    int32_t  maxn, minn;                   /* as PatternModel::maxlength()/minlength()                       */
    uint64_t windows[COLIBRI_MAX_ORDER];   /* W_n: n-token windows inside sentences (line.ngrams(), :1063)    */
    uint64_t admitted[COLIBRI_MAX_ORDER];  /* P_n: windows that passed the look-back and were counted        */
    uint64_t found[COLIBRI_MAX_ORDER];     /* distinct new patterns of order n before pruning                */
    uint64_t pruned[COLIBRI_MAX_ORDER];    /* erased by prune()/pruneskipgrams() at order n                  */
    uint64_t kept[COLIBRI_MAX_ORDER];      /* found - pruned                                                 */
    double   train_ms;                     /* wall time of the device work of train(), host clock            */
    /* ABI 4: which engines counted the run (COLIBRI_PATH_* bits, of the LAST attempt — the one whose model this is), and whether the run was repeated: an order that does
     * not fit the engine it started on (a record region, a final bin's LDS table, the key bits) ends the attempt, and the whole run is done again, exactly, on the
     * next engine down — a 2-10 x cost a caller could only see with COLIBRI_DEBUG_OVERFLOW before. fallback_reason: COLIBRI_FALLBACK_* of the FIRST repeat (0: none);
     * retries: repeats in all (the result buffers growing counts as one). */
    int32_t  path, fallback_reason, retries, reserved_;
} colibri_stats;
enum {
    COLIBRI_PATH_TABLE   = 1,  /* the global open-addressed table (device atomics): table_mode = 1, or the last resort                                   */
    COLIBRI_PATH_RADIX   = 2,  /* the radix path: records partitioned twice, counted in LDS (first-generation kernels for the orders the bits below do not name) */
    COLIBRI_PATH_BI2     = 4,  /* ... order 2 on the second-generation engine (8-byte records, dense head, one wave per final bin)                        */
    COLIBRI_PATH_CHAIN   = 8,  /* ... orders >= 3 on that engine too (keys = (number of the leading (n-1)-gram, class))                                    */
    COLIBRI_PATH_WIDE    = 16, /* ... in its form for 2.15 - 4.3 x 10^8 positions (eight sub-regions, 2048-slot tables)                                     */
    COLIBRI_PATH_SLICED  = 32, /* an order counted in several passes over slices of its keys (corpora beyond one pass)                                      */
    COLIBRI_PATH_PER_PASS = 64 /* the id-keeping kinds' loop: one host look-up per pass (indexed / skipgram models off the enqueued loop, constrained runs)   */
};
enum {
    COLIBRI_FALLBACK_NONE      = 0,
    COLIBRI_FALLBACK_REGION    = 1,  /* an A-bin region of the first-generation radix path outgrew its room   -> global table                */
    COLIBRI_FALLBACK_BIN       = 2,  /* a final bin outgrew its LDS table (hash skew)                          -> global table                */
    COLIBRI_FALLBACK_IDS       = 3,  /* survivor id range exhausted                                            -> global table                */
    COLIBRI_FALLBACK_ORDER2    = 4,  /* the second-generation order 2 could not hold the corpus               -> smaller passes, or first-generation kernels */
    COLIBRI_FALLBACK_SPLIT     = 8,  /* a run of a sliced order's direct split outgrew its room                -> the exact split             */
    COLIBRI_FALLBACK_CHAIN     = 16, /* an order >= 3 did not fit the chained engine (key bits, region, bin)   -> first-generation orders >= 3 */
    COLIBRI_FALLBACK_RESULTS   = 32, /* the result buffers were too small (duplicated text)                    -> four times the room         */
    COLIBRI_FALLBACK_PAIRS     = 64, /* an index with more references than the pair buffer started with       -> a larger buffer             */
    COLIBRI_FALLBACK_LDS_ORDER = 128 /* a checked rank of the forward index's build disagreed (ranks from LDS adds) -> every rank matched with ballots */
};

/* kernel classes for colibri_kernel_time */
enum {
    COLIBRI_K_TOKENISE = 0, /* byte stream -> token-start vector (upload time, not train)        */
    COLIBRI_K_CLEAR    = 1, /* table reset                                                       */
    COLIBRI_K_COUNT    = 2, /* scan + SpookyHash + hash-table build: the dominant kernel         */
    COLIBRI_K_PRUNE    = 3, /* threshold prune + survivor compaction                             */
    COLIBRI_K_RESOLVE  = 4, /* per-position survivor ids for the next order                      */
    COLIBRI_K_SKIPGRAM = 5, /* skipgram counting                                                 */
    COLIBRI_K_INDEX    = 6, /* forward-index build                                               */
    COLIBRI_K_EXPORT   = 7,
    COLIBRI_K_EMIT     = 8,  /* binned path: scan + SpookyHash + block-local reduce -> (key, position, count) records */
    COLIBRI_K_SCATTER  = 9,  /* binned path: two-level radix partition of the records by hash bits            */
    COLIBRI_K_BINCOUNT = 10, /* binned path: per-bin LDS hash build + threshold + survivor ids                */
    COLIBRI_K_EMIT2    = 11, /* order 2, second generation: scan -> dense head histogram | 8-byte records by A bin */
    COLIBRI_K_LEVELB2  = 12, /* ... one block partitions one (sub-region, A bin) slot by B bin                  */
    COLIBRI_K_COUNT2   = 13, /* ... one wave per final bin: LDS table, threshold, survivors, positions          */
    COLIBRI_K_LISTS2   = 14, /* ... survivor positions -> bucket lists -> bitmap -> active list of order 3      */
    COLIBRI_K_NCLASSES = 15
};

/* ---- lifecycle ------------------------------------------------------------------------------- */
int         colibri_abi_version(void);
int         colibri_create(colibri_ctx** out, int device); /* device = HIP ordinal                  */
void        colibri_destroy(colibri_ctx* ctx);
const char* colibri_last_error(const colibri_ctx* ctx);   /* never NULL                            */

/* ---- corpus: replaces IndexedCorpus::load (reference src/pattern.cpp:1916-1967) ------------------
 * payload = the .colibri.dat v2 file minus its 2-byte header (A2 02). first_sentence = the index given
 * to the first sentence (train()'s `firstsentence`, patternmodel.h:880-881,896; 1 by default) — this is
 * also how a sentence-sharded rank keeps global sentence numbers. The bytes are copied to HBM and
 * tokenised there (token-start vector + sentence table). */
int colibri_upload_corpus(colibri_ctx* ctx, const uint8_t* payload, uint64_t nbytes, uint32_t first_sentence);
/* same, for bytes that already live in this device's HBM (no PCIe copy; a device-to-device copy is made) */
int colibri_upload_corpus_device(colibri_ctx* ctx, const void* device_payload, uint64_t nbytes, uint32_t first_sentence);
/* tokens (delimiters excluded), sentences (empty ones included), highest class id */
int colibri_corpus_info(const colibri_ctx* ctx, uint64_t* ntokens, uint64_t* nsentences, uint64_t* maxclass);

/* ---- training: replaces PatternModel::train (reference include/patternmodel.h:880-1345) and, for
 * indexed models, IndexedPatternModel::train/trainskipgrams (:2828-2844, :2969-3010). All orders run on
 * the device without a host round trip per order. */
int colibri_train(colibri_ctx* ctx, const colibri_options* opt, colibri_stats* stats);

/* ---- results: replaces iteration over PatternMap + valuehandler.write (patternstore.h:534-542,
 * datatypes.h:219-221,263-270). Caller allocates from colibri_result_sizes():
 *   key_off[npatterns+1], key_bytes[keybytes], counts[npatterns]
 *   indexed: ref_off[npatterns+1], ref_sentence[nrefs], ref_token[nrefs]; refs sorted by (sentence, token).
 * Keys are the pattern's bytes without the trailing 00 (a skipgram's gaps are the byte 03). Order: by
 * pattern length n, unspecified inside an order (the reference's order is unordered_map order). */
int colibri_result_sizes(colibri_ctx* ctx, uint64_t* npatterns, uint64_t* keybytes, uint64_t* nrefs); /* the first call after a train computes the key lengths on the device */
int colibri_export_unindexed(colibri_ctx* ctx, uint64_t* key_off, uint8_t* key_bytes, uint32_t* counts);
int colibri_export_indexed(colibri_ctx* ctx, uint64_t* key_off, uint8_t* key_bytes, uint32_t* counts, uint64_t* ref_off,
                           uint32_t* ref_sentence, uint16_t* ref_token);

/* ---- sentence-sharded multi-GPU training (one context per rank; the collectives themselves are the caller's: RCCL through
 * torch.distributed in colibri_amd.dist, or any all-to-all). The reference has no counterpart — it is single-threaded; these
 * entry points split PatternModel::train's order loop (patternmodel.h:981-1270) and the skipgram passes (:1163-1171, :2969-3010)
 * at their only cross-shard dependency: the global count of a candidate pattern. Survivor ids are GLOBAL (handed out by the owner
 * rank of a key), so the 64-bit keys of all ranks are comparable and counts are summed exactly. All *_dev arguments are DEVICE
 * pointers. A "pass" is identified by (n, mask, level): mask 0 = the n-gram pass of order n; mask != 0 = level 1..parts-1 of the
 * skipgram (n, mask), whose identity is built by pairing the global ids of its contiguous parts left to right.
 *   per pass:  shard_count -> [all-to-all sizes] -> shard_send -> [all-to-all keys, counts (, aux)] -> shard_merge ->
 *              [all-gather found, kept] -> shard_reply -> [all-to-all replies back] -> shard_apply ;  at the end: shard_finish.
 * Each rank then exports (colibri_export_unindexed + colibri_shard_export_gids) the patterns it was named exporter of — the
 * union over ranks is the model — and, for indexed models, its LOCAL forward index keyed by global id
 * (colibri_shard_export_index); a pattern's index is the concatenation of the ranks' runs in rank order (sentence ranges of
 * the ranks are disjoint and ascending, so that concatenation is already sorted). */
int colibri_shard_begin(colibri_ctx* ctx, const colibri_options* opt, int world);
/* local count of pass (n, mask, level), then this rank's distinct candidates partitioned by owner = hash(key) % world */
int colibri_shard_count(colibri_ctx* ctx, int n, uint32_t mask, int level, uint64_t* ncandidates, uint64_t* per_owner /* [world] */);
/* copy the partitioned candidates into the caller's send buffers: keys u64[ncandidates], counts u32[ncandidates],
 * aux u32[ncandidates] (distinct-source counts of indexed skipgrams; zeros otherwise; may be NULL when not wanted) */
int colibri_shard_send(colibri_ctx* ctx, void* keys_dev, void* counts_dev, void* aux_dev);
/* the same without a copy: the library's own partitioned buffers (device pointers, valid until the next colibri_shard_count; *aux_dev = NULL when the pass has none) */
int colibri_shard_send_view(colibri_ctx* ctx, void** keys_dev, void** counts_dev, void** aux_dev);
/* owner side: merge the records received from every rank (concatenated in rank order; per_src[r] records from rank r) */
int colibri_shard_merge(colibri_ctx* ctx, const void* keys_dev, const void* counts_dev, const void* aux_dev, const uint64_t* per_src /* [world] */,
                        uint64_t* found, uint64_t* kept);
/* owner side: assign global survivor ids gid_base.. to the kept keys and fill one reply per received record
 * (reply_gid: u32 global id | bit 31 = "you export it", 0xFFFFFFFF = pruned; reply_cnt: u32 global count) */
int colibri_shard_reply(colibri_ctx* ctx, uint32_t gid_base, void* reply_gid_dev, void* reply_cnt_dev);
/* contributor side: apply the replies (same order as the records sent) and write global survivor ids per position */
int colibri_shard_apply(colibri_ctx* ctx, const void* reply_gid_dev, const void* reply_cnt_dev, uint64_t* exported, uint64_t* admitted);
/* order 1 on class-indexed arrays (the north star's "all-reduce of the per-bucket count tables before the prune"): when every rank's
 * encoding is canonical (eligible) the unigram pass needs no key exchange — each rank fills cnt[nclasses] (u32 per class id) and
 * minrank[nclasses] (its rank where it saw the class, else 0x7FFFFFFF), the caller all-reduces them (SUM / MIN), and every rank
 * applies the reduced arrays. The global id of a unigram is its class id; later passes must start their ids at nclasses. */
int colibri_shard_uni_info(const colibri_ctx* ctx, int* eligible, uint64_t* maxclass);
int colibri_shard_uni_count(colibri_ctx* ctx, void* cnt_dev, void* minrank_dev, uint32_t nclasses, int rank);
int colibri_shard_uni_apply(colibri_ctx* ctx, const void* cnt_global_dev, const void* minrank_global_dev, uint32_t nclasses, int rank, uint64_t* found, uint64_t* kept,
                            uint64_t* exported);
/* close the run: global per-order found / kept (caller-reduced), global token count; fills stats like colibri_train */
int colibri_shard_finish(colibri_ctx* ctx, const uint64_t* found_global, const uint64_t* kept_global, uint64_t totaltokens_global, int maxn, colibri_stats* stats);
/* global id of every pattern this rank exports, in the order of colibri_export_unindexed (gids[npatterns]) */
int colibri_shard_export_gids(colibri_ctx* ctx, uint32_t* gids);
/* indexed models: this rank's forward index — gids[ngids] ascending, ref_off[ngids+1], sentence/token[nrefs] */
int colibri_shard_index_sizes(const colibri_ctx* ctx, uint64_t* ngids, uint64_t* nrefs);
int colibri_shard_export_index(colibri_ctx* ctx, uint32_t* gids, uint64_t* ref_off, uint32_t* ref_sentence, uint16_t* ref_token);

/* ---- key-sharded multi-GPU training of the plain n-gram model (BASELINE.json configs[2]; csrc/kshard.hpp). The reference has no counterpart (it is
 * single-threaded); what is distributed is PatternModel::train's order loop (include/patternmodel.h:981-1270) at its only cross-shard dependency, the GLOBAL count
 * of a candidate before the prune of its order (:1195-1245). Unlike colibri_shard_* above, no rank counts candidates it does not own: every rank scans its own
 * sentences into RECORDS (the single-device emit kernels), the records travel to the owner of their key (owner = top bits of the key's mix), the owner counts them
 * with the single-device kernels and applies the threshold to the exact global count, and only what survived travels back (order 2: positions of surviving windows;
 * order >= 3: global survivor ids of surviving records; plus one (representative, count) per kept pattern to the rank that exports it). Order 1 is an all-reduce
 * (SUM) of the dense per-class count array — the north star's "all-reduce of the per-bucket count tables before the prune". world = 1, 2, 4 or 8.
 * The collectives are the caller's (host/src/sharded.cpp: RCCL, or device copies); all device work is enqueued on colibri_stream(ctx), which the caller hands to
 * its collectives too: an order costs two host look-ups (the sizes of the two exchanges). All *_dev arguments are DEVICE pointers owned by the library.
 *   colibri_kshard_info on every rank -> [the caller checks that all ranks are eligible, takes the maxima] -> colibri_kshard_begin
 *   order 1:    colibri_kshard_uni_count -> [all-reduce SUM, u32 x nclasses, in place] -> colibri_kshard_uni_apply
 *   order n>=2: colibri_kshard_emit -> [sizes] -> colibri_kshard_recv_buffers -> [all-to-all records, tables; order 2: all-reduce head rows SUM / MIN] ->
 *               colibri_kshard_count -> [sizes] -> colibri_kshard_feedback_buffers -> [all-to-all feedback, exports] -> colibri_kshard_apply
 *               (the loop ends after the order at which no rank admitted a window: the reference's "None found", :1189-1194)
 *   end:        colibri_kshard_local_stats -> [sums over ranks] -> colibri_kshard_finish; then colibri_result_sizes / colibri_export_unindexed on every rank:
 *               each pattern of the model is exported by exactly one rank (the lowest that holds an occurrence; rank 0 for unigrams). */
void* colibri_stream(colibri_ctx* ctx); /* the hipStream_t all of the context's device work is enqueued on */
int colibri_kshard_info(colibri_ctx* ctx, const colibri_options* opt, int* eligible, uint64_t* maxclass, uint64_t* npositions);
int colibri_kshard_begin(colibri_ctx* ctx, const colibri_options* opt, int world, int rank, uint64_t maxclass_global, uint64_t maxpositions_global);
int colibri_kshard_uni_count(colibri_ctx* ctx, void** cnt_dev, uint32_t* nclasses);
int colibri_kshard_uni_apply(colibri_ctx* ctx);
/* est_records: an upper bound of the order's records over ALL ranks (order 2: the corpus' tokens; above: the windows that survived the order below), ids_global: the
 * numbers the order below handed out over all ranks (colibri_kshard_apply's *ids_global) — the same values on every rank; more: order n + 1 follows.
 * per_owner[world]: keys for each rank, in rank order in *send_dev (*recbytes = 4 each: the source has partitioned its records completely, a key is what the final bin
 * does not fix); *tab_dev: u32[world][*tab_words] (the runs' lengths per (owner, bin)), row d goes to rank d;
 * *head_dev: order 2: u32[2][4096] (row 0: all-reduce SUM, row 1: all-reduce MIN, in place), else NULL; *admitted: windows this rank counted at order n */
int colibri_kshard_emit(colibri_ctx* ctx, int n, uint64_t est_records, uint64_t ids_global, int more, void** send_dev, void** tab_dev, uint32_t* tab_words, uint64_t* per_owner,
                        uint32_t* recbytes, void** head_dev, uint64_t* admitted);
int colibri_kshard_recv_buffers(colibri_ctx* ctx, uint64_t nrecords, void** recv_dev /* keys, concatenated in source order */, void** tab_recv_dev /* u32[world][tab_words], row s from rank s */);
/* per_src[world]: keys received from each rank. more: order n + 1 follows (feedback is produced). fb_per_dst [world]: feedback for each rank, in rank order in *fb_dev, in
 * units of *fb_bytes (= 4): per source, in the order it sent, one bit per key, then the dense number of every surviving key's window; ex_per_dst: exports (8 bytes each) in
 * *ex_dev; *kept_bins: the keys this owner kept (the caller gathers them over the ranks: colibri_kshard_apply's kept_per_owner) */
int colibri_kshard_count(colibri_ctx* ctx, int n, const uint64_t* per_src, int more, void** fb_dev, uint64_t* fb_per_dst, uint32_t* fb_bytes, void** ex_dev, uint64_t* ex_per_dst,
                         uint64_t* kept_bins);
/* after colibri_kshard_count of order 2: the windows of the surviving head pairs (both classes < 64: counted densely, on no owner's list) over ALL ranks — the same value
 * on every rank (the head counts are all-reduced); with the feedback sizes it bounds order 3's records exactly. No device look-up: read with the count step's sizes. */
int colibri_kshard_head_windows(const colibri_ctx* ctx, uint64_t* windows);
int colibri_kshard_feedback_buffers(colibri_ctx* ctx, uint64_t nfeedback, uint64_t nexports, void** fb_recv_dev, void** ex_recv_dev); /* each concatenated in owner order */
/* fb_src / ex_src [world]: feedback units / exports received from each owner; kept_per_owner[world]; *ids_global: the numbers order n handed out over all ranks */
int colibri_kshard_apply(colibri_ctx* ctx, int n, const uint64_t* fb_src, const uint64_t* ex_src, const uint64_t* kept_per_owner, int more, uint64_t* ids_global);
/* arrays of COLIBRI_MAX_ORDER: distinct keys / survivors among the keys this rank OWNS (order 1: global, on rank 0 only), windows it counted; *syncs: host look-ups so far */
int colibri_kshard_local_stats(colibri_ctx* ctx, uint64_t* found, uint64_t* kept, uint64_t* admitted, uint32_t* syncs);
int colibri_kshard_finish(colibri_ctx* ctx, const uint64_t* found_global, const uint64_t* kept_global, const uint64_t* admitted_global, uint64_t totaltokens_global, int maxn,
                          colibri_stats* stats);

/* ---- parity / measurement hooks ------------------------------------------------------------------ */
/* SpookyHash::Hash64 (reference include/SpookyV2.h:59-66) of every n-token window, computed by the same
 * device routine the count kernel uses: out[i] for token position i (delimiters are positions too);
 * 0 where no n-token window starts. out has colibri_positions() entries. */
int colibri_hash_windows(colibri_ctx* ctx, int n, uint64_t* out_host);
int colibri_positions(const colibri_ctx* ctx, uint64_t* npositions);
/* Which implementation of the counting stage the last colibri_train ran (the reference has one, include/patternmodel.h:1078-1178; here the choice follows
 * the corpus and colibri_options.table_mode, and an overflow of the radix path repeats the run on the table): 1 = global open-addressed table,
 * 2 = radix partition + LDS count, 0 = neither (pattern list, or nothing trained). *passes (optional) = passes over key slices the order-2 stage used. */
int colibri_last_mode(const colibri_ctx* ctx, int* passes);
/* What the dominant kernel of the last plain run processed at order 2 (second-generation kernels, colibri_last_mode = 2): *records = the 8-byte records
 * bi2_count_kernel read (in a key-sharded run: the records this rank counted as an owner); *head_windows = the admitted bigram windows whose two classes are both
 * < 64 — counted in the scan's dense LDS histogram, never records (0 in a key-sharded run). bench.py prices the kernel by these. */
int colibri_order2_records(colibri_ctx* ctx, uint64_t* records, uint64_t* head_windows);
/* SpookyHash::Hash64 of nkeys independent byte strings (off[nkeys+1] into bytes) on the device */
int colibri_hash_keys(colibri_ctx* ctx, const uint8_t* bytes, const uint64_t* off, uint64_t nkeys, uint64_t* out_host);
/* accumulated HIP-event time and launch count of one kernel class since the last colibri_train() began
 * (only when options.profile = 1; events are recorded on the library's own stream) */
int colibri_kernel_time(const colibri_ctx* ctx, int kernel_class, double* total_ms, uint64_t* launches);

/* ---- constrained training -------------------------------------------------------------------------------------------
 * Replaces PatternModel::train(..., constrainbymodel) (reference include/patternmodel.h:1062-1072, :1088-1089, :1209-1217): while a
 * constraint set is installed, colibri_train counts — in one pass per length MINLENGTH..MAXLENGTH, without look-back — exactly the
 * windows whose key bytes are a member, and keeps those that reach MINTOKENS (any value >= 1). key_off[npatterns + 1] / key_bytes: the
 * patterns' keys as in colibri_export_unindexed (skipgram / flexgram keys never equal a window and are simply never hit).
 * npatterns = 0 removes the constraint. stats.totaltypes is 0 after a constrained run (the reference leaves it unset there). */
int colibri_set_constraint(colibri_ctx* ctx, const uint64_t* key_off, const uint8_t* key_bytes, uint64_t npatterns);
/* Continued training: replaces PatternModel::train(..., continued = true) on a model that already holds patterns (reference
 * include/patternmodel.h:983-995 "Skipping n-grams, already in model", colibri-patternmodeller -E). The patterns of that model are installed
 * like a constraint set; the next colibri_train then counts only the orders the model has no n-grams of, and the look-back of those orders
 * (:1139-1152) finds the loaded patterns as well as the new survivors. The results are the NEW patterns only (the caller already has the
 * others); stats.totaltokens is the corpus' (the reference leaves the model's total untouched: the C++ face ignores it). MINTOKENS >= 2,
 * no skipgrams. npatterns = 0 (or colibri_set_constraint) ends the mode. */
int colibri_set_continuation(colibri_ctx* ctx, const uint64_t* key_off, const uint8_t* key_bytes, uint64_t npatterns);
/* Filtered training: replaces PatternModel::train(..., filter) (reference include/patternmodel.h:899-914, :1106-1137). While a filter set is installed, colibri_train
 * counts at every order exactly the windows that contain one of the set's n-grams (as a sub-n-gram of any length) or are an instance of one of its skipgrams
 * (same length, non-gap tokens equal; flexgrams match nothing, src/pattern.cpp:1760-1785) — without look-back, then prunes by MINTOKENS per order; the run ends at
 * the first order that finds nothing (MINTOKENS = 1: after MAXLENGTH). MINLENGTH = 1, no skipgrams. npatterns = 0 (or colibri_set_constraint) ends the mode. */
int colibri_set_filter(colibri_ctx* ctx, const uint64_t* key_off, const uint8_t* key_bytes, uint64_t npatterns);

/* ---- class encoder: plain text -> word frequency list -> class-encoded corpus -------------------------------------
 * Replaces the corpus-proportional work of ClassEncoder::processcorpus (reference src/classencoder.cpp:156-188: the word frequency
 * list) and ClassEncoder::encodefile / encodestring (:369-436, :550-600: words -> varint classes, one 00 per line). The class of
 * every DISTINCT word is decided by the caller between colibri_text_words and colibri_text_encode (buildclasses :213-229, the
 * unknown-word policy of encodestring :412-424) — that step is proportional to the vocabulary, not to the corpus.
 * Word rules, `rules` = 0 (frequency list) / 1 (encoder): see colibri-core_amd/csrc/textenc.hpp. */
int colibri_text_upload(colibri_ctx* ctx, const uint8_t* text, uint64_t nbytes);                 /* plain text, '\n' ends a line; < 2 GiB */
int colibri_text_count(colibri_ctx* ctx, int rules, uint64_t* nwords, uint64_t* ndistinct);      /* counts every word under `rules` */
/* one entry per distinct word, in no particular order: byte offset of its FIRST occurrence in the text, its byte length, its count.
 * (The reference fills its unordered_map in first-occurrence order; sorting by first_start reproduces that insertion order.) */
int colibri_text_words(colibri_ctx* ctx, uint32_t* first_start, uint32_t* length, uint32_t* count);
/* cls[k], repeat[k] for distinct word k of the last colibri_text_count(rules = 1): the word becomes `repeat` copies of varint(cls)
 * (0 drops it; "{*3*}" is 3 x class 3). Every '\n' becomes one 00; text after the last '\n' is not encoded (encodefile :569-570).
 * The result is a .colibri.dat v2 payload (no A2 02 header) kept on the device. */
int colibri_text_encode(colibri_ctx* ctx, const uint32_t* cls, const uint32_t* repeat, uint64_t* outbytes, uint64_t* ntokens, uint64_t* nlines);
int colibri_text_fetch(colibri_ctx* ctx, uint8_t* out);                                          /* the encoded payload -> host */
int colibri_text_as_corpus(colibri_ctx* ctx, uint32_t first_sentence);                           /* ... or straight into colibri_upload_corpus_device */

/* ---- flexgrams abstracted from skipgrams ---------------------------------------------------------
 * Replaces IndexedPatternModel::computeflexgrams_fromskipgrams (reference include/patternmodel.h:3724-3744) with
 * Pattern::toflexgram (src/pattern.cpp:145-180) and Pattern::category (:107-127): every SKIPGRAM among the given patterns hands
 * all its references to the flexgram whose key is the skipgram's with each run of {*} (03) tokens replaced by one {**} (04).
 * Input = an indexed model in the layout colibri_export_indexed writes (patterns of other categories are ignored; no corpus is
 * needed). Result, kept on the device until the next call: one entry per distinct flexgram, count = number of references,
 * references ascending by (sentence, token) with duplicates kept (IndexedData::insert is a push_back, datatypes.h:117-119; the
 * reference's own order is its unordered_map's iteration order, and its loop inserts into the map it iterates — the specification
 * here is the loop without that hazard). Grouping is exact (64-bit hash of the collapsed bytes, bytes verified, reseeded on a
 * collision). colibri_flexgrams_fetch copies the result into caller-allocated buffers sized from the three totals
 * (key_off / ref_off: nflexgrams + 1 entries). */
int colibri_flexgrams(colibri_ctx* ctx, const uint64_t* key_off, const uint8_t* key_bytes, const uint64_t* ref_off, const uint32_t* ref_sentence, const uint16_t* ref_token,
                      uint64_t npatterns, uint64_t* nflexgrams, uint64_t* keybytes, uint64_t* nrefs);
/* The same on the indexed model of the last colibri_train of this context, which is still resident in HBM (keys, counts and the forward
 * index never leave the device; only the flexgrams come back through colibri_flexgrams_fetch). COLIBRI_ERR_STATE unless the context holds
 * a trained indexed model of a non-sharded run. */
int colibri_flexgrams_resident(colibri_ctx* ctx, uint64_t* nflexgrams, uint64_t* keybytes, uint64_t* nrefs);
int colibri_flexgrams_fetch(colibri_ctx* ctx, uint64_t* key_off, uint8_t* key_bytes, uint32_t* counts, uint64_t* ref_off, uint32_t* ref_sentence, uint16_t* ref_token);

#ifdef __cplusplus
}
#endif
#endif
