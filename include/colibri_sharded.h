/* include/colibri_sharded.h — C face of the multi-GPU trainer (colibri-core_amd/host/src/sharded.cpp -> lib/libcolibri_sharded.so).
 *
 * PatternModel::train (reference include/patternmodel.h:880-1345) across the GPUs of one node: the corpus is cut into contiguous sentence ranges, one rank per
 * GPU, RCCL over xGMI for the exchange steps (include/colibri_hip.h: colibri_kshard_* for the plain n-gram model, colibri_shard_* for the other model kinds).
 * The reference has no counterpart (it is single-threaded); inside the reference's API this sits behind PatternModel::train itself (colibri_host::set_gpus /
 * colibri-patternmodeller --gpus N, which run the same code). This header exists for callers that want the shards to stay resident in HBM between runs — the
 * benchmark (bench.py --gpus N) and the tests. Plain pointers and sizes; every function returns COLIBRI_OK (0) or a negative status, and
 * colibri_sharded_last_error() has the message.
 *
 * A trainer holds `nlocal` ranks of a run of `world`:
 *   nlocal == world   every rank in this process, one host thread each (ncclCommInitAll; ranks made to share a device exchange by device copies instead);
 *   nlocal == 1       one process per rank: every process passes the same 128-byte id from colibri_sharded_unique_id (distributed by the caller) and its rank.
 * colibri_sharded_train is collective: every trainer of the run calls it with the same options. Afterwards each local rank exports its share of the model
 * (colibri_sharded_result_sizes / colibri_sharded_export_unindexed, layout as colibri_export_unindexed); every pattern is exported by exactly one rank.
 */
#ifndef COLIBRI_SHARDED_H
#define COLIBRI_SHARDED_H
#include "colibri_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define COLIBRI_SHARDED_ID_BYTES 128

typedef struct colibri_sharded colibri_sharded;

typedef struct colibri_sharded_info {
    int32_t  protocol;      /* 0: key-sharded counting (colibri_kshard_*), 1: candidate exchange (colibri_shard_*)                          */
    int32_t  rccl;          /* 1: RCCL collectives, 0: device copies between contexts sharing a device                                     */
    uint32_t host_lookups;  /* times a rank's host waited for its device during the run (key-sharded protocol: two per order + two)        */
    uint32_t pad;
    double   wall_ms;       /* host clock around the whole collective call                                                                */
    uint64_t alltoall_bytes;         /* bytes the first local rank put into all-to-alls during the run (records, tables, feedback, exports) ...     */
    uint64_t alltoall_bytes_to_self; /* ... of which addressed to itself (never leave the device)                                                  */
    uint64_t allreduce_bytes;        /* bytes it put into all-reduces (dense class counts, dense head)                                             */
} colibri_sharded_info;

int         colibri_sharded_unique_id(void* out128);
int         colibri_sharded_create(colibri_sharded** out, int world, int nlocal, int first_rank, const int* devices /* [nlocal] or NULL */, const void* unique_id /* or NULL */);
void        colibri_sharded_destroy(colibri_sharded* t);
const char* colibri_sharded_last_error(const colibri_sharded* t);
/* the shard of one local rank: a .colibri.dat v2 payload (header stripped) of whole sentences; first_sentence = global number of its first sentence */
int colibri_sharded_upload(colibri_sharded* t, int local_rank, const uint8_t* payload, uint64_t nbytes, uint32_t first_sentence);
/* nlocal == world: cut a whole payload into `world` contiguous sentence ranges of about equal bytes and upload them */
int colibri_sharded_upload_split(colibri_sharded* t, const uint8_t* payload, uint64_t nbytes, uint32_t first_sentence);
/* 0 (default): key-sharded counting where the run allows it; 1: always the candidate exchange */
int colibri_sharded_set_protocol(colibri_sharded* t, int protocol);
/* stats: found / kept / admitted / tokens / types are the model's (global); npatterns, nsentences, windows are summed over this trainer's local ranks */
int colibri_sharded_train(colibri_sharded* t, const colibri_options* opt, colibri_stats* stats, colibri_sharded_info* info);
/* colibri_kernel_time of one local rank's context (options.profile = 1 or 2 in the last colibri_sharded_train) */
int colibri_sharded_kernel_time(colibri_sharded* t, int local_rank, int kernel_class, double* total_ms, uint64_t* launches);
int colibri_sharded_result_sizes(colibri_sharded* t, int local_rank, uint64_t* npatterns, uint64_t* keybytes);
int colibri_sharded_export_unindexed(colibri_sharded* t, int local_rank, uint64_t* key_off, uint8_t* key_bytes, uint32_t* counts);
/* indexed models (options.indexed; replaces what IndexedPatternModel keeps per pattern, reference include/patternmodel.h:2789-2800, include/datatypes.h:247-297): the
 * references stay with the rank that holds their sentences, keyed by the patterns' GLOBAL numbers.
 *   colibri_sharded_export_gids:  the global number of every pattern this rank exports, in the order of colibri_sharded_export_unindexed (gids[npatterns]);
 *   colibri_sharded_index_sizes / _export_index: this rank's forward index — gids[ngids] ascending, ref_off[ngids + 1], ref_sentence / ref_token[nrefs], every run in
 *   corpus order, sentences numbered globally (colibri_sharded_upload's first_sentence).
 * A pattern's reference list = the runs of its global number on rank 0, 1, ... concatenated: the ranks hold disjoint, ascending sentence ranges, so that is sorted. */
int colibri_sharded_export_gids(colibri_sharded* t, int local_rank, uint32_t* gids);
int colibri_sharded_index_sizes(colibri_sharded* t, int local_rank, uint64_t* ngids, uint64_t* nrefs);
int colibri_sharded_export_index(colibri_sharded* t, int local_rank, uint32_t* gids, uint64_t* ref_off, uint32_t* ref_sentence, uint16_t* ref_token);

#ifdef __cplusplus
}
#endif
#endif
