/* oracle/colibri_oracle.h — TEST INFRASTRUCTURE (CPU restatement of the reference hot path).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker. Nothing under colibri-core_amd/ links, loads or executes it.
 *
 * Parity status: PINNED. liboracle.so is checked (tests/test_oracle.py) against
 *   - the reference's own fixtures and known answers (exp/hamlet.v1.colibri.dat +
 *     exp/hamlet.v1.colibri.patternmodel: 111 patterns / 186 types / 354 tokens; src/test.cpp:1214-1221,
 *     :1268-1283 (385), :1327-1337 (133); SpookyHash values), committed as tests/golden/;
 *   - outputs of the real reference built from its own sources (oracle/_ref/ref_driver), on hamlet,
 *     crafted edge corpora and seeded Zipf corpora, in six training modes.
 */
#ifndef COLIBRI_ORACLE_H
#define COLIBRI_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirror of the PatternModelOptions fields the hot path reads (reference include/patternmodel.h:103-213). */
typedef struct co_options {
    int32_t mintokens;              /* MINTOKENS (-1 -> 2, 0 -> 1)                        */
    int32_t maxlength;              /* MAXLENGTH                                           */
    int32_t mintokens_skipgrams;    /* MINTOKENS_SKIPGRAMS (raised to MINTOKENS if lower)  */
    int32_t minskiptypes;           /* MINSKIPTYPES                                        */
    int32_t maxskips;               /* MAXSKIPS                                            */
    int32_t doskipgrams;            /* DOSKIPGRAMS (indexed models only)                   */
    int32_t doskipgrams_exhaustive; /* DOSKIPGRAMS_EXHAUSTIVE                              */
    int32_t indexed;                /* 0: PatternModel<uint32_t>, 1: IndexedPatternModel<> */
    int32_t mintokens_unigrams;     /* MINTOKENS_UNIGRAMS (-W): words of longer patterns must occur this often */
    int32_t maxbackofflength;       /* MAXBACKOFFLENGTH (-b): the look-back checks the sub-patterns of min(n-1, this) tokens; 0 = no limit */
} co_options;

typedef struct co_model co_model;

/* SpookyHash V2 Hash64, seed 0 (reference include/SpookyV2.h:59-66 -> src/SpookyV2.cpp:21-113). len < 192. */
uint64_t co_spooky64(const uint8_t* data, uint64_t len);

/* Skipgram gap masks for patterns of n tokens (reference src/algorithms.cpp:79-94). Returns count. */
int co_skip_configurations(int n, int maxskips, uint32_t* out, int cap);

/* v1 -> v2 corpus conversion (reference src/classencoder.cpp:602-765). out may be NULL to size. */
int co_v1_to_v2(const uint8_t* in, uint64_t nin, uint8_t* out, uint64_t* nout);

/* PatternModel::train on a v2 payload (file minus the 2-byte header); reference patternmodel.h:880-1345. */
co_model* co_train(const uint8_t* payload, uint64_t nbytes, const co_options* opt, uint32_t firstsentence);
void      co_free(co_model* m);

uint64_t co_npatterns(const co_model* m);
uint64_t co_totaltokens(const co_model* m);
uint64_t co_totaltypes(const co_model* m);
uint64_t co_keybytes(const co_model* m); /* sum of key byte lengths            */
uint64_t co_nrefs(const co_model* m);    /* sum of index lengths (indexed)     */
int      co_maxn(const co_model* m);
uint64_t co_windows(const co_model* m);  /* sum over orders of windows visited */
/* per-order training log (what the reference prints on stderr, patternmodel.h:1195-1245): which = 0 found, 1 pruned, 2 kept */
int64_t  co_order_stat(const co_model* m, int n, int which);

/* Canonical export: patterns sorted by key bytes (memcmp, shorter first on ties).
 * key_off[np+1], key_bytes[co_keybytes], counts[np]; for indexed: ref_off[np+1], ref_sentence[nrefs], ref_token[nrefs]. */
void co_export(const co_model* m, uint64_t* key_off, uint8_t* key_bytes, uint32_t* counts, uint64_t* ref_off, uint32_t* ref_sentence,
               uint16_t* ref_token);

#ifdef __cplusplus
}
#endif
#endif
