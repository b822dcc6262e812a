/* oracle/colibri_oracle.c — TEST INFRASTRUCTURE: plain-C CPU restatement of the reference hot path.
 *
 * This is NOT product code. It states, from scratch and in the reference's own terms (one hash map
 * keyed by pattern bytes, order loop -> sentence loop -> window loop -> look-back -> add -> prune),
 * what `PatternModel::train` computes, so that the HIP path (which computes the same thing a very
 * different way) can be checked against it. Every function cites the reference file:line it follows.
 * See colibri_oracle.h for the pinning status.
 */
#include "colibri_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * SpookyHash V2, short-message form (public-domain algorithm by Bob Jenkins).
 * reference: include/SpookyV2.h:59-66 (Hash64) -> src/SpookyV2.cpp:116-120 (Hash128 dispatches to
 * Short below 192 bytes) -> src/SpookyV2.cpp:21-113 (Short), include/SpookyV2.h:277-314 (ShortMix),
 * :328-362 (ShortEnd), :392 (sc_const). Seeds are 0/0 for Hash64(ptr,len) as called from
 * Pattern::hash (src/pattern.cpp:234-238).
 * The mixing schedules are written here as rotation tables driving one generic 4-lane step.
 * ---------------------------------------------------------------------------------------------- */
#define SPOOKY_CONST 0xdeadbeefdeadbeefULL

static inline uint64_t rotl64(uint64_t x, unsigned k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t load_le(const uint8_t* p, unsigned nbytes) { /* little-endian, 0..8 bytes */
    uint64_t v = 0;
    for (unsigned i = 0; i < nbytes; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

/* ShortMix: 12 steps; step s works on lanes (s+2)&3 [rotated], (s+3)&3 [added], s&3 [xored]. */
static const unsigned char MIX_ROT[12] = {50, 52, 30, 41, 54, 48, 38, 37, 62, 34, 5, 36};
static void short_mix(uint64_t h[4]) {
    for (int s = 0; s < 12; ++s) {
        const int r = (s + 2) & 3, a = (s + 3) & 3, x = s & 3;
        h[r] = rotl64(h[r], MIX_ROT[s]);
        h[r] += h[a];
        h[x] ^= h[r];
    }
}
/* ShortEnd: 11 steps; step s: lane t=(s+3)&3 ^= lane u=(s+2)&3; u = rot(u); t += u. */
static const unsigned char END_ROT[11] = {15, 52, 26, 51, 28, 9, 47, 54, 32, 25, 63};
static void short_end(uint64_t h[4]) {
    for (int s = 0; s < 11; ++s) {
        const int t = (s + 3) & 3, u = (s + 2) & 3;
        h[t] ^= h[u];
        h[u] = rotl64(h[u], END_ROT[s]);
        h[t] += h[u];
    }
}

uint64_t co_spooky64(const uint8_t* p, uint64_t len) {
    uint64_t h[4] = {0, 0, SPOOKY_CONST, SPOOKY_CONST}; /* a,b = seeds; c,d = sc_const */
    uint64_t left = len;
    if (len > 15) {
        while (left >= 32) { /* whole 32-byte groups */
            h[2] += load_le(p, 8);
            h[3] += load_le(p + 8, 8);
            short_mix(h);
            h[0] += load_le(p + 16, 8);
            h[1] += load_le(p + 24, 8);
            p += 32;
            left -= 32;
        }
        if (left >= 16) {
            h[2] += load_le(p, 8);
            h[3] += load_le(p + 8, 8);
            short_mix(h);
            p += 16;
            left -= 16;
        }
    }
    /* last 0..15 bytes: the reference's switch is a little-endian pack of bytes 0..7 into c and 8..14 into d */
    h[3] += len << 56;
    if (left == 0) {
        h[2] += SPOOKY_CONST;
        h[3] += SPOOKY_CONST;
    } else {
        h[2] += load_le(p, left < 8 ? (unsigned)left : 8);
        if (left > 8) h[3] += load_le(p + 8, (unsigned)left - 8);
    }
    short_end(h);
    return h[0];
}

/* ------------------------------------------------------------------------------------------------
 * gap masks — reference src/algorithms.cpp:79-94 (compute_skip_configurations), :33-54 (mask2vector)
 * ---------------------------------------------------------------------------------------------- */
static int count_gap_runs(uint32_t mask, int n) {
    int runs = 0, in = 0;
    for (int i = 0; i < n; ++i) {
        const int g = (mask >> i) & 1;
        if (g && !in) ++runs;
        in = g;
    }
    return runs;
}
int co_skip_configurations(int n, int maxskips, uint32_t* out, int cap) {
    int cnt = 0;
    if (n < 3) return 0;
    const uint32_t order = 1u << (n - 2);
    for (uint32_t i = 1; i < order; ++i) {
        const uint32_t mask = i << 1; /* never a gap at either end */
        if (n - 2 >= maxskips && count_gap_runs(mask, n) > maxskips) continue;
        if (out && cnt < cap) out[cnt] = mask;
        ++cnt;
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------------------
 * v1 -> v2 — reference src/classencoder.cpp:602-647 (per-line conversion), :22-42 (inttobytes),
 * src/classdecoder.cpp:76-82 (bytestoint_v1). v1: token = length byte (1..127) + little-endian
 * base-256 digits; 0 ends a line; 128 / 129 are the skip / flex markers; other bytes >= 130 ignored.
 * ---------------------------------------------------------------------------------------------- */
static unsigned put_varint(uint8_t* dst, uint32_t cls) {
    unsigned n = 0;
    do {
        uint8_t b = cls & 127;
        cls >>= 7;
        if (cls) b |= 128;
        if (dst) dst[n] = b;
        ++n;
    } while (cls);
    return n;
}
int co_v1_to_v2(const uint8_t* in, uint64_t nin, uint8_t* out, uint64_t* nout) {
    uint64_t i = 0, o = 0;
    while (i < nin) {
        const uint8_t c = in[i];
        if (c == 0) {
            if (out) out[o] = 0;
            ++o;
            ++i;
        } else if (c < 128) {
            if (i + 1 + c > nin) return -1;
            uint32_t cls = 0;
            for (unsigned k = 0; k < c && k < 4; ++k) cls |= (uint32_t)in[i + 1 + k] << (8 * k);
            o += put_varint(out ? out + o : NULL, cls);
            i += (uint64_t)c + 1;
        } else if (c == 128 || c == 129) {
            if (out) out[o] = (c == 128) ? 3 : 4;
            ++o;
            ++i;
        } else {
            ++i;
        }
    }
    *nout = o;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * The pattern map: PatternMap<ValueType> = std::unordered_map<Pattern,V> keyed by key bytes, hashed
 * with Spooky (reference include/patternstore.h:937-1011, include/pattern.h:563-597). Distinct keys
 * never merge: equality is byte equality (src/pattern.cpp:990-1004).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t sentence;
    uint16_t token;
} co_ref;

typedef struct {
    uint64_t hash;
    uint64_t keyoff;
    uint32_t keylen;
    uint32_t count; /* unindexed: the count; indexed: nrefs */
    uint16_t n;     /* tokens */
    uint8_t  skipgram;
    uint8_t  alive;
    co_ref*  refs; /* indexed only */
    uint32_t refcap;
    uint32_t nsrc; /* indexed skipgrams: number of distinct source n-grams */
} co_entry;

struct co_model {
    co_entry* e;
    uint64_t  ne, ecap;
    uint64_t* bucket; /* index+1 into e, 0 = empty */
    uint64_t  nbucket;
    uint8_t*  arena;
    uint64_t  arenalen, arenacap;
    uint64_t  alive;
    uint64_t  totaltokens, totaltypes, windows;
    int       maxn, indexed;
    int64_t   stat[3][128];
};

static void map_grow(co_model* m) {
    const uint64_t nb = m->nbucket ? m->nbucket * 2 : 1024;
    uint64_t*      b  = (uint64_t*)calloc(nb, sizeof(uint64_t));
    for (uint64_t i = 0; i < m->ne; ++i) {
        uint64_t s = m->e[i].hash & (nb - 1);
        while (b[s]) s = (s + 1) & (nb - 1);
        b[s] = i + 1;
    }
    free(m->bucket);
    m->bucket  = b;
    m->nbucket = nb;
}

static co_entry* map_find(const co_model* m, const uint8_t* key, uint32_t len, uint64_t h) {
    if (!m->nbucket) return NULL;
    uint64_t s = h & (m->nbucket - 1);
    while (m->bucket[s]) {
        co_entry* e = &m->e[m->bucket[s] - 1];
        if (e->hash == h && e->keylen == len && memcmp(m->arena + e->keyoff, key, len) == 0) return e;
        s = (s + 1) & (m->nbucket - 1);
    }
    return NULL;
}

/* PatternMap::has (patternstore.h:957-959): present and not erased */
static int map_has(const co_model* m, const uint8_t* key, uint32_t len) {
    const co_entry* e = map_find(m, key, len, co_spooky64(key, len));
    return e && e->alive;
}

/* getdata(pattern, makeifnew=true) (patternmodel.h:1675-1684) */
static co_entry* map_get_or_insert(co_model* m, const uint8_t* key, uint32_t len, uint16_t n, int skipgram) {
    const uint64_t h = co_spooky64(key, len);
    co_entry*      e = map_find(m, key, len, h);
    if (e) {
        if (!e->alive) { /* re-insert of an erased key */
            e->alive = 1;
            e->count = 0;
            ++m->alive;
        }
        return e;
    }
    if ((m->ne + 1) * 2 > m->nbucket) map_grow(m);
    if (m->ne == m->ecap) {
        m->ecap = m->ecap ? m->ecap * 2 : 1024;
        m->e    = (co_entry*)realloc(m->e, m->ecap * sizeof(co_entry));
    }
    if (m->arenalen + len > m->arenacap) {
        m->arenacap = (m->arenacap + len) * 2 + 4096;
        m->arena    = (uint8_t*)realloc(m->arena, m->arenacap);
    }
    e = &m->e[m->ne];
    memset(e, 0, sizeof *e);
    e->hash     = h;
    e->keyoff   = m->arenalen;
    e->keylen   = len;
    e->n        = n;
    e->skipgram = (uint8_t)skipgram;
    e->alive    = 1;
    memcpy(m->arena + m->arenalen, key, len);
    m->arenalen += len;
    uint64_t s = h & (m->nbucket - 1);
    while (m->bucket[s]) s = (s + 1) & (m->nbucket - 1);
    m->bucket[s] = ++m->ne;
    ++m->alive;
    return e;
}

/* valuehandler.add: +1 (datatypes.h:228-230) or push_back(ref) (datatypes.h:283-289) */
static void entry_add(co_model* m, co_entry* e, co_ref ref) {
    if (m->indexed) {
        if (e->count == e->refcap) {
            e->refcap = e->refcap ? e->refcap * 2 : 4;
            e->refs   = (co_ref*)realloc(e->refs, e->refcap * sizeof(co_ref));
        }
        e->refs[e->count] = ref;
    }
    ++e->count;
}

static void entry_erase(co_model* m, co_entry* e) {
    e->alive = 0;
    free(e->refs);
    e->refs   = NULL;
    e->refcap = 0;
    e->count  = 0;
    --m->alive;
}

/* prune(threshold, n) (patternmodel.h:2107-2128): erase size-n patterns (any category) under threshold */
static uint64_t prune(co_model* m, uint32_t threshold, int n) {
    uint64_t pruned = 0;
    for (uint64_t i = 0; i < m->ne; ++i) {
        co_entry* e = &m->e[i];
        if (e->alive && e->n == n && e->count < threshold) {
            entry_erase(m, e);
            ++pruned;
        }
    }
    return pruned;
}

/* base pruneskipgrams (patternmodel.h:2167-2186): no-op when minskiptypes <= 1, else threshold only */
static uint64_t pruneskipgrams_unindexed(co_model* m, uint32_t threshold, int minskiptypes, int n) {
    uint64_t pruned = 0;
    if (minskiptypes <= 1) return 0;
    for (uint64_t i = 0; i < m->ne; ++i) {
        co_entry* e = &m->e[i];
        if (e->alive && e->n == n && e->skipgram && e->count < threshold) {
            entry_erase(m, e);
            ++pruned;
        }
    }
    return pruned;
}

/* ------------------------------------------------------------------------------------------------
 * Sentence / token scanning. A token ends at a byte < 128; a byte 0 that does not follow a high
 * byte is the sentence delimiter (src/pattern.cpp:74-105 datasize, :1947-1958 sentence index,
 * :483-518 streaming line read). Every delimiter closes a sentence, empty ones included.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t* start; /* start[k] = byte offset of token k; start[ntok] = end of last token */
    uint32_t  ntok, cap;
} tokbuf;

static void tok_push(tokbuf* t, uint64_t off) {
    if (t->ntok + 1 >= t->cap) {
        t->cap   = t->cap ? t->cap * 2 : 256;
        t->start = (uint64_t*)realloc(t->start, t->cap * sizeof(uint64_t));
    }
    t->start[t->ntok++] = off;
}

/* scans one sentence starting at *pos; returns 0 at end of data */
static int next_sentence(const uint8_t* p, uint64_t nbytes, uint64_t* pos, tokbuf* t) {
    if (*pos >= nbytes) return 0;
    uint64_t i = *pos, tokstart = *pos;
    int      prevhigh = 0;
    t->ntok = 0;
    while (i < nbytes) {
        const uint8_t c = p[i];
        if (!prevhigh && c == 0) break; /* delimiter */
        if (c < 128) {
            tok_push(t, tokstart);
            tokstart = i + 1;
            prevhigh = 0;
        } else {
            prevhigh = 1;
        }
        ++i;
    }
    /* sentinel: end of the last complete token */
    if (t->ntok + 1 >= t->cap) {
        t->cap   = t->cap ? t->cap * 2 : 256;
        t->start = (uint64_t*)realloc(t->start, t->cap * sizeof(uint64_t));
    }
    t->start[t->ntok] = tokstart;
    *pos              = (i < nbytes) ? i + 1 : i;
    return 1;
}

/* Pattern(PatternPointer) for a skipgram (src/pattern.cpp:886-908): every gapped token becomes the
 * single byte 03 (skipclass); other tokens are copied. Returns the key length. */
static uint32_t materialise_skipgram(const uint8_t* p, const uint64_t* start, int n, uint32_t mask, uint8_t* out) {
    uint32_t o = 0;
    for (int k = 0; k < n; ++k) {
        if ((mask >> k) & 1) {
            out[o++] = 3;
        } else {
            const uint32_t len = (uint32_t)(start[k + 1] - start[k]);
            memcpy(out + o, p + start[k], len);
            o += len;
        }
    }
    return o;
}

static int cmp_ref(const void* a, const void* b) {
    const co_ref *x = (const co_ref*)a, *y = (const co_ref*)b;
    if (x->sentence != y->sentence) return x->sentence < y->sentence ? -1 : 1;
    if (x->token != y->token) return x->token < y->token ? -1 : 1;
    return 0;
}

/* IndexedPatternModel::trainskipgrams (patternmodel.h:2969-3010) with computeskipgrams
 * (:1370-1527, multiplerefs branch :1508-1512) and the derived pruneskipgrams (:3362-3383) whose
 * criterion is the number of distinct skip contents (getskipcontent :3029-3059) = number of distinct
 * plain n-grams spanning first..last gap over the skipgram's occurrences. The reference inserts into
 * the map it iterates; the specification followed here is the clean one (iterate the n-grams that
 * existed when the order started), see SURVEY.md §8 a-10. */
static void train_indexed_skipgrams(co_model* m, const co_options* opt, uint32_t thr) {
    uint32_t* masks = NULL;
    uint8_t  keybuf[4096];
    uint64_t tokstart[128];
    for (int n = 3; n <= opt->maxlength && n < 32; ++n) { /* a gap mask is a uint32_t: patterns of up to 31 tokens (include/pattern.h:368) */
        const int nmasks = co_skip_configurations(n, opt->maxskips, NULL, 0);
        masks            = (uint32_t*)realloc(masks, sizeof(uint32_t) * (size_t)(nmasks + 1));
        co_skip_configurations(n, opt->maxskips, masks, nmasks);
        const uint64_t ne0    = m->ne; /* the n-grams that exist when this order starts */
        uint64_t       found  = 0;
        for (uint64_t i = 0; i < ne0; ++i) {
            if (!(m->e[i].alive && m->e[i].n == n && !m->e[i].skipgram)) continue;
            for (int q = 0; q < nmasks; ++q) {
                const uint8_t* key = m->arena + m->e[i].keyoff; /* re-fetched: arena / entries may move on insert */
                int            k   = 0;
                tokstart[0]        = 0;
                for (uint32_t b = 0; b < m->e[i].keylen; ++b)
                    if (key[b] < 128) tokstart[++k] = b + 1;
                const uint32_t klen = materialise_skipgram(key, tokstart, n, masks[q], keybuf);
                if (!map_has(m, keybuf, klen)) ++found; /* :1504-1505 */
                co_entry* s = map_get_or_insert(m, keybuf, klen, (uint16_t)n, 1);
                ++s->nsrc; /* one more distinct source n-gram = one more distinct skip content */
                const co_entry* src = &m->e[i];
                for (uint32_t r = 0; r < src->count; ++r) entry_add(m, s, src->refs[r]); /* :1508-1512 */
            }
        }
        if (!found) break; /* " None found" :2992-2994 */
        uint64_t pruned = prune(m, thr, n); /* :3000 */
        /* derived pruneskipgrams(int, minskiptypes, n) :3362-3383 — threshold argument ignored */
        if (opt->minskiptypes > 1) {
            for (uint64_t i = 0; i < m->ne; ++i) {
                co_entry* e = &m->e[i];
                if (e->alive && e->n == n && e->skipgram && (int)e->nsrc < opt->minskiptypes) {
                    entry_erase(m, e);
                    ++pruned;
                }
            }
        }
        if (n < 128) {
            m->stat[0][n] += (int64_t)found;
            m->stat[1][n] += (int64_t)pruned;
            m->stat[2][n] += (int64_t)found - (int64_t)pruned;
        }
        if (n > m->maxn) m->maxn = n;
    }
    free(masks);
}

/* ------------------------------------------------------------------------------------------------
 * PatternModel::train — reference include/patternmodel.h:880-1345, accelerated subset
 * (constrainbymodel==NULL, filter==NULL, continued==false, DOPATTERNPERLINE==false, MINTOKENS>=2,
 * MINLENGTH==1, MAXBACKOFFLENGTH>=MAXLENGTH; MINTOKENS_UNIGRAMS > MINTOKENS without skipgrams).
 * ---------------------------------------------------------------------------------------------- */
co_model* co_train(const uint8_t* payload, uint64_t nbytes, const co_options* opt_in, uint32_t firstsentence) {
    co_options opt = *opt_in;
    if (opt.mintokens == -1) opt.mintokens = 2; /* :883-886 */
    if (opt.mintokens == 0) opt.mintokens = 1;
    if (opt.mintokens_skipgrams < opt.mintokens) opt.mintokens_skipgrams = opt.mintokens; /* :887-888 */
    /* MINTOKENS == 1 makes the reference count all lengths in one pass without look-back (:1069-1072); with nothing ever pruned the
     * order loop below admits every window and gives the same model. Skipgrams at threshold 1 stay outside the restated subset. */
    if (opt.mintokens < 1) return NULL;
    const uint32_t thr = (uint32_t)opt.mintokens;
    /* secondary word threshold (:1090-1104): with MINLENGTH == 1 and MINTOKENS > 1 the unigrams themselves are still pruned at MINTOKENS
     * (:1220), but a longer window is only counted if every one of its words occurs at least MINTOKENS_UNIGRAMS times */
    const uint32_t wthr = (opt.mintokens_unigrams > opt.mintokens) ? (uint32_t)opt.mintokens_unigrams : 0u;
    if (wthr && opt.mintokens < 2) return NULL;

    co_model* m = (co_model*)calloc(1, sizeof(co_model));
    m->indexed  = opt.indexed;
    tokbuf   t  = {0};
    uint32_t* masks = NULL;
    uint8_t  keybuf[4096];
    uint64_t prevsize = 0;

    for (int n = 1; n <= opt.maxlength; ++n) { /* ORDER LOOP :981 */
        uint64_t foundskipgrams = 0;
        int      nmasks         = 0;
        if (opt.doskipgrams_exhaustive && n >= 3 && n < 32) { /* :1021-1022; a gap mask is a uint32_t: up to 31 tokens */
            nmasks = co_skip_configurations(n, opt.maxskips, NULL, 0);
            masks  = (uint32_t*)realloc(masks, sizeof(uint32_t) * (size_t)(nmasks + 1));
            co_skip_configurations(n, opt.maxskips, masks, nmasks);
        }
        uint32_t sentence = firstsentence - 1;
        uint64_t pos      = 0;
        while (next_sentence(payload, nbytes, &pos, &t)) { /* SENTENCE LOOP :1030 */
            ++sentence;
            if (t.ntok == 0) continue;                    /* :1042-1045 */
            if (n == 1) m->totaltokens += t.ntok;         /* :1047-1048 */
            if ((uint32_t)n > t.ntok) continue;           /* ngrams(): n > _n -> none (pattern.cpp:1286-1287) */
            for (uint32_t i = 0; i + n <= t.ntok; ++i) {  /* WINDOW LOOP :1078 */
                ++m->windows;
                const uint8_t* w    = payload + t.start[i];
                const uint32_t wlen = (uint32_t)(t.start[i + n] - t.start[i]);
                int            found = 1;
                if (n > 1 && wthr) { /* unigram check :1093-1104 */
                    for (int k = 0; k < n && found; ++k) {
                        const uint8_t*  u  = payload + t.start[i + k];
                        const uint32_t  ul = (uint32_t)(t.start[i + k + 1] - t.start[i + k]);
                        const co_entry* e  = map_find(m, u, ul, co_spooky64(u, ul));
                        if (!(e && e->alive && e->count >= wthr)) found = 0;
                    }
                }
                if (found && n > 1 && thr > 1) { /* look-back :1139-1152: every sub-pattern of backoffn = min(n-1, MAXBACKOFFLENGTH) tokens must be in the model
                                                    (both (n-1)-grams when the back-off length does not bite); no look-back at MINTOKENS = 1 */
                    int backoffn = n - 1;
                    if (opt.maxbackofflength > 0 && backoffn > opt.maxbackofflength) backoffn = opt.maxbackofflength;
                    for (int k = 0; k + backoffn <= n && found; ++k) {
                        const uint32_t l = (uint32_t)(t.start[i + k + backoffn] - t.start[i + k]);
                        found = map_has(m, payload + t.start[i + k], l);
                    }
                }
                const co_ref ref = {sentence, (uint16_t)i}; /* :1155 */
                if (found) entry_add(m, map_get_or_insert(m, w, wlen, (uint16_t)n, 0), ref); /* :1160 */
                if (n >= 3 && opt.doskipgrams_exhaustive) { /* :1163-1171 -> computeskipgrams :1370-1527 */
                    /* validity as executed: slices drop the mask (pattern.cpp:853-855), so the test is
                     * has(both plain (n-1)-grams) = `found` above, evaluated for every window */
                    if (found) {
                        for (int q = 0; q < nmasks; ++q) {
                            const uint32_t klen = materialise_skipgram(payload, t.start + i, n, masks[q], keybuf);
                            if (!map_has(m, keybuf, klen)) ++foundskipgrams; /* :1504-1505 */
                            entry_add(m, map_get_or_insert(m, keybuf, klen, (uint16_t)n, 1), ref);
                        }
                    }
                }
            }
        }
        const int64_t foundngrams = (int64_t)m->alive - (int64_t)foundskipgrams - (int64_t)prevsize; /* :1182 */
        if (foundngrams || foundskipgrams) {
            if (n > m->maxn) m->maxn = n;
        } else {
            break; /* "None found" :1189-1194 */
        }
        if (n == 1) m->totaltypes = m->alive; /* :1199-1201, before pruning */
        uint64_t pruned = prune(m, thr, n);   /* :1220 */
        if (foundskipgrams) pruned += pruneskipgrams_unindexed(m, (uint32_t)opt.mintokens_skipgrams, opt.minskiptypes, n); /* :1233-1243 */
        if (n < 128) {
            m->stat[0][n] = foundngrams + (int64_t)foundskipgrams;
            m->stat[1][n] = (int64_t)pruned;
            m->stat[2][n] = foundngrams + (int64_t)foundskipgrams - (int64_t)pruned;
        }
        prevsize = m->alive; /* :1269 */
    }
    if (opt.doskipgrams && !opt.doskipgrams_exhaustive && opt.indexed) train_indexed_skipgrams(m, &opt, thr); /* :1271-1273 */
    if (opt.indexed) { /* posttrain: sort every index (patternmodel.h:2699-2705) */
        for (uint64_t i = 0; i < m->ne; ++i)
            if (m->e[i].alive) qsort(m->e[i].refs, m->e[i].count, sizeof(co_ref), cmp_ref);
    }
    free(t.start);
    free(masks);
    return m;
}

void co_free(co_model* m) {
    if (!m) return;
    for (uint64_t i = 0; i < m->ne; ++i) free(m->e[i].refs);
    free(m->e);
    free(m->bucket);
    free(m->arena);
    free(m);
}

uint64_t co_npatterns(const co_model* m) { return m->alive; }
uint64_t co_totaltokens(const co_model* m) { return m->totaltokens; }
uint64_t co_totaltypes(const co_model* m) { return m->totaltypes; }
int      co_maxn(const co_model* m) { return m->maxn; }
uint64_t co_windows(const co_model* m) { return m->windows; }
uint64_t co_keybytes(const co_model* m) {
    uint64_t s = 0;
    for (uint64_t i = 0; i < m->ne; ++i)
        if (m->e[i].alive) s += m->e[i].keylen;
    return s;
}
uint64_t co_nrefs(const co_model* m) {
    uint64_t s = 0;
    if (!m->indexed) return 0;
    for (uint64_t i = 0; i < m->ne; ++i)
        if (m->e[i].alive) s += m->e[i].count;
    return s;
}
int64_t co_order_stat(const co_model* m, int n, int which) {
    if (n < 0 || n >= 128 || which < 0 || which > 2) return -1;
    return m->stat[which][n];
}

static const co_model* g_sort_model;
static int cmp_entry(const void* a, const void* b) {
    const co_entry* x = &g_sort_model->e[*(const uint64_t*)a];
    const co_entry* y = &g_sort_model->e[*(const uint64_t*)b];
    const uint32_t  l = x->keylen < y->keylen ? x->keylen : y->keylen;
    const int       c = memcmp(g_sort_model->arena + x->keyoff, g_sort_model->arena + y->keyoff, l);
    if (c) return c;
    return (x->keylen > y->keylen) - (x->keylen < y->keylen);
}

void co_export(const co_model* m, uint64_t* key_off, uint8_t* key_bytes, uint32_t* counts, uint64_t* ref_off, uint32_t* ref_sentence,
               uint16_t* ref_token) {
    uint64_t* idx = (uint64_t*)malloc((m->alive + 1) * sizeof(uint64_t));
    uint64_t  k   = 0;
    for (uint64_t i = 0; i < m->ne; ++i)
        if (m->e[i].alive) idx[k++] = i;
    g_sort_model = m;
    qsort(idx, k, sizeof(uint64_t), cmp_entry);
    uint64_t ko = 0, ro = 0;
    for (uint64_t j = 0; j < k; ++j) {
        const co_entry* e = &m->e[idx[j]];
        key_off[j]        = ko;
        memcpy(key_bytes + ko, m->arena + e->keyoff, e->keylen);
        ko += e->keylen;
        counts[j] = e->count;
        if (m->indexed && ref_off) {
            ref_off[j] = ro;
            for (uint32_t r = 0; r < e->count; ++r) {
                ref_sentence[ro] = e->refs[r].sentence;
                ref_token[ro]    = e->refs[r].token;
                ++ro;
            }
        }
    }
    key_off[k] = ko;
    if (m->indexed && ref_off) ref_off[k] = ro;
    free(idx);
}
