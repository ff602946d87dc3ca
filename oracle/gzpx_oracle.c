/*
 * gzpx_oracle.c -- CPU restatement (plain C) of the gzp ParCompress<Bgzf/Mgzip> hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see gzpx_oracle.h).  Scalar, sequential, written the way the
 * reference's backend works (one block at a time, hash table walked position by position) so
 * that it checks the GPU path's *parallel* reformulation independently.
 *
 * Reference anchors (relative to /root/reference):
 *   framing        src/bgzf.rs:204-237,274-303 ; src/mgzip.rs:187-218,246-275
 *   EOF marker     src/bgzf.rs:24-38 ; appended by Bgzf::encode src/deflate.rs:622-624
 *   stream cutting src/par/compress.rs:413-463 (write), :332-362 (flush_last), :377-388 (finish)
 *   arithmetic     libdeflate (Cargo.lock:414-430, libdeflate-sys 1.24.0; source not vendored):
 *                  call sites src/bgzf.rs:214-216 (deflate_compress), :224-225 (crc32).
 *                  Restated from SURVEY.md Appendix A.0-A.6; pinned against the v1.10 binary.
 */
#include "gzpx_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------ CRC-32 */

static uint32_t crc_table[256];
static int crc_table_ready;

static void crc_init(void)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++)
            c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        crc_table[i] = c;
    }
    crc_table_ready = 1;
}

/* libdeflate_crc32 semantics (src/bgzf.rs:224-225 via libdeflater::Crc). */
uint32_t gzpx_oracle_crc32(uint32_t crc, const void *buf, size_t n)
{
    const uint8_t *p = (const uint8_t *)buf;
    if (!crc_table_ready)
        crc_init();
    crc = ~crc;
    for (size_t i = 0; i < n; i++)
        crc = (crc >> 8) ^ crc_table[(crc ^ p[i]) & 0xFF];
    return ~crc;
}

/* ------------------------------------------------------------------ DEFLATE constants */

#define NUM_LITLEN_SYMS 288
#define NUM_OFFSET_SYMS 32
#define NUM_PRECODE_SYMS 19
#define MAX_LITLEN_CODEWORD_LEN 14
#define MAX_OFFSET_CODEWORD_LEN 15
#define MAX_PRE_CODEWORD_LEN 7
#define END_OF_BLOCK 256
#define FIRST_LEN_SYM 257
#define MIN_MATCH_LEN 3
#define MAX_MATCH_LEN 258
#define WINDOW_SIZE 32768

#define MIN_BLOCK_LENGTH 5000
#define FAST_SOFT_MAX_BLOCK_LENGTH 65535
#define FAST_SEQ_STORE_LENGTH 8192

static const unsigned length_slot_base[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,
                                              15, 17, 19, 23, 27, 31, 35, 43, 51,  59,
                                              67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t extra_length_bits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                              2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const unsigned offset_slot_base[30] = {1,    2,    3,    4,    5,    7,     9,     13,
                                              17,   25,   33,   49,   65,   97,    129,   193,
                                              257,  385,  513,  769,  1025, 1537,  2049,  3073,
                                              4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t extra_offset_bits[30] = {0, 0, 0, 0, 1, 1, 2,  2,  3,  3,  4,  4,  5,  5,  6,
                                              6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t extra_precode_bits[NUM_PRECODE_SYMS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                                             0, 0, 0, 0, 0, 0, 2, 3, 7};
static const uint8_t precode_lens_permutation[NUM_PRECODE_SYMS] = {
    16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* slot lookup tables (libdeflate keeps the same kind of tables), built on first use */
static uint8_t length_slot_tab[MAX_MATCH_LEN + 1];
static uint8_t offset_slot_tab[32768 + 1];
static int slot_tabs_ready;

static void init_slot_tabs(void)
{
    for (unsigned len = MIN_MATCH_LEN; len <= MAX_MATCH_LEN; len++) {
        unsigned s = 28;
        while (length_slot_base[s] > len)
            s--;
        length_slot_tab[len] = (uint8_t)s;
    }
    for (unsigned off = 1; off <= 32768; off++) {
        unsigned s = 29;
        while (offset_slot_base[s] > off)
            s--;
        offset_slot_tab[off] = (uint8_t)s;
    }
    slot_tabs_ready = 1;
}

static inline unsigned length_slot(unsigned len) { return length_slot_tab[len]; }
static inline unsigned offset_slot(unsigned off) { return offset_slot_tab[off]; }

/* ------------------------------------------------------------------ output bitstream */

struct bitwriter {
    uint8_t *out;
    size_t cap;
    size_t pos;      /* bytes written */
    uint64_t bitbuf; /* pending bits, LSB first */
    unsigned bitcount;
    int overflow;
};

static void bw_put_byte(struct bitwriter *w, uint8_t b)
{
    if (w->pos < w->cap)
        w->out[w->pos] = b;
    else
        w->overflow = 1;
    w->pos++;
}

static void bw_add(struct bitwriter *w, uint32_t bits, unsigned n)
{
    w->bitbuf |= (uint64_t)bits << w->bitcount;
    w->bitcount += n;
    while (w->bitcount >= 8) {
        bw_put_byte(w, (uint8_t)w->bitbuf);
        w->bitbuf >>= 8;
        w->bitcount -= 8;
    }
}

static void bw_align(struct bitwriter *w)
{
    if (w->bitcount) {
        bw_put_byte(w, (uint8_t)w->bitbuf);
        w->bitbuf = 0;
        w->bitcount = 0;
    }
}

/* ------------------------------------------------------------------ Huffman code construction (A.6) */

#define NUM_SYMBOL_BITS 10
#define SYMBOL_MASK ((1u << NUM_SYMBOL_BITS) - 1)

static uint32_t reverse_codeword(uint32_t cw, unsigned len)
{
    uint32_t r = 0;
    for (unsigned i = 0; i < len; i++)
        r |= ((cw >> i) & 1u) << (len - 1 - i);
    return r;
}

/* entries: sym | (key << NUM_SYMBOL_BITS); sorted ascending by (freq, sym). */
static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

void gzpx_oracle_make_huffman_code(unsigned num_syms, unsigned max_len, int compat,
                                   const uint32_t *freqs, uint8_t *lens, uint32_t *codewords)
{
    uint64_t A[NUM_LITLEN_SYMS];     /* sorted leaves: (freq << 10) | sym              */
    uint64_t NF[NUM_LITLEN_SYMS];    /* internal node frequencies                      */
    unsigned parent[NUM_LITLEN_SYMS]; /* parent index of internal node                  */
    unsigned depth[NUM_LITLEN_SYMS];
    unsigned len_counts[16];
    unsigned num_used = 0;

    for (unsigned s = 0; s < num_syms; s++) {
        lens[s] = 0;
        codewords[s] = 0;
        if (freqs[s])
            A[num_used++] = ((uint64_t)freqs[s] << NUM_SYMBOL_BITS) | s;
    }
    qsort(A, num_used, sizeof(A[0]), cmp_u64);

    if (num_used == 0 && compat == GZPX_ORACLE_COMPAT_1_10)
        return; /* v1.10: empty code, all lengths zero (SURVEY A.7 delta 1) */
    if (num_used < 2) {
        unsigned sym = num_used ? (unsigned)(A[0] & SYMBOL_MASK) : 0;
        unsigned other = sym ? sym : 1;
        lens[0] = 1;
        codewords[0] = 0;
        lens[other] = 1;
        codewords[other] = 1;
        return;
    }

    /* two-queue tree build; ties prefer leaves */
    {
        const unsigned last = num_used - 1;
        unsigned i = 0, b = 0, e = 0;
        do {
            uint64_t nf;
            if (i + 1 <= last && (b == e || (A[i + 1] >> NUM_SYMBOL_BITS) <= NF[b])) {
                nf = (A[i] >> NUM_SYMBOL_BITS) + (A[i + 1] >> NUM_SYMBOL_BITS);
                i += 2;
            } else if (b + 2 <= e && (i > last || NF[b + 1] < (A[i] >> NUM_SYMBOL_BITS))) {
                nf = NF[b] + NF[b + 1];
                parent[b] = e;
                parent[b + 1] = e;
                b += 2;
            } else {
                nf = (A[i] >> NUM_SYMBOL_BITS) + NF[b];
                parent[b] = e;
                i++;
                b++;
            }
            NF[e] = nf;
        } while (++e < last);

        /* length counts from internal-node depths, clamping over-long codewords */
        for (unsigned l = 0; l <= max_len; l++)
            len_counts[l] = 0;
        len_counts[1] = 2;
        const unsigned root = last - 1;
        depth[root] = 0;
        for (int node = (int)root - 1; node >= 0; node--) {
            unsigned d = depth[parent[node]] + 1;
            unsigned l = d;
            depth[node] = d;
            if (l >= max_len) {
                l = max_len;
                do {
                    l--;
                } while (len_counts[l] == 0);
            }
            len_counts[l]--;
            len_counts[l + 1] += 2;
        }
    }

    /* longest lengths to the least frequent symbols */
    {
        unsigned k = 0;
        for (unsigned l = max_len; l >= 1; l--)
            for (unsigned c = len_counts[l]; c; c--)
                lens[A[k++] & SYMBOL_MASK] = (uint8_t)l;
    }
    /* canonical codewords, bit-reversed */
    {
        uint32_t next[16];
        next[0] = 0;
        next[1] = 0;
        for (unsigned l = 2; l <= max_len; l++)
            next[l] = (next[l - 1] + len_counts[l - 1]) << 1;
        for (unsigned s = 0; s < num_syms; s++)
            if (lens[s])
                codewords[s] = reverse_codeword(next[lens[s]]++, lens[s]);
    }
}

/* ------------------------------------------------------------------ block flush (A.5) */

struct codes {
    uint8_t litlen_lens[NUM_LITLEN_SYMS + NUM_OFFSET_SYMS]; /* offset lens may be moved adjacent */
    uint8_t offset_lens[NUM_OFFSET_SYMS];
    uint32_t litlen_cw[NUM_LITLEN_SYMS];
    uint32_t offset_cw[NUM_OFFSET_SYMS];
};

struct freqs {
    uint32_t litlen[NUM_LITLEN_SYMS];
    uint32_t offset[NUM_OFFSET_SYMS];
};

static struct codes static_codes;
static int static_codes_ready;

static void init_static_codes(void)
{
    uint32_t f[NUM_LITLEN_SYMS];
    unsigned i;
    /* frequencies chosen so that make_code yields the fixed code lengths (libdeflate idiom) */
    for (i = 0; i < 144; i++)
        f[i] = 1 << (9 - 8);
    for (; i < 256; i++)
        f[i] = 1 << (9 - 9);
    for (; i < 280; i++)
        f[i] = 1 << (9 - 7);
    for (; i < 288; i++)
        f[i] = 1 << (9 - 8);
    gzpx_oracle_make_huffman_code(NUM_LITLEN_SYMS, MAX_LITLEN_CODEWORD_LEN, 0, f,
                                  static_codes.litlen_lens, static_codes.litlen_cw);
    for (i = 0; i < NUM_OFFSET_SYMS; i++)
        f[i] = 1;
    gzpx_oracle_make_huffman_code(NUM_OFFSET_SYMS, MAX_OFFSET_CODEWORD_LEN, 0, f,
                                  static_codes.offset_lens, static_codes.offset_cw);
    static_codes_ready = 1;
}

/* precode run-length items: sym | extra << 5 */
static unsigned compute_precode_items(const uint8_t *lens, unsigned num_lens, uint32_t *pfreq,
                                      unsigned *items)
{
    unsigned n = 0, run_start = 0;
    memset(pfreq, 0, NUM_PRECODE_SYMS * sizeof(pfreq[0]));
    do {
        uint8_t len = lens[run_start];
        unsigned run_end = run_start, extra;
        do {
            run_end++;
        } while (run_end != num_lens && len == lens[run_end]);
        if (len == 0) {
            while (run_end - run_start >= 11) {
                extra = run_end - run_start - 11;
                if (extra > 0x7F)
                    extra = 0x7F;
                pfreq[18]++;
                items[n++] = 18 | (extra << 5);
                run_start += 11 + extra;
            }
            if (run_end - run_start >= 3) {
                extra = run_end - run_start - 3;
                if (extra > 0x7)
                    extra = 0x7;
                pfreq[17]++;
                items[n++] = 17 | (extra << 5);
                run_start += 3 + extra;
            }
        } else if (run_end - run_start >= 4) {
            pfreq[len]++;
            items[n++] = len;
            run_start++;
            do {
                extra = run_end - run_start - 3;
                if (extra > 0x3)
                    extra = 0x3;
                pfreq[16]++;
                items[n++] = 16 | (extra << 5);
                run_start += 3 + extra;
            } while (run_end - run_start >= 3);
        }
        while (run_start != run_end) {
            pfreq[len]++;
            items[n++] = len;
            run_start++;
        }
    } while (run_start != num_lens);
    return n;
}

#define TOKEN_MATCH 0x80000000u
#define TOKEN_LEN(t) ((t)&0x1FFu)
#define TOKEN_OFF(t) (((t) >> 9) & 0xFFFFu)

static void write_stored(struct bitwriter *w, const uint8_t *data, size_t len, int is_final)
{
    do {
        size_t chunk = len > 65535 ? 65535 : len;
        int last = (chunk == len);
        bw_add(w, (is_final && last) ? 1 : 0, 1);
        bw_add(w, 0, 2);
        bw_align(w);
        bw_put_byte(w, (uint8_t)(chunk & 0xFF));
        bw_put_byte(w, (uint8_t)(chunk >> 8));
        bw_put_byte(w, (uint8_t)(~chunk & 0xFF));
        bw_put_byte(w, (uint8_t)((~chunk >> 8) & 0xFF));
        for (size_t i = 0; i < chunk; i++)
            bw_put_byte(w, data[i]);
        data += chunk;
        len -= chunk;
    } while (len);
}

static void flush_block(struct bitwriter *w, int compat, const uint8_t *block_begin,
                        size_t block_length, const uint32_t *tokens, size_t n_tokens,
                        struct freqs *fr, int is_final)
{
    struct codes dyn;
    uint32_t pfreq[NUM_PRECODE_SYMS];
    uint8_t plens[NUM_PRECODE_SYMS];
    uint32_t pcw[NUM_PRECODE_SYMS];
    unsigned items[NUM_LITLEN_SYMS + NUM_OFFSET_SYMS];
    unsigned num_items, num_litlen, num_offset, num_explicit, sym;
    uint32_t dynamic_cost = 0, static_cost = 0, uncompressed_cost = 0;
    const struct codes *codes;

    if (!static_codes_ready)
        init_static_codes();
    if (!slot_tabs_ready)
        init_slot_tabs();

    fr->litlen[END_OF_BLOCK]++;
    gzpx_oracle_make_huffman_code(NUM_LITLEN_SYMS, MAX_LITLEN_CODEWORD_LEN, compat, fr->litlen,
                                  dyn.litlen_lens, dyn.litlen_cw);
    gzpx_oracle_make_huffman_code(NUM_OFFSET_SYMS, MAX_OFFSET_CODEWORD_LEN, compat, fr->offset,
                                  dyn.offset_lens, dyn.offset_cw);

    for (num_litlen = NUM_LITLEN_SYMS; num_litlen > 257; num_litlen--)
        if (dyn.litlen_lens[num_litlen - 1] != 0)
            break;
    for (num_offset = NUM_OFFSET_SYMS; num_offset > 1; num_offset--)
        if (dyn.offset_lens[num_offset - 1] != 0)
            break;
    {
        uint8_t all[NUM_LITLEN_SYMS + NUM_OFFSET_SYMS];
        memcpy(all, dyn.litlen_lens, num_litlen);
        memcpy(all + num_litlen, dyn.offset_lens, num_offset);
        num_items = compute_precode_items(all, num_litlen + num_offset, pfreq, items);
    }
    gzpx_oracle_make_huffman_code(NUM_PRECODE_SYMS, MAX_PRE_CODEWORD_LEN, compat, pfreq, plens,
                                  pcw);
    for (num_explicit = NUM_PRECODE_SYMS; num_explicit > 4; num_explicit--)
        if (plens[precode_lens_permutation[num_explicit - 1]] != 0)
            break;

    dynamic_cost += 5 + 5 + 4 + 3 * num_explicit;
    for (sym = 0; sym < NUM_PRECODE_SYMS; sym++)
        dynamic_cost += pfreq[sym] * (extra_precode_bits[sym] + plens[sym]);
    for (sym = 0; sym < 144; sym++) {
        dynamic_cost += fr->litlen[sym] * dyn.litlen_lens[sym];
        static_cost += fr->litlen[sym] * 8;
    }
    for (; sym < 256; sym++) {
        dynamic_cost += fr->litlen[sym] * dyn.litlen_lens[sym];
        static_cost += fr->litlen[sym] * 9;
    }
    dynamic_cost += dyn.litlen_lens[END_OF_BLOCK];
    static_cost += 7;
    for (sym = FIRST_LEN_SYM; sym < FIRST_LEN_SYM + 29; sym++) {
        uint32_t extra = extra_length_bits[sym - FIRST_LEN_SYM];
        dynamic_cost += fr->litlen[sym] * (extra + dyn.litlen_lens[sym]);
        static_cost += fr->litlen[sym] * (extra + static_codes.litlen_lens[sym]);
    }
    for (sym = 0; sym < 30; sym++) {
        uint32_t extra = extra_offset_bits[sym];
        dynamic_cost += fr->offset[sym] * (extra + dyn.offset_lens[sym]);
        static_cost += fr->offset[sym] * (extra + 5);
    }
    uncompressed_cost += ((0u - (w->bitcount + 3)) & 7) + 32 +
                         40 * (uint32_t)((block_length + 65534) / 65535 - 1) +
                         8 * (uint32_t)block_length;

    if (dynamic_cost < (static_cost < uncompressed_cost ? static_cost : uncompressed_cost)) {
        codes = &dyn;
        bw_add(w, is_final ? 1 : 0, 1);
        bw_add(w, 2, 2);
        bw_add(w, num_litlen - 257, 5);
        bw_add(w, num_offset - 1, 5);
        bw_add(w, num_explicit - 4, 4);
        for (unsigned i = 0; i < num_explicit; i++)
            bw_add(w, plens[precode_lens_permutation[i]], 3);
        for (unsigned i = 0; i < num_items; i++) {
            unsigned psym = items[i] & 0x1F, extra = items[i] >> 5;
            bw_add(w, pcw[psym], plens[psym]);
            if (psym >= 16)
                bw_add(w, extra, extra_precode_bits[psym]);
        }
    } else if (static_cost < uncompressed_cost) {
        codes = &static_codes;
        bw_add(w, is_final ? 1 : 0, 1);
        bw_add(w, 1, 2);
    } else {
        write_stored(w, block_begin, block_length, is_final);
        return;
    }

    for (size_t i = 0; i < n_tokens; i++) {
        uint32_t t = tokens[i];
        if (t & TOKEN_MATCH) {
            unsigned len = TOKEN_LEN(t), off = TOKEN_OFF(t);
            unsigned ls = length_slot(len), os = offset_slot(off);
            bw_add(w, codes->litlen_cw[FIRST_LEN_SYM + ls], codes->litlen_lens[FIRST_LEN_SYM + ls]);
            bw_add(w, len - length_slot_base[ls], extra_length_bits[ls]);
            bw_add(w, codes->offset_cw[os], codes->offset_lens[os]);
            bw_add(w, off - offset_slot_base[os], extra_offset_bits[os]);
        } else {
            bw_add(w, codes->litlen_cw[t], codes->litlen_lens[t]);
        }
    }
    bw_add(w, codes->litlen_cw[END_OF_BLOCK], codes->litlen_lens[END_OF_BLOCK]);
}

/* ------------------------------------------------------------------ level 1: ht_matchfinder (A.1-A.3) */

#define HT_HASH_ORDER 15
#define HT_BUCKET 2
#define HT_REQUIRED_NBYTES 5

struct ht_mf {
    int16_t tab[1u << HT_HASH_ORDER][HT_BUCKET];
};

static uint32_t le32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

static uint32_t lz_hash(uint32_t v, unsigned bits)
{
    return (uint32_t)(v * 0x1E35A7BDu) >> (32 - bits);
}

static void ht_slide(struct ht_mf *mf)
{
    int16_t *t = &mf->tab[0][0];
    for (size_t i = 0; i < (size_t)HT_BUCKET << HT_HASH_ORDER; i++)
        t[i] = (int16_t)(t[i] >= 0 ? t[i] - WINDOW_SIZE : -WINDOW_SIZE);
}

static unsigned lz_extend(const uint8_t *a, const uint8_t *b, unsigned len, unsigned max_len)
{
    while (len + 8 <= max_len) {
        uint64_t x, y;
        memcpy(&x, a + len, 8);
        memcpy(&y, b + len, 8);
        if (x != y)
            return len + (unsigned)(__builtin_ctzll(x ^ y) >> 3);
        len += 8;
    }
    while (len < max_len && a[len] == b[len])
        len++;
    return len;
}

static unsigned ht_longest_match(struct ht_mf *mf, const uint8_t **in_base_p, const uint8_t *in_next,
                                 unsigned max_len, unsigned nice_len, uint32_t *next_hash,
                                 unsigned *offset_ret)
{
    unsigned best_len = 0;
    const uint8_t *best_ptr = in_next;
    uint32_t cur_pos = (uint32_t)(in_next - *in_base_p);
    const uint8_t *in_base;
    int32_t cutoff;
    uint32_t hash, seq;
    int16_t cur_node, to_insert;
    const uint8_t *matchptr;

    if (cur_pos == WINDOW_SIZE) {
        ht_slide(mf);
        *in_base_p += WINDOW_SIZE;
        cur_pos = 0;
    }
    in_base = *in_base_p;
    cutoff = (int32_t)cur_pos - WINDOW_SIZE;

    hash = *next_hash;
    *next_hash = lz_hash(le32(in_next + 1), HT_HASH_ORDER);
    seq = le32(in_next);

    /* bucket size 2, libdeflate's hand-unrolled form: entry 0 is copied to entry 1 even when
     * nice_len is reached on entry 0 */
    cur_node = mf->tab[hash][0];
    mf->tab[hash][0] = (int16_t)cur_pos;
    if (cur_node <= cutoff)
        goto out;
    matchptr = in_base + cur_node;
    to_insert = cur_node;
    cur_node = mf->tab[hash][1];
    mf->tab[hash][1] = to_insert;

    if (le32(matchptr) == seq) {
        best_len = lz_extend(in_next, matchptr, 4, max_len);
        best_ptr = matchptr;
        if (cur_node <= cutoff || best_len >= nice_len)
            goto out;
        matchptr = in_base + cur_node;
        if (le32(matchptr) == seq) {
            unsigned len = lz_extend(in_next, matchptr, 4, max_len);
            if (len > best_len) {
                best_len = len;
                best_ptr = matchptr;
            }
        }
    } else {
        if (cur_node <= cutoff)
            goto out;
        matchptr = in_base + cur_node;
        if (le32(matchptr) == seq) {
            best_len = lz_extend(in_next, matchptr, 4, max_len);
            best_ptr = matchptr;
        }
    }
out:
    *offset_ret = (unsigned)(in_next - best_ptr);
    return best_len;
}

static void ht_skip_bytes(struct ht_mf *mf, const uint8_t **in_base_p, const uint8_t *in_next,
                          const uint8_t *in_end, unsigned count, uint32_t *next_hash)
{
    int32_t cur_pos = (int32_t)(in_next - *in_base_p);
    uint32_t hash;
    unsigned remaining = count;

    if ((size_t)count + HT_REQUIRED_NBYTES > (size_t)(in_end - in_next))
        return;
    if (cur_pos + (int32_t)count - 1 >= WINDOW_SIZE) {
        ht_slide(mf);
        *in_base_p += WINDOW_SIZE;
        cur_pos -= WINDOW_SIZE;
    }
    hash = *next_hash;
    do {
        mf->tab[hash][1] = mf->tab[hash][0];
        mf->tab[hash][0] = (int16_t)cur_pos;
        hash = lz_hash(le32(++in_next), HT_HASH_ORDER);
        cur_pos++;
    } while (--remaining);
    *next_hash = hash;
}

typedef void (*block_sink_fn)(void *ctx, const uint8_t *block_begin, size_t block_length,
                              const uint32_t *tokens, size_t n_tokens, struct freqs *fr,
                              int is_final);

static void tally_literal(struct freqs *fr, uint32_t *tokens, size_t *nt, uint8_t lit)
{
    fr->litlen[lit]++;
    tokens[(*nt)++] = lit;
}

static void tally_match(struct freqs *fr, uint32_t *tokens, size_t *nt, unsigned len, unsigned off)
{
    fr->litlen[FIRST_LEN_SYM + length_slot(len)]++;
    fr->offset[offset_slot(off)]++;
    tokens[(*nt)++] = TOKEN_MATCH | (off << 9) | len;
}

/* The compressor object of the reference lives as long as its worker thread
 * (src/par/compress.rs:278); its big arrays are likewise kept per thread here instead of being
 * re-allocated for every block. */
static __thread struct ht_mf *tl_mf;
static __thread uint32_t *tl_tokens;

/* deflate_compress_fastest: greedy parse with ht_matchfinder; sub-block per 8192 sequences. */
static int compress_fastest(const uint8_t *in, size_t n, block_sink_fn sink, void *ctx)
{
    const uint8_t *in_next = in, *in_end = in + n, *in_cur_base = in;
    unsigned max_len = MAX_MATCH_LEN, nice_len = 32;
    uint32_t next_hash = 0;
    struct ht_mf *mf;
    uint32_t *tokens;
    if (!slot_tabs_ready)
        init_slot_tabs();
    if (!tl_mf)
        tl_mf = (struct ht_mf *)malloc(sizeof(*tl_mf));
    /* worst case one token per byte in a sub-block of <= 65535+5000 bytes */
    if (!tl_tokens)
        tl_tokens = (uint32_t *)malloc(sizeof(uint32_t) * (FAST_SOFT_MAX_BLOCK_LENGTH + MIN_BLOCK_LENGTH + 300));
    mf = tl_mf;
    tokens = tl_tokens;
    if (!mf || !tokens)
        return -1;
    for (size_t i = 0; i < (1u << HT_HASH_ORDER); i++)
        mf->tab[i][0] = mf->tab[i][1] = -WINDOW_SIZE;

    do {
        const uint8_t *block_begin = in_next;
        const uint8_t *max_block_end =
            ((size_t)(in_end - in_next) < FAST_SOFT_MAX_BLOCK_LENGTH + MIN_BLOCK_LENGTH)
                ? in_end
                : in_next + FAST_SOFT_MAX_BLOCK_LENGTH;
        struct freqs fr;
        size_t nt = 0;
        unsigned nseq = 0;
        memset(&fr, 0, sizeof(fr));
        do {
            unsigned length, offset;
            size_t remaining = (size_t)(in_end - in_next);
            if (remaining < MAX_MATCH_LEN) {
                max_len = (unsigned)remaining;
                if (max_len < HT_REQUIRED_NBYTES) {
                    do {
                        tally_literal(&fr, tokens, &nt, *in_next++);
                    } while (--max_len);
                    break;
                }
                if (nice_len > max_len)
                    nice_len = max_len;
            }
            length = ht_longest_match(mf, &in_cur_base, in_next, max_len, nice_len, &next_hash,
                                      &offset);
            if (length) {
                tally_match(&fr, tokens, &nt, length, offset);
                nseq++;
                ht_skip_bytes(mf, &in_cur_base, in_next + 1, in_end, length - 1, &next_hash);
                in_next += length;
            } else {
                tally_literal(&fr, tokens, &nt, *in_next++);
            }
        } while (in_next < max_block_end && nseq < FAST_SEQ_STORE_LENGTH);
        sink(ctx, block_begin, (size_t)(in_next - block_begin), tokens, nt, &fr, in_next == in_end);
    } while (in_next != in_end);
    return 0;
}

/* ------------------------------------------------------------------ levels 2-4: hc_matchfinder, greedy (A.8) */

#define SOFT_MAX_BLOCK_LENGTH 300000
#define SEQ_STORE_LENGTH 50000
#define HC_HASH3_ORDER 15
#define HC_HASH4_ORDER 16
#define NUM_LITERAL_OBSERVATION_TYPES 8
#define NUM_MATCH_OBSERVATION_TYPES 2
#define NUM_OBSERVATION_TYPES (NUM_LITERAL_OBSERVATION_TYPES + NUM_MATCH_OBSERVATION_TYPES)
#define NUM_OBSERVATIONS_PER_BLOCK_CHECK 512

struct hc_mf {
    int16_t hash3_tab[1u << HC_HASH3_ORDER];
    int16_t hash4_tab[1u << HC_HASH4_ORDER];
    int16_t next_tab[WINDOW_SIZE];
};

struct split_stats {
    uint32_t new_observations[NUM_OBSERVATION_TYPES];
    uint32_t observations[NUM_OBSERVATION_TYPES];
    uint32_t num_new_observations;
    uint32_t num_observations;
};

static void hc_slide(struct hc_mf *mf)
{
    int16_t *t = (int16_t *)mf;
    const size_t cnt = sizeof(*mf) / sizeof(int16_t);
    for (size_t i = 0; i < cnt; i++)
        t[i] = (int16_t)(t[i] >= 0 ? t[i] - WINDOW_SIZE : -WINDOW_SIZE);
}

static unsigned hc_longest_match(struct hc_mf *mf, const uint8_t **in_base_p, const uint8_t *in_next,
                                 unsigned best_len, unsigned max_len, unsigned nice_len,
                                 unsigned max_search_depth, uint32_t next_hashes[2], unsigned *offset_ret)
{
    unsigned depth_remaining = max_search_depth;
    const uint8_t *best_matchptr = in_next;
    int16_t cur_node3, cur_node4;
    uint32_t hash3, hash4, next_hashseq, seq4;
    const uint8_t *matchptr;
    unsigned len;
    uint32_t cur_pos = (uint32_t)(in_next - *in_base_p);
    const uint8_t *in_base;
    int32_t cutoff;

    if (cur_pos == WINDOW_SIZE) {
        hc_slide(mf);
        *in_base_p += WINDOW_SIZE;
        cur_pos = 0;
    }
    in_base = *in_base_p;
    cutoff = (int32_t)cur_pos - WINDOW_SIZE;

    if (max_len < 5) /* cannot read 4 bytes from in_next + 1 */
        goto out;

    hash3 = next_hashes[0];
    hash4 = next_hashes[1];
    cur_node3 = mf->hash3_tab[hash3];
    cur_node4 = mf->hash4_tab[hash4];
    mf->hash3_tab[hash3] = (int16_t)cur_pos;
    mf->hash4_tab[hash4] = (int16_t)cur_pos;
    mf->next_tab[cur_pos] = cur_node4;

    next_hashseq = le32(in_next + 1);
    next_hashes[0] = lz_hash(next_hashseq & 0xFFFFFF, HC_HASH3_ORDER);
    next_hashes[1] = lz_hash(next_hashseq, HC_HASH4_ORDER);

    if (best_len < 4) {
        if (cur_node3 <= cutoff)
            goto out;
        seq4 = le32(in_next);
        if (best_len < 3) {
            matchptr = in_base + cur_node3;
            if ((le32(matchptr) & 0xFFFFFF) == (seq4 & 0xFFFFFF)) {
                best_len = 3;
                best_matchptr = matchptr;
            }
        }
        if (cur_node4 <= cutoff)
            goto out;
        for (;;) {
            matchptr = in_base + cur_node4;
            if (le32(matchptr) == seq4)
                break;
            cur_node4 = mf->next_tab[cur_node4 & (WINDOW_SIZE - 1)];
            if (cur_node4 <= cutoff || !--depth_remaining)
                goto out;
        }
        best_matchptr = matchptr;
        best_len = lz_extend(in_next, best_matchptr, 4, max_len);
        if (best_len >= nice_len)
            goto out;
        cur_node4 = mf->next_tab[cur_node4 & (WINDOW_SIZE - 1)];
        if (cur_node4 <= cutoff || !--depth_remaining)
            goto out;
    } else {
        if (cur_node4 <= cutoff || best_len >= nice_len)
            goto out;
    }

    for (;;) {
        for (;;) {
            matchptr = in_base + cur_node4;
            if (le32(matchptr + best_len - 3) == le32(in_next + best_len - 3) &&
                le32(matchptr) == le32(in_next))
                break;
            cur_node4 = mf->next_tab[cur_node4 & (WINDOW_SIZE - 1)];
            if (cur_node4 <= cutoff || !--depth_remaining)
                goto out;
        }
        len = lz_extend(in_next, matchptr, 4, max_len);
        if (len > best_len) {
            best_len = len;
            best_matchptr = matchptr;
            if (best_len >= nice_len)
                goto out;
        }
        cur_node4 = mf->next_tab[cur_node4 & (WINDOW_SIZE - 1)];
        if (cur_node4 <= cutoff || !--depth_remaining)
            goto out;
    }
out:
    *offset_ret = (unsigned)(in_next - best_matchptr);
    return best_len;
}

static void hc_skip_bytes(struct hc_mf *mf, const uint8_t **in_base_p, const uint8_t *in_next,
                          const uint8_t *in_end, unsigned count, uint32_t next_hashes[2])
{
    uint32_t cur_pos, hash3, hash4, next_hashseq;
    unsigned remaining = count;

    if ((size_t)count + 5 > (size_t)(in_end - in_next))
        return;
    cur_pos = (uint32_t)(in_next - *in_base_p);
    hash3 = next_hashes[0];
    hash4 = next_hashes[1];
    do {
        if (cur_pos == WINDOW_SIZE) {
            hc_slide(mf);
            *in_base_p += WINDOW_SIZE;
            cur_pos = 0;
        }
        mf->hash3_tab[hash3] = (int16_t)cur_pos;
        mf->next_tab[cur_pos] = mf->hash4_tab[hash4];
        mf->hash4_tab[hash4] = (int16_t)cur_pos;
        next_hashseq = le32(++in_next);
        hash3 = lz_hash(next_hashseq & 0xFFFFFF, HC_HASH3_ORDER);
        hash4 = lz_hash(next_hashseq, HC_HASH4_ORDER);
        cur_pos++;
    } while (--remaining);
    next_hashes[0] = hash3;
    next_hashes[1] = hash4;
}

static unsigned choose_min_match_len(unsigned num_used_literals, unsigned max_search_depth)
{
    static const uint8_t min_lens[] = {
        9, 9, 9, 9, 9, 9, 8, 8, 7, 7, 6, 6, 6, 6, 6, 6, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
        5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4,
        4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    };
    unsigned min_len = MIN_MATCH_LEN;
    if (num_used_literals < sizeof(min_lens))
        min_len = min_lens[num_used_literals];
    if (max_search_depth < 16) {
        unsigned cap = max_search_depth < 5 ? 4 : max_search_depth < 10 ? 5 : 7;
        if (min_len > cap)
            min_len = cap;
    }
    return min_len;
}

static unsigned calculate_min_match_len(const uint8_t *data, size_t data_len, unsigned max_search_depth,
                                        int compat)
{
    uint8_t used[256];
    unsigned num_used = 0;
    memset(used, 0, sizeof(used));
    /* libdeflate >= 1.1x: a scan shorter than 512 bytes always uses 3 (SURVEY A.7 delta 2) */
    if (compat != GZPX_ORACLE_COMPAT_1_10 && data_len < 512)
        return MIN_MATCH_LEN;
    if (data_len > 4096)
        data_len = 4096;
    for (size_t i = 0; i < data_len; i++)
        used[data[i]] = 1;
    for (unsigned i = 0; i < 256; i++)
        num_used += used[i];
    return choose_min_match_len(num_used, max_search_depth);
}

static int do_end_block_check(struct split_stats *st, uint32_t block_length)
{
    if (st->num_observations > 0) {
        uint32_t total_delta = 0, num_items, cutoff;
        for (int i = 0; i < NUM_OBSERVATION_TYPES; i++) {
            uint32_t expected = st->observations[i] * st->num_new_observations;
            uint32_t actual = st->new_observations[i] * st->num_observations;
            total_delta += actual > expected ? actual - expected : expected - actual;
        }
        num_items = st->num_observations + st->num_new_observations;
        cutoff = st->num_new_observations * 200 / 512 * st->num_observations;
        if (block_length < 10000 && num_items < 8192)
            cutoff += (uint32_t)((uint64_t)cutoff * (8192 - num_items) / 8192);
        if (total_delta + (block_length / 4096) * st->num_observations >= cutoff)
            return 1;
    }
    for (int i = 0; i < NUM_OBSERVATION_TYPES; i++) {
        st->num_observations += st->new_observations[i];
        st->observations[i] += st->new_observations[i];
        st->new_observations[i] = 0;
    }
    st->num_new_observations = 0;
    return 0;
}

static __thread struct hc_mf *tl_hc;
static __thread uint32_t *tl_tokens_hc;

/* deflate_compress_greedy */
static int compress_greedy(const uint8_t *in, size_t n, unsigned nice_match_length,
                           unsigned max_search_depth, int compat, block_sink_fn sink, void *ctx)
{
    const uint8_t *in_next = in, *in_end = in + n, *in_cur_base = in;
    unsigned max_len = MAX_MATCH_LEN;
    unsigned nice_len = nice_match_length < max_len ? nice_match_length : max_len;
    uint32_t next_hashes[2] = {0, 0};
    struct hc_mf *mf;
    uint32_t *tokens;
    if (!slot_tabs_ready)
        init_slot_tabs();
    if (!tl_hc)
        tl_hc = (struct hc_mf *)malloc(sizeof(*tl_hc));
    if (!tl_tokens_hc)
        tl_tokens_hc = (uint32_t *)malloc(sizeof(uint32_t) * (SOFT_MAX_BLOCK_LENGTH + MIN_BLOCK_LENGTH + 300));
    mf = tl_hc;
    tokens = tl_tokens_hc;
    if (!mf || !tokens)
        return -1;
    {
        int16_t *t = (int16_t *)mf;
        for (size_t i = 0; i < sizeof(*mf) / sizeof(int16_t); i++)
            t[i] = -WINDOW_SIZE;
    }
    do {
        const uint8_t *block_begin = in_next;
        const uint8_t *max_block_end =
            ((size_t)(in_end - in_next) < SOFT_MAX_BLOCK_LENGTH + MIN_BLOCK_LENGTH)
                ? in_end
                : in_next + SOFT_MAX_BLOCK_LENGTH;
        struct freqs fr;
        struct split_stats st;
        size_t nt = 0;
        unsigned nseq = 0, min_len;
        memset(&fr, 0, sizeof(fr));
        memset(&st, 0, sizeof(st));
        min_len = calculate_min_match_len(in_next, (size_t)(max_block_end - in_next), max_search_depth,
                                          compat);
        for (;;) {
            unsigned length, offset;
            size_t remaining = (size_t)(in_end - in_next);
            if (remaining < MAX_MATCH_LEN) {
                max_len = (unsigned)remaining;
                if (nice_len > max_len)
                    nice_len = max_len;
            }
            length = hc_longest_match(mf, &in_cur_base, in_next, min_len - 1, max_len, nice_len,
                                      max_search_depth, next_hashes, &offset);
            if (length >= min_len && (length > MIN_MATCH_LEN || offset <= 4096)) {
                tally_match(&fr, tokens, &nt, length, offset);
                nseq++;
                st.new_observations[NUM_LITERAL_OBSERVATION_TYPES + (length >= 9)]++;
                st.num_new_observations++;
                hc_skip_bytes(mf, &in_cur_base, in_next + 1, in_end, length - 1, next_hashes);
                in_next += length;
            } else {
                uint8_t lit = *in_next++;
                tally_literal(&fr, tokens, &nt, lit);
                st.new_observations[((lit >> 5) & 0x6) | (lit & 1)]++;
                st.num_new_observations++;
            }
            if (!(in_next < max_block_end && nseq < SEQ_STORE_LENGTH))
                break;
            if (st.num_new_observations >= NUM_OBSERVATIONS_PER_BLOCK_CHECK &&
                (size_t)(in_next - block_begin) >= MIN_BLOCK_LENGTH &&
                (size_t)(in_end - in_next) >= MIN_BLOCK_LENGTH &&
                do_end_block_check(&st, (uint32_t)(in_next - block_begin)))
                break;
        }
        sink(ctx, block_begin, (size_t)(in_next - block_begin), tokens, nt, &fr, in_next == in_end);
    } while (in_next != in_end);
    return 0;
}

/* recalculate_min_match_len: the same choice from the literal frequencies seen so far in the block
 * (a literal counts as "used" when it is more frequent than 1/1024 of all literals) */
static unsigned recalculate_min_match_len(const struct freqs *fr, unsigned max_search_depth)
{
    uint32_t literal_freq = 0, cutoff;
    unsigned num_used = 0;
    for (unsigned i = 0; i < 256; i++)
        literal_freq += fr->litlen[i];
    cutoff = literal_freq >> 10;
    for (unsigned i = 0; i < 256; i++)
        if (fr->litlen[i] > cutoff)
            num_used++;
    return choose_min_match_len(num_used, max_search_depth);
}

static unsigned bsr32(uint32_t v) { return 31u - (unsigned)__builtin_clz(v); }

/* deflate_compress_lazy_generic (levels 5-7: lazy; 8-9: lazy2).  hc_matchfinder as the greedy parser;
 * a match is only taken if the next position (and, for lazy2, the one after it) has nothing clearly
 * better, searched with half (a quarter of) the depth. */
static int compress_lazy(const uint8_t *in, size_t n, unsigned nice_match_length, unsigned max_search_depth,
                         int lazy2, int compat, block_sink_fn sink, void *ctx)
{
    const uint8_t *in_next = in, *in_end = in + n, *in_cur_base = in;
    unsigned max_len = MAX_MATCH_LEN;
    unsigned nice_len = nice_match_length < max_len ? nice_match_length : max_len;
    uint32_t next_hashes[2] = {0, 0};
    struct hc_mf *mf;
    uint32_t *tokens;
    if (!slot_tabs_ready)
        init_slot_tabs();
    if (!tl_hc)
        tl_hc = (struct hc_mf *)malloc(sizeof(*tl_hc));
    if (!tl_tokens_hc)
        tl_tokens_hc = (uint32_t *)malloc(sizeof(uint32_t) * (SOFT_MAX_BLOCK_LENGTH + MIN_BLOCK_LENGTH + 300));
    mf = tl_hc;
    tokens = tl_tokens_hc;
    if (!mf || !tokens)
        return -1;
    {
        int16_t *t = (int16_t *)mf;
        for (size_t i = 0; i < sizeof(*mf) / sizeof(int16_t); i++)
            t[i] = -WINDOW_SIZE;
    }
#define ADJUST_LENS()                                          \
    do {                                                       \
        size_t remaining_ = (size_t)(in_end - in_next);        \
        if (remaining_ < MAX_MATCH_LEN) {                      \
            max_len = (unsigned)remaining_;                    \
            if (nice_len > max_len)                            \
                nice_len = max_len;                            \
        }                                                      \
    } while (0)
#define LITERAL(b)                                                   \
    do {                                                             \
        uint8_t lit_ = (b);                                          \
        tally_literal(&fr, tokens, &nt, lit_);                       \
        st.new_observations[((lit_ >> 5) & 0x6) | (lit_ & 1)]++;     \
        st.num_new_observations++;                                   \
    } while (0)
#define MATCH(len_, off_)                                                          \
    do {                                                                           \
        tally_match(&fr, tokens, &nt, (len_), (off_));                             \
        nseq++;                                                                    \
        st.new_observations[NUM_LITERAL_OBSERVATION_TYPES + ((len_) >= 9)]++;      \
        st.num_new_observations++;                                                 \
    } while (0)
    do {
        const uint8_t *block_begin = in_next;
        const uint8_t *max_block_end =
            ((size_t)(in_end - in_next) < SOFT_MAX_BLOCK_LENGTH + MIN_BLOCK_LENGTH)
                ? in_end
                : in_next + SOFT_MAX_BLOCK_LENGTH;
        const uint8_t *next_recalc_min_len =
            in_next + ((size_t)(in_end - in_next) < 10000 ? (size_t)(in_end - in_next) : 10000);
        struct freqs fr;
        struct split_stats st;
        size_t nt = 0;
        unsigned nseq = 0, min_len;
        memset(&fr, 0, sizeof(fr));
        memset(&st, 0, sizeof(st));
        min_len = calculate_min_match_len(in_next, (size_t)(max_block_end - in_next), max_search_depth, compat);
        do {
            unsigned cur_len, cur_offset, next_len, next_offset;
            if (in_next >= next_recalc_min_len) {
                size_t a = (size_t)(in_end - next_recalc_min_len), b = (size_t)(in_next - block_begin);
                min_len = recalculate_min_match_len(&fr, max_search_depth);
                next_recalc_min_len += a < b ? a : b;
            }
            ADJUST_LENS();
            cur_len = hc_longest_match(mf, &in_cur_base, in_next, min_len - 1, max_len, nice_len, max_search_depth,
                                       next_hashes, &cur_offset);
            if (cur_len < min_len || (cur_len == MIN_MATCH_LEN && cur_offset > 8192)) {
                LITERAL(*in_next);
                in_next++;
                continue;
            }
            in_next++;
        have_cur_match:
            if (cur_len >= nice_len) {
                MATCH(cur_len, cur_offset);
                hc_skip_bytes(mf, &in_cur_base, in_next, in_end, cur_len - 1, next_hashes);
                in_next += cur_len - 1;
                continue;
            }
            ADJUST_LENS();
            next_len = hc_longest_match(mf, &in_cur_base, in_next++, cur_len - 1, max_len, nice_len,
                                        max_search_depth >> 1, next_hashes, &next_offset);
            if (next_len >= cur_len &&
                4 * (int)(next_len - cur_len) + ((int)bsr32(cur_offset) - (int)bsr32(next_offset)) > 2) {
                LITERAL(*(in_next - 2));
                cur_len = next_len;
                cur_offset = next_offset;
                goto have_cur_match;
            }
            if (lazy2) {
                ADJUST_LENS();
                next_len = hc_longest_match(mf, &in_cur_base, in_next++, cur_len - 1, max_len, nice_len,
                                            max_search_depth >> 2, next_hashes, &next_offset);
                if (next_len >= cur_len &&
                    4 * (int)(next_len - cur_len) + ((int)bsr32(cur_offset) - (int)bsr32(next_offset)) > 6) {
                    LITERAL(*(in_next - 3));
                    LITERAL(*(in_next - 2));
                    cur_len = next_len;
                    cur_offset = next_offset;
                    goto have_cur_match;
                }
                MATCH(cur_len, cur_offset);
                if (cur_len > 3) {
                    hc_skip_bytes(mf, &in_cur_base, in_next, in_end, cur_len - 3, next_hashes);
                    in_next += cur_len - 3;
                }
            } else {
                MATCH(cur_len, cur_offset);
                hc_skip_bytes(mf, &in_cur_base, in_next, in_end, cur_len - 2, next_hashes);
                in_next += cur_len - 2;
            }
        } while (in_next < max_block_end && nseq < SEQ_STORE_LENGTH &&
                 !(st.num_new_observations >= NUM_OBSERVATIONS_PER_BLOCK_CHECK &&
                   (size_t)(in_next - block_begin) >= MIN_BLOCK_LENGTH &&
                   (size_t)(in_end - in_next) >= MIN_BLOCK_LENGTH &&
                   do_end_block_check(&st, (uint32_t)(in_next - block_begin))));
        sink(ctx, block_begin, (size_t)(in_next - block_begin), tokens, nt, &fr, in_next == in_end);
    } while (in_next != in_end);
#undef ADJUST_LENS
#undef LITERAL
#undef MATCH
    return 0;
}

/* ------------------------------------------------------------------ entry points */

/* ------------------------------------------------------------------------------------------
 * Levels 10-12: deflate_compress_near_optimal (libdeflate v1.10), restated: bt_matchfinder (hash3
 * 2-way + hash4 -> binary trees of the window's positions), every position's matches cached, block
 * splitting on the observation statistics with a rewind to the previous check, then per block an
 * iterated minimum-cost path over the cached matches (costs from default tables blended with the
 * previous block's, then from the Huffman codes of the previous pass).
 * The three default-cost tables are libdeflate's default_litlen_costs[]: int(-log2((1 - p) / max(j, 1))
 * * BIT_COST) for match probabilities p = 0.25 / 0.5 / 0.75 -- the same bytes sit in the v1.10 binary's
 * read-only data (tests/test_oracle_near_optimal.py compares them when the library is there).
 * ------------------------------------------------------------------------------------------ */
#define BT_HASH3_ORDER 16
#define BT_HASH3_WAYS 2
#define BT_HASH4_ORDER 16
#define BT_REQUIRED_NBYTES 5
#define MATCH_CACHE_LENGTH (SOFT_MAX_BLOCK_LENGTH * 5)
#define MAX_MATCHES_PER_POS (MAX_MATCH_LEN - MIN_MATCH_LEN + 1)
#define NO_MAX_BLOCK_LENGTH (SOFT_MAX_BLOCK_LENGTH + MIN_BLOCK_LENGTH - 1)
#define BIT_COST 16
#define LITERAL_NOSTAT_BITS 13
#define LENGTH_NOSTAT_BITS 13
#define OFFSET_NOSTAT_BITS 10
#define OPTIMUM_OFFSET_SHIFT 9
#define OPTIMUM_LEN_MASK ((1u << OPTIMUM_OFFSET_SHIFT) - 1)

struct bt_mf {
    int16_t hash3_tab[1u << BT_HASH3_ORDER][BT_HASH3_WAYS];
    int16_t hash4_tab[1u << BT_HASH4_ORDER];
    int16_t child_tab[2 * WINDOW_SIZE];
};

struct lz_match {
    uint16_t length;
    uint16_t offset;
};

struct optimum_node {
    uint32_t cost_to_end;
    uint32_t item; /* literal: (byte << 9) | 1; match: (offset << 9) | length */
};

struct no_costs {
    uint32_t literal[256];
    uint32_t length[MAX_MATCH_LEN + 1];
    uint32_t offset_slot[30];
};

struct no_state {
    struct bt_mf mf;
    struct lz_match match_cache[MATCH_CACHE_LENGTH + MAX_MATCHES_PER_POS + MAX_MATCH_LEN - 1];
    struct optimum_node optimum_nodes[NO_MAX_BLOCK_LENGTH + 1];
    struct no_costs costs;
    struct split_stats split;
    uint32_t new_match_len_freqs[MAX_MATCH_LEN + 1];
    uint32_t match_len_freqs[MAX_MATCH_LEN + 1];
    uint32_t prev_observations[NUM_OBSERVATION_TYPES];
    uint32_t prev_num_observations;
    uint32_t tokens[NO_MAX_BLOCK_LENGTH + 1];
    unsigned max_search_depth, nice_match_length, num_optim_passes;
    int compat;
};

static __thread struct no_state *tl_no;

/* default_litlen_costs[i].used_lits_to_lit_cost[j] / .len_sym_cost (scripts/gen_default_litlen_costs.py) */
static uint8_t no_default_lit_cost[3][257];
static uint8_t no_default_len_sym_cost[3];
static int no_tables_ready;

void gzpx_oracle_default_litlen_costs(uint8_t lit[3][257], uint8_t len_sym[3])
{
    static const double probs[3] = {0.25, 0.5, 0.75};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j <= 256; j++)
            lit[i][j] = (uint8_t)(int)(-log2((1.0 - probs[i]) / (double)(j ? j : 1)) * BIT_COST);
        len_sym[i] = (uint8_t)(int)(-log2(probs[i] / 29.0) * BIT_COST);
    }
}

static inline int16_t *bt_left(struct bt_mf *mf, int32_t node) { return &mf->child_tab[2 * (node & (WINDOW_SIZE - 1))]; }
static inline int16_t *bt_right(struct bt_mf *mf, int32_t node) { return &mf->child_tab[2 * (node & (WINDOW_SIZE - 1)) + 1]; }

static void bt_init(struct bt_mf *mf)
{
    int16_t *t = (int16_t *)mf;
    for (size_t i = 0; i < sizeof(*mf) / sizeof(int16_t); i++)
        t[i] = -WINDOW_SIZE;
}

static void bt_slide(struct bt_mf *mf)
{
    int16_t *t = (int16_t *)mf;
    for (size_t i = 0; i < sizeof(*mf) / sizeof(int16_t); i++)
        t[i] = (int16_t)(t[i] >= 0 ? t[i] - WINDOW_SIZE : -WINDOW_SIZE);
}

static uint32_t le24(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }

/* bt_matchfinder_advance_one_byte: record_matches = get_matches, else skip_byte */
static struct lz_match *bt_advance(struct bt_mf *mf, const uint8_t *in_base, ptrdiff_t cur_pos, uint32_t max_len,
                                   uint32_t nice_len, uint32_t max_search_depth, uint32_t next_hashes[2],
                                   struct lz_match *lz_matchptr, int record_matches)
{
    const uint8_t *in_next = in_base + cur_pos;
    uint32_t depth_remaining = max_search_depth;
    const int32_t cutoff = (int32_t)cur_pos - WINDOW_SIZE;
    uint32_t next_hashseq, hash3, hash4;
    int32_t cur_node, cur_node_2;
    const uint8_t *matchptr;
    int16_t *pending_lt_ptr, *pending_gt_ptr;
    uint32_t best_lt_len, best_gt_len, len, best_len = 3;

    next_hashseq = le32(in_next + 1);
    hash3 = next_hashes[0];
    hash4 = next_hashes[1];
    next_hashes[0] = lz_hash(next_hashseq & 0xFFFFFF, BT_HASH3_ORDER);
    next_hashes[1] = lz_hash(next_hashseq, BT_HASH4_ORDER);

    cur_node = mf->hash3_tab[hash3][0];
    mf->hash3_tab[hash3][0] = (int16_t)cur_pos;
    cur_node_2 = mf->hash3_tab[hash3][1];
    mf->hash3_tab[hash3][1] = (int16_t)cur_node;
    if (record_matches && cur_node > cutoff) {
        uint32_t seq3 = le24(in_next);
        if (seq3 == le24(&in_base[cur_node])) {
            lz_matchptr->length = 3;
            lz_matchptr->offset = (uint16_t)(in_next - &in_base[cur_node]);
            lz_matchptr++;
        } else if (cur_node_2 > cutoff && seq3 == le24(&in_base[cur_node_2])) {
            lz_matchptr->length = 3;
            lz_matchptr->offset = (uint16_t)(in_next - &in_base[cur_node_2]);
            lz_matchptr++;
        }
    }

    cur_node = mf->hash4_tab[hash4];
    mf->hash4_tab[hash4] = (int16_t)cur_pos;

    pending_lt_ptr = bt_left(mf, (int32_t)cur_pos);
    pending_gt_ptr = bt_right(mf, (int32_t)cur_pos);

    if (cur_node <= cutoff) {
        *pending_lt_ptr = -WINDOW_SIZE;
        *pending_gt_ptr = -WINDOW_SIZE;
        return lz_matchptr;
    }

    best_lt_len = 0;
    best_gt_len = 0;
    len = 0;

    for (;;) {
        matchptr = &in_base[cur_node];
        if (matchptr[len] == in_next[len]) {
            len = lz_extend(in_next, matchptr, len + 1, max_len);
            if (!record_matches || len > best_len) {
                if (record_matches) {
                    best_len = len;
                    lz_matchptr->length = (uint16_t)len;
                    lz_matchptr->offset = (uint16_t)(in_next - matchptr);
                    lz_matchptr++;
                }
                if (len >= nice_len) {
                    *pending_lt_ptr = *bt_left(mf, cur_node);
                    *pending_gt_ptr = *bt_right(mf, cur_node);
                    return lz_matchptr;
                }
            }
        }
        if (matchptr[len] < in_next[len]) {
            *pending_lt_ptr = (int16_t)cur_node;
            pending_lt_ptr = bt_right(mf, cur_node);
            cur_node = *pending_lt_ptr;
            best_lt_len = len;
            if (best_gt_len < len)
                len = best_gt_len;
        } else {
            *pending_gt_ptr = (int16_t)cur_node;
            pending_gt_ptr = bt_left(mf, cur_node);
            cur_node = *pending_gt_ptr;
            best_gt_len = len;
            if (best_lt_len < len)
                len = best_lt_len;
        }
        if (cur_node <= cutoff || !--depth_remaining) {
            *pending_lt_ptr = -WINDOW_SIZE;
            *pending_gt_ptr = -WINDOW_SIZE;
            return lz_matchptr;
        }
    }
}

static void no_init_stats(struct no_state *c)
{
    memset(&c->split, 0, sizeof(c->split));
    memset(c->new_match_len_freqs, 0, sizeof(c->new_match_len_freqs));
    memset(c->match_len_freqs, 0, sizeof(c->match_len_freqs));
}

static void no_merge_stats(struct no_state *c)
{
    for (int i = 0; i < NUM_OBSERVATION_TYPES; i++) {
        c->split.num_observations += c->split.new_observations[i];
        c->split.observations[i] += c->split.new_observations[i];
        c->split.new_observations[i] = 0;
    }
    c->split.num_new_observations = 0;
    for (unsigned i = 0; i <= MAX_MATCH_LEN; i++) {
        c->match_len_freqs[i] += c->new_match_len_freqs[i];
        c->new_match_len_freqs[i] = 0;
    }
}

static void no_save_stats(struct no_state *c)
{
    for (int i = 0; i < NUM_OBSERVATION_TYPES; i++)
        c->prev_observations[i] = c->split.observations[i];
    c->prev_num_observations = c->split.num_observations;
}

static void no_clear_old_stats(struct no_state *c)
{
    for (int i = 0; i < NUM_OBSERVATION_TYPES; i++)
        c->split.observations[i] = 0;
    c->split.num_observations = 0;
    memset(c->match_len_freqs, 0, sizeof(c->match_len_freqs));
}

/* deflate_choose_default_litlen_costs */
static void no_choose_default_litlen_costs(struct no_state *c, const uint8_t *block_begin, uint32_t block_length,
                                           uint32_t *lit_cost, uint32_t *len_sym_cost)
{
    uint32_t litfreq[256];
    unsigned num_used_literals = 0;
    uint32_t literal_freq = block_length, match_freq = 0, cutoff, i;
    memset(litfreq, 0, sizeof(litfreq));
    cutoff = literal_freq >> 11; /* ignore literals used very rarely */
    for (i = 0; i < block_length; i++)
        litfreq[block_begin[i]]++;
    for (i = 0; i < 256; i++)
        if (litfreq[i] > cutoff)
            num_used_literals++;
    if (num_used_literals == 0)
        num_used_literals = 1;
    /* the match frequency of a greedy parse, with the min_len heuristic of the greedy / lazy parsers */
    i = choose_min_match_len(num_used_literals, c->max_search_depth);
    for (; i <= MAX_MATCH_LEN; i++) {
        match_freq += c->match_len_freqs[i];
        literal_freq -= i * c->match_len_freqs[i];
    }
    if ((int32_t)literal_freq < 0)
        literal_freq = 0;
    if (match_freq > literal_freq)
        i = 2; /* many matches */
    else if (match_freq * 4 > literal_freq)
        i = 1; /* neutral */
    else
        i = 0; /* few matches */
    *lit_cost = no_default_lit_cost[i][num_used_literals];
    *len_sym_cost = no_default_len_sym_cost[i];
}

static uint32_t no_default_length_cost(unsigned len, uint32_t len_sym_cost)
{
    return len_sym_cost + extra_length_bits[length_slot(len)] * BIT_COST;
}

static uint32_t no_default_offset_slot_cost(unsigned slot)
{
    /* int(-log2(1/30) * BIT_COST): all 30 offset symbols equally probable */
    const uint32_t offset_sym_cost = 4 * BIT_COST + (907 * BIT_COST) / 1000;
    return offset_sym_cost + extra_offset_bits[slot] * BIT_COST;
}

static void no_set_default_costs(struct no_state *c, uint32_t lit_cost, uint32_t len_sym_cost)
{
    unsigned i;
    for (i = 0; i < 256; i++)
        c->costs.literal[i] = lit_cost;
    for (i = MIN_MATCH_LEN; i <= MAX_MATCH_LEN; i++)
        c->costs.length[i] = no_default_length_cost(i, len_sym_cost);
    for (i = 0; i < 30; i++)
        c->costs.offset_slot[i] = no_default_offset_slot_cost(i);
}

static void no_adjust_cost(uint32_t *cost_p, uint32_t default_cost, int change_amount)
{
    if (change_amount == 0) /* block is very similar to the previous one: prefer the previous costs */
        *cost_p = (default_cost + 3 * *cost_p) / 4;
    else if (change_amount == 1)
        *cost_p = (default_cost + *cost_p) / 2;
    else if (change_amount == 2)
        *cost_p = (5 * default_cost + 3 * *cost_p) / 8;
    else /* block differs greatly from the previous one: prefer the default costs */
        *cost_p = (3 * default_cost + *cost_p) / 4;
}

static void no_adjust_costs(struct no_state *c, uint32_t lit_cost, uint32_t len_sym_cost)
{
    uint64_t total_delta = 0, cutoff;
    int change_amount;
    unsigned i;
    for (i = 0; i < NUM_OBSERVATION_TYPES; i++) {
        uint64_t prev = (uint64_t)c->prev_observations[i] * c->split.num_observations;
        uint64_t cur = (uint64_t)c->split.observations[i] * c->prev_num_observations;
        total_delta += prev > cur ? prev - cur : cur - prev;
    }
    cutoff = ((uint64_t)c->prev_num_observations * c->split.num_observations * 200) / 512;
    if (4 * total_delta > 9 * cutoff)
        change_amount = 3;
    else if (2 * total_delta > 3 * cutoff)
        change_amount = 2;
    else if (2 * total_delta > cutoff)
        change_amount = 1;
    else
        change_amount = 0;
    for (i = 0; i < 256; i++)
        no_adjust_cost(&c->costs.literal[i], lit_cost, change_amount);
    for (i = MIN_MATCH_LEN; i <= MAX_MATCH_LEN; i++)
        no_adjust_cost(&c->costs.length[i], no_default_length_cost(i, len_sym_cost), change_amount);
    for (i = 0; i < 30; i++)
        no_adjust_cost(&c->costs.offset_slot[i], no_default_offset_slot_cost(i), change_amount);
}

static void no_set_costs_from_codes(struct no_state *c, const uint8_t *litlen_lens, const uint8_t *offset_lens)
{
    unsigned i;
    for (i = 0; i < 256; i++)
        c->costs.literal[i] = (litlen_lens[i] ? litlen_lens[i] : LITERAL_NOSTAT_BITS) * BIT_COST;
    for (i = MIN_MATCH_LEN; i <= MAX_MATCH_LEN; i++) {
        unsigned slot = length_slot(i), sym = FIRST_LEN_SYM + slot;
        uint32_t bits = litlen_lens[sym] ? litlen_lens[sym] : LENGTH_NOSTAT_BITS;
        c->costs.length[i] = (bits + extra_length_bits[slot]) * BIT_COST;
    }
    for (i = 0; i < 30; i++) {
        uint32_t bits = offset_lens[i] ? offset_lens[i] : OFFSET_NOSTAT_BITS;
        c->costs.offset_slot[i] = (bits + extra_offset_bits[i]) * BIT_COST;
    }
}

/* deflate_find_min_cost_path + deflate_tally_item_list: `cache_ptr` = the end of the block's cache entries */
static void no_find_min_cost_path(struct no_state *c, uint32_t block_length, const struct lz_match *cache_ptr,
                                  struct freqs *fr)
{
    struct optimum_node *end_node = &c->optimum_nodes[block_length];
    struct optimum_node *cur_node = end_node;
    cur_node->cost_to_end = 0;
    do {
        unsigned num_matches, literal;
        uint32_t best_cost_to_end;
        cur_node--;
        cache_ptr--;
        num_matches = cache_ptr->length;
        literal = cache_ptr->offset;
        /* it is always possible to choose a literal */
        best_cost_to_end = c->costs.literal[literal] + (cur_node + 1)->cost_to_end;
        cur_node->item = ((uint32_t)literal << OPTIMUM_OFFSET_SHIFT) | 1;
        if (num_matches) {
            /* every length from 3 to the longest match found here, each with the smallest offset that has it */
            const struct lz_match *match = cache_ptr - num_matches;
            unsigned len = MIN_MATCH_LEN;
            do {
                unsigned offset = match->offset;
                uint32_t offset_cost = c->costs.offset_slot[offset_slot(offset)];
                do {
                    uint32_t cost_to_end = offset_cost + c->costs.length[len] + (cur_node + len)->cost_to_end;
                    if (cost_to_end < best_cost_to_end) {
                        best_cost_to_end = cost_to_end;
                        cur_node->item = ((uint32_t)offset << OPTIMUM_OFFSET_SHIFT) | len;
                    }
                } while (++len <= match->length);
            } while (++match != cache_ptr);
            cache_ptr -= num_matches;
        }
        cur_node->cost_to_end = best_cost_to_end;
    } while (cur_node != &c->optimum_nodes[0]);

    memset(fr, 0, sizeof(*fr));
    for (uint32_t pos = 0; pos < block_length;) {
        uint32_t item = c->optimum_nodes[pos].item;
        unsigned length = item & OPTIMUM_LEN_MASK, offset = item >> OPTIMUM_OFFSET_SHIFT;
        if (length == 1) {
            fr->litlen[offset]++;
        } else {
            fr->litlen[FIRST_LEN_SYM + length_slot(length)]++;
            fr->offset[offset_slot(offset)]++;
        }
        pos += length;
    }
}

/* deflate_optimize_block, then the block goes to the sink as tokens (deflate_flush_block on the item list) */
static void no_optimize_and_flush(struct no_state *c, const uint8_t *block_begin, uint32_t block_length,
                                  const struct lz_match *cache_ptr, int is_first_block, int is_final_block,
                                  block_sink_fn sink, void *ctx)
{
    unsigned num_passes_remaining = c->num_optim_passes;
    uint32_t i, lit_cost, len_sym_cost;
    struct freqs fr;
    size_t nt = 0;

    /* the block really ends at block_length, even if matches reach beyond it */
    for (i = block_length;
         i <= (block_length - 1 + MAX_MATCH_LEN < NO_MAX_BLOCK_LENGTH ? block_length - 1 + MAX_MATCH_LEN : NO_MAX_BLOCK_LENGTH);
         i++)
        c->optimum_nodes[i].cost_to_end = 0x80000000u;

    no_choose_default_litlen_costs(c, block_begin, block_length, &lit_cost, &len_sym_cost);
    if (is_first_block)
        no_set_default_costs(c, lit_cost, len_sym_cost);
    else
        no_adjust_costs(c, lit_cost, len_sym_cost);

    do {
        /* a pass: the minimum-cost path under the current costs, the Huffman codes of what it uses (with the
         * end-of-block symbol tallied), and the costs those codes imply -- after the LAST pass as well: they are
         * what the next block blends its default costs with.  (Both details pinned on the v1.10 binary: without
         * the end-of-block symbol 141 of 165 single-block cases match it, with it all; with the costs left at
         * those of the last pass 106 of 132 two-block cases, updated 132.) */
        uint8_t litlen_lens[NUM_LITLEN_SYMS], offset_lens[NUM_OFFSET_SYMS];
        uint32_t litlen_cw[NUM_LITLEN_SYMS], offset_cw[NUM_OFFSET_SYMS];
        no_find_min_cost_path(c, block_length, cache_ptr, &fr);
        fr.litlen[END_OF_BLOCK]++;
        gzpx_oracle_make_huffman_code(NUM_LITLEN_SYMS, MAX_LITLEN_CODEWORD_LEN, c->compat, fr.litlen, litlen_lens, litlen_cw);
        gzpx_oracle_make_huffman_code(NUM_OFFSET_SYMS, MAX_OFFSET_CODEWORD_LEN, c->compat, fr.offset, offset_lens, offset_cw);
        no_set_costs_from_codes(c, litlen_lens, offset_lens);
        fr.litlen[END_OF_BLOCK]--; /* (flush_block tallies it itself) */
    } while (--num_passes_remaining);

    for (uint32_t pos = 0; pos < block_length;) {
        uint32_t item = c->optimum_nodes[pos].item;
        unsigned length = item & OPTIMUM_LEN_MASK, offset = item >> OPTIMUM_OFFSET_SHIFT;
        c->tokens[nt++] = length == 1 ? offset : (TOKEN_MATCH | (offset << 9) | length);
        pos += length;
    }
    sink(ctx, block_begin, block_length, c->tokens, nt, &fr, is_final_block);
}

static int compress_near_optimal(const uint8_t *in, size_t n, unsigned nice_match_length, unsigned max_search_depth,
                                 unsigned num_optim_passes, int compat, block_sink_fn sink, void *ctx)
{
    struct no_state *c;
    const uint8_t *in_next = in, *in_block_begin = in, *in_end = in + n, *in_cur_base = in, *in_next_slide;
    unsigned max_len = MAX_MATCH_LEN;
    unsigned nice_len = nice_match_length < max_len ? nice_match_length : max_len;
    struct lz_match *cache_ptr;
    uint32_t next_hashes[2] = {0, 0};

    if (!slot_tabs_ready)
        init_slot_tabs();
    if (!no_tables_ready) {
        gzpx_oracle_default_litlen_costs(no_default_lit_cost, no_default_len_sym_cost);
        no_tables_ready = 1;
    }
    if (!tl_no)
        tl_no = (struct no_state *)malloc(sizeof(*tl_no));
    c = tl_no;
    if (!c)
        return -1;
    c->max_search_depth = max_search_depth;
    c->nice_match_length = nice_match_length;
    c->num_optim_passes = num_optim_passes;
    c->compat = compat;
    memset(c->prev_observations, 0, sizeof(c->prev_observations));
    c->prev_num_observations = 0;
    cache_ptr = c->match_cache;
    in_next_slide = in_next + ((size_t)(in_end - in_next) < WINDOW_SIZE ? (size_t)(in_end - in_next) : WINDOW_SIZE);

    bt_init(&c->mf);
    no_init_stats(c);

    do {
        /* starting a new DEFLATE block */
        const uint8_t *in_max_block_end =
            ((size_t)(in_end - in_block_begin) < SOFT_MAX_BLOCK_LENGTH + MIN_BLOCK_LENGTH)
                ? in_end
                : in_block_begin + SOFT_MAX_BLOCK_LENGTH;
        const uint8_t *prev_end_block_check = NULL;
        int change_detected = 0;
        const uint8_t *next_observation = in_next;
        unsigned min_len = calculate_min_match_len(in_block_begin, (size_t)(in_max_block_end - in_block_begin),
                                                   max_search_depth, compat);

        for (;;) {
            struct lz_match *matches;
            unsigned best_len;
            size_t remaining = (size_t)(in_end - in_next);

            if (in_next == in_next_slide) {
                bt_slide(&c->mf);
                in_cur_base = in_next;
                in_next_slide = in_next + (remaining < WINDOW_SIZE ? remaining : WINDOW_SIZE);
            }
            matches = cache_ptr;
            best_len = 0;
            if (remaining < MAX_MATCH_LEN) {
                max_len = (unsigned)remaining;
                if (nice_len > max_len)
                    nice_len = max_len;
            }
            if (max_len >= BT_REQUIRED_NBYTES) {
                cache_ptr = bt_advance(&c->mf, in_cur_base, in_next - in_cur_base, max_len, nice_len, max_search_depth,
                                       next_hashes, matches, 1);
                if (cache_ptr > matches)
                    best_len = cache_ptr[-1].length;
            }
            if (in_next >= next_observation) {
                if (best_len >= min_len) {
                    c->split.new_observations[NUM_LITERAL_OBSERVATION_TYPES + (best_len >= 9)]++;
                    c->split.num_new_observations++;
                    next_observation = in_next + best_len;
                    c->new_match_len_freqs[best_len]++;
                } else {
                    uint8_t lit = *in_next;
                    c->split.new_observations[((lit >> 5) & 0x6) | (lit & 1)]++;
                    c->split.num_new_observations++;
                    next_observation = in_next + 1;
                }
            }
            cache_ptr->length = (uint16_t)(cache_ptr - matches);
            cache_ptr->offset = *in_next;
            in_next++;
            cache_ptr++;

            /* a very long match: no matches are cached for the bytes it covers */
            if (best_len >= MIN_MATCH_LEN && best_len >= nice_len) {
                --best_len;
                do {
                    remaining = (size_t)(in_end - in_next);
                    if (in_next == in_next_slide) {
                        bt_slide(&c->mf);
                        in_cur_base = in_next;
                        in_next_slide = in_next + (remaining < WINDOW_SIZE ? remaining : WINDOW_SIZE);
                    }
                    if (remaining < MAX_MATCH_LEN) {
                        max_len = (unsigned)remaining;
                        if (nice_len > max_len)
                            nice_len = max_len;
                    }
                    if (max_len >= BT_REQUIRED_NBYTES)
                        bt_advance(&c->mf, in_cur_base, in_next - in_cur_base, max_len, nice_len, max_search_depth,
                                   next_hashes, NULL, 0);
                    cache_ptr->length = 0;
                    cache_ptr->offset = *in_next;
                    in_next++;
                    cache_ptr++;
                } while (--best_len);
            }
            if (in_next >= in_max_block_end)
                break;
            if (cache_ptr >= &c->match_cache[MATCH_CACHE_LENGTH])
                break;
            if (!(c->split.num_new_observations >= NUM_OBSERVATIONS_PER_BLOCK_CHECK &&
                  (size_t)(in_next - in_block_begin) >= MIN_BLOCK_LENGTH && (size_t)(in_end - in_next) >= MIN_BLOCK_LENGTH))
                continue;
            if (do_end_block_check(&c->split, (uint32_t)(in_next - in_block_begin))) {
                change_detected = 1;
                break;
            }
            no_merge_stats(c);
            prev_end_block_check = in_next;
        }

        if (change_detected && prev_end_block_check != NULL) {
            /* a recent chunk differs from the rest of the block: rewind to just before it */
            struct lz_match *orig_cache_ptr = cache_ptr;
            const uint8_t *in_block_end = prev_end_block_check;
            uint32_t block_length = (uint32_t)(in_block_end - in_block_begin);
            uint32_t num_bytes_to_rewind = (uint32_t)(in_next - in_block_end);
            size_t cache_len_rewound;
            do {
                cache_ptr--;
                cache_ptr -= cache_ptr->length;
            } while (--num_bytes_to_rewind);
            cache_len_rewound = (size_t)(orig_cache_ptr - cache_ptr);
            no_optimize_and_flush(c, in_block_begin, block_length, cache_ptr, in_block_begin == in, 0, sink, ctx);
            memmove(c->match_cache, cache_ptr, cache_len_rewound * sizeof(*cache_ptr));
            cache_ptr = &c->match_cache[cache_len_rewound];
            no_save_stats(c);
            no_clear_old_stats(c);
            in_block_begin = in_block_end;
        } else {
            uint32_t block_length = (uint32_t)(in_next - in_block_begin);
            no_merge_stats(c);
            no_optimize_and_flush(c, in_block_begin, block_length, cache_ptr, in_block_begin == in, in_next == in_end,
                                  sink, ctx);
            cache_ptr = &c->match_cache[0];
            no_save_stats(c);
            no_init_stats(c);
            in_block_begin = in_next;
        }
    } while (in_next != in_end);
    return 0;
}

struct emit_ctx {
    struct bitwriter w;
    int compat;
};

static void emit_sink(void *vctx, const uint8_t *block_begin, size_t block_length,
                      const uint32_t *tokens, size_t n_tokens, struct freqs *fr, int is_final)
{
    struct emit_ctx *c = (struct emit_ctx *)vctx;
    flush_block(&c->w, c->compat, block_begin, block_length, tokens, n_tokens, fr, is_final);
}

size_t gzpx_oracle_deflate_bound(size_t n)
{
    /* libdeflate_deflate_compress_bound (v1.10): 5 bytes per 10000-byte stored block + 1 + 8 */
    size_t max_blocks = (n + 10000 - 1) / 10000;
    if (max_blocks < 1)
        max_blocks = 1;
    return 5 * max_blocks + n + 1 + 8;
}

size_t gzpx_oracle_deflate_compress(int level, int compat, const uint8_t *in, size_t n,
                                    uint8_t *out, size_t cap)
{
    struct emit_ctx c;
    memset(&c, 0, sizeof(c));
    c.w.out = out;
    c.w.cap = cap;
    c.compat = compat;
    if (level < 0 || level > 12)
        return 0;
    /* A.0: very short inputs (and level 0) are emitted as stored blocks only */
    if (level == 0 || n <= (size_t)(55 - 4 * level)) {
        write_stored(&c.w, in, n, 1);
    } else if (level == 1) {
        if (compress_fastest(in, n, emit_sink, &c) != 0)
            return 0;
        bw_align(&c.w);
    } else if (level <= 4) {
        /* level 2: depth 6 nice 10; level 3: depth 12 nice 14; level 4: depth 16 nice 30 */
        static const unsigned depth[5] = {0, 0, 6, 12, 16}, nice[5] = {0, 0, 10, 14, 30};
        if (compress_greedy(in, n, nice[level], depth[level], compat, emit_sink, &c) != 0)
            return 0;
        bw_align(&c.w);
    } else if (level >= 10) {
        /* levels 10-12: near-optimal parsing (max_search_depth / nice_match_length / num_optim_passes) */
        static const unsigned depth[3] = {35, 70, 150}, nice[3] = {75, 150, 258}, passes[3] = {2, 3, 4};
        if (compress_near_optimal(in, n, nice[level - 10], depth[level - 10], passes[level - 10], compat, emit_sink,
                                  &c) != 0)
            return 0;
        bw_align(&c.w);
    } else {
        /* levels 5-7 lazy, 8-9 lazy2 */
        static const unsigned depth[10] = {0, 0, 0, 0, 0, 16, 35, 100, 300, 600};
        static const unsigned nice[10] = {0, 0, 0, 0, 0, 30, 65, 130, 258, 258};
        if (compress_lazy(in, n, nice[level], depth[level], level >= 8, compat, emit_sink, &c) != 0)
            return 0;
        bw_align(&c.w);
    }
    if (c.w.overflow)
        return 0;
    return c.w.pos;
}

struct tok_ctx {
    uint32_t *tokens;
    size_t max_tokens, n_tokens;
    uint32_t *first;
    size_t max_sub, n_sub;
};

static void tok_sink(void *vctx, const uint8_t *block_begin, size_t block_length,
                     const uint32_t *tokens, size_t n_tokens, struct freqs *fr, int is_final)
{
    struct tok_ctx *c = (struct tok_ctx *)vctx;
    (void)block_begin;
    (void)block_length;
    (void)fr;
    (void)is_final;
    if (c->n_sub < c->max_sub)
        c->first[c->n_sub] = (uint32_t)c->n_tokens;
    c->n_sub++;
    for (size_t i = 0; i < n_tokens; i++) {
        if (c->n_tokens < c->max_tokens)
            c->tokens[c->n_tokens] = tokens[i];
        c->n_tokens++;
    }
}

size_t gzpx_oracle_l1_tokens(const uint8_t *in, size_t n, uint32_t *tokens, size_t max_tokens,
                             uint32_t *sub_block_first_token, size_t max_sub_blocks,
                             size_t *n_sub_blocks)
{
    struct tok_ctx c = {tokens, max_tokens, 0, sub_block_first_token, max_sub_blocks, 0};
    if (n_sub_blocks)
        *n_sub_blocks = 0;
    if (n <= 51)
        return 0;
    if (compress_fastest(in, n, tok_sink, &c) != 0)
        return 0;
    if (n_sub_blocks)
        *n_sub_blocks = c.n_sub;
    return c.n_tokens;
}

/* ------------------------------------------------------------------ framing */

/* src/bgzf.rs:24-38 */
static const uint8_t BGZF_EOF[28] = {0x1f, 0x8b, 0x08, 0x04, 0x00, 0x00, 0x00, 0x00, 0x00, 0xff,
                                     0x06, 0x00, 0x42, 0x43, 0x02, 0x00, 0x1b, 0x00, 0x03, 0x00,
                                     0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00};

static void put_le16(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
}

static void put_le32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}

static size_t extra_amount(size_t n)
{
    size_t e = (size_t)((double)n * 0.1); /* src/bgzf.rs:45,50-52 ; src/mgzip.rs:28-30 */
    return e < 128 ? 128 : e;
}

size_t gzpx_oracle_encode_block(int fmt, int level, int compat, const uint8_t *in, size_t n,
                                int is_last, uint8_t *out, size_t cap, int *err)
{
    const size_t hdr = (fmt == GZPX_ORACLE_FMT_BGZF) ? 18 : 20;
    size_t payload_cap, c, total;
    uint8_t xfl;
    int e = 0;
    if (err)
        *err = 0;
    if (level < 0 || level > 12) {
        if (err)
            *err = 3;
        return 0;
    }
    /* the reference sizes its Vec as hdr + n + extra_amount(n) + 8 and hands libdeflate
     * the slice after the header (src/bgzf.rs:211-216): capacity n + extra + 8 */
    payload_cap = n + extra_amount(n) + 8;
    if (cap < hdr + payload_cap) {
        if (err)
            *err = 1;
        return 0;
    }
    c = gzpx_oracle_deflate_compress(level, compat, in, n, out + hdr, payload_cap);
    if (c == 0)
        e = 1;
    else if (fmt == GZPX_ORACLE_FMT_BGZF && c >= 65536)
        e = 2; /* src/bgzf.rs:218-223 */
    if (e) {
        if (err)
            *err = e;
        return 0;
    }
    xfl = (level >= 9) ? 2 : (level <= 1) ? 4 : 0; /* src/bgzf.rs:278-284 */
    out[0] = 0x1f;
    out[1] = 0x8b;
    out[2] = 8;
    out[3] = 4;
    put_le32(out + 4, 0);
    out[8] = xfl;
    out[9] = 255;
    if (fmt == GZPX_ORACLE_FMT_BGZF) {
        put_le16(out + 10, 6);
        out[12] = 'B';
        out[13] = 'C';
        put_le16(out + 14, 2);
        put_le16(out + 16, (uint32_t)(c + 26 - 1)); /* src/bgzf.rs:298-300 */
    } else {
        put_le16(out + 10, 8);
        out[12] = 'I';
        out[13] = 'G';
        put_le16(out + 14, 4);
        put_le32(out + 16, (uint32_t)(c + 28)); /* src/mgzip.rs:270-272 */
    }
    total = hdr + c;
    put_le32(out + total, gzpx_oracle_crc32(0, in, n));
    put_le32(out + total + 4, (uint32_t)n);
    total += 8;
    if (is_last && fmt == GZPX_ORACLE_FMT_BGZF) { /* src/deflate.rs:622-624 */
        if (cap < total + sizeof(BGZF_EOF)) {
            if (err)
                *err = 1;
            return 0;
        }
        memcpy(out + total, BGZF_EOF, sizeof(BGZF_EOF));
        total += sizeof(BGZF_EOF);
    }
    return total;
}

size_t gzpx_oracle_compress_stream(int fmt, int level, int compat, size_t buffer_size,
                                   const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                                   uint32_t *block_sizes, size_t max_blocks, size_t *n_blocks,
                                   int *err)
{
    size_t pos = 0, opos = 0, nb = 0;
    uint8_t *tmp;
    size_t tmp_cap = 20 + buffer_size + extra_amount(buffer_size) + 8 + 28;
    if (err)
        *err = 0;
    if (n_blocks)
        *n_blocks = 0;
    tmp = (uint8_t *)malloc(tmp_cap);
    if (!tmp)
        return 0;
    /* write(): dispatch full blocks while strictly more than buffer_size is buffered
     * (src/par/compress.rs:415); flush_last(true): at least one block, is_last on the final
     * piece (src/par/compress.rs:333-341). */
    for (;;) {
        size_t remaining = n - pos;
        size_t take = remaining > buffer_size ? buffer_size : remaining;
        int is_last = (take == remaining);
        int e = 0;
        size_t got = gzpx_oracle_encode_block(fmt, level, compat, in + pos, take, is_last, tmp,
                                              tmp_cap, &e);
        if (got == 0) {
            if (err)
                *err = e ? e : 1;
            free(tmp);
            return 0;
        }
        if (opos + got > cap) {
            if (err)
                *err = 1;
            free(tmp);
            return 0;
        }
        memcpy(out + opos, tmp, got);
        opos += got;
        if (block_sizes && nb < max_blocks)
            block_sizes[nb] = (uint32_t)got;
        nb++;
        pos += take;
        if (is_last)
            break;
    }
    free(tmp);
    if (n_blocks)
        *n_blocks = nb;
    return opos;
}
