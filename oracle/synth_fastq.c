/* synth_fastq.c -- CPU statement of the synthetic FASTQ stream of BASELINE.json configs[3]
 * (SURVEY.md 8(d) "Config 4 input").  TEST INFRASTRUCTURE: the checker of the device generator
 * gzpx_synth_fastq_device (gzp_amd/csrc/gzpx_synth.hip) and the source of the full-size golden
 * digests; never linked into the product.
 *
 * The stream is defined in independent 64 KiB pages so that any byte range of a 32 GiB stream can
 * be produced anywhere (one GPU thread per page, any rank's shard) without the bytes before it:
 *   page c:  rng = splitmix64 started at (seed ^ c * 0xD6E8FEB86659FD93); records until the page is
 *            full, the last one cut at the page end (one broken record per ~200: the stream stays
 *            FASTQ-shaped, which is all the compressor sees).
 *   record k of page c, read number r = c * 512 + k + 1:
 *            "@SRR" %07d(seed % 10^7) "." r " " r "/1\n"
 *            L = 100 + next() % 51 bases: 5 per next() (12 bits each: low 10 bits zero -> 'N',
 *            else "ACGT"[bits 10..11]), "\n+\n", L qualities: a 4-state chain over "F:,#"
 *            (start 'F', 21 steps per next(), 3 bits each through the table below), "\n".
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define PAGE 65536u

static const char kQual[4] = {'F', ':', ',', '#'};
/* next state by (state, 3 random bits) */
static const uint8_t kStep[4][8] = {
    {0, 0, 0, 0, 0, 0, 0, 1},
    {0, 0, 0, 1, 1, 1, 1, 2},
    {0, 1, 1, 2, 2, 2, 2, 3},
    {2, 2, 3, 3, 3, 3, 3, 3},
};

typedef struct {
    uint64_t s;
} rng_t;

static uint64_t next64(rng_t *r) {
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static unsigned put_dec(uint8_t *p, uint64_t v) {
    char tmp[24];
    unsigned n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    for (unsigned i = 0; i < n; i++) p[i] = (uint8_t)tmp[n - 1 - i];
    return n;
}

static void gen_page(uint64_t seed, uint64_t c, uint8_t *page) {
    rng_t r = {seed ^ (c * 0xD6E8FEB86659FD93ull)};
    uint8_t rec[512];
    uint32_t fill = 0;
    for (uint32_t k = 0; fill < PAGE; k++) {
        unsigned n = 0;
        const uint64_t readno = c * 512 + k + 1;
        memcpy(rec, "@SRR", 4);
        n = 4;
        {
            uint32_t run = (uint32_t)(seed % 10000000ull);
            for (int d = 6; d >= 0; d--) {
                rec[n + d] = (uint8_t)('0' + run % 10);
                run /= 10;
            }
            n += 7;
        }
        rec[n++] = '.';
        n += put_dec(rec + n, readno);
        rec[n++] = ' ';
        n += put_dec(rec + n, readno);
        rec[n++] = '/';
        rec[n++] = '1';
        rec[n++] = '\n';
        const uint32_t L = 100 + (uint32_t)(next64(&r) % 51);
        for (uint32_t i = 0; i < L; i += 5) {
            uint64_t v = next64(&r);
            for (uint32_t j = 0; j < 5 && i + j < L; j++, v >>= 12)
                rec[n++] = (v & 1023) == 0 ? 'N' : (uint8_t)"ACGT"[(v >> 10) & 3];
        }
        rec[n++] = '\n';
        rec[n++] = '+';
        rec[n++] = '\n';
        uint32_t st = 0;
        for (uint32_t i = 0; i < L; i += 21) {
            uint64_t v = next64(&r);
            for (uint32_t j = 0; j < 21 && i + j < L; j++, v >>= 3) {
                st = kStep[st][v & 7];
                rec[n++] = (uint8_t)kQual[st];
            }
        }
        rec[n++] = '\n';
        const uint32_t take = n < PAGE - fill ? n : PAGE - fill;
        memcpy(page + fill, rec, take);
        fill += take;
    }
}

/* bytes [offset, offset + n) of the stream with this seed */
void gzpx_oracle_fastq(uint64_t seed, uint64_t offset, uint8_t *out, size_t n) {
    uint8_t page[PAGE];
    size_t done = 0;
    while (done < n) {
        const uint64_t pos = offset + done, c = pos / PAGE;
        const uint32_t in_page = (uint32_t)(pos % PAGE);
        size_t take = PAGE - in_page;
        if (take > n - done) take = n - done;
        gen_page(seed, c, page);
        memcpy(out + done, page + in_page, take);
        done += take;
    }
}

/* bytes [offset, offset + n) of the printable-ASCII noise of BASELINE configs[2]:
 * byte i = 0x20 + (splitmix64 output i >> 56) % 95 (gzp_amd/synth.py ascii_random, gzpx_synth_ascii_device) */
void gzpx_oracle_ascii(uint64_t seed, uint64_t offset, uint8_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint64_t z = seed + (offset + i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        out[i] = (uint8_t)(0x20 + (z >> 56) % 95);
    }
}
