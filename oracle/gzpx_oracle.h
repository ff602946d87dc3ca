/*
 * gzpx_oracle.h -- CPU restatement of the gzp ParCompress<Bgzf/Mgzip> hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.
 *
 * What it restates (reference file:line, relative to /root/reference):
 *   - bgzf::compress / header_inner / BGZF_EOF        src/bgzf.rs:20-52,204-237,274-303
 *   - mgzip::compress / header_inner                  src/mgzip.rs:187-218,246-275
 *   - Bgzf::encode / Mgzip::encode (is_last -> EOF)   src/deflate.rs:613-626,463-472
 *   - ParCompress::write / flush_last / finish cuts   src/par/compress.rs:332-362,377-388,413-463
 *   - libdeflater::Compressor::deflate_compress and libdeflater::Crc, i.e. the C
 *     library libdeflate pinned by Cargo.lock:414-430 (libdeflate-sys 1.24.0).
 *     Its source is NOT vendored under /root/reference; the algorithm is restated
 *     from its published behaviour and pinned byte-for-byte
 *     against the libdeflate v1.10 binary of this image (see oracle/README.md and
 *     tests/golden/make_golden.py).  compat selects the one known v1.10/v1.24 delta.
 */
#ifndef GZPX_ORACLE_H
#define GZPX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GZPX_ORACLE_COMPAT_1_24 0 /* libdeflate >= 1.1x rule: never emit an empty Huffman code   */
#define GZPX_ORACLE_COMPAT_1_10 1 /* libdeflate 1.10 rule: empty offset code has all lengths zero */

#define GZPX_ORACLE_FMT_BGZF 0
#define GZPX_ORACLE_FMT_MGZIP 1

/* gzip CRC-32 (libdeflate_crc32 semantics: pass 0 to start). */
uint32_t gzpx_oracle_crc32(uint32_t crc, const void *buf, size_t n);

/* Worst-case raw DEFLATE size (libdeflate_deflate_compress_bound semantics). */
size_t gzpx_oracle_deflate_bound(size_t n);

/*
 * Raw DEFLATE of in[0..n) at `level` (0 = stored only, 1 = "fastest" ht_matchfinder parse,
 * 2..4 = greedy hc_matchfinder parse, 5..9 = lazy / lazy2, 10..12 = the near-optimal parser as in
 * libdeflate v1.10 -- later versions changed it: pinned for that version, tests/golden/l1012_vectors.json).
 * Returns the number of bytes written to out, or 0 if they do not fit in cap
 * (libdeflate_deflate_compress semantics) or the level is out of range.
 */
size_t gzpx_oracle_deflate_compress(int level, int compat, const uint8_t *in, size_t n,
                                    uint8_t *out, size_t cap);

/* libdeflate's default_litlen_costs[] (levels 10-12), computed: lit[i][j] for match probability 0.25 / 0.5 / 0.75 and
 * j used literals, len_sym[i] the cost of a length symbol (the same bytes sit in the v1.10 binary's read-only data) */
void gzpx_oracle_default_litlen_costs(uint8_t lit[3][257], uint8_t len_sym[3]);

/*
 * One framed block: bgzf::compress / mgzip::compress (+ BGZF_EOF when is_last and fmt==BGZF).
 * Returns bytes written, 0 on "does not fit"; *err (optional) gets 0 ok, 1 insufficient space,
 * 2 BlockSizeExceeded (BGZF payload >= 65536), 3 bad level.
 */
size_t gzpx_oracle_encode_block(int fmt, int level, int compat, const uint8_t *in, size_t n,
                                int is_last, uint8_t *out, size_t cap, int *err);

/*
 * Whole stream as ParCompress<fmt> would emit it for ONE write_all(in[0..n)) followed by
 * finish(): blocks are cut by the strict `>` rule of write() and flush_last(true) emits at
 * least one (possibly empty) block; EOF marker after the last block for BGZF.
 * block_sizes (optional, capacity max_blocks) receives each framed block's byte size
 * (the last one including the EOF marker).  Returns total bytes or 0 on error.
 */
size_t gzpx_oracle_compress_stream(int fmt, int level, int compat, size_t buffer_size,
                                   const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                                   uint32_t *block_sizes, size_t max_blocks, size_t *n_blocks,
                                   int *err);

/* ---- intermediate products of the level-1 parse, for per-kernel parity tests ---- */

/*
 * Level-1 token stream of in[0..n): tokens[i] = literal byte (bit 31 clear) or
 * (1u<<31) | (offset << 9) | length for a match.  sub_block_first_token[k] is the index of
 * the first token of DEFLATE sub-block k.  Returns the number of tokens (n <= 51 inputs take
 * the stored-only path in deflate_compress and have no token stream: returns 0).
 */
size_t gzpx_oracle_l1_tokens(const uint8_t *in, size_t n, uint32_t *tokens, size_t max_tokens,
                             uint32_t *sub_block_first_token, size_t max_sub_blocks,
                             size_t *n_sub_blocks);

/*
 * make_code of libdeflate (length-limited canonical Huffman): freqs[num_syms] ->
 * lens[num_syms], codewords[num_syms] (bit-reversed, ready to OR in LSB-first).
 */
void gzpx_oracle_make_huffman_code(unsigned num_syms, unsigned max_len, int compat,
                                   const uint32_t *freqs, uint8_t *lens, uint32_t *codewords);

#ifdef __cplusplus
}
#endif
#endif
