// gzpx_device.h -- data layout shared by the HIP kernels and the host pipeline.
//
// Vocabulary (follows the reference): a *block* is one BGZF/Mgzip member = one `buffer_size`
// cut of the caller's stream (src/par/compress.rs:415-416); a *sub-block* is one DEFLATE block
// inside its payload (libdeflate starts a new one every 8192 matches); a *token* is one
// literal or one match of the level-1 greedy parse.
#pragma once
#include <stdint.h>

namespace gzpx {

constexpr unsigned kTile = 65536;         // positions handled per LDS tile (k_parse) / stage window
constexpr unsigned kMaxBlockSize = 1u << 26;  // largest buffer_size the kernels accept (64 MiB; round 6: tested at 32 MiB)
constexpr unsigned kSeqPerSub = 8192;      // FAST_SEQ_STORE_LENGTH
constexpr unsigned kSoftMaxSub = 65535;    // FAST_SOFT_MAX_BLOCK_LENGTH
constexpr unsigned kMinBlockLen = 5000;    // MIN_BLOCK_LENGTH
constexpr unsigned kNumLitlen = 288;
constexpr unsigned kNumOffset = 32;
constexpr unsigned kHistStride = kNumLitlen + kNumOffset;  // 320 u32 per sub-block
constexpr unsigned kHdrWords = 160;                        // dynamic header bit string, <= 4554 bits
constexpr unsigned kCodeWords = kNumLitlen + kNumOffset;   // (codeword | len << 16) per symbol

constexpr uint32_t kTokMatch = 0x80000000u;  // token = kTokMatch | offset << 9 | length

enum SubType : uint32_t { kStored = 0, kStatic = 1, kDynamic = 2 };

enum BlockStatus : uint32_t {
    kStatusOk = 0,
    kStatusBlockSizeExceeded = 2,  // BGZF payload >= 65536 (src/bgzf.rs:218-223)
    kStatusInternal = 3,           // a capacity this implementation sizes was exceeded (reported as a device error)
};

struct SubMeta {
    uint32_t type;       // SubType
    uint32_t tok_begin;  // token range in the block's token array
    uint32_t tok_end;
    uint32_t byte_begin;  // input range covered (for stored blocks and the uncompressed cost)
    uint32_t byte_len;
    uint32_t bit_begin;  // first bit of this sub-block inside the payload
    uint32_t hdr_bits;   // bits of BFINAL+BTYPE(+dynamic header) held in hdr[]
    uint32_t is_final;
};

struct BlockMeta {
    uint32_t n;              // input bytes
    uint32_t is_last;        // append BGZF_EOF
    uint32_t ntok;
    uint32_t nsub;
    uint32_t payload_bytes;  // raw DEFLATE size c
    uint32_t framed_bytes;   // header + c + footer (+ EOF)
    uint32_t crc;
    uint32_t status;
};

// What the host needs back from one batch (written by k_scan): 16 bytes instead of every BlockMeta.
struct SlabResult {
    uint64_t total;        // framed bytes of the slab up to and including this batch
    uint32_t fail_block;   // first block whose status is not kStatusOk, 0xFFFFFFFF = none
    uint32_t fail_status;  // that block's BlockStatus
};

// Levels 2-4: per-block parse state carried between tiles and between match/parse rounds (a
// round ends early when a new DEFLATE sub-block needs another minimum match length).
struct HcState {
    uint32_t done;        // the block's token stream is complete
    uint32_t min_len;     // min_len of the sub-block that starts at resume_pos (0 = not computed yet)
    uint32_t resume_pos;  // where match results must be (re)computed and the parse resumes
    uint32_t tok_carry;   // tokens emitted before resume_pos
    uint32_t mat_carry;   // matches emitted before resume_pos
    uint32_t cur_sub;     // index of the sub-block that starts at resume_pos
    uint32_t rounds;      // diagnostics
    uint32_t pad;
    // what k_match_hc's arrays (len8 / which / alt) hold for this block (round 5, k_match_hc_sparse):
    //   0 nothing yet, or not usable: the dense k_match_hc must run (from resume_pos)
    //   1 the full search ONLY for the token starts of the greedy parse from position 0 with HcState.min_len (every other
    //     position: the search's first chain node) -- all a parse needs as long as min_len does not change
    //   2 a later sub-block needs another min_len (k_parse_hc found out, and listed the block for k_match_hc_stale): as 0
    //     until that kernel has been over the block, as 3 from resume_pos on behind it (the value stays 2)
    //   3 the full search of every position (the dense kernel ran)
    uint32_t sparse;
};
constexpr uint32_t kHcArraysNone = 0, kHcArraysPath = 1, kHcArraysStale = 2, kHcArraysDense = 3;

struct CrcConsts {
    uint32_t pow64[10];  // x^(8*64*2^l) mod P (reflected), l = 0..9
    uint32_t pow_tile;   // x^(8*65536) mod P: appends one full 64 KiB chunk
    uint32_t pow_small;  // x^(8*16384) mod P: the 256-thread routine's 16 KiB chunk
};

struct Config {
    uint32_t format;      // 0 BGZF, 1 Mgzip
    uint32_t level;       // 1
    uint32_t compat;      // 0: libdeflate >= 1.1x Huffman rule, 1: libdeflate 1.10
    uint32_t block_size;  // buffer_size of the reference's builder
    uint32_t xfl;         // gzip XFL byte derived from level (src/bgzf.rs:278-284)
    uint32_t debug;       // diagnostics only: bit 0 = k_candidates' order-independent fallback on every block, bit 1 =
                          // level 1 through the dense k_match / k_parse pair instead of k_mparse, bit 2 =
                          // k_mparse hands every block back (exercises the redo list); bits 8-9 (host side,
                          // experiments): where the CRC kernel is forked onto the side stream
    uint32_t stride;      // per-block stride (positions) of cand / len8 / alt / tok: >= block_size + 1024
    uint32_t max_sub;     // per-block capacity of sub / hist / codes / hdr (sub-blocks >= 32768 bytes)
    uint32_t passthrough; // n <= 55 - 4*level is emitted as stored blocks only (deflate_compress_none)
    uint32_t hc_depth;    // levels 2-9: max_search_depth
    uint32_t hc_nice;     // levels 2-9: nice_match_length
    uint32_t lazy;        // 0: greedy parser (levels 2-4), 1: lazy (5-7), 2: lazy2 (8-9)
    uint32_t n_cu;        // compute units of the device (persistent kernels launch one workgroup per CU)
    uint32_t no_passes;   // levels 10-12: num_optim_passes (hc_depth / hc_nice hold max_search_depth / nice_match_length)
};

// Device scratch for one batch of blocks.
struct Scratch {
    BlockMeta *meta;      // [nb]
    SubMeta *sub;         // [nb][max_sub]
    uint16_t *cand;       // [nb][stride]        d0: distance to the bucket predecessor (0 = none)
    uint8_t *len8;        // [nb][stride]        0 = no match at p, else match length - 3
    uint32_t *which;      // [nb][stride/32]     bit p: the older candidate won at p
    uint16_t *alt;        // [nb][stride]        match distance at p where that bit is set
    uint16_t *d4;         // [nb][stride]        levels 2-9: distance to the hash4 chain predecessor
    uint8_t *lz_len;      // [nb][2][stride]     levels 5-9: length - 3 of the half / quarter depth searches
    uint16_t *lz_dist;    // [nb][2][stride]     levels 5-9: their distances (0 = no match)
    HcState *hc;          // [nb]                levels 2-4: parse state
    uint32_t *pending;    // [1]                 levels 2-4: blocks that need another round
    uint32_t *tok;        // [nb][stride]        worst case one token per byte
    uint32_t *redo;       // [1 + nb]            level 1: blocks k_mparse hands back to k_match / k_parse
    uint32_t *claim;      // [8]                 ticket counters of the persistent kernels that claim their blocks (behind redo's list)
    uint32_t *hist;       // [nb][max_sub][kHistStride]
    uint32_t *codes;      // [nb][max_sub][kCodeWords]
    uint32_t *hdr;        // [nb][max_sub][kHdrWords]
    uint64_t *out_off;    // [nb + 1] byte offset of each framed block in the output
    uint32_t *sizes;      // [nb]     framed size of each block (compact copy for the host / the index)
    // levels 10-12 (gzpx_nearopt.hip): the state of the blocks in flight, one lane each
    uint32_t no_lanes;    // how many
    void *no_state;       // [no_lanes] NoLane: trees, hash tables, costs, frequencies, Huffman scratch
    uint8_t *no_cache;    // [no_lanes] match cache (libdeflate's MATCH_CACHE_LENGTH entries + slack)
    uint8_t *no_nodes;    // [no_lanes] minimum-cost path nodes of a DEFLATE block
};

// Host-side launchers (gzpx_kernels.hip).  All asynchronous on `stream`.
void launch_init_meta(const Config &cfg, uint64_t slab_len, uint32_t nb, int is_last,
                      const Scratch &s, hipStream_t stream);
void launch_candidates(const Config &cfg, const uint8_t *slab, uint64_t slab_len, uint32_t nb, int is_last,
                       const Scratch &s, hipStream_t stream);  // (fills BlockMeta too: k_init_meta's work)
void launch_match(const Config &cfg, const uint8_t *slab, uint64_t slab_len, uint32_t nb,
                  const Scratch &s, hipStream_t stream);
void launch_parse(const Config &cfg, const uint8_t *slab, uint64_t slab_len, uint32_t nb,
                  const Scratch &s, hipStream_t stream);
void launch_hc(const Config &cfg, const uint8_t *slab, uint32_t nb, const Scratch &s, hipStream_t stream);
void launch_lazy(const Config &cfg, const uint8_t *slab, uint32_t nb, const Scratch &s, hipStream_t stream);
// levels 10-12 (gzpx_nearopt.hip)
size_t no_lane_bytes();
size_t no_cache_bytes();
size_t no_nodes_bytes(uint32_t block_size);
void launch_near_optimal_tables(void *lanes, uint32_t n_lanes, const uint8_t *d_tables, hipStream_t stream);
void launch_near_optimal(const Config &cfg, const uint8_t *slab, uint32_t nb, const Scratch &s, hipStream_t stream);
void launch_hist(const Config &cfg, uint32_t nb, const Scratch &s, hipStream_t stream);
void launch_huffman(const Config &cfg, uint32_t nb, const Scratch &s, hipStream_t stream);
void launch_crc32(const Config &cfg, const uint8_t *slab, uint64_t slab_len, uint32_t nb,
                  const Scratch &s, const CrcConsts &cc, hipStream_t stream);
void launch_scan(uint32_t nb, const Scratch &s, const SlabResult *prev, SlabResult *result, hipStream_t stream);
void launch_emit(const Config &cfg, const uint8_t *slab, uint64_t slab_len, uint32_t nb,
                 const Scratch &s, uint8_t *out, uint64_t out_cap, hipStream_t stream);

// ParDecompress side: one record per block (filled by k_dinit / k_inflate)
struct DBlockHost {
    uint64_t in_off;
    uint32_t size, isize, crc, status, produced, nmatch;
    uint32_t cyc[8];  // debug launches: see DBlock in gzpx_kernels.hip
};
// Scratch of the two-kernel inflate (gzpx_inflate_seg.h): the members' match records, the first record of every
// 32 KiB output tile, and the list of members handed back to k_inflate ([0] = how many).
struct InflateScratch {
    void *mlist = nullptr;       // inflate_mlist_bytes(out_cap, nb)
    uint32_t *tfirst = nullptr;  // inflate_tfirst_bytes(out_cap, nb)
    uint32_t *redo = nullptr;    // [1 + nb] the list, [1 + nb] behind it k_inflate_seg's ticket counter
    uint32_t *summary = nullptr; // [12] k_dsummary: the first failing member and its checksums (both routes); [15] k_inflate_seg's first-block hint
    int n_cu = 0;                // compute units of the device (the size of the persistent launch)
    int big_members = 0;         // the slab's members average >= 128 KiB compressed: several waves work on each
};
enum { kInflateRouteSeg = 0, kInflateRouteWave = 1 };  // k_inflate_seg + k_lzcopy (default) | k_inflate for every member
size_t inflate_mlist_bytes(uint64_t out_cap, uint64_t nb);
size_t inflate_tfirst_bytes(uint64_t out_cap, uint64_t nb);
void launch_inflate(uint32_t hdr_len, const uint8_t *d_in, const uint64_t *d_offsets, const uint32_t *d_sizes,
                    uint32_t nb, void *d_blk, uint64_t *d_out_off, uint8_t *d_out, uint64_t out_cap,
                    uint32_t *d_crc_found, const CrcConsts &cc, int debug, hipEvent_t ev_begin,
                    hipEvent_t ev_end, hipStream_t stream, const InflateScratch &sc, int route, hipEvent_t ev_mid = nullptr);

// gzpx_check.hip: (s1, s2, n) of every 64 KiB tile of d_in[0..n) -> d_out3[3 * tile + ..]
void launch_adler32(const uint8_t *d_in, uint64_t n, uint32_t *d_out3, hipStream_t stream);

}  // namespace gzpx
