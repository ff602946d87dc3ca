// gzpx_inflate_seg.h -- the two-kernel form of ParDecompress's decode_block (src/par/decompress.rs:162-186,
// src/bgzf.rs:151-170: libdeflate_deflate_decompress of one member).  Included by gzpx_kernels.hip inside
// namespace gzpx, behind k_inflate (whose DBlock and status codes it shares).
//
//   k_inflate_seg  Huffman decode only: bitstream -> literal bytes in place + a list of (position, length,
//                  distance) records; it never reads the window.  One wave per member.  The bits of a DEFLATE
//                  block are cut into 64 segments, one per lane; every lane decodes ITS segment serially from
//                  a speculative start (pass 1), lane i then replays from lane i-1's exit beside its own
//                  speculative path until the two meet (pass 2: a DEFLATE stream re-synchronises after ~100
//                  bits, so this is short), a prefix sum of the byte / match counts gives every lane its
//                  output offsets, and the lanes decode their segment once more from the true entry (pass 3),
//                  storing literals at their final place and matches in the member's record list.
//                  The symbol step is straight-line code: two-level tables (no slow path), positional reads of
//                  the lane's own compressed dwords from a lane-private LDS ring that is refilled for all
//                  lanes at once, 256 bits at a time (no per-lane memory instruction inside the loop).
//   k_lzcopy       one workgroup per member: the member's output (literals in place, holes where matches go)
//                  goes through LDS 32 KiB at a time; matches are resolved one per lane, 1,024 at a time in list
//                  order, by waves that poll a "byte is final" bitmap (no barrier between dependency levels);
//                  sources in front of the tile are final and come from HBM; finished tiles go back in 16-byte
//                  stores.
//
// Anything out of the ordinary on the true path (invalid codes, a distance before the start, output that does
// not fit, a stream that ends early, lanes that do not re-synchronise) is NOT judged here: the member is put
// on the redo list and k_inflate, which holds libdeflate's exact error classes, decodes it again.

constexpr uint32_t kInfRedo = 0x80u;     // DBlock.status while a member waits for k_inflate
constexpr uint32_t kSegMinBits = 512u;   // segment length per lane: remaining bits / 64, within these bounds
constexpr uint32_t kSegMaxBits = 8192u;
constexpr uint32_t kSegMaxFix = 8u;      // pass-2 iterations before the member is handed to k_inflate
constexpr uint32_t kSegWinDw = 4u;       // dwords per refill window (128 bits of every lane's stream)
constexpr uint32_t kSegRingDw = 2u * kSegWinDw;  // the ring: two windows
constexpr uint32_t kSegRingStride = kSegRingDw + 2u;  // dwords per lane: + copies of slots 0 and 1 behind the last slot (a step reads three consecutive dwords)
constexpr uint32_t kSegLRoot = 10u, kSegORoot = 8u, kSegPRoot = 7u;  // root bits of the litlen / offset / precode tables
constexpr uint32_t kSegLSub = 320u, kSegOSub = 160u;  // second-level entries (ENOUGH(288,10,15) - 1024 = 310, (32,8,15): 146)
#ifndef GZPX_LZ_TILE_SHIFT
#define GZPX_LZ_TILE_SHIFT 14
#endif
constexpr uint32_t kLzTileShift = GZPX_LZ_TILE_SHIFT;
constexpr uint32_t kLzTile = 1u << kLzTileShift;  // k_lzcopy works on this much output at a time (16 KiB: six workgroups per CU; measured 1.84 ms at 32 KiB / four, 1.56 at 16 / six, 1.61 at 8 / eight)

struct __attribute__((aligned(8))) LzMatch {
    uint32_t pos;       // first output byte, relative to the member's output
    uint32_t len_dist;  // length << 16 | distance (1..32768)
};

// Table entries (32 bit).  bits 0-3: codeword length; 0 = a pointer (bits 12-15: index bits of the second level,
// bits 16-24: its first entry) -- an unused codeword of the root is the all-zero pointer to second-level entry 0, which
// holds the "invalid" entry like every second-level slot no codeword fills.
//   litlen:  bit 4 a length, bit 5 end of block, bit 6 a literal (none of the three: an invalid codeword); bits 8-10 extra
//            bits (0 unless a length), bits 16-24 the literal byte or the base length, bits 25-29 codeword length + extra bits;
//   offset:  bit 4 invalid, bits 8-11 extra bits, bits 16-30 base distance;
//   precode: bits 16-20 the symbol.
// An invalid codeword is ONE BIT LONG and produces nothing (kSegBadL / kSegBadO): a path that meets one steps over it.
// Only speculative paths do in a valid stream, and they are better off walking on until they meet the true path than
// stopping (pass 2 would have to replay their whole segment); on the true path pass 3 sees the missing type bits / bit 4
// and hands the member to k_inflate.  Every pass applies the same rule, so their bit positions and counts agree.
enum SegKind { kSegLitlen = 0, kSegOffset = 1, kSegPrecode = 2 };
constexpr uint32_t kSegLen = 1u << 4, kSegEob = 1u << 5, kSegLit = 1u << 6;
constexpr uint32_t kSegBadL = 1u | (1u << 25), kSegBadO = 1u | (1u << 4);
template <int KIND>
__device__ __forceinline__ uint32_t seg_entry(uint32_t sym, uint32_t cl) {
    if (KIND == kSegPrecode) return cl | (sym << 16);
    if (KIND == kSegLitlen) {
        if (sym < 256) return cl | kSegLit | (sym << 16) | (cl << 25);
        if (sym == 256) return cl | kSegEob | (cl << 25);
        if (sym > 285) return kSegBadL;
        const uint32_t slot = sym - 257;
        uint32_t base, xb = 0;
        if (slot < 8) {
            base = 3 + slot;
        } else if (slot == 28) {
            base = 258;
        } else {
            xb = (slot - 4) >> 2;
            base = 3 + ((4 + (slot & 3)) << xb);
        }
        return cl | kSegLen | (xb << 8) | (base << 16) | ((cl + xb) << 25);
    }
    if (sym > 29) return kSegBadO;
    uint32_t base, xb = 0;
    if (sym < 4) {
        base = 1 + sym;
    } else {
        xb = (sym - 2) >> 1;
        base = 1 + ((2 + (sym & 1)) << xb);
    }
    return cl | (xb << 8) | (base << 16);
}

struct InfSegLds {
    uint32_t lfast[1u << kSegLRoot];  // (its first 416 bytes double as code-length scratch + staged header while a header is parsed)
    uint32_t lsub[kSegLSub];
    uint32_t ofast[1u << kSegORoot];  // (and the 7-bit precode table)
    uint32_t osub[kSegOSub];
    uint8_t lens[320];                // code lengths: litlen then offset
    uint32_t first[16];               // builder scratch: first canonical code of every length
    uint32_t alloc;                   // builder scratch: next free second-level entry
};
// A member's LDS: the tables (one copy for the W waves that work on it), every lane's ring of its own compressed dwords
// (lane i of wave w at [(64 w + i) * stride + ((dword - first dword) & (ring - 1))]; while a table is built, wave 0's
// part holds the symbols' codewords), what wave 0 found in a block header, and what the waves tell each other.
template <int W>
struct InfSegLdsW {
    InfSegLds t;
    uint32_t ring[W * 64 * kSegRingStride];
    uint32_t hres[8];
    uint32_t x_ex[W], x_first[W], x_fl[W], x_n[W], x_m[W];
};

// Build the two-level decode table of one code from lens[0 .. nsyms) (nsyms <= 320): `root` index bits in
// main[], longer codewords behind pointer entries in sub[] (sub[0] stays "invalid": where unused codewords land).  All 64
// lanes call it.  Returns false for an over-subscribed code or a second level that does not fit.
template <int KIND>
__device__ __attribute__((noinline)) bool seg_build(InfSegLds &h, uint16_t *cwtab, const uint8_t *lens, uint32_t nsyms, uint32_t root,
                                       uint32_t *main, uint32_t *sub, uint32_t sub_cap, uint32_t lane) {
    const uint64_t lane_below = (1ull << lane) - 1ull;
    const uint32_t rounds = (nsyms + 63) >> 6;
    uint32_t cnt[16];
#pragma unroll
    for (uint32_t l = 0; l < 16; l++) cnt[l] = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t s = 64 * r + lane;
        const uint32_t myl = s < nsyms ? lens[s] : 0;
#pragma unroll
        for (uint32_t l = 1; l <= 15; l++) cnt[l] += (uint32_t)__popcll(__ballot(myl == l));
    }
    uint32_t code = 0, kraft = 0;
    wave_sync();
#pragma unroll
    for (uint32_t l = 1; l <= 15; l++) {
        code <<= 1;
        if (lane == 0) h.first[l] = code;
        code += cnt[l];
        kraft += cnt[l] << (15 - l);
    }
    if (kraft > (1u << 15)) return false;
    for (uint32_t i = lane; i < (1u << root); i += 64) main[i] = 0;
    if (sub)
        for (uint32_t i = lane; i < sub_cap; i += 64) sub[i] = KIND == kSegLitlen ? kSegBadL : kSegBadO;
    if (lane == 0) h.alloc = 1;
    wave_sync();
    // codewords by rank among the symbols of the same length; short ones fill main[], long ones leave their
    // length at main[first `root` bits] (the longest of the group stays: the size of its second level)
    uint32_t run[16];
#pragma unroll
    for (uint32_t l = 0; l < 16; l++) run[l] = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t s = 64 * r + lane;
        const uint32_t myl = s < nsyms ? lens[s] : 0;
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t l = 1; l <= 15; l++) {
            const uint64_t m = __ballot(myl == l);
            if (myl == l) rank = run[l] + (uint32_t)__popcll(m & lane_below);
            run[l] += (uint32_t)__popcll(m);
        }
        if (myl) {
            const uint32_t cw = __brev(h.first[myl] + rank) >> (32 - myl);
            cwtab[s] = (uint16_t)cw;
            if (myl <= root) {
                const uint32_t e = seg_entry<KIND>(s, myl);
                for (uint32_t k = cw; k < (1u << root); k += 1u << myl) main[k] = e;
            } else {
                atomicMax(&main[cw & ((1u << root) - 1u)], myl);
            }
        }
    }
    wave_sync();
    if (cnt[11] + cnt[12] + cnt[13] + cnt[14] + cnt[15] + (root < 10 ? cnt[9] + cnt[10] : 0u) + (root < 8 ? cnt[8] : 0u) == 0)
        return true;  // (nothing beyond the root: the usual precode / offset code)
    if (!sub) return false;
    bool fits = true;
    for (uint32_t i = lane; i < (1u << root); i += 64) {
        const uint32_t v = main[i];
        if (v > root && v <= 15u) {  // (an entry of a short codeword has a length <= root in these bits, or more bits set)
            const uint32_t sb = v - root;
            const uint32_t base = atomicAdd(&h.alloc, 1u << sb);
            if (base + (1u << sb) > sub_cap) fits = false;
            main[i] = (base << 16) | (sb << 12);
        }
    }
    if (__ballot(!fits)) return false;
    wave_sync();
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t s = 64 * r + lane;
        const uint32_t myl = s < nsyms ? lens[s] : 0;
        if (myl > root) {
            const uint32_t cw = cwtab[s];
            const uint32_t p = main[cw & ((1u << root) - 1u)];
            const uint32_t base = p >> 16, sb = (p >> 12) & 15u;
            const uint32_t e = seg_entry<KIND>(s, myl);
            for (uint32_t k = cw >> root; k < (1u << sb); k += 1u << (myl - root)) sub[base + k] = e;
        }
    }
    wave_sync();
    return true;
}

// ---- the lane's compressed bits: relative positions (bit 0 = bit 0 of the lane's first dword `w0`), 32 bits at a time
struct SegWin {
    const uint32_t *pay32;
    uint32_t last_w;  // last readable dword of the member
    uint32_t *rl;     // the lane's ring dwords (kSegRingStride of them)
    uint32_t w0;      // payload dword at relative bit 0
};
__device__ __forceinline__ dword4 seg_load16(const SegWin &s, uint32_t w) {
    dword4 v;
    if (w + 3 <= s.last_w) {
        v = *(const dword4 *)(s.pay32 + w);
    } else {  // (the last dwords of a member: clamped, bits past the payload never count)
        v.x = s.pay32[w < s.last_w ? w : s.last_w];
        v.y = s.pay32[w + 1 < s.last_w ? w + 1 : s.last_w];
        v.z = s.pay32[w + 2 < s.last_w ? w + 2 : s.last_w];
        v.w = s.pay32[w + 3 < s.last_w ? w + 3 : s.last_w];
    }
    return v;
}
// window k (relative dwords 4k .. 4k+3) into its half of the ring
__device__ __forceinline__ void seg_win_put(const SegWin &s, uint32_t k, const dword4 &a) {
    uint32_t *d = s.rl + kSegWinDw * (k & 1u);
    d[0] = a.x;
    d[1] = a.y;
    d[2] = a.z;
    d[3] = a.w;
    if ((k & 1u) == 0) {
        s.rl[kSegRingDw] = a.x;
        s.rl[kSegRingDw + 1u] = a.y;
    }
}
__device__ __forceinline__ void seg_win_load(const SegWin &s, uint32_t k, dword4 &a) { a = seg_load16(s, s.w0 + kSegWinDw * k); }
// One symbol at relative position rp, straight-line.  y.e = its litlen entry (kSegLit: y.lv the byte; kSegLen: y.lv the
// length, y.dist the distance, y.obad != 0 an invalid offset codeword; kSegEob; none of the three: an invalid codeword),
// y.used = its bits, y.outlen = the bytes it produces.
//   PAIR: a literal takes the literal BEHIND it along in the same step (y.pair, y.val2 the second byte, y.u1 where it
//   starts) when that one's codeword sits in the root table and starts in front of `r_end`, the segment's end.  The bits
//   behind the first symbol are fetched for the offset code anyway, so the second literal costs one more table read; and a
//   wave's step count is that of its lane with the most symbols, which is the one in a run of literals (a match is
//   20-odd bits a step, a literal 6-9).  The rule depends on the position only, not on where a path started, so a
//   segment's exit -- the first symbol that starts at or behind r_end -- is what single steps give.
struct SegSym {
    uint32_t e, lv, dist, obad, used, outlen, u1, pair, val2;
};
template <bool PAIR>
__device__ __forceinline__ void seg_sym(const InfSegLds &h, const SegWin &s, uint32_t rp, uint32_t r_end, SegSym &y) {
    // 96 bits from rp's dword on: the litlen symbol (<= 20 bits) and what follows it (<= 28) start inside the first 51
    const uint32_t *d = s.rl + ((rp >> 5) & (kSegRingDw - 1u));
    const uint32_t d0 = d[0], d1 = d[1], d2 = d[2];
    const uint32_t b = __builtin_amdgcn_alignbit(d1, d0, rp & 31u);
    uint32_t e = h.lfast[b & ((1u << kSegLRoot) - 1u)];
    {
        const uint32_t sb = (e >> 12) & 15u;
        const uint32_t e2 = h.lsub[((e >> 16) & 511u) + ((b >> kSegLRoot) & ((1u << sb) - 1u))];  // (a plain entry: sb = 0, index <= 258)
        e = (e & 15u) ? e : e2;
    }
    const uint32_t cl = e & 15u, xb = (e >> 8) & 7u, u1 = e >> 25;  // (u1 = cl + xb, <= 20)
    y.lv = ((e >> 16) & 511u) + ((b >> cl) & ((1u << xb) - 1u));
    const uint32_t sh = (rp & 31u) + u1;  // (<= 51; alignbit takes the low five bits)
    const uint32_t b2 = __builtin_amdgcn_alignbit(sh >= 32u ? d2 : d1, sh >= 32u ? d1 : d0, sh);
    uint32_t oe = h.ofast[b2 & ((1u << kSegORoot) - 1u)];
    {
        const uint32_t sb = (oe >> 12) & 15u;
        const uint32_t idx = ((oe & 15u) ? 0u : (oe >> 16)) + ((b2 >> kSegORoot) & ((1u << sb) - 1u));
        const uint32_t oe2 = h.osub[idx];
        oe = (oe & 15u) ? oe : oe2;
    }
    const uint32_t dcl = oe & 15u, dxb = (oe >> 8) & 15u;
    y.dist = (oe >> 16) + ((b2 >> dcl) & ((1u << dxb) - 1u));
    const uint32_t is_len = (e >> 4) & 1u;
    y.e = e;
    y.obad = oe & (is_len << 4);
    y.u1 = u1;
    y.used = u1 + (is_len ? dcl + dxb : 0u);
    y.outlen = is_len ? y.lv : (e >> 6) & 1u;
    y.pair = 0;
    y.val2 = 0;
    if (PAIR) {
        const uint32_t en = h.lfast[b2 & ((1u << kSegLRoot) - 1u)];  // the symbol behind this one, if this one is a literal
        // (bit 6 in both: this one is a literal, and the entry behind it is a literal's own -- a pointer has no type bits)
        const uint32_t pair = ((e & en) >> 6) & (rp + u1 < r_end ? 1u : 0u);
        y.used += pair ? (en & 15u) : 0u;
        y.outlen += pair;
        y.pair = pair;
        y.val2 = (en >> 16) & 511u;
    }
}

__device__ __forceinline__ void seg_redo(DBlock *blk, uint32_t *redo, uint32_t b, uint32_t lane) {
    if (lane == 0) {
        blk->status = kInfRedo;
        const uint32_t slot = atomicAdd(&redo[0], 1u);
        redo[1 + slot] = b;
    }
}

#ifndef GZPX_SEG_WAVES
#define GZPX_SEG_WAVES 4
#endif
#ifndef GZPX_SEG_SMALLW
#define GZPX_SEG_SMALLW 1
#endif
constexpr int kSegSmallW = GZPX_SEG_SMALLW;    // waves per member for BGZF-sized members
constexpr int kSegBigW = 8;                    // waves per member in the launch form for large members (Mgzip)
constexpr uint32_t kSegBigBytes = 131072u;     // compressed bytes per member (average of the slab) from which it is used

// A DEFLATE block header, by one wave: block type, code lengths, the three tables.  What it found goes to res[]:
// [0] bad, [1] final block, [2] type, [3] the bit behind the header (stored: behind LEN / NLEN), [4] stored length.
__device__ __attribute__((noinline)) void seg_header(InfSegLds &h, uint16_t *cwtab, uint32_t *res, const uint32_t *__restrict__ pay32,
                                                     uint32_t last_w, uint32_t bit_end, uint32_t bp, uint32_t lane) {
    uint32_t *hdr_w = h.lfast + 128;  // 96 staged dwords of a block header (lfast[0..80) is the code-length scratch)
    uint32_t hbase = 0;
    auto hbits = [&](uint32_t p) -> uint32_t {
        const uint32_t w = (p >> 5) - hbase;
        return __builtin_amdgcn_alignbit(hdr_w[w + 1], hdr_w[w], p & 31u);
    };
    bool bad = false, final_block = false;
    uint32_t btype = 3, slen = 0;
    do {
        if (bp + 3 > bit_end) {
            bad = true;
            break;
        }
        wave_sync();
        hbase = bp >> 5;
        {
            const uint32_t w0 = hbase + lane, w1 = hbase + 64 + lane;
            hdr_w[lane] = pay32[w0 < last_w ? w0 : last_w];
            if (lane < 32) hdr_w[64 + lane] = pay32[w1 < last_w ? w1 : last_w];
        }
        wave_sync();
        const uint32_t hb = uniform(hbits(bp));
        final_block = (hb & 1u) != 0;
        btype = (hb >> 1) & 3u;
        bp += 3;
        if (btype == 0) {  // stored: LEN, NLEN
            bp = (bp + 7u) & ~7u;
            if (bp + 32 > bit_end) {
                bad = true;
                break;
            }
            const uint32_t x = uniform(hbits(bp));
            const uint32_t len = x & 0xFFFFu, nlen = x >> 16;
            bp += 32;
            if ((len ^ 0xFFFFu) != nlen) bad = true;
            slen = len;
            break;
        }
        if (btype == 3) {
            bad = true;
            break;
        }
        // ---- code lengths
        if (btype == 1) {
            for (uint32_t i = lane; i < 320; i += 64)
                h.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5);
            wave_sync();
        } else {
            const uint32_t nlit = ((hb >> 3) & 31u) + 257, ndist = ((hb >> 8) & 31u) + 1;
            const uint32_t nclen = ((hb >> 13) & 15u) + 4;
            bp += 14;
            if (nlit > 286 + 2 || ndist > 32) {
                bad = true;
                break;
            }
            {
                const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const uint32_t v = hbits(bp + 3u * (lane < 19 ? lane : 0)) & 7u;
                if (lane < 19) h.lens[order[lane]] = (uint8_t)(lane < nclen ? v : 0u);
            }
            bp += 3u * nclen;
            wave_sync();
            if (!seg_build<kSegPrecode>(h, cwtab, h.lens, 19, kSegPRoot, h.ofast, nullptr, 0, lane)) {
                bad = true;
                break;
            }
            uint32_t i = 0;
            const uint32_t total = nlit + ndist;
            uint32_t prev = 0;
            uint8_t *tmp = (uint8_t *)h.lfast;  // 320 bytes of scratch in front of the staged header
            // The code lengths are a serial chain of precode symbols, but WHAT a symbol is depends only on where it
            // starts: every lane decodes the one that would start at bit bp + lane (two LDS round trips for all 64),
            // and the chain through them is scalar work on lane values -- a dozen symbols per round instead of one
            // symbol per pair of round trips (a fifth of a BGZF member's decode time went into these headers).
            while (i < total && !bad) {
                if (bp + 64u - 32u * hbase > 96u * 32u - 64u) {  // (a header never needs this much: garbage)
                    bad = true;
                    break;
                }
                const uint32_t bb = hbits(bp + lane);
                const uint32_t e = h.ofast[bb & 127u];
                const uint32_t cl = e & 15u, sym = e >> 16;
                uint32_t rep = 1, val = sym, used = cl;
                if (sym == 16) {
                    rep = 3 + ((bb >> cl) & 3u);
                    used += 2;
                } else if (sym == 17) {
                    rep = 3 + ((bb >> cl) & 7u);
                    used += 3;
                    val = 0;
                } else if (sym == 18) {
                    rep = 11 + ((bb >> cl) & 127u);
                    used += 7;
                    val = 0;
                }
                // used (<= 14) | rep (<= 138) << 8 | val (<= 16: 16 = the length before) << 16; 0 = an unused codeword
                const uint32_t pack = cl ? used | (rep << 8) | (val << 16) : 0u;
                uint32_t cur = 0;
                while (cur < 64u && i < total) {
                    const uint32_t pk = rdlane(pack, cur);
                    if (pk == 0) {
                        bad = true;
                        break;
                    }
                    const uint32_t r = (pk >> 8) & 255u;
                    uint32_t v = pk >> 16;
                    if (v == 16) {
                        if (i == 0) {
                            bad = true;
                            break;
                        }
                        v = prev;
                    }
                    if (i + r > total) {
                        bad = true;
                        break;
                    }
                    if (lane < r) tmp[i + lane] = (uint8_t)v;
                    if (lane + 64 < r) tmp[i + lane + 64] = (uint8_t)v;
                    if (lane + 128 < r) tmp[i + lane + 128] = (uint8_t)v;
                    prev = v;
                    i += r;
                    cur += pk & 255u;
                }
                bp += cur;
            }
            if (bad) break;
            wave_sync();
            uint8_t mine[5];
            for (uint32_t k = 0; k < 5; k++) {
                const uint32_t s = lane + 64 * k;
                uint32_t v = 0;
                if (s < 288) v = s < nlit ? tmp[s] : 0;
                else if (s < 320) v = (s - 288) < ndist ? tmp[nlit + (s - 288)] : 0;
                mine[k] = (uint8_t)v;
            }
            wave_sync();
            for (uint32_t k = 0; k < 5; k++) h.lens[lane + 64 * k] = mine[k];
            wave_sync();
            if (h.lens[256] == 0) {
                bad = true;
                break;
            }
        }
        if (!seg_build<kSegLitlen>(h, cwtab, h.lens, 288, kSegLRoot, h.lfast, h.lsub, kSegLSub, lane) ||
            !seg_build<kSegOffset>(h, cwtab, h.lens + 288, 32, kSegORoot, h.ofast, h.osub, kSegOSub, lane))
            bad = true;
    } while (false);
    wave_sync();
    if (lane == 0) {
        res[0] = bad ? 1u : 0u;
        res[1] = final_block ? 1u : 0u;
        res[2] = btype;
        res[3] = bp;
        res[4] = slen;
    }
    wave_sync();
}

// DBlock.cyc of a debug launch (wave 0's clocks): [0] whole member, [1] headers + tables, [2] pass 1, [3] pass 2,
// [4] pass 3, counts [5] spans, [6] pass-2 iterations, [7] symbol steps of passes 1 and 3 (wave iterations)
//
// W waves work on one member: wave 0 reads the block headers and builds the tables, a span is 64 W segments, the
// entries chain from lane to lane and from wave to wave.  W = 1 for BGZF-sized members (one wave each: thousands of
// members fill the chip), kSegBigW for Mgzip members (a 1 MiB member is a million symbols: one wave would take its
// lanes through 16 thousand steps each, and a slab holds too few members for the chip).
template <bool DBG, int W>
__device__ __attribute__((noinline)) void seg_member(const uint32_t b, const uint32_t tid, uint32_t hdr_len, const uint8_t *__restrict__ in_all,
                                                     DBlock *__restrict__ blk_all, const uint64_t *__restrict__ out_off, uint8_t *out_all,
                                                     uint64_t out_cap, LzMatch *__restrict__ mlist_all, uint32_t *__restrict__ tfirst_all,
                                                     uint32_t *__restrict__ redo, uint32_t *hint_p) {
    __shared__ InfSegLdsW<W> hh;  // (its own LDS object, so that every access stays an LDS instruction)
    InfSegLds &h = hh.t;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    constexpr uint32_t NT = 64u * W;
    DBlock *blk = blk_all + b;
    const uint32_t isize = blk->isize;
    if (tid == 0) blk->nmatch = 0;
    if (isize == 0) return;  // src/par/decompress.rs:163-171: nothing to decode
    const uint64_t ooff = out_off[b];
    if (ooff + isize > out_cap) {
        if (tid == 0) blk->status = kInfInsufficientSpace;
        return;
    }
    uint8_t *out = out_all + ooff;
    const uint8_t *pay = in_all + blk->in_off + hdr_len;
    const uint32_t pay_len = blk->size - hdr_len - 8;
    LzMatch *ml = mlist_all + (ooff / 3u + b);
    uint32_t *tf = tfirst_all + ((ooff >> kLzTileShift) + 2ull * b);
    const bool multi = isize > kLzTile;  // more than one k_lzcopy tile: the first record of every tile is noted
    if (multi && tid == 0) tf[0] = 0;
    const long long t_begin = DBG ? clock64() : 0;
    uint32_t dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    const uint32_t pmis = (uint32_t)((uintptr_t)pay & 3u);
    const uint32_t *pay32 = (const uint32_t *)(pay - pmis);
    const uint32_t pay_words = (pmis + pay_len + 8 + 3) >> 2;  // the 8 footer bytes are readable too
    const uint32_t last_w = pay_words - 1;
    const uint32_t bit0 = 8u * pmis, bit_end = bit0 + 8u * pay_len;
    SegWin win;
    win.pay32 = pay32;
    win.last_w = last_w;
    win.rl = hh.ring + tid * kSegRingStride;
    win.w0 = 0;

    uint32_t bp = bit0, o = 0, mtot = 0, prev_blk_bits = 0;  // (prev_blk_bits: the symbols of the block before, in bits)
    bool final_block = false, bad = false;
#define SEG_BAD() do { bad = true; } while (0)  // (the member goes to k_inflate)
    while (!final_block && !bad) {
        bp = uniform(bp);
        o = uniform(o);
        mtot = uniform(mtot);
        const long long t_hdr = DBG ? clock64() : 0;
        if (W > 1) __syncthreads();  // (the other waves are done with the tables and with hres)
        if (wave == 0) seg_header(h, (uint16_t *)hh.ring, hh.hres, pay32, last_w, bit_end, bp, lane);
        if (W > 1) __syncthreads();
        bad = uniform(hh.hres[0]) != 0;
        final_block = uniform(hh.hres[1]) != 0;
        const uint32_t btype = uniform(hh.hres[2]);
        bp = uniform(hh.hres[3]);
        if (bad) break;
        if (btype == 0) {
            // stored: raw bytes straight from the payload to their place (final bytes, like literals)
            const uint32_t len = uniform(hh.hres[4]);
            const uint32_t src = (bp - bit0) >> 3;
            if (src + len > pay_len || o + len > isize) {
                SEG_BAD();
                break;
            }
            const uint8_t *sp = pay + src;
            uint8_t *dp = out + o;
            uint32_t head = (uint32_t)((4u - ((uintptr_t)dp & 3u)) & 3u);
            if (head > len) head = len;
            if (tid < head) dp[tid] = sp[tid];
            const uint32_t nw = (len - head) >> 2;
            const uint32_t smis = (uint32_t)((uintptr_t)(sp + head) & 3u);
            const uint32_t *s32 = (const uint32_t *)(sp + head - smis);
            for (uint32_t k = tid; k < nw; k += NT) {
                const uint32_t lo = s32[k], hi = smis ? s32[k + 1] : 0u;  // (hi: inside the member, the footer follows)
                *(uint32_t *)(dp + head + 4 * k) = __builtin_amdgcn_alignbyte(hi, lo, smis);
            }
            const uint32_t donew = head + 4 * nw;
            if (tid < 3 && donew + tid < len) dp[donew + tid] = sp[donew + tid];
            if (multi && tid == 0)
                for (uint32_t k = (o >> kLzTileShift) + 1; k <= ((o + len) >> kLzTileShift); k++) tf[k] = mtot;
            o += len;
            bp += 8u * len;
            continue;
        }
        if (DBG) dbg[1] += (uint32_t)(clock64() - t_hdr);

        // ---- the block's symbols, one span of 64 W segments after the other until its end-of-block code
        // How long a span is: a DEFLATE block does not say where it ends, and every lane of a span behind the block's end
        // is work thrown away (passes 1 and 2 cannot know).  Spans cut from what is left of the MEMBER did that to every
        // block but a member's last: eighteen blocks per 1 MiB Mgzip member of text at level 1 -- eight ninths of the
        // work.  So a block's spans are HALF as long as the block before it (two spans for a block like the last one,
        // at most a quarter of the second wasted), a block that outlives its span goes on with spans of the same length,
        // and a member's first block starts with the whole member (one wave per member: BGZF members hold one to three
        // blocks, and a half measured 4 % slower, also with short spans for what outlives it) or a sixteenth of it (Mgzip).  Measured, 512 MiB
        // in 1 MiB members, decode ms: text level 1 8.05 -> 3.22, level 3 2.98 -> 1.92, FASTQ 10.9 -> 4.39, printable
        // noise 4.55 -> 3.80 (tools/gpu_r6_ab_mgzip_inflate.py; a whole / 5/4 / 3/4 / a quarter of the block before and
        // a half ... a thirty-second of the member all within 8 % of this).
        bool eob = false;
        const uint32_t blk_bp0 = bp;
        uint32_t span_bits = prev_blk_bits ? prev_blk_bits / 2u : (bit_end - bp) / (W > 1 ? 16u : 1u);
        // (a FINAL block ends where the member ends: what is left IS its length -- without this the rule above cost BGZF
        // members of DNA / FASTQ, whose last block is not half of the one before it, 8 %)
        if (final_block) span_bits = bit_end - bp;
        // (a member's first block that is not its last, one wave per member: where the first block of the members before
        // this one ended, per mille of the member -- members of one stream are alike: libdeflate's level 1 ends the first
        // of two blocks at 0.98 of a member of text, 0.58 of DNA, 0.65 of FASTQ, member after member.  The hint is a
        // word of the context that any member writes and any reads; a wrong one costs what no hint costs.)
        const bool hinted = W == 1 && !final_block && !prev_blk_bits && hint_p != nullptr;
        uint32_t hint = 0;
        if (hinted) {
            hint = uniform(__hip_atomic_load(hint_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (hint >= 1u && hint <= 1000u) span_bits = (uint32_t)((uint64_t)(bit_end - bp) * (hint + 40u) / 1000u);
        }
        while (!eob && !bad) {
            bp = uniform(bp);
            o = uniform(o);
            mtot = uniform(mtot);
            if (bp >= bit_end) {  // no end-of-block code before the payload's end
                SEG_BAD();
                break;
            }
            if (DBG) dbg[5]++;
            const uint32_t rem = bit_end - bp;
            const uint32_t want = span_bits < rem ? span_bits : rem;
            uint32_t S = (((want + NT - 1u) / NT) + 31u) & ~31u;
            S = S < kSegMinBits ? kSegMinBits : S > kSegMaxBits ? kSegMaxBits : S;
            const uint32_t my_start = bp + tid * S;
            win.w0 = my_start >> 5;
            const uint32_t rel0 = 32u * win.w0;           // absolute position of relative bit 0
            const uint32_t r_start = my_start - rel0;      // 0..31
            const uint32_t r_end = r_start + S;            // the segment's end, relative
            const uint32_t nwin = (S + 31u + 32u * kSegWinDw - 1u) / (32u * kSegWinDw);  // windows that cover any lane's segment
            const bool has_data = my_start < bit_end;
            // ---- pass 1: from the speculative start (lane 0: the true one) to the segment's end
            const long long t_p1 = DBG ? clock64() : 0;
            uint32_t rp = r_start, n1 = 0, m1 = 0;
            const uint32_t fl1 = has_data ? 0u : 2u;  // fl: 0 runs / ran through, 1 end of block, 2 no data
            // A speculative path does not stop at an end-of-block code either: in front of the point where it meets the true
            // path the code is as likely garbage as not (15 bits that happen to match, once or twice per span of ASCII
            // noise), and a path that stopped there left pass 2 its whole segment to replay alone, with the wave -- the
            // workgroup, for W > 1 -- waiting.  It notes its FIRST one (exit | matches in front << 16, bytes in front) and
            // the exit of its LAST one and walks on; which one counts, if any, is decided where the true path joins it
            // (pass 2).  (Selects, not a branch: the rare branch cost the loop thirty register copies per step.)
            uint32_t e1p = 0, e1n = 0, eLx = 0;
            {
                dword4 a0, a1;
                seg_win_load(win, 0, a0);
                seg_win_load(win, 1, a1);
                wave_sync();
                seg_win_put(win, 0, a0);
                seg_win_put(win, 1, a1);
                wave_sync();
                for (uint32_t k = 0; k < nwin; k++) {
                    dword4 na;
                    if (k + 2 < nwin + 1) seg_win_load(win, k + 2, na);
                    const uint32_t wend = 32u * kSegWinDw * (k + 1);
                    const uint32_t lim = r_end < wend ? r_end : wend;
                    while (__ballot(fl1 == 0 && rp < lim)) {
                        if (fl1 == 0 && rp < lim) {  // (one exec mask around the step instead of a select per result)
                            SegSym y;
                            seg_sym<true>(h, win, rp, r_end, y);
                            rp += y.used;
                            const bool is_eob = (y.e & kSegEob) != 0, first = is_eob && e1p == 0;
                            e1p = first ? rp | (m1 << 16) : e1p;  // (rp < 2^14, m1 < 2^16)
                            e1n = first ? n1 : e1n;
                            eLx = is_eob ? rp : eLx;
                            n1 += y.outlen;
                            m1 += (y.e >> 4) & 1u;
                        }
                        if (DBG) dbg[7]++;
                    }
                    if (k + 2 < nwin + 1) {
                        wave_sync();
                        seg_win_put(win, k + 2, na);
                        wave_sync();
                    }
                }
            }
            const uint32_t exit1 = rel0 + rp;
            if (DBG) dbg[2] += (uint32_t)(clock64() - t_p1);
            // ---- pass 2: the true entries.  Lane i enters where lane i - 1 leaves; it replays from there beside
            // its speculative path until the two meet (then the rest of pass 1 holds) or its segment ends.
            const long long t_p2 = DBG ? clock64() : 0;
            // (until pass 2 says otherwise a path's first end-of-block code counts: it does on the true path, lane 0's)
            const uint32_t e1x = e1p & 0xFFFFu, e1m = e1p >> 16;
            uint32_t entry = my_start, ex = e1p ? rel0 + e1x : exit1, nn = e1p ? e1n : n1, mm = e1p ? e1m : m1, fl = e1p ? 1u : fl1;
            uint32_t iters = 0;
            for (;;) {
                // (every lane with data follows its predecessor's exit, also behind a lane that stopped: a speculative path
                // that ran into an end-of-block code or an unused codeword in front of its meeting point is put right like any
                // other, and does not hold up the lanes behind it for an iteration of its own)
                if (W > 1) {  // the wave's last exit, for the wave behind it
                    __syncthreads();
                    if (lane == 63) hh.x_ex[wave] = ex;
                    __syncthreads();
                }
                uint32_t new_entry = (uint32_t)__shfl_up((int)ex, 1);
                if (lane == 0) new_entry = wave == 0 ? bp : hh.x_ex[wave > 0 ? wave - 1 : 0];
                const bool need = has_data && new_entry != entry && new_entry >= my_start;
                if (W > 1 ? __syncthreads_or(need ? 1 : 0) == 0 : __ballot(need) == 0) break;
                if (++iters > kSegMaxFix) {
                    SEG_BAD();
                    break;
                }
                // both replays run in the lane's relative coordinates, window by window like pass 1
                uint32_t a = new_entry - rel0, bq = r_start, na = 0, ma = 0, nb2 = 0, mb = 0, fa = 0, fb = has_data ? 0u : 2u;
                bool synced = false, done = !need;
                dword4 a0, a1;
                seg_win_load(win, 0, a0);
                seg_win_load(win, 1, a1);
                wave_sync();
                seg_win_put(win, 0, a0);
                seg_win_put(win, 1, a1);
                wave_sync();
                for (uint32_t k = 0; k < nwin; k++) {
                    dword4 wa;
                    if (k + 2 < nwin + 1) seg_win_load(win, k + 2, wa);
                    const uint32_t wend = 32u * kSegWinDw * (k + 1);
                    for (;;) {
                        // whose turn: the replay from the true entry (a) unless the speculative one (b) is behind it
                        const bool a_fin = fa != 0 || a >= r_end;
                        const bool b_fin = fb != 0 || bq >= r_end;
                        if (!done && a_fin) done = true;
                        if (!done && !b_fin && a == bq) {
                            // the paths have met, and from here on the speculative one IS the true one: its first
                            // end-of-block code ends the block if it lies behind this point; if its last one lies in
                            // front, there is none and pass 1's exit holds.  (In between -- a garbage code in front and
                            // another one behind, of which pass 1 kept no counts -- the replay goes on alone.)
                            if ((e1p != 0 && e1x > a) || eLx <= a) {
                                synced = true;
                                done = true;
                            } else {
                                fb = 2u;
                            }
                        }
                        const bool step_a = b_fin || a < bq;
                        const uint32_t p = step_a ? a : bq, other = step_a ? bq : a;
                        const bool go = !done && p < wend;
                        if (__ballot(go) == 0) break;
                        if (go) {
                            SegSym y;
                            seg_sym<true>(h, win, p, r_end, y);
                            // a pair whose second literal starts where the other path stands: the paths have met there,
                            // this one takes the first literal only and lands on it (they could leap over each other for
                            // ever otherwise; a meeting point in the middle of the OTHER path's last step shows one step
                            // later, as the end of that path's next step or the middle of this one's)
                            const bool half = y.pair && p + y.u1 == other;
                            const uint32_t used = half ? y.u1 : y.used, outlen = half ? 1u : y.outlen;
                            const uint32_t isl = (y.e >> 4) & 1u, f = (y.e >> 5) & 1u;
                            if (step_a) {
                                a += used;
                                na += outlen;
                                ma += isl;
                                fa = f;  // (this one stops at an end-of-block code: from a true entry it is the block's)
                            } else {
                                bq += used;
                                nb2 += outlen;
                                mb += isl;
                            }
                        }
                    }
                    if (__ballot(!done) == 0) break;
                    if (k + 2 < nwin + 1) {
                        wave_sync();
                        seg_win_put(win, k + 2, wa);
                        wave_sync();
                    }
                }
                if (need) {
                    if (synced) {
                        if (e1p != 0 && e1x > a) {
                            ex = rel0 + e1x;
                            nn = e1n - nb2 + na;
                            mm = e1m - mb + ma;
                            fl = 1u;
                        } else {
                            ex = exit1;
                            nn = n1 - nb2 + na;
                            mm = m1 - mb + ma;
                            fl = fl1;
                        }
                    } else {
                        ex = rel0 + a;
                        nn = na;
                        mm = ma;
                        fl = fa;
                    }
                    entry = new_entry;
                }
            }
            if (DBG) {
                dbg[3] += (uint32_t)(clock64() - t_p2);
                dbg[6] += iters;
            }
            if (bad) break;
            // ---- what the span holds: live lanes up to the first one that stopped
            const uint64_t stopped = __ballot(fl != 0);
            const uint32_t wstop = stopped ? (uint32_t)__ffsll((long long)stopped) - 1u : 64u;  // this wave's first stop
            uint32_t stop_wave = wave, first_stop = wstop;  // the span's: (wave, lane)
            uint32_t stop_fl = wstop < 64u ? rdlane(fl, wstop) : 0u, stop_ex = rdlane(ex, wstop < 64u ? wstop : 63u);
            if (W > 1) {
                __syncthreads();
                if (lane == 0) {
                    hh.x_first[wave] = wstop;
                    hh.x_fl[wave] = stop_fl;
                    hh.x_ex[wave] = stop_ex;  // (the stop lane's exit, or the wave's last)
                }
                __syncthreads();
                stop_wave = (uint32_t)W;
                first_stop = 64u;
                for (uint32_t w2 = 0; w2 < (uint32_t)W; w2++)
                    if (stop_wave == (uint32_t)W && hh.x_first[w2] < 64u) {
                        stop_wave = w2;
                        first_stop = hh.x_first[w2];
                    }
                const uint32_t from = stop_wave < (uint32_t)W ? stop_wave : (uint32_t)W - 1u;
                stop_fl = hh.x_fl[from];
                stop_ex = hh.x_ex[from];
            } else if (wstop == 64u) {
                stop_wave = 1u;  // (= W: no stop in the span)
            }
            const bool live = wave < stop_wave || (wave == stop_wave && lane <= first_stop);
            if (stop_wave < (uint32_t)W) {
                if (stop_fl != 1u) {  // an invalid symbol on the true path
                    SEG_BAD();
                    break;
                }
                eob = true;
            }
            const uint32_t my_n = live ? nn : 0u, my_m = live ? mm : 0u;
            uint32_t in_n = wave_incl_add(my_n), in_m = wave_incl_add(my_m);
            uint32_t tot_n = rdlane(in_n, 63), tot_m = rdlane(in_m, 63);
            if (W > 1) {  // the waves' totals -> every wave's base
                if (lane == 0) {
                    hh.x_n[wave] = tot_n;
                    hh.x_m[wave] = tot_m;
                }
                __syncthreads();
                uint32_t base_n = 0, base_m = 0;
                tot_n = tot_m = 0;
                for (uint32_t w2 = 0; w2 < (uint32_t)W; w2++) {
                    if (w2 < wave) {
                        base_n += hh.x_n[w2];
                        base_m += hh.x_m[w2];
                    }
                    tot_n += hh.x_n[w2];
                    tot_m += hh.x_m[w2];
                }
                in_n += base_n;
                in_m += base_m;
            }
            const uint32_t new_bp = stop_ex;
            if (o + tot_n > isize || new_bp > bit_end + (eob ? 0u : 64u)) {
                SEG_BAD();
                break;
            }
            // ---- pass 3: the segment again from its true entry, now writing
            const long long t_p3 = DBG ? clock64() : 0;
            bool bad_dist = false;
            {
                uint32_t pos = o + in_n - my_n, mi = mtot + in_m - my_m;
                uint32_t f3 = live ? 0u : 2u;
                rp = entry - rel0;
                dword4 a0, a1;
                seg_win_load(win, 0, a0);
                seg_win_load(win, 1, a1);
                wave_sync();
                seg_win_put(win, 0, a0);
                seg_win_put(win, 1, a1);
                wave_sync();
                for (uint32_t k = 0; k < nwin; k++) {
                    dword4 na;
                    if (k + 2 < nwin + 1) seg_win_load(win, k + 2, na);
                    const uint32_t wend = 32u * kSegWinDw * (k + 1);
                    const uint32_t lim = r_end < wend ? r_end : wend;
                    while (__ballot(f3 == 0 && rp < lim)) {
                        if (f3 == 0 && rp < lim) {
                            SegSym y;
                            seg_sym<true>(h, win, rp, r_end, y);
                            if (y.e & kSegLit) {
                                out[pos] = (uint8_t)y.lv;
                                if (y.pair) out[pos + 1] = (uint8_t)y.val2;
                            } else if (y.e & kSegLen) {
                                if (y.dist > pos || y.obad) bad_dist = true;
                                LzMatch rec;
                                rec.pos = pos;
                                rec.len_dist = (y.lv << 16) | y.dist;
                                ml[mi] = rec;
                                mi++;
                            } else if (!(y.e & kSegEob)) {
                                bad_dist = true;  // an invalid codeword on the true path
                            }
                            const uint32_t np = pos + y.outlen;
                            if (multi && ((pos ^ np) >> kLzTileShift)) tf[np >> kLzTileShift] = mi;  // the next record is the tile's first
                            pos = np;
                            rp += y.used;
                            f3 = (y.e >> 5) & 1u;
                        }
                        if (DBG) dbg[7]++;
                    }
                    if (k + 2 < nwin + 1) {
                        wave_sync();
                        seg_win_put(win, k + 2, na);
                        wave_sync();
                    }
                }
            }
            if (DBG) dbg[4] += (uint32_t)(clock64() - t_p3);
            if (W > 1 ? __syncthreads_or(bad_dist ? 1 : 0) != 0 : __ballot(bad_dist) != 0) {
                SEG_BAD();
                break;
            }
            o += tot_n;
            mtot += tot_m;
            bp = new_bp;
            if (hinted) span_bits = bit_end - bp;  // (the guess was short: the rest of the member, as without a hint)
        }
        prev_blk_bits = bp - blk_bp0;
        if (hinted && !bad && tid == 0 && bit_end > blk_bp0) {
            const uint32_t pm = (uint32_t)((uint64_t)prev_blk_bits * 1000u / (bit_end - blk_bp0));
            __hip_atomic_store(hint_p, pm < 1u ? 1u : pm > 1000u ? 1000u : pm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!bad && (o != isize || bp > bit_end)) SEG_BAD();
    if (bad) {
        seg_redo(blk, redo, b, tid);
    } else if (tid == 0) {
        blk->status = kInfOk;
        blk->produced = o;
        blk->nmatch = mtot;
        if (multi) tf[(isize >> kLzTileShift) + 1] = mtot;
    }
    if (DBG && tid == 0) {
        dbg[0] = (uint32_t)(clock64() - t_begin);
        for (uint32_t k = 0; k < 8; k++) blk->cyc[k] = dbg[k];
    }
}

#undef SEG_BAD

// The members are claimed from a ticket counter (redo[1 + nb]): the launch holds as many workgroups as the chip keeps
// resident, and one that finishes a member takes the next -- 8,835 members on 4,096 wave slots are 2.16 member times, not three.
template <bool DBG, int W>
__global__ __launch_bounds__(64 * W, GZPX_SEG_WAVES) void k_inflate_seg(uint32_t hdr_len, const uint8_t *__restrict__ in_all,
                                                                       DBlock *__restrict__ blk_all,
                                                                       const uint64_t *__restrict__ out_off, uint8_t *out_all,
                                                                       uint64_t out_cap, LzMatch *__restrict__ mlist_all,
                                                                       uint32_t *__restrict__ tfirst_all,
                                                                       uint32_t *__restrict__ redo, uint32_t nb, uint32_t *hint_p) {
    __shared__ uint32_t s_ticket;
    const uint32_t tid = threadIdx.x;
    for (;;) {
        if (W > 1) __syncthreads();  // (everybody has read the last ticket)
        if (tid == 0) s_ticket = atomicAdd(&redo[1 + nb], 1u);
        if (W > 1) __syncthreads();
        else wave_sync();
        const uint32_t b = uniform(s_ticket);
        if (b >= nb) break;
        wave_sync();
        seg_member<DBG, W>(b, tid, hdr_len, in_all, blk_all, out_off, out_all, out_cap, mlist_all, tfirst_all, redo, hint_p);
    }
}

// ------------------------------------------------------------------------------------------
// k_lzcopy
// ------------------------------------------------------------------------------------------
constexpr uint32_t kLcThreads = 256;
constexpr uint32_t kLcPad = 16u;           // bytes in front of the tile in LDS (a copy may read the dword before its source)
constexpr uint32_t kLcBytes = kLzTile + 16u + kLcPad + 32u;  // tile + address phase + front pad + read-ahead slack
constexpr uint32_t kLcK = 4u;              // records per lane and chunk: a chunk is 1,024 consecutive matches
constexpr uint32_t kLcChunk = kLcThreads * kLcK;
constexpr uint32_t kLcShort = 16u;         // bytes a lane copies itself; longer or self-overlapping matches: the wave together
constexpr uint32_t kLcMaxSpins = 1u << 22;
constexpr uint32_t kLcReps = 4u;           // polls of the short matches per iteration of the polling loop (while they make progress)

// Tile coordinates: byte p of the member's output lives at LDS byte X = p - ts + phase + kLcPad (phase = the low four
// address bits of the member's first output byte, so that X and the byte's address agree modulo 16); bit X of the
// bitmap says "final".
struct LcLds {
    uint32_t tile[kLcBytes / 4];
    uint32_t bm[kLcBytes / 32 + 2];
    uint32_t chunk_lo;
    uint32_t gave_up;
    // the member's CRC-32, taken from the finished tiles while they are in LDS (k_dcrc32 then skips the member)
    uint32_t crc_tab[4][256];       // slice-by-4 tables
    uint32_t crc_pw[kLcThreads];    // x^(8 * 64 * (255 - t)): what thread t's 64-byte piece of a tile is shifted by
    uint32_t crc_part[kLcThreads / 64];
    uint32_t crc_total;
};
constexpr uint32_t kLcCrcDone = 0x80000000u;  // DBlock.nmatch, top bit: crc_found[] holds the member's CRC already

// MEM = (MEM & ~mask) | data on an LDS dword, atomically: lanes that write different bytes of one dword in the same
// instruction (neighbouring matches) do not lose each other's bytes
__device__ __forceinline__ void lds_mskor(uint32_t *p, uint32_t mask, uint32_t data) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t addr = (uint32_t)(size_t)p;
    asm volatile("ds_mskor_b32 %0, %1, %2" : : "v"(addr), "v"(mask), "v"(data) : "memory");
#else
    *p = (*p & ~mask) | data;
#endif
}

// bits [a, e) of the bitmap: all ones?
__device__ __forceinline__ bool lc_all_set(const uint32_t *bm, uint32_t a, uint32_t e) {
    bool ok = true;
    while (a < e) {
        const uint32_t w = a >> 5, lo = a & 31u;
        const uint32_t n = (e - a) < (32u - lo) ? (e - a) : (32u - lo);
        const uint32_t mask = (n == 32u ? 0xFFFFFFFFu : ((1u << n) - 1u)) << lo;
        if ((bm[w] & mask) != mask) ok = false;
        a += n;
    }
    return ok;
}
template <bool SET>
__device__ __forceinline__ void lc_mark(uint32_t *bm, uint32_t a, uint32_t e) {
    while (a < e) {
        const uint32_t w = a >> 5, lo = a & 31u;
        const uint32_t n = (e - a) < (32u - lo) ? (e - a) : (32u - lo);
        const uint32_t mask = (n == 32u ? 0xFFFFFFFFu : ((1u << n) - 1u)) << lo;
        if (SET) atomicOr(&bm[w], mask);
        else atomicAnd(&bm[w], ~mask);
        a += n;
    }
}

// four bits -> four bytes of ones
__device__ __forceinline__ uint32_t lc_bytes(uint32_t nib) { return (((nib & 15u) * 0x00204081u) & 0x01010101u) * 0xFFu; }

// n <= 16 bytes to tile bytes [D, D + n) from six dwords w0..w5 of which byte `r` of w0 is the byte for tile byte D & ~3:
// moved to the destination's alignment, up to five masked dword writes
__device__ __forceinline__ void lc_put16(uint32_t *tile, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5,
                                         uint32_t r, uint32_t D, uint32_t n) {
    const uint32_t v0 = __builtin_amdgcn_alignbyte(w1, w0, r), v1 = __builtin_amdgcn_alignbyte(w2, w1, r),
                   v2 = __builtin_amdgcn_alignbyte(w3, w2, r), v3 = __builtin_amdgcn_alignbyte(w4, w3, r),
                   v4 = __builtin_amdgcn_alignbyte(w5, w4, r);
    const uint32_t m20 = ((1u << n) - 1u) << (D & 3u);  // one bit per byte of the five destination dwords
    const uint32_t d = D >> 2;
    const uint32_t m0 = lc_bytes(m20), m1 = lc_bytes(m20 >> 4), m2 = lc_bytes(m20 >> 8);
    lds_mskor(&tile[d + 0], m0, v0 & m0);
    lds_mskor(&tile[d + 1], m1, v1 & m1);
    lds_mskor(&tile[d + 2], m2, v2 & m2);
    if (m20 >> 12) {
        const uint32_t m3 = lc_bytes(m20 >> 12), m4 = lc_bytes(m20 >> 16);
        lds_mskor(&tile[d + 3], m3, v3 & m3);
        lds_mskor(&tile[d + 4], m4, v4 & m4);
    }
}
// bits [a, a + n) := 1, n <= 32
__device__ __forceinline__ void lc_set32(uint32_t *bm, uint32_t a, uint32_t n) {
    const uint64_t m = ((n >= 32u ? 0xFFFFFFFFull : ((1ull << n) - 1ull))) << (a & 31u);
    atomicOr(&bm[a >> 5], (uint32_t)m);
    if ((uint32_t)(m >> 32)) atomicOr(&bm[(a >> 5) + 1], (uint32_t)(m >> 32));
}

// One match, clipped to the tile [ts, te): its part [k0, k1), where that part's bytes come from, and which of the
// three ways it goes: kLcTile (<= 16 bytes from inside the tile, disjoint from the destination: the polling lane
// copies them), kLcWindow (<= 16 bytes from in front of the tile: final, fetched from HBM when the chunk is set up),
// kLcSlow (long, self-overlapping or across the tile's start: the wave together, byte by byte).
enum { kLcNone = 0, kLcTile = 1, kLcWindow = 2, kLcSlow = 3 };
struct LcWork {
    uint32_t pos, dist;
    uint32_t k0, k1;
    uint32_t kind;
};
__device__ __forceinline__ void lc_prepare(LcWork &w, const LzMatch &m, bool valid, uint32_t ts, uint32_t te, uint32_t gmis) {
    const uint32_t len = m.len_dist >> 16;
    w.pos = m.pos;
    w.dist = m.len_dist & 0xFFFFu;
    w.k0 = w.pos < ts ? ts - w.pos : 0u;
    w.k1 = w.pos + len > te ? te - w.pos : len;
    w.kind = kLcNone;
    if (valid && w.k0 < w.k1) {
        const uint32_t n = w.k1 - w.k0, s0 = w.pos - w.dist + w.k0;  // first source byte (dist <= pos: checked by k_inflate_seg)
        if (n > kLcShort || w.dist < n) w.kind = kLcSlow;
        else if (s0 >= ts) w.kind = kLcTile;
        else if (s0 + n <= ts && s0 + gmis >= 4u) w.kind = kLcWindow;
        else w.kind = kLcSlow;
    }
}

// DBlock.cyc of a debug launch of k_lzcopy (wave 0's clocks): [0] whole member, [1] tile staged in, [2] chunk set-up
// (records, bitmap cleared, sources in front of the tile fetched, barrier), [3] polling loop, [4] tile written out,
// counts [5] polling iterations of all waves, [6] of them without progress, [7] matches
#ifndef GZPX_LC_WAVES
#define GZPX_LC_WAVES 6
#endif
template <bool DBG>
__global__ __launch_bounds__(kLcThreads, GZPX_LC_WAVES) void k_lzcopy(DBlock *__restrict__ blk_all, const uint64_t *__restrict__ out_off,
                                                         uint8_t *out_all, const LzMatch *__restrict__ mlist_all,
                                                         const uint32_t *__restrict__ tfirst_all, uint32_t *__restrict__ redo,
                                                         uint32_t *__restrict__ crc_found, CrcConsts cc) {
    __shared__ LcLds l;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t b = blockIdx.x;
    DBlock *blk = blk_all + b;
    if (blk->status != kInfOk) return;
    const uint32_t isize = blk->isize, nmatch = blk->nmatch;
    if (isize == 0 || nmatch == 0) return;  // literals and stored bytes are in place already
    const uint64_t ooff = out_off[b];
    uint8_t *out = out_all + ooff;
    const LzMatch *ml = mlist_all + (ooff / 3u + b);
    const uint32_t *tf = tfirst_all + ((ooff >> kLzTileShift) + 2ull * b);
    const bool multi = isize > kLzTile;
    const uint32_t phase = (uint32_t)((uintptr_t)out & 15u);
    const uint32_t gmis = (uint32_t)((uintptr_t)out & 3u);
    const uint32_t *out32 = (const uint32_t *)(out - gmis);  // the member's output as aligned dwords: byte p is byte p + gmis of it
    uint8_t *tile8 = (uint8_t *)l.tile;
    for (uint32_t i = tid; i < kLcBytes / 32 + 2; i += kLcThreads) l.bm[i] = 0xFFFFFFFFu;
    if (tid == 0) {
        l.gave_up = 0;
        l.crc_total = 0;
    }
    {  // CRC tables (one entry of each per thread) and the thread's shift: the product of the x^(512 * 2^k) of its bits
        uint32_t c = tid;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        l.crc_tab[0][tid] = c;
        uint32_t pw = 0x80000000u;  // x^0
        const uint32_t q = kLcThreads - 1u - tid;
        for (uint32_t k = 0; k < 8; k++)
            if ((q >> k) & 1u) pw = gf2_multmodp(cc.pow64[k], pw);
        l.crc_pw[tid] = pw;
        __syncthreads();
        const uint32_t t0 = l.crc_tab[0][tid];
        const uint32_t t1 = (t0 >> 8) ^ l.crc_tab[0][t0 & 0xFFu];
        const uint32_t t2 = (t1 >> 8) ^ l.crc_tab[0][t1 & 0xFFu];
        const uint32_t t3 = (t2 >> 8) ^ l.crc_tab[0][t2 & 0xFFu];
        l.crc_tab[1][tid] = t1;
        l.crc_tab[2][tid] = t2;
        l.crc_tab[3][tid] = t3;
    }
    uint32_t spins = 0;
    bool give_up = false;
    const long long t_begin = DBG ? clock64() : 0;
    uint32_t dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    const uint32_t ntiles = (isize + kLzTile - 1) / kLzTile;
    for (uint32_t t = 0; t < ntiles && !give_up; t++) {
        const uint32_t ts = t * kLzTile;
        const uint32_t te = ts + kLzTile < isize ? ts + kLzTile : isize;
        const uint32_t xo = phase + kLcPad - ts;  // X = p + xo
        // ---- the tile's matches: list entries [m0, m1), and the one before if it reaches into the tile
        uint32_t m0 = multi ? tf[t] : 0u;
        const uint32_t m1 = (multi && t + 1 < ntiles) ? tf[t + 1] : nmatch;
        if (t > 0 && m0 > 0) {
            const LzMatch pm = ml[m0 - 1];
            if (pm.pos + (pm.len_dist >> 16) > ts) m0--;
        }
        LzMatch nx[kLcK];  // the next chunk's records, on their way while this one is worked on
#pragma unroll
        for (uint32_t j = 0; j < kLcK; j++) {
            const uint32_t mi = m0 + j * kLcThreads + tid;
            nx[j] = ml[mi < m1 ? mi : m1 - 1 + (m1 == 0)];
        }
        __syncthreads();  // (the previous tile's stores have read the LDS copy)
        const long long t_in = DBG ? clock64() : 0;
        // ---- stage the tile: [ts, te) -> LDS, 16 bytes per lane where the address allows
        {
            const uint32_t a0 = ts + xo, a1 = te + xo;                 // X coordinates (address = out - xo + X)
            const uint32_t v0 = (a0 + 15u) & ~15u, v1 = a1 & ~15u;     // the 16-byte-aligned middle
            const uint8_t *src = out + ts - (phase + kLcPad);  // (the address of X = 0)
            if (v0 < v1) {
                for (uint32_t a = v0 + 16u * tid; a < v1; a += 16u * kLcThreads) *(uint4 *)(tile8 + a) = *(const uint4 *)(src + a);
                for (uint32_t a = a0 + tid; a < v0; a += kLcThreads) tile8[a] = src[a];
                for (uint32_t a = v1 + tid; a < a1; a += kLcThreads) tile8[a] = src[a];
            } else {
                for (uint32_t a = a0 + tid; a < a1; a += kLcThreads) tile8[a] = src[a];
            }
        }
        if (DBG) {
            __syncthreads();
            dbg[1] += (uint32_t)(clock64() - t_in);
        }
        for (uint32_t c0 = m0; c0 < m1 && !give_up; c0 += kLcChunk) {
            const long long t_c = DBG ? clock64() : 0;
            // this chunk's records: lane `tid` works on c0 + tid, c0 + 256 + tid, ... one after the other.  They stay
            // where they are (loop-invariant registers); `jn` counts the ones the lane has taken up.
            LzMatch rc[kLcK];
#pragma unroll
            for (uint32_t j = 0; j < kLcK; j++) rc[j] = nx[j];
            const uint32_t navail = c0 + tid >= m1 ? 0u : (m1 - c0 - tid + kLcThreads - 1) / kLcThreads;  // records of this lane in the chunk
            const uint32_t nmine = navail < kLcK ? navail : kLcK;
#pragma unroll
            for (uint32_t j = 0; j < kLcK; j++) {
                const uint32_t mi = c0 + kLcChunk + j * kLcThreads + tid;
                nx[j] = ml[mi < m1 ? mi : m1 - 1];
            }
            // every record's destination bits cleared; sources in front of the tile (final: the tile before this one was
            // written out and fenced) fetched from HBM for all four records at once, written and marked final
            uint32_t wdone = 0;  // bit j: record j needs no polling (outside the tile, or done here)
            {
                LcWork w[kLcK];
                uint32_t g[kLcK][6];
#pragma unroll
                for (uint32_t j = 0; j < kLcK; j++) {
                    lc_prepare(w[j], rc[j], j < nmine, ts, te, gmis);
                    if (w[j].kind == kLcNone) wdone |= 1u << j;
                    else lc_mark<false>(l.bm, w[j].pos + w[j].k0 + xo, w[j].pos + w[j].k1 + xo);
                    if (w[j].kind == kLcWindow) {
                        const uint32_t D = w[j].pos + w[j].k0 + xo;
                        const uint32_t sb = w[j].pos - w[j].dist + w[j].k0 + gmis - (D & 3u);  // >= 1: the byte of out32 for tile byte D & ~3
                        const uint32_t *q = out32 + (sb >> 2);
                        const dword4 v = *(const dword4 *)q;
                        g[j][0] = v.x;
                        g[j][1] = v.y;
                        g[j][2] = v.z;
                        g[j][3] = v.w;
                        g[j][4] = q[4];
                        g[j][5] = q[5];
                    }
                }
                if (tid == 0) l.chunk_lo = w[0].pos + w[0].k0 + xo;  // (the chunk's first record exists and lies in the tile)
                __syncthreads();
#pragma unroll
                for (uint32_t j = 0; j < kLcK; j++) {
                    if (w[j].kind == kLcWindow) {
                        const uint32_t D = w[j].pos + w[j].k0 + xo, n = w[j].k1 - w[j].k0;
                        const uint32_t sb = w[j].pos - w[j].dist + w[j].k0 + gmis - (D & 3u);
                        lc_put16(l.tile, g[j][0], g[j][1], g[j][2], g[j][3], g[j][4], g[j][5], sb & 3u, D, n);
                        lc_set32(l.bm, D, n);
                        wdone |= 1u << j;
                    }
                }
            }
            const uint32_t chunk_lo = l.chunk_lo;
            const long long t_p = DBG ? clock64() : 0;
            if (DBG) dbg[2] += (uint32_t)(t_p - t_c);
            // ---- the wave polls: no barrier between dependency levels, a lane moves on as soon as its match is done.
            // Short matches from inside the tile take the straight-line path, with everything that does not change
            // between polls worked out when the match is taken up.
            LzMatch cm = rc[0];        // the lane's current match
            bool have = false, is_short = false;
            uint32_t jn = 0;
            uint32_t f_bw = 0, f_want_lo = 0, f_want_hi = 0;  // bitmap dword of the first byte it needs, the bits it needs there
            uint32_t f_u = 0, f_r = 0, f_D = 0, f_n = 0;      // source dword, byte shift, destination byte, length
            for (;;) {
                // a lane without a match takes up its next record (one that needs no polling is dropped at once)
                if (__ballot(!have && jn < nmine) != 0) {
                    if (!have && jn < nmine) {
                        cm = rc[0];
#pragma unroll
                        for (uint32_t j = 1; j < kLcK; j++)
                            if (jn == j) cm = rc[j];
                        have = ((wdone >> jn) & 1u) == 0;
                        jn++;
                        LcWork w;
                        lc_prepare(w, cm, true, ts, te, gmis);
                        is_short = w.kind == kLcTile;
                        const uint32_t n = w.k1 - w.k0;
                        const uint32_t D = w.pos + w.k0 + xo, S = D - w.dist;
                        uint32_t need_a = S, need_e = S + n;  // (short: disjoint from the destination)
                        if (need_a < chunk_lo) need_a = chunk_lo;  // everything in front of the chunk is final
                        const uint32_t cnt = need_e > need_a ? need_e - need_a : 0u;  // <= 16 when short
                        const uint64_t want = ((1ull << (cnt & 31u)) - 1ull) << (need_a & 31u);
                        f_bw = need_a >> 5;
                        f_want_lo = (uint32_t)want;
                        f_want_hi = (uint32_t)(want >> 32);
                        f_u = (S - (D & 3u)) >> 2;
                        f_r = (S - (D & 3u)) & 3u;
                        f_D = D;
                        f_n = n;
                    }
                }
                if (__ballot(have) == 0) {
                    if (__ballot(jn < nmine) == 0) break;
                    continue;
                }
                bool progress = false;
                // (asked again at once, up to kLcReps times, while somebody copied: what a lane has just made final is
                // often what its neighbours -- the next matches of the list -- were waiting for, and a whole iteration
                // of this loop, with its take-up and long-match ballots, is four times this check.  Measured, 256 MiB
                // of text / DNA / FASTQ: 0.94 / 1.87 / 2.01 ms once, 0.82 / 1.47 / 1.63 twice, 0.76 / 1.25 / 1.45 four
                // times, the same at eight and sixteen; with the take-up inside as well 0.96 / 1.70 / 1.89.)
                for (uint32_t rep = 0; rep < kLcReps; rep++) {
                    const bool mine = have && is_short;
                    const uint32_t bw = mine ? f_bw : 0u, u = mine ? f_u : 0u;
                    const uint32_t b_lo = l.bm[bw], b_hi = l.bm[bw + 1];
                    const bool go = mine && (b_lo & f_want_lo) == f_want_lo && (b_hi & f_want_hi) == f_want_hi;
                    if (go) {
                        lc_put16(l.tile, l.tile[u], l.tile[u + 1], l.tile[u + 2], l.tile[u + 3], l.tile[u + 4], l.tile[u + 5], f_r, f_D, f_n);
                        // (no wait between the bytes and their "final" bits: a wave's LDS operations are carried out in the
                        // order it issues them, so whoever sees the bits sees the bytes)
                        lc_set32(l.bm, f_D, f_n);
                        have = false;
                    }
                    if (__ballot(go) == 0) break;
                    progress = true;
                }
                // ---- the rest (long, self-overlapping, or with a source across the tile's start): checked the long way,
                // copied by the wave together, 64 bytes per step.  An overlapping source repeats with period `dist` and
                // [src, dst) is final; bytes in front of the tile come from HBM.
                if (__ballot(have && !is_short) != 0) {
                    LcWork w;
                    w.pos = w.dist = w.k0 = w.k1 = 0;
                    w.kind = kLcNone;
                    bool mine = have && !is_short, open_range = false;
                    uint32_t need_a = 0, need_e = 0;
                    if (mine) {
                        lc_prepare(w, cm, true, ts, te, gmis);
                        const uint32_t s0 = w.pos - w.dist + w.k0, s1 = w.pos - w.dist + w.k1, d0 = w.pos + w.k0;
                        const uint32_t pa = s0 > ts ? s0 : ts, pe = s1 < d0 ? s1 : d0;  // the part inside the tile that it does not write itself
                        need_a = pa + xo;
                        need_e = pe + xo;
                        if (need_a < chunk_lo) need_a = chunk_lo;  // everything in front of the chunk is final
                        open_range = pa < pe && need_a < need_e;
                    }
                    // (again and again while something became ready: a run of long matches that each copy from the one
                    // before -- the 254 matches of a block of zeros -- goes link by link here, not one polling iteration each)
                    for (;;) {
                        const bool ready = mine && (!open_range || lc_all_set(l.bm, need_a, need_e));
                        uint64_t longs = __ballot(ready);
                        if (longs == 0) break;
                        progress = true;
                        while (longs) {
                            const uint32_t jl = (uint32_t)__ffsll((long long)longs) - 1u;
                            longs &= longs - 1ull;
                            const uint32_t jp = rdlane(w.pos, jl), jdist = rdlane(w.dist, jl), jk = rdlane(w.k0, jl), jk1 = rdlane(w.k1, jl);
                            for (uint32_t i = jk + lane; i < jk1; i += 64) {
                                const uint32_t r = i < jdist ? i : i % jdist;
                                const uint32_t sp = jp - jdist + r;
                                const uint32_t v = sp < ts ? (uint32_t)out[sp] : (uint32_t)tile8[sp + xo];
                                tile8[jp + i + xo] = (uint8_t)v;
                            }
                        }
                        wave_sync();  // (the wave's byte stores above are other lanes' stores for the owning lane)
                        if (ready) {
                            lc_mark<true>(l.bm, w.pos + w.k0 + xo, w.pos + w.k1 + xo);
                            have = false;
                            mine = false;
                        }
                    }
                }
                if (DBG) dbg[5]++;
                if (!progress) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kLcMaxSpins) l.gave_up = 1;  // (cannot happen: the lowest pending match is always ready)
                    if (l.gave_up) break;
                    if (DBG) dbg[6]++;
                }
            }
            __syncthreads();
            if (DBG) dbg[3] += (uint32_t)(clock64() - t_p);
            give_up = l.gave_up != 0;
        }
        __syncthreads();
        const long long t_out = DBG ? clock64() : 0;
        // ---- the tile's CRC-32 while it is here: thread t takes the 64-byte piece that ends 64 (255 - t) bytes in front
        // of the tile's end (only the first pieces of a short tile are empty), shifts its CRC to the tile's end, the
        // pieces are XORed, and thread 0 appends the tile to the member's running CRC
        if (!give_up) {
            const uint32_t tl = te - ts, a0 = ts + xo;
            const int32_t e_i = (int32_t)tl - 64 * (int32_t)(kLcThreads - 1u - tid);
            uint32_t piece = 0;
            if (e_i > 0) {
                uint32_t pos = a0 + (e_i > 64 ? (uint32_t)(e_i - 64) : 0u);
                const uint32_t end = a0 + (uint32_t)e_i;
                uint32_t c = 0xFFFFFFFFu;
                while (pos < end && (pos & 3u)) {
                    c = (c >> 8) ^ l.crc_tab[0][(c ^ tile8[pos]) & 0xFFu];
                    pos++;
                }
                while (pos + 4 <= end) {
                    c ^= l.tile[pos >> 2];
                    c = l.crc_tab[3][c & 0xFFu] ^ l.crc_tab[2][(c >> 8) & 0xFFu] ^ l.crc_tab[1][(c >> 16) & 0xFFu] ^ l.crc_tab[0][c >> 24];
                    pos += 4;
                }
                while (pos < end) {
                    c = (c >> 8) ^ l.crc_tab[0][(c ^ tile8[pos]) & 0xFFu];
                    pos++;
                }
                piece = gf2_multmodp(l.crc_pw[tid], ~c);
            }
            for (int m = 32; m >= 1; m >>= 1) piece ^= __shfl_xor(piece, m);
            if (lane == 0) l.crc_part[tid >> 6] = piece;
            __syncthreads();
            if (tid == 0) {
                uint32_t tile_crc = 0;
                for (uint32_t w2 = 0; w2 < kLcThreads / 64; w2++) tile_crc ^= l.crc_part[w2];
                uint32_t tot = l.crc_total;
                if (t > 0) {  // crc(A || B) = crc(A) x^(8 |B|) + crc(B): 64-byte steps from the table, the rest byte by byte
                    const uint32_t q = tl >> 6;
                    tot = gf2_multmodp(q >= kLcThreads ? cc.pow64[8] : l.crc_pw[kLcThreads - 1u - q], tot);
                    for (uint32_t r = 0; r < (tl & 63u); r++) tot = (tot >> 8) ^ l.crc_tab[0][tot & 0xFFu];
                }
                l.crc_total = tot ^ tile_crc;
            }
        }
        // ---- the finished tile back to HBM; what the next tile reads of it comes from there
        if (!give_up) {
            const uint32_t a0 = ts + xo, a1 = te + xo;
            const uint32_t v0 = (a0 + 15u) & ~15u, v1 = a1 & ~15u;
            uint8_t *dstp = out + ts - (phase + kLcPad);
            if (v0 < v1) {
                for (uint32_t a = v0 + 16u * tid; a < v1; a += 16u * kLcThreads) *(uint4 *)(dstp + a) = *(const uint4 *)(tile8 + a);
                for (uint32_t a = a0 + tid; a < v0; a += kLcThreads) dstp[a] = tile8[a];
                for (uint32_t a = v1 + tid; a < a1; a += kLcThreads) dstp[a] = tile8[a];
            } else {
                for (uint32_t a = a0 + tid; a < a1; a += kLcThreads) dstp[a] = tile8[a];
            }
            if (t + 1 < ntiles) {
                // (workgroup scope is all it takes: the waves of a workgroup share their CU's write-through vector cache;
                // an agent-scope fence here writes back and invalidates caches, measured at ~0.1 ms per tile)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
        }
        if (DBG) {
            __syncthreads();
            dbg[4] += (uint32_t)(clock64() - t_out);
        }
    }
    if (give_up) {
        seg_redo(blk, redo, b, tid);
    } else if (tid == 0) {  // (thread 0 wrote crc_total last: its own program order)
        crc_found[b] = l.crc_total;
        blk->nmatch = nmatch | kLcCrcDone;
    }
    if (DBG) {
        if (tid == 0) {
            blk->cyc[0] = (uint32_t)(clock64() - t_begin);
            for (uint32_t k = 1; k < 5; k++) blk->cyc[k] = dbg[k];
            blk->cyc[5] = blk->cyc[6] = 0;
            blk->cyc[7] = nmatch;
        }
        __syncthreads();
        if (lane == 0) {
            atomicAdd(&blk->cyc[5], dbg[5]);
            atomicAdd(&blk->cyc[6], dbg[6]);
        }
    }
}
