// gzpx_nearopt.hip -- levels 10-12: libdeflate's near-optimal parser (deflate_compress_near_optimal,
// v1.10) on the GPU.  What it replaces: the same call site as every other level
// (libdeflater::Compressor::deflate_compress, src/bgzf.rs:214-216 / src/mgzip.rs:201-203) with
// CompressionLvl 10..12 (src/deflate.rs:596-599).
//
// The parser is sequential in every part that costs: bt_matchfinder keeps the window's positions
// in binary trees that every position re-roots (a data-dependent walk of up to max_search_depth
// nodes), every position's matches are cached, blocks are split on running statistics with a
// rewind to the previous check, and each DEFLATE block is then parsed two to four times backwards
// (minimum-cost path) with costs taken from the Huffman codes of the pass before.  There is no
// position-parallel restatement of the tree walk (a node's place depends on every earlier walk),
// so this first form runs the algorithm as it stands with ONE LANE PER BLOCK: BGZF / Mgzip blocks are
// independent, a slab has thousands of them, and a lane's dependent-load chain is hidden behind
// the other lanes' and waves'.  State per block in flight: the trees and hash tables (1 MiB: 32-bit
// absolute positions, so no window slide is needed -- an entry is alive iff it is less than 32768
// behind, exactly what the reference's 16-bit rebasing computes), the match cache (libdeflate's
// MATCH_CACHE_LENGTH: its overflow ends a block, so the size is part of the result), the
// minimum-cost path nodes, costs, frequencies and the Huffman scratch.  Output: the token stream and
// DEFLATE block boundaries (SubMeta) the level-independent back end expects -- k_hist, k_huffman
// (libdeflate's flush_block: codes from the final path's frequencies, dynamic / static / stored
// choice), k_scan, k_emit follow as at every level.
//
// Rates are those of a serial algorithm on 64-wide hardware; the point of this file is
// that levels 10-12 exist on the device path, bit-exact with libdeflate 1.10 (later versions
// changed this parser: tests/golden/l1012_vectors.json is the v1.10 binary's output).
#include <hip/hip_runtime.h>
#include <cstring>

#include "gzpx_device.h"

namespace gzpx {

namespace {

constexpr uint32_t kNoWindow = 32768;
constexpr uint32_t kNoMinMatch = 3, kNoMaxMatch = 258;
constexpr uint32_t kNoSoftMaxBlock = 300000;       // SOFT_MAX_BLOCK_LENGTH
constexpr uint32_t kNoMaxBlock = kNoSoftMaxBlock + kMinBlockLen - 1;
constexpr uint32_t kNoCacheLen = kNoSoftMaxBlock * 5;  // MATCH_CACHE_LENGTH
constexpr uint32_t kNoCacheSlack = (kNoMaxMatch - kNoMinMatch + 1) + kNoMaxMatch - 1;
constexpr uint32_t kNoBitCost = 16;
constexpr uint32_t kNoLitNostat = 13, kNoLenNostat = 13, kNoOffNostat = 10;
constexpr uint32_t kNoNumObs = 10, kNoNumLitObs = 8, kNoObsPerCheck = 512;
constexpr uint32_t kNoNumLitlen = 288, kNoNumOffset = 32, kNoEob = 256, kNoFirstLen = 257;
constexpr int32_t kNoDead = -32768;  // MATCHFINDER_INITVAL: never within the window of any position

__device__ const uint16_t kNoLenBase[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                            31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t kNoLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint8_t kNoOffExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
// choose_min_match_len (deflate_compress.c): by the number of distinct literals in use
__device__ const uint8_t kNoMinLens[80] = {9, 9, 9, 9, 9, 9, 8, 8, 7, 7, 6, 6, 6, 6, 6, 6, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
                                           5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4,
                                           4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};

struct NoMatch {
    uint16_t length, offset;
};
struct NoNode {
    uint32_t cost_to_end, item;  // item: literal (byte << 9) | 1, match (offset << 9) | length
};
struct __attribute__((aligned(8))) NoKids {
    int32_t x, y;  // a tree node's children: less-than side, greater-than side
};

}  // namespace

// Everything a block in flight owns besides its match cache and path nodes (one per lane).
struct NoLane {
    int32_t hash3[1u << 16][2];  // bt_matchfinder: two most recent positions per 3-byte hash
    int32_t hash4[1u << 16];     // the root of the 4-byte hash's tree
    int32_t child[2 * kNoWindow];
    uint32_t cost_lit[256], cost_len[kNoMaxMatch + 1], cost_off[30];
    uint32_t fr_litlen[kNoNumLitlen], fr_offset[kNoNumOffset];
    uint32_t new_mlf[kNoMaxMatch + 1], mlf[kNoMaxMatch + 1];  // (new_)match_len_freqs
    uint32_t litfreq[256];
    unsigned long long hA[kNoNumLitlen], hNF[kNoNumLitlen];  // Huffman scratch: sorted leaves, node frequencies
    uint16_t hparent[kNoNumLitlen];
    uint8_t hdepth[kNoNumLitlen];
    uint8_t lens_litlen[kNoNumLitlen], lens_offset[kNoNumOffset];
    // libdeflate's default_litlen_costs[]: int(-log2((1 - p) / max(j, 1)) * BIT_COST), int(-log2(p / 29) * BIT_COST) for
    // p = 0.25 / 0.5 / 0.75 (computed by the host, gzpx_api.cpp; the same bytes sit in the v1.10 binary's read-only data)
    uint8_t default_lit[3][257], default_len_sym[3];
};

size_t no_lane_bytes() { return sizeof(NoLane); }
size_t no_cache_bytes() { return (size_t)(kNoCacheLen + kNoCacheSlack) * sizeof(NoMatch); }
size_t no_nodes_bytes(uint32_t block_size) {
    // (a DEFLATE block is at most min(block, MAX_BLOCK_LENGTH) long; nodes up to MAX_MATCH_LEN - 1 behind its end are
    // marked unreachable before every parse)
    return (size_t)((block_size < kNoMaxBlock ? block_size : kNoMaxBlock) + 1 + kNoMaxMatch) * sizeof(NoNode);
}

namespace {

__device__ __forceinline__ uint32_t no_le32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ uint32_t no_le24(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}
__device__ __forceinline__ uint32_t no_hash(uint32_t v, uint32_t bits) { return (v * 0x1E35A7BDu) >> (32 - bits); }
__device__ __forceinline__ uint32_t no_length_slot(uint32_t len) {
    uint32_t s = 0;
    for (uint32_t i = 1; i < 29; i++)
        if (len >= kNoLenBase[i]) s = i;
    return s;
}
__device__ __forceinline__ uint32_t no_offset_slot(uint32_t off) {
    // DEFLATE offset slots: 0..3 for 1..4, then two slots per power of two
    if (off <= 4) return off - 1;
    const uint32_t m = off - 1, hb = 31u - (uint32_t)__clz((int)m);
    return 2 * hb + ((m >> (hb - 1)) & 1u);
}
// lz_extend, four bytes per step (unaligned dword loads are fine in global memory; both pointers stay inside the block:
// len + 4 <= max_len <= bytes left)
__device__ uint32_t no_lz_extend(const uint8_t *a, const uint8_t *b, uint32_t len, uint32_t max_len) {
    while (len + 4 <= max_len) {
        uint32_t va, vb;
        __builtin_memcpy(&va, a + len, 4);
        __builtin_memcpy(&vb, b + len, 4);
        const uint32_t x = va ^ vb;
        if (x) return len + (((uint32_t)__ffs((int)x) - 1u) >> 3);
        len += 4;
    }
    while (len < max_len && a[len] == b[len]) len++;
    return len;
}

__device__ uint32_t no_choose_min_match_len(uint32_t num_used_literals, uint32_t max_search_depth) {
    uint32_t min_len = kNoMinMatch;
    if (num_used_literals < 80) min_len = kNoMinLens[num_used_literals];
    if (max_search_depth < 16) {
        const uint32_t cap = max_search_depth < 5 ? 4 : max_search_depth < 10 ? 5 : 7;
        if (min_len > cap) min_len = cap;
    }
    return min_len;
}

// calculate_min_match_len (the first 4096 bytes of the block; libdeflate >= 1.1x: scans shorter than 512 bytes use 3)
__device__ uint32_t no_calc_min_len(const uint8_t *data, uint32_t data_len, uint32_t depth, uint32_t compat) {
    uint32_t used[8] = {0, 0, 0, 0, 0, 0, 0, 0}, num_used = 0;
    if (compat == 0 && data_len < 512) return kNoMinMatch;
    if (data_len > 4096) data_len = 4096;
    for (uint32_t i = 0; i < data_len; i++) used[data[i] >> 5] |= 1u << (data[i] & 31u);
    for (uint32_t k = 0; k < 8; k++) num_used += (uint32_t)__popc(used[k]);
    return no_choose_min_match_len(num_used, depth);
}

// bt_matchfinder_advance_one_byte (record = get_matches, else skip_byte); positions are absolute in the block
__device__ NoMatch *no_bt_advance(NoLane &L, const uint8_t *in, uint32_t cur_pos, uint32_t max_len, uint32_t nice_len,
                                  uint32_t max_search_depth, uint32_t (&next_hashes)[2], NoMatch *mp, bool record) {
    const uint8_t *in_next = in + cur_pos;
    uint32_t depth_remaining = max_search_depth;
    const int32_t cutoff = (int32_t)cur_pos - (int32_t)kNoWindow;
    const uint32_t next_hashseq = no_le32(in_next + 1);
    const uint32_t hash3 = next_hashes[0], hash4 = next_hashes[1];
    next_hashes[0] = no_hash(next_hashseq & 0xFFFFFFu, 16);
    next_hashes[1] = no_hash(next_hashseq, 16);

    int32_t cur_node = L.hash3[hash3][0];
    L.hash3[hash3][0] = (int32_t)cur_pos;
    const int32_t cur_node_2 = L.hash3[hash3][1];
    L.hash3[hash3][1] = cur_node;
    if (record && cur_node > cutoff) {
        const uint32_t seq3 = no_le24(in_next);
        if (seq3 == no_le24(in + cur_node)) {
            mp->length = 3;
            mp->offset = (uint16_t)(cur_pos - (uint32_t)cur_node);
            mp++;
        } else if (cur_node_2 > cutoff && seq3 == no_le24(in + cur_node_2)) {
            mp->length = 3;
            mp->offset = (uint16_t)(cur_pos - (uint32_t)cur_node_2);
            mp++;
        }
    }
    cur_node = L.hash4[hash4];
    L.hash4[hash4] = (int32_t)cur_pos;

    int32_t *pending_lt = &L.child[2 * (cur_pos & (kNoWindow - 1))];
    int32_t *pending_gt = pending_lt + 1;
    if (cur_node <= cutoff) {
        *pending_lt = kNoDead;
        *pending_gt = kNoDead;
        return mp;
    }
    uint32_t best_lt_len = 0, best_gt_len = 0, len = 0, best_len = 3;
    for (;;) {
        const uint8_t *matchptr = in + cur_node;
        int32_t *node_children = &L.child[2 * ((uint32_t)cur_node & (kNoWindow - 1))];
        // the node's two children are read NOW, beside its bytes (one round trip to HBM per node instead of two: the
        // walk is a chain of dependent loads).  Nothing below writes this node's own slots before they are used: the
        // pending pointers belong to nodes visited earlier, or to the position being inserted.
        const NoKids kids = *(const NoKids *)node_children;
        if (matchptr[len] == in_next[len]) {
            len = no_lz_extend(in_next, matchptr, len + 1, max_len);
            if (!record || len > best_len) {
                if (record) {
                    best_len = len;
                    mp->length = (uint16_t)len;
                    mp->offset = (uint16_t)(cur_pos - (uint32_t)cur_node);
                    mp++;
                }
                if (len >= nice_len) {
                    *pending_lt = kids.x;
                    *pending_gt = kids.y;
                    return mp;
                }
            }
        }
        if (matchptr[len] < in_next[len]) {
            *pending_lt = cur_node;
            pending_lt = node_children + 1;
            cur_node = kids.y;
            best_lt_len = len;
            if (best_gt_len < len) len = best_gt_len;
        } else {
            *pending_gt = cur_node;
            pending_gt = node_children;
            cur_node = kids.x;
            best_gt_len = len;
            if (best_lt_len < len) len = best_lt_len;
        }
        if (cur_node <= cutoff || !--depth_remaining) {
            *pending_lt = kNoDead;
            *pending_gt = kNoDead;
            return mp;
        }
    }
}

// deflate_make_huffman_code, code lengths only (libdeflate's rules: leaves sorted by (frequency, symbol), two-queue
// tree build with leaf preference, length limiting by the clamp on the length counts; a code of fewer than two used
// symbols gets two one-bit codewords -- in the 1.10 compat mode an empty code keeps all lengths zero)
__device__ void no_make_code_lens(NoLane &L, uint32_t num_syms, uint32_t max_len, uint32_t compat_1_10,
                                  const uint32_t *freqs, uint8_t *lens) {
    unsigned long long *A = L.hA, *NF = L.hNF;
    uint32_t num_used = 0;
    for (uint32_t s = 0; s < num_syms; s++) {
        lens[s] = 0;
        if (freqs[s]) {  // insertion sort by (freq << 10 | sym)
            const unsigned long long key = ((unsigned long long)freqs[s] << 10) | s;
            uint32_t k = num_used++;
            while (k > 0 && A[k - 1] > key) {
                A[k] = A[k - 1];
                k--;
            }
            A[k] = key;
        }
    }
    if (num_used == 0 && compat_1_10) return;
    if (num_used < 2) {
        const uint32_t sym = num_used ? (uint32_t)(A[0] & 1023u) : 0u;
        lens[0] = 1;
        lens[sym ? sym : 1] = 1;
        return;
    }
    uint32_t len_counts[16];
    const uint32_t last = num_used - 1;
    {
        uint32_t i = 0, b = 0, e = 0;
        do {
            unsigned long long nf;
            if (i + 1 <= last && (b == e || (A[i + 1] >> 10) <= NF[b])) {
                nf = (A[i] >> 10) + (A[i + 1] >> 10);
                i += 2;
            } else if (b + 2 <= e && (i > last || NF[b + 1] < (A[i] >> 10))) {
                nf = NF[b] + NF[b + 1];
                L.hparent[b] = (uint16_t)e;
                L.hparent[b + 1] = (uint16_t)e;
                b += 2;
            } else {
                nf = (A[i] >> 10) + NF[b];
                L.hparent[b] = (uint16_t)e;
                i++;
                b++;
            }
            NF[e] = nf;
        } while (++e < last);
    }
    for (uint32_t l = 0; l < 16; l++) len_counts[l] = 0;
    len_counts[1] = 2;
    const uint32_t root = last - 1;
    L.hdepth[root] = 0;
    for (int node = (int)root - 1; node >= 0; node--) {
        const uint32_t d = (uint32_t)L.hdepth[L.hparent[node]] + 1;
        uint32_t l = d;
        L.hdepth[node] = (uint8_t)d;
        if (l >= max_len) {
            l = max_len;
            do {
                l--;
            } while (len_counts[l] == 0);
        }
        len_counts[l]--;
        len_counts[l + 1] += 2;
    }
    uint32_t k = 0;
    for (uint32_t l = max_len; l >= 1; l--)
        for (uint32_t c = len_counts[l]; c; c--) lens[A[k++] & 1023u] = (uint8_t)l;
}

struct NoStats {
    uint32_t new_obs[kNoNumObs], obs[kNoNumObs], num_new, num;
    uint32_t prev_obs[kNoNumObs], prev_num;
};

// do_end_block_check (shared by every splitting parser; restated in k_parse_hc's form as well)
__device__ bool no_end_block_check(NoStats &st, uint32_t block_length) {
    if (st.num > 0) {
        uint32_t total_delta = 0;
        for (uint32_t i = 0; i < kNoNumObs; i++) {
            const uint32_t expected = st.obs[i] * st.num_new, actual = st.new_obs[i] * st.num;
            total_delta += actual > expected ? actual - expected : expected - actual;
        }
        const uint32_t num_items = st.num + st.num_new;
        uint32_t cutoff = st.num_new * 200 / 512 * st.num;
        if (block_length < 10000 && num_items < 8192)
            cutoff += (uint32_t)((unsigned long long)cutoff * (8192 - num_items) / 8192);
        if (total_delta + (block_length / 4096) * st.num >= cutoff) return true;
    }
    for (uint32_t i = 0; i < kNoNumObs; i++) {
        st.num += st.new_obs[i];
        st.obs[i] += st.new_obs[i];
        st.new_obs[i] = 0;
    }
    st.num_new = 0;
    return false;
}

__device__ void no_merge_stats(NoLane &L, NoStats &st) {
    for (uint32_t i = 0; i < kNoNumObs; i++) {
        st.num += st.new_obs[i];
        st.obs[i] += st.new_obs[i];
        st.new_obs[i] = 0;
    }
    st.num_new = 0;
    for (uint32_t i = 0; i <= kNoMaxMatch; i++) {
        L.mlf[i] += L.new_mlf[i];
        L.new_mlf[i] = 0;
    }
}

__device__ __forceinline__ uint32_t no_default_length_cost(uint32_t len, uint32_t len_sym_cost) {
    return len_sym_cost + kNoLenExtra[no_length_slot(len)] * kNoBitCost;
}
__device__ __forceinline__ uint32_t no_default_offset_cost(uint32_t slot) {
    return 4 * kNoBitCost + (907 * kNoBitCost) / 1000 + kNoOffExtra[slot] * kNoBitCost;
}
__device__ __forceinline__ uint32_t no_blend(uint32_t cost, uint32_t def, int change) {
    return change == 0 ? (def + 3 * cost) / 4 : change == 1 ? (def + cost) / 2 : change == 2 ? (5 * def + 3 * cost) / 8
                                                                                             : (3 * def + cost) / 4;
}

// deflate_optimize_block + the block's tokens and SubMeta entry (deflate_flush_block's input)
__device__ void no_optimize_block(NoLane &L, NoStats &st, NoNode *nodes, const uint8_t *block_begin,
                                  uint32_t block_length, const NoMatch *cache_end, bool is_first, uint32_t depth,
                                  uint32_t num_passes, uint32_t compat_1_10, uint32_t *tok, uint32_t &ntok) {
    // the block really ends at block_length, even if matches reach beyond it
    {
        const uint32_t hi = block_length - 1 + kNoMaxMatch < kNoMaxBlock ? block_length - 1 + kNoMaxMatch : kNoMaxBlock;
        for (uint32_t i = block_length; i <= hi; i++) nodes[i].cost_to_end = 0x80000000u;
    }
    // deflate_choose_default_litlen_costs
    uint32_t lit_cost, len_sym_cost;
    {
        uint32_t num_used = 0, literal_freq = block_length, match_freq = 0, i;
        for (i = 0; i < 256; i++) L.litfreq[i] = 0;
        const uint32_t cutoff = literal_freq >> 11;
        for (i = 0; i < block_length; i++) L.litfreq[block_begin[i]]++;
        for (i = 0; i < 256; i++)
            if (L.litfreq[i] > cutoff) num_used++;
        if (num_used == 0) num_used = 1;
        for (i = no_choose_min_match_len(num_used, depth); i <= kNoMaxMatch; i++) {
            match_freq += L.mlf[i];
            literal_freq -= i * L.mlf[i];
        }
        if ((int32_t)literal_freq < 0) literal_freq = 0;
        i = match_freq > literal_freq ? 2u : match_freq * 4 > literal_freq ? 1u : 0u;
        lit_cost = L.default_lit[i][num_used];
        len_sym_cost = L.default_len_sym[i];
    }
    if (is_first) {
        for (uint32_t i = 0; i < 256; i++) L.cost_lit[i] = lit_cost;
        for (uint32_t i = kNoMinMatch; i <= kNoMaxMatch; i++) L.cost_len[i] = no_default_length_cost(i, len_sym_cost);
        for (uint32_t i = 0; i < 30; i++) L.cost_off[i] = no_default_offset_cost(i);
    } else {  // deflate_adjust_costs: the more the block differs from the previous one, the more the defaults count
        unsigned long long total_delta = 0;
        for (uint32_t i = 0; i < kNoNumObs; i++) {
            const unsigned long long prev = (unsigned long long)st.prev_obs[i] * st.num;
            const unsigned long long cur = (unsigned long long)st.obs[i] * st.prev_num;
            total_delta += prev > cur ? prev - cur : cur - prev;
        }
        const unsigned long long cutoff = ((unsigned long long)st.prev_num * st.num * 200) / 512;
        const int change = 4 * total_delta > 9 * cutoff ? 3 : 2 * total_delta > 3 * cutoff ? 2 : 2 * total_delta > cutoff ? 1 : 0;
        for (uint32_t i = 0; i < 256; i++) L.cost_lit[i] = no_blend(L.cost_lit[i], lit_cost, change);
        for (uint32_t i = kNoMinMatch; i <= kNoMaxMatch; i++)
            L.cost_len[i] = no_blend(L.cost_len[i], no_default_length_cost(i, len_sym_cost), change);
        for (uint32_t i = 0; i < 30; i++) L.cost_off[i] = no_blend(L.cost_off[i], no_default_offset_cost(i), change);
    }

    for (uint32_t pass = 0; pass < num_passes; pass++) {
        // deflate_find_min_cost_path
        const NoMatch *cp = cache_end;
        NoNode *cur = nodes + block_length;
        cur->cost_to_end = 0;
        do {
            cur--;
            cp--;
            const uint32_t num_matches = cp->length, literal = cp->offset;
            uint32_t best = L.cost_lit[literal] + (cur + 1)->cost_to_end;
            uint32_t item = (literal << 9) | 1u;
            if (num_matches) {
                const NoMatch *match = cp - num_matches;
                uint32_t len = kNoMinMatch;
                do {
                    const uint32_t offset = match->offset, mlen = match->length;
                    const uint32_t offset_cost = L.cost_off[no_offset_slot(offset)];
                    do {
                        const uint32_t c = offset_cost + L.cost_len[len] + (cur + len)->cost_to_end;
                        if (c < best) {
                            best = c;
                            item = (offset << 9) | len;
                        }
                    } while (++len <= mlen);
                } while (++match != cp);
                cp -= num_matches;
            }
            cur->cost_to_end = best;
            cur->item = item;
        } while (cur != nodes);
        // deflate_tally_item_list (+ the end-of-block symbol), the codes, the costs they imply -- after the last
        // pass as well: the next block blends its defaults with them (both pinned on the v1.10 binary: DESIGN 7)
        for (uint32_t i = 0; i < kNoNumLitlen; i++) L.fr_litlen[i] = 0;
        for (uint32_t i = 0; i < kNoNumOffset; i++) L.fr_offset[i] = 0;
        for (uint32_t pos = 0; pos < block_length;) {
            const uint32_t it = nodes[pos].item, length = it & 511u, offset = it >> 9;
            if (length == 1) {
                L.fr_litlen[offset]++;
            } else {
                L.fr_litlen[kNoFirstLen + no_length_slot(length)]++;
                L.fr_offset[no_offset_slot(offset)]++;
            }
            pos += length;
        }
        L.fr_litlen[kNoEob]++;
        no_make_code_lens(L, kNoNumLitlen, 14, compat_1_10, L.fr_litlen, L.lens_litlen);
        no_make_code_lens(L, kNoNumOffset, 15, compat_1_10, L.fr_offset, L.lens_offset);
        for (uint32_t i = 0; i < 256; i++) L.cost_lit[i] = (L.lens_litlen[i] ? L.lens_litlen[i] : kNoLitNostat) * kNoBitCost;
        for (uint32_t i = kNoMinMatch; i <= kNoMaxMatch; i++) {
            const uint32_t slot = no_length_slot(i), l = L.lens_litlen[kNoFirstLen + slot];
            L.cost_len[i] = ((l ? l : kNoLenNostat) + kNoLenExtra[slot]) * kNoBitCost;
        }
        for (uint32_t i = 0; i < 30; i++)
            L.cost_off[i] = ((L.lens_offset[i] ? L.lens_offset[i] : kNoOffNostat) + kNoOffExtra[i]) * kNoBitCost;
    }
    for (uint32_t pos = 0; pos < block_length;) {
        const uint32_t it = nodes[pos].item, length = it & 511u, offset = it >> 9;
        tok[ntok++] = length == 1 ? offset : (kTokMatch | (offset << 9) | length);
        pos += length;
    }
}

}  // namespace

// One lane per block; a lane walks the blocks lane_id, + lanes in flight, ...
#ifndef GZPX_NO_WAVES
#define GZPX_NO_WAVES 1  // waves per SIMD the kernel is compiled for
#endif
__global__ __launch_bounds__(64, GZPX_NO_WAVES) void k_near_optimal(Config cfg, const uint8_t *__restrict__ slab,
                                                     BlockMeta *__restrict__ meta_all, SubMeta *__restrict__ sub_all,
                                                     uint32_t *__restrict__ tok_all, uint32_t nb, NoLane *lanes,
                                                     uint8_t *cache_all, uint8_t *nodes_all, size_t nodes_stride,
                                                     uint32_t n_lanes) {
    const uint32_t lane_id = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane_id >= n_lanes) return;
    NoLane &L = lanes[lane_id];
    NoMatch *const cache = (NoMatch *)(cache_all + (size_t)lane_id * ((size_t)(kNoCacheLen + kNoCacheSlack) * sizeof(NoMatch)));
    NoNode *const nodes = (NoNode *)(nodes_all + (size_t)lane_id * nodes_stride);
    const uint32_t depth = cfg.hc_depth, nice_level = cfg.hc_nice, num_passes = cfg.no_passes;
    const uint32_t compat_1_10 = cfg.compat != 0;

    for (uint32_t b = lane_id; b < nb; b += n_lanes) {
        BlockMeta *meta = meta_all + b;
        const uint32_t n = meta->n;
        if (n <= cfg.passthrough) continue;  // deflate_compress_none: k_huffman writes the stored block
        const uint8_t *in = slab + (uint64_t)b * cfg.block_size;
        SubMeta *sub = sub_all + (uint64_t)b * cfg.max_sub;
        uint32_t *tok = tok_all + (uint64_t)b * cfg.stride;
        uint32_t ntok = 0, nsub = 0;

        // bt_matchfinder_init + deflate_near_optimal_init_stats
        for (uint32_t i = 0; i < (1u << 16); i++) {
            L.hash3[i][0] = kNoDead;
            L.hash3[i][1] = kNoDead;
            L.hash4[i] = kNoDead;
            L.child[i] = kNoDead;
        }
        NoStats st;
        for (uint32_t i = 0; i < kNoNumObs; i++) st.new_obs[i] = st.obs[i] = st.prev_obs[i] = 0;
        st.num_new = st.num = st.prev_num = 0;
        for (uint32_t i = 0; i <= kNoMaxMatch; i++) L.new_mlf[i] = L.mlf[i] = 0;

        uint32_t in_next = 0, block_begin = 0;
        uint32_t max_len = kNoMaxMatch, nice_len = nice_level < max_len ? nice_level : max_len;
        NoMatch *cache_ptr = cache;
        uint32_t next_hashes[2] = {0, 0};
        do {
            // a new DEFLATE block
            const uint32_t max_block_end = (n - block_begin < kNoSoftMaxBlock + kMinBlockLen) ? n : block_begin + kNoSoftMaxBlock;
            uint32_t prev_check = 0xFFFFFFFFu;  // prev_end_block_check (none yet)
            bool change_detected = false;
            uint32_t next_observation = in_next;
            const uint32_t min_len = no_calc_min_len(in + block_begin, max_block_end - block_begin, depth, cfg.compat == 0 ? 0u : 1u);
            for (;;) {
                NoMatch *matches = cache_ptr;
                uint32_t best_len = 0;
                uint32_t remaining = n - in_next;
                if (remaining < kNoMaxMatch) {
                    max_len = remaining;
                    if (nice_len > max_len) nice_len = max_len;
                }
                if (max_len >= 5) {
                    cache_ptr = no_bt_advance(L, in, in_next, max_len, nice_len, depth, next_hashes, matches, true);
                    if (cache_ptr > matches) best_len = cache_ptr[-1].length;
                }
                if (in_next >= next_observation) {
                    if (best_len >= min_len) {
                        st.new_obs[kNoNumLitObs + (best_len >= 9 ? 1u : 0u)]++;
                        st.num_new++;
                        next_observation = in_next + best_len;
                        L.new_mlf[best_len]++;
                    } else {
                        const uint32_t lit = in[in_next];
                        st.new_obs[((lit >> 5) & 0x6u) | (lit & 1u)]++;
                        st.num_new++;
                        next_observation = in_next + 1;
                    }
                }
                cache_ptr->length = (uint16_t)(cache_ptr - matches);
                cache_ptr->offset = in[in_next];
                in_next++;
                cache_ptr++;
                // a very long match: no matches are cached for the bytes it covers
                if (best_len >= kNoMinMatch && best_len >= nice_len) {
                    --best_len;
                    do {
                        remaining = n - in_next;
                        if (remaining < kNoMaxMatch) {
                            max_len = remaining;
                            if (nice_len > max_len) nice_len = max_len;
                        }
                        if (max_len >= 5) no_bt_advance(L, in, in_next, max_len, nice_len, depth, next_hashes, nullptr, false);
                        cache_ptr->length = 0;
                        cache_ptr->offset = in[in_next];
                        in_next++;
                        cache_ptr++;
                    } while (--best_len);
                }
                if (in_next >= max_block_end) break;
                if (cache_ptr >= cache + kNoCacheLen) break;
                if (!(st.num_new >= kNoObsPerCheck && in_next - block_begin >= kMinBlockLen && n - in_next >= kMinBlockLen)) continue;
                if (no_end_block_check(st, in_next - block_begin)) {
                    change_detected = true;
                    break;
                }
                no_merge_stats(L, st);
                prev_check = in_next;
            }
            uint32_t block_end, block_length;
            bool is_final;
            const NoMatch *block_cache_end;
            NoMatch *orig_cache_ptr = cache_ptr;
            const bool rewind = change_detected && prev_check != 0xFFFFFFFFu;
            if (rewind) {
                // a recent chunk differs from the rest of the block: the block ends just before it
                block_end = prev_check;
                uint32_t num_bytes_to_rewind = in_next - block_end;
                do {
                    cache_ptr--;
                    cache_ptr -= cache_ptr->length;
                } while (--num_bytes_to_rewind);
                is_final = false;
            } else {
                block_end = in_next;
                no_merge_stats(L, st);
                is_final = in_next == n;
            }
            block_length = block_end - block_begin;
            block_cache_end = cache_ptr;
            const uint32_t tok_begin = ntok;
            no_optimize_block(L, st, nodes, in + block_begin, block_length, block_cache_end, block_begin == 0, depth, num_passes,
                              compat_1_10, tok, ntok);
            if (nsub < cfg.max_sub) {
                sub[nsub].tok_begin = tok_begin;
                sub[nsub].tok_end = ntok;
                sub[nsub].byte_begin = block_begin;
                sub[nsub].byte_len = block_length;
                sub[nsub].is_final = is_final ? 1u : 0u;
            }
            nsub++;
            // deflate_near_optimal_save_stats
            for (uint32_t i = 0; i < kNoNumObs; i++) st.prev_obs[i] = st.obs[i];
            st.prev_num = st.num;
            if (rewind) {
                const size_t keep = (size_t)(orig_cache_ptr - cache_ptr);
                for (size_t i = 0; i < keep; i++) cache[i] = cache_ptr[i];  // (memmove towards lower addresses)
                cache_ptr = cache + keep;
                // deflate_near_optimal_clear_old_stats: only the chunk behind the block's end stays
                for (uint32_t i = 0; i < kNoNumObs; i++) st.obs[i] = 0;
                st.num = 0;
                for (uint32_t i = 0; i <= kNoMaxMatch; i++) L.mlf[i] = 0;
            } else {
                cache_ptr = cache;
                for (uint32_t i = 0; i < kNoNumObs; i++) st.new_obs[i] = st.obs[i] = 0;
                st.num_new = st.num = 0;
                for (uint32_t i = 0; i <= kNoMaxMatch; i++) L.new_mlf[i] = L.mlf[i] = 0;
            }
            block_begin = block_end;
        } while (in_next != n);
        meta->ntok = ntok;
        meta->nsub = nsub <= cfg.max_sub ? nsub : cfg.max_sub;
        if (nsub > cfg.max_sub) meta->status = kStatusInternal;  // (cannot happen: a block is at least 5000 bytes)
    }
}

// the three default-cost tables into every lane's state (host-computed once: gzpx_api.cpp)
__global__ void k_no_tables(NoLane *lanes, uint32_t n_lanes, const uint8_t *tables /* 3 x 258 */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lanes) return;
    for (uint32_t k = 0; k < 3; k++) {
        for (uint32_t j = 0; j <= 256; j++) lanes[i].default_lit[k][j] = tables[258 * k + j];
        lanes[i].default_len_sym[k] = tables[258 * k + 257];
    }
}

void launch_near_optimal_tables(void *lanes, uint32_t n_lanes, const uint8_t *d_tables, hipStream_t stream) {
    hipLaunchKernelGGL(k_no_tables, dim3((n_lanes + 63) / 64), dim3(64), 0, stream, (NoLane *)lanes, n_lanes, d_tables);
}

void launch_near_optimal(const Config &cfg, const uint8_t *slab, uint32_t nb, const Scratch &s, hipStream_t stream) {
    const uint32_t n_lanes = s.no_lanes < nb ? s.no_lanes : nb;
    // Blocks per wave: the lanes of a wave run different blocks, so the wave executes the union of their paths (tree
    // walks of different depths, extension loops of different lengths).  The chip has far more wave slots than a slab
    // has 64-block groups, so the blocks are spread over as many waves as there are slots -- 8,835 blocks = 9 per wave
    // instead of 64: 91 -> 194 MiB/s at level 10 (16 per wave 149-169, 5 per wave -- a second round of waves -- 126).
    const uint32_t slots = (cfg.n_cu ? cfg.n_cu : 256u) * 4u * GZPX_NO_WAVES;
    uint32_t per_wave = (n_lanes + slots - 1) / slots;
    if (per_wave < 1) per_wave = 1;
    if (per_wave > 64) per_wave = 64;
    hipLaunchKernelGGL(k_near_optimal, dim3((n_lanes + per_wave - 1) / per_wave), dim3(per_wave), 0, stream, cfg, slab, s.meta,
                       s.sub, s.tok, nb, (NoLane *)s.no_state, s.no_cache, s.no_nodes, no_nodes_bytes(cfg.block_size), n_lanes);
}

}  // namespace gzpx
