// Table building for mj_plan_create (host side, no device code): the canonical code book + 9-bit LUT of a DHT, the resolved
// AC tables of the lane walk (huffman_lanes13.hip / fused.hip) and the counting walks' tables of the synchronisation form
// (huffman_sync.hip: k_count).  jpeg_decoder.py:305-324, :366-377 (Huffman table form), :1636-1646 (EXTEND).
#include "plan.h"

namespace mj {

// DHT -> canonical code book + 9-bit LUT (jpeg_decoder.py:366-377)
void build_dev_huff(const mj_huff_spec &spec, mj::DevHuff &h) {
    memset(&h, 0, sizeof(h));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        code <<= 1;
        h.first_code[l] = code;
        h.count[l] = spec.bits[l - 1];
        h.first_sym[l] = k;
        for (int i = 0; i < spec.bits[l - 1] && k < 256; ++i, ++k, ++code) {
            h.vals[k] = spec.vals[k];
            if (l <= mj::kLutBits && code < (1 << l)) {
                int shift = mj::kLutBits - l;
                for (int f = 0; f < (1 << shift); ++f) {
                    int idx = (code << shift) | f;
                    if (h.lut[idx] == 0) h.lut[idx] = (uint16_t)((l << 8) | spec.vals[k]);   // first (shortest) key wins
                }
            }
        }
    }
}

// Resolved AC tables (huffman_lanes13.hip's entry format) for every table of the batch used as an AC table, table (LDS slot) s with
// ab_of_slot[s] index bits: a main level of 2^AB entries — the FINISHED symbol wherever code + value bits fit the index
// (jpeg_decoder.py:834-866 and bin_twos_complement :1636-1646 evaluated here), else what the arithmetic step needs — and second-level
// tables of 2^(16 - AB) entries for the prefixes of longer codes.  fixed_slot_bytes: the stride of a table in `out` (the stage-1
// kernel's), or 0: back to back, each as small as its codes allow (a fused launch's).  slot_off / total_bytes: where each lies.
// false: does not fit.
bool build_resolved_tables(const mj_batch *b, const std::vector<int> &role, uint64_t ac_pk, int n_ac, const int ab_of_slot[4], int fixed_slot_bytes,
                           std::vector<uint32_t> &out, int slot_off[4], int &total_bytes) {
    if (n_ac > 4) return false;
    int subs_of_slot[4] = {1, 1, 1, 1}, words_of_slot[4] = {0, 0, 0, 0};
    for (int t = 0; t < b->n_huff; ++t) {
        if (role[t] != 2) continue;
        const int slot = (int)((ac_pk >> (8 * t)) & 0xFF), AB = ab_of_slot[slot], AS = 1 << AB, SUB = 1 << (16 - AB);
        if (fixed_slot_bytes) {
            subs_of_slot[slot] = (fixed_slot_bytes / 4 - AS) / SUB;
            words_of_slot[slot] = fixed_slot_bytes / 4;
            continue;
        }
        std::vector<char> seen(AS, 0);
        int n = 1, code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            code <<= 1;
            for (int i = 0; i < b->huff[t].bits[l - 1] && k < 256; ++i, ++k, ++code) {
                if (code >= (1 << l) || l <= AB) continue;
                const int prefix = code >> (l - AB);
                if (!seen[prefix]) { seen[prefix] = 1; ++n; }
            }
        }
        subs_of_slot[slot] = n;
        words_of_slot[slot] = ((AS + n * SUB) * 4 + 15) / 16 * 4;
    }
    int at = 0;
    for (int sl = 0; sl < n_ac; ++sl) {
        if ((size_t)words_of_slot[sl] * 4 > 65535u) return false;        // (second-level tables are addressed by a 16-bit byte offset)
        slot_off[sl] = at * 4;
        at += words_of_slot[sl];
    }
    total_bytes = at * 4;
    out.assign((size_t)at, 0xFFFFFFFFu);
    for (int t = 0; t < b->n_huff; ++t) {
        if (role[t] != 2) continue;
        const int slot = (int)((ac_pk >> (8 * t)) & 0xFF), AB = ab_of_slot[slot], AS = 1 << AB, SUB = 1 << (16 - AB);
        const int SLOT = words_of_slot[slot], max_sub = subs_of_slot[slot];
        uint32_t *tab = out.data() + slot_off[slot] / 4;
        // second-level tables behind the main one: for the 16 - AB bits that follow an AB-bit prefix of longer codes;
        // table 0 = "no such code" (where every other unset main entry points as well)
        int n_sub = 1;
        for (int i = 0; i < SUB; ++i) tab[AS + i] = 0x8000u;
        int code = 0, k = 0;
        auto put = [&](uint32_t *base, uint32_t first, uint32_t count, uint32_t entry) {     // the shortest code wins (first fit)
            for (uint32_t f = 0; f < count; ++f)
                if (base[first + f] == 0xFFFFFFFFu) base[first + f] = entry;
        };
        for (int l = 1; l <= 16; ++l) {
            code <<= 1;
            for (int i = 0; i < b->huff[t].bits[l - 1] && k < 256; ++i, ++k, ++code) {
                if (code >= (1 << l)) continue;
                const int hv = b->huff[t].vals[k], run = hv >> 4, size = hv & 15;
                const uint32_t adv = hv == 0 ? 127u : 2u * (uint32_t)(run + 1);
                const uint32_t open_entry = ((uint32_t)(31 - size) << 24) | ((uint32_t)l << 16) | 0x8000u | ((hv == 0 ? 0u : (uint32_t)(run + 1)) << 8);   // value bits taken arithmetically
                if (l > AB) {
                    const uint32_t prefix = (uint32_t)code >> (l - AB);
                    uint32_t &m = tab[prefix];
                    if (m == 0xFFFFFFFFu) {                       // first long code under this prefix: a new table
                        if (n_sub >= max_sub) return false;
                        for (int j = 0; j < SUB; ++j) tab[AS + n_sub * SUB + j] = 0xFFFFFFFFu;
                        m = ((uint32_t)(AS * 4 + n_sub * SUB * 4) << 16) | 0xC000u;
                        ++n_sub;
                    }
                    if ((m & 0xC0FFu) != 0xC000u) continue;       // a shorter code owns the prefix (over-subscribed table)
                    uint32_t *sub = tab + ((m >> 16) / 4);
                    put(sub, ((uint32_t)code << (16 - l)) & (uint32_t)(SUB - 1), 1u << (16 - l), open_entry);
                } else if (hv == 0 ? l <= AB : l + size <= AB) {
                    const int n = hv == 0 ? 0 : size, rest = AB - l - n;
                    for (uint32_t vb = 0; vb < (1u << n); ++vb) {
                        // bin_twos_complement (:1636-1646): leading 1 = the value itself, leading 0 = value - (2^n - 1)
                        const int val = n == 0 ? 0 : ((vb >> (n - 1)) ? (int)vb : (int)vb - ((1 << n) - 1));
                        put(tab, (((uint32_t)code << n) | vb) << rest, 1u << rest,
                            ((uint32_t)(uint16_t)(int16_t)val << 16) | (adv << 8) | (uint32_t)(l + n));
                    }
                } else {
                    put(tab, (uint32_t)code << (AB - l), 1u << (AB - l), open_entry);
                }
            }
        }
        for (int i = 0; i < AS; ++i)
            if (tab[i] == 0xFFFFFFFFu) tab[i] = ((uint32_t)(AS * 4) << 16) | 0xC000u;          // no such code: the empty second-level table
        for (int i = AS; i < SLOT; ++i)
            if (tab[i] == 0xFFFFFFFFu) tab[i] = 0x8000u;
    }
    return true;
}

// Every table of a batch of at most 8 as the counting walks of the synchronisation form want it (huffman_sync.hip: k_count): all
// with W index bits, `tab_bytes` apart.  A 32-bit entry: bits consumed — code AND value — (0..5) | run + 1, 128 = end of block
// (8..15) | DC tables: the EXTENDed difference (16..30; jpeg_decoder.py:818-820, bin_twos_complement :1636-1646) — finished
// wherever the code fits the index (AC tables: counting does not look at AC values) or code + value bits do (DC tables).  Bit 31
// = not finished: bit 30 set = a code longer than the index, (0..15) the byte offset of the second-level table (2^(16 - W)
// entries for the bits behind the index) for its prefix; else the open form, which second-level tables hold throughout: code
// length (0..4; 0 = no such code) | run + 1 / 128 (8..15) | size (16..19).  false: does not fit (or a DC size above 15).
bool build_count_tables(const mj_batch *b, const std::vector<int> &role, int W, std::vector<uint32_t> &out, int &tab_bytes) {
    if (b->n_huff > 8) return false;
    const int AS = 1 << W, SUB = 1 << (16 - W);
    auto walk_codes = [&](const mj_huff_spec &spec, auto &&f) {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            code <<= 1;
            for (int i = 0; i < spec.bits[l - 1] && k < 256; ++i, ++k, ++code)
                if (code < (1 << l)) f(l, code, (int)spec.vals[k]);
        }
    };
    int max_words = AS + SUB;
    for (int t = 0; t < b->n_huff; ++t) {
        if (role[t] != 1 && role[t] != 2) return false;
        std::vector<char> seen((size_t)AS, 0);
        int n = 1;
        walk_codes(b->huff[t], [&](int l, int code, int) {
            if (l > W && !seen[(size_t)(code >> (l - W))]) { seen[(size_t)(code >> (l - W))] = 1; ++n; }
        });
        max_words = std::max(max_words, AS + n * SUB);
    }
    tab_bytes = (max_words * 4 + 15) / 16 * 16;
    if (tab_bytes > 65535) return false;
    const int TW = tab_bytes / 4;
    out.assign((size_t)b->n_huff * TW, 0xFFFFFFFFu);
    for (int t = 0; t < b->n_huff; ++t) {
        uint32_t *tab = out.data() + (size_t)t * TW;
        const bool is_dc = role[t] == 1;
        int n_sub = 1;                                            // table 0 = "no such code"
        for (int i = 0; i < SUB; ++i) tab[AS + i] = 0x80000000u;
        bool ok = true;
        auto put = [&](uint32_t *base, uint32_t first, uint32_t count, uint32_t entry) {     // the shortest code wins (first fit)
            for (uint32_t f = 0; f < count; ++f)
                if (base[first + f] == 0xFFFFFFFFu) base[first + f] = entry;
        };
        walk_codes(b->huff[t], [&](int l, int code, int hv) {
            const int run = is_dc ? 0 : hv >> 4, size = is_dc ? hv : (hv & 15);
            if (size > 15) { ok = false; return; }
            const bool eob = !is_dc && hv == 0;
            const uint32_t adv = eob ? 128u : (uint32_t)(run + 1);
            const uint32_t open_entry = 0x80000000u | ((uint32_t)size << 16) | (adv << 8) | (uint32_t)l;
            if (l > W) {
                uint32_t &m = tab[(uint32_t)code >> (l - W)];
                if (m == 0xFFFFFFFFu) {                           // first long code under this prefix: a new table
                    for (int j = 0; j < SUB; ++j) tab[AS + n_sub * SUB + j] = 0xFFFFFFFFu;
                    m = 0xC0000000u | (uint32_t)((AS + n_sub * SUB) * 4);
                    ++n_sub;
                }
                if ((m & 0xC0000000u) != 0xC0000000u) return;     // a shorter code owns the prefix (over-subscribed table)
                put(tab + (m & 0xFFFFu) / 4, ((uint32_t)code << (16 - l)) & (uint32_t)(SUB - 1), 1u << (16 - l), open_entry);
            } else if (!is_dc) {
                put(tab, (uint32_t)code << (W - l), 1u << (W - l), (adv << 8) | (uint32_t)(l + size));
            } else if (l + size <= W) {
                const int rest = W - l - size;
                for (uint32_t vb = 0; vb < (1u << size); ++vb) {
                    const int val = size == 0 ? 0 : ((vb >> (size - 1)) ? (int)vb : (int)vb - ((1 << size) - 1));
                    put(tab, (((uint32_t)code << size) | vb) << rest, 1u << rest, (((uint32_t)val & 0x7FFFu) << 16) | (adv << 8) | (uint32_t)(l + size));
                }
            } else {
                put(tab, (uint32_t)code << (W - l), 1u << (W - l), open_entry);
            }
        });
        if (!ok) return false;
        for (int i = 0; i < AS; ++i)
            if (tab[i] == 0xFFFFFFFFu) tab[i] = 0x80000000u;       // no such code
        for (int i = AS; i < TW; ++i)
            if (tab[i] == 0xFFFFFFFFu) tab[i] = 0x80000000u;
    }
    return true;
}

}  // namespace mj

extern "C" {

int mj_debug_count_tables(const mj_huff_spec *huff, int32_t n_huff, const int32_t *roles, int32_t wbits, uint32_t *out, int64_t cap_words,
                          int32_t *tab_bytes) {
    if (!huff || !roles || !tab_bytes || n_huff < 1 || n_huff > 8 || wbits < 10 || wbits > 13) return MJ_ERR_INVALID;
    mj_batch b{};
    b.n_huff = n_huff; b.huff = huff;
    std::vector<int> role(roles, roles + n_huff);
    std::vector<uint32_t> t;
    int tb = 0;
    if (!mj::build_count_tables(&b, role, wbits, t, tb)) return MJ_ERR_UNSUPPORTED;
    *tab_bytes = tb;
    if (out) {
        if ((int64_t)t.size() > cap_words) return MJ_ERR_INVALID;
        memcpy(out, t.data(), t.size() * 4);
    }
    return MJ_OK;
}

}  // extern "C"
