// Wave-uniform bit reader and Huffman symbol decode shared by the wave-per-segment stage-1 kernels
// (baseline: huffman.hip, progressive: progressive.hip).  See huffman.hip for the design notes.
#pragma once
#include "mijpeg_internal.h"

namespace mj {
namespace wavebits {

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct BitReader {
    // all members are wave-uniform
    const uint32_t *words;   // 4-byte aligned start of the stream
    uint32_t cur, nxt;       // per-lane: dwords [cd0 + lane] and [cd0 + 63 + lane]
    int cd0;                 // dword index of lane 0 of `cur`
    int pos;                 // next byte to load, relative to `words`
    int end;                 // one past the last byte of the segment, relative to `words`
    uint64_t bb;             // bit buffer, next bit = bit 63
    int bc;                  // valid bits in bb
    int pad;                 // bits of zero padding appended past `end`
    int lane;

    __device__ __forceinline__ void init(const uint8_t *blob, int64_t begin, int len, int lane_) {
        lane = lane_;
        int64_t abase = begin & ~(int64_t)3;
        words = reinterpret_cast<const uint32_t *>(blob + abase);
        pos = (int)(begin - abase);
        end = pos + len;
        cd0 = 0;
        cur = words[lane];
        nxt = words[63 + lane];
        bb = 0;
        bc = 0;
        pad = 0;
    }

    // continue where an earlier launch stopped (after init() with the same segment)
    __device__ __forceinline__ void restore(int pos_, uint64_t bb_, int bc_, int pad_) {
        pos = pos_; bb = bb_; bc = bc_; pad = pad_;
        cd0 = pos_ >> 2;
        cur = words[cd0 + lane];
        nxt = words[cd0 + 63 + lane];
    }

    // dwords [i], [i+1] of the stream around byte `p`, shifted so that byte p is the low byte
    __device__ __forceinline__ uint32_t peek_raw(int p) {
        int i = (p >> 2) - cd0;
        if (i >= 63) {            // uniform branch: step to the next 252-byte window
            cd0 += 63;
            cur = nxt;
            nxt = words[cd0 + 63 + lane];
            i -= 63;
        }
        uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)cur, i);
        uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)cur, i + 1);
        uint64_t w = ((uint64_t)hi << 32) | lo;
        return (uint32_t)(w >> ((p & 3) * 8));
    }

    // make at least 32 bits available (zero-padded past the end of the segment)
    __device__ __forceinline__ void refill() {
        while (bc <= 32) {
            if (pos + 4 <= end) {
                uint32_t w = peek_raw(pos);
                // any byte == 0xFF ?  (zero byte in ~w)
                uint32_t nw = ~w;
                if (((nw - 0x01010101u) & ~nw & 0x80808080u) == 0) {
                    uint32_t be = __builtin_bswap32(w);
                    bb |= (uint64_t)be << (32 - bc);
                    bc += 32;
                    pos += 4;
                    continue;
                }
                uint32_t b = w & 0xFFu;                 // byte-wise, reference semantics (:673-677)
                pos += (b == 0xFFu) ? 2 : 1;
                bb |= (uint64_t)b << (56 - bc);
                bc += 8;
            } else if (pos < end) {
                uint32_t b = peek_raw(pos) & 0xFFu;
                pos += (b == 0xFFu) ? 2 : 1;
                bb |= (uint64_t)b << (56 - bc);
                bc += 8;
            } else {
                pad += 8;                                // past the end: feed zeros, remember how many
                bc += 8;
            }
        }
    }

    __device__ __forceinline__ uint32_t peek16() const { return (uint32_t)(bb >> 48); }
    __device__ __forceinline__ void skip(int n) { bb <<= n; bc -= n; }
    __device__ __forceinline__ uint32_t take(int n) {   // 1 <= n <= 16
        uint32_t v = (uint32_t)(bb >> (64 - n));
        bb <<= n;
        bc -= n;
        return v;
    }
};

// next_huffval (:712-722).  Returns the symbol, or -1 if no code matches within 16 bits.
__device__ __forceinline__ int decode_symbol(BitReader &br, const uint16_t *lut, const DevHuff *tab) {
    uint32_t p16 = br.peek16();
    int e = rfl((int)lut[p16 >> (16 - kLutBits)]);
    int len = e >> 8;
    int sym = e & 0xFF;
    if (len == 0) {
        sym = -1;
        for (int l = kLutBits + 1; l <= 16; ++l) {
            int d = (int)(p16 >> (16 - l)) - tab->first_code[l];
            if (d >= 0 && d < tab->count[l]) {
                sym = tab->vals[tab->first_sym[l] + d];
                len = l;
                break;
            }
        }
        if (sym < 0) return -1;
    }
    br.skip(len);
    return sym;
}

// bin_twos_complement (:1636-1646)
__device__ __forceinline__ int extend(uint32_t raw, int n) {
    return (raw >> (n - 1)) ? (int)raw : (int)raw - ((1 << n) - 1);
}


}  // namespace wavebits
}  // namespace mj
