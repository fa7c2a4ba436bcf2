// Stage 1 — baseline Huffman entropy decode on gfx950: one wavefront per restart segment.
//
// Replaces the entropy part of JpegDecoder.baseline_dct_scan (jpeg_decoder.py:734-866, :894-900) with its
// helpers get_bits (:654-695), next_huffval (:712-722) and bin_twos_complement (:1636-1646).
//
// Mapping to the machine
//   * A restart segment is the largest unit with no serial dependence on its neighbours (DC predictors
//     reset, bit reader re-aligned, :898-900), so each wavefront owns one segment and walks it serially.
//     The walk is wave-uniform: the bit buffer, positions, run/size arithmetic all live in SGPRs and
//     issue on the scalar unit; 8 waves per SIMD hide the dependent-lookup latency of each other.
//   * The 64 lanes are used for what is parallel:
//       - the bitstream is fetched 256 B at a time, one dword per lane, and the serial walker pulls
//         dwords out of that register with v_readlane (no LDS or memory access on the critical path);
//         the next 256 B are already in flight while the current ones are consumed;
//       - lane l owns zig-zag coefficient l of the current block: a decoded value is "scattered" with a
//         single compare+select, and the finished block leaves as one 128-B line (lanes permuted to the
//         natural [v][u] order stage 2 wants).
//   * Huffman tables: the 9-bit primary LUT of every table the image uses is copied into LDS once per
//     wave (1 KiB each); codes longer than 9 bits (rare) take a canonical search through L2-resident
//     first_code/count arrays with scalar loads.
//   * Byte stuffing follows the reference literally: whatever follows a 0xFF byte is skipped (:676-677).
//     Four bytes are consumed per refill when none of them is 0xFF, else the refill goes byte-wise.
//   * Restart handling is count-driven in the reference (:667-669); here the host has located the RSTn
//     markers, and a segment whose MCUs do not end exactly at its marker is reported (MJ_ST_DESYNC /
//     MJ_ST_OVERRUN) instead of silently decoding garbage.
#include "mijpeg_internal.h"
#include "wave_bits.h"

namespace mj {

namespace {

// zig-zag index -> position inside the stored block.  Blocks are kept in HBM as [v][u] (row = vertical
// frequency, the usual "natural order"), which is what stage 2's 8-lane groups load 16 bytes at a time.
__constant__ uint8_t c_nat_of_zz[64] = {
    0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
   12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
   58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

using namespace wavebits;

}  // namespace

__global__ __launch_bounds__(256) void k_huffman(const uint8_t *__restrict__ blob,
                                                 const DevSegment *__restrict__ segs, int64_t n_segs,
                                                 const DevImage *__restrict__ images,
                                                 const DevHuff *__restrict__ huff, int16_t *__restrict__ coef,
                                                 int32_t *__restrict__ status, int lut_slots, int tr) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s_lut[];   // [4 waves][lut_slots][kLutSize]
    const int lane = threadIdx.x & 63;
    const int wave = rfl((int)(threadIdx.x >> 6));
    const int64_t seg_id = (int64_t)blockIdx.x * 4 + wave;
    if (seg_id >= n_segs) return;

    const DevSegment *sg = segs + seg_id;
    const int image = sg->image;
    const DevImage *im = images + image;
    const int n_tabs = im->n_tabs;
    uint16_t *my_lut = s_lut + (size_t)wave * lut_slots * kLutSize;

    // stage this image's primary LUTs: 1 KiB each = one 16-byte load per lane
    for (int t = 0; t < n_tabs; ++t) {
        const uint4 *src = reinterpret_cast<const uint4 *>(huff[im->tab_index[t]].lut);
        reinterpret_cast<uint4 *>(my_lut + t * kLutSize)[lane] = src[lane];
    }

    BitReader br;
    br.init(blob, sg->begin, sg->len, lane);

    const int bpm = im->blocks_per_mcu;
    const int n_mcu = sg->n_mcu;
    // per-block tables packed into scalars: 8 blocks x 8 bits each, twice (unusual sampling layouts have up to 16 blocks per MCU)
    const uint64_t comp_pk0 = reinterpret_cast<const uint64_t *>(im->blk_comp)[0], comp_pk1 = reinterpret_cast<const uint64_t *>(im->blk_comp)[1];
    const uint64_t dc_pk0 = reinterpret_cast<const uint64_t *>(im->blk_dc_slot)[0], dc_pk1 = reinterpret_cast<const uint64_t *>(im->blk_dc_slot)[1];
    const uint64_t ac_pk0 = reinterpret_cast<const uint64_t *>(im->blk_ac_slot)[0], ac_pk1 = reinterpret_cast<const uint64_t *>(im->blk_ac_slot)[1];

    // tr: the plan keeps blocks transposed ([u][v]) for the row-major stage 2
    const int nat0 = c_nat_of_zz[lane], nat = tr ? ((nat0 & 7) << 3 | nat0 >> 3) : nat0;
    int16_t *out = coef + (im->block_off + (int64_t)sg->mcu0 * bpm) * 64 + nat;
    int pred0 = 0, pred1 = 0, pred2 = 0;   // previous_dc (:735), int16 arithmetic
    int err = 0;

    for (int m = 0; m < n_mcu; ++m) {
        for (int b = 0; b < bpm; ++b) {
            const int sh8 = 8 * (b & 7);
            const int comp = (int)(((b < 8 ? comp_pk0 : comp_pk1) >> sh8) & 0xFF);
            const int dslot = (int)(((b < 8 ? dc_pk0 : dc_pk1) >> sh8) & 0xFF);
            const int aslot = (int)(((b < 8 ? ac_pk0 : ac_pk1) >> sh8) & 0xFF);
            const uint16_t *dc_lut = my_lut + dslot * kLutSize;
            const uint16_t *ac_lut = my_lut + aslot * kLutSize;
            const DevHuff *dc_tab = huff + im->tab_index[dslot];
            const DevHuff *ac_tab = huff + im->tab_index[aslot];
            int v = 0;   // this lane's coefficient of the block (lane = zig-zag index)

            // ---- DC (:810-820)
            br.refill();
            int s = decode_symbol(br, dc_lut, dc_tab);
            if (s < 0 || s > 16) { err = MJ_ST_BAD_CODE; s = 0; }
            int diff = 0;
            if (s > 0) diff = extend(br.take(s), s);
            int pred = comp == 0 ? pred0 : (comp == 1 ? pred1 : pred2);
            int dcv = (int)(int16_t)(diff + pred);
            if (comp == 0) pred0 = dcv; else if (comp == 1) pred1 = dcv; else pred2 = dcv;
            if (lane == 0) v = dcv;

            // ---- AC (:833-866)
            int k = 1;
            while (k < 64) {
                br.refill();
                int hv = decode_symbol(br, ac_lut, ac_tab);
                if (hv < 0) { err = MJ_ST_BAD_CODE; break; }
                if (hv == 0) break;                 // EOB
                k += hv >> 4;
                if (k >= 64) break;
                int n = hv & 15;
                if (n > 0) {
                    int val = extend(br.take(n), n);
                    if (lane == k) v = val;
                }
                ++k;
            }
            *out = (int16_t)v;
            out += 64;
            if (err) break;
        }
        if (err) break;
    }

    if (!err) {
        if (br.pad > 0 && br.bc < br.pad) err = MJ_ST_OVERRUN;             // consumed bits that are not there
        else if (!sg->last && (((br.bc - br.pad) >> 3) > 0 || br.pos < br.end)) err = MJ_ST_DESYNC;
    }
    if (err && lane == 0) atomicMax(status + image, err);
}

hipError_t launch_huffman(hipStream_t stream, const uint8_t *blob, const DevSegment *segs, int64_t n_segs,
                          const DevImage *images, const DevHuff *huff, int16_t *coef, int32_t *status,
                          int lut_slots, int transposed) {
    if (n_segs == 0) return hipSuccess;
    const int64_t blocks = (n_segs + 3) / 4;
    const size_t lds = (size_t)4 * lut_slots * kLutSize * sizeof(uint16_t);
    hipLaunchKernelGGL(k_huffman, dim3((unsigned)blocks), dim3(256), lds, stream, blob, segs, n_segs, images, huff,
                       coef, status, lut_slots, transposed);
    return hipGetLastError();
}

}  // namespace mj
