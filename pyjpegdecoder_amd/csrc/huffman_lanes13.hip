// Stage 1, lane-parallel form with RESOLVED 13-bit tables — baseline Huffman entropy decode, one restart segment per lane.
//
// Same contract and output as huffman_lanes.hip (the 11-bit form: its header describes the lock-step at block granularity and
// the LDS block rows, both kept).  What changes is everything on a wave's serial path, the only thing that matters here:
// the launch lasts as long as one segment's walk.
//   * The AC table of a component is a 13-bit LUT of 32-bit entries that carry the FINISHED symbol whenever code and
//     value bits together fit the index (99.6 % of the symbols of the benchmark's files; jpeg_decoder.py:834-866 and
//     bin_twos_complement :1636-1646 are evaluated when the table is built, api.hip):
//         byte 0      bits consumed (code + value bits)
//         byte 1      how far the write position moves: 2 * (run + 1) bytes; end of block = 127 (odd, and past the row
//                     wherever the position is)
//         bits 31:16  the coefficient, EXTENDed, ready for ds_write_b16_d16_hi
//     Not resolved (bit 15): byte 0 = 0 — nothing is consumed and the position leaves the row by 128 or more —, byte 1 =
//     0x80 | run + 1 (0 = end of block), byte 2 = code length, byte 3 = 31 - size for a code of <= 13 bits whose value bits
//     do not fit; byte 1 = 0xC0 and the high word = where the 8-entry second-level table of its 13-bit prefix starts, for
//     codes of 14..16 bits (entries of the same kind; code length 0 = no such code: the canonical search,
//     jpeg_decoder.py:366-377 semantics, reports it).
//   * The symbol step is hand-written (inline asm, the 64-bit bit buffer in fixed registers): LUT address from the top of
//     the buffer (v_bfe, v_lshl_add), ds_read_b32, position += byte 1 (SDWA), buffer <<= byte 0 (v_lshlrev_b64 reads its 6
//     low bits of the entry as it is), count -= byte 0 (SDWA), one v_cmpx that keeps the lanes still inside their block —
//     the lanes that store this symbol —, and the store straight from the entry's high half, issued behind the NEXT symbol's
//     LUT read.  Finished lanes are simply off (exec): nothing is selected or clamped.  10 instructions per symbol against
//     44 in the 11-bit form.  What the step does not test is put right later: a symbol that lands on coefficient 63 is
//     stored after the loop (its lane left before the store, the entry is still in its register); an entry that overshoots
//     the block has consumed value bits the reference leaves unread (:849, :855-856) — found after the block by the parity
//     of the final position, damaged files only; entries that are not resolved are looked for once per turn of the loop
//     (four refills, eight symbols) on every lane at once, and their lanes take an arithmetic step.
//   * The stream is read through a per-lane window in LDS, topped up 16 or 32 bytes per turn one turn ahead: a refill of the
//     bit buffer is a ds_read_b32, no vector-memory wait sits in the loop except the window's own, a turn old.
//   * The flush moves 16 bytes per lane (8 blocks per store instruction), by hand for full rounds.
// Used when the batch's distinct tables fit LDS in this format (<= 3 AC tables of 37 KiB + <= 4 DC tables of 4 KiB: every
// batch of files with the standard tables); other batches keep the 11-bit form with its per-workgroup table lists.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "mijpeg_internal.h"
#include "lanes13_walk.h"

namespace mj {

using namespace lanes13;

#ifdef MJ_DIAGNOSTIC     // when does every wave of the launch finish?  (100 MHz wall clock; diagnostic build only)
void dbg_lanes13_waves_report() {
    std::vector<unsigned long long> t(4096 * 3), z(4096 * 3, 0);
    (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_dbg13_waves), t.size() * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg13_waves), z.data(), z.size() * 8);
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < t.size(); i += 3) if (t[i] && t[i] < t0) t0 = t[i];
    std::vector<double> en, by_slot[16];
    for (size_t i = 0; i < t.size(); i += 3) if (t[i]) { const double e = (double)(t[i + 1] - t0) * 0.01; en.push_back(e); by_slot[(i / 3) % 16].push_back(e); }
    if (en.empty()) return;
    std::sort(en.begin(), en.end());
    auto q = [&](double f) { return en[(size_t)(f * (en.size() - 1))]; };
    fprintf(stderr, "[diag lanes13] %zu waves; end us: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f; mean by wave of the workgroup:", en.size(), q(0), q(0.1), q(0.5), q(0.9), q(1));
    for (int w = 0; w < 16; ++w) if (!by_slot[w].empty()) { double m = 0; for (double x : by_slot[w]) m += x; fprintf(stderr, " %.0f", m / by_slot[w].size()); }
    fprintf(stderr, "\n");
}
#endif
// (the walk is lanes13_walk.h: stage() = tables into LDS, walk() = one wavefront's segments)
__global__ __launch_bounds__(1024) void k_huffman_lanes13(lanes13::Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = (int)(blockDim.x >> 6);
    lanes13::stage<false>(A, smem, tid, (int)blockDim.x, nw);
    __syncthreads();
    lanes13::walk<0>(A, smem, lane, wave, nw, (int)blockIdx.x, (int)gridDim.x, 0u);
}

// LDS bytes of a launch with `nw` waves of `lpw` lanes
static size_t lds13(int n_ac, int n_dc, int nw, int lpw, int kRing = 128) { return lanes13::lds_bytes(n_ac * kLanes13SlotBytes, n_dc, nw, lpw, kRing); }

bool lanes13_fits(int n_ac, int n_dc) { return n_ac >= 1 && n_ac <= 3 && n_dc >= 1 && n_dc <= 4 && lds13(n_ac, n_dc, 4, 8, 64) <= 160 * 1024; }

hipError_t launch_huffman_lanes13(hipStream_t stream, const uint32_t *dstream, const int32_t *seg_bits, const DevSegment *segs, int64_t n_segs,
                                  const DevImage *images, const DevHuff *huff, const uint16_t *lut11, const uint32_t *lut13,
                                  int n_ac, int n_dc, uint64_t ac_slot_pk, uint64_t dc_slot_pk, uint64_t dc_tab_pk,
                                  int16_t *coef, int32_t *status, int transposed, const DevVSeg *vsegs, const int32_t *by_length, int order_mode) {
    if (n_segs == 0) return hipSuccess;
    const int cus = device_cus();
    // tuning switches (mj_set_option; tools/stage_probe.py): waves per workgroup, lanes per wave
    int env_nw = -1, env_lpw = -1;
    if (const char *e = opt("MJ_LANES_WAVES")) { env_nw = atoi(e); if (env_nw < 1 || env_nw > 16) env_nw = -1; }
    if (const char *e = opt("MJ_LANES_PER_WAVE")) { env_lpw = atoi(e); if (env_lpw < 1 || env_lpw > 64) env_lpw = -1; }
    // One workgroup per CU (the tables take most of its LDS), all workgroups resident at once and equally loaded: the
    // segments a CU gets are spread over `nw` waves (the kernel is bound by instruction issue and LDS latency: about three
    // waves per SIMD keep a SIMD busy, more lanes per wave cost lock-step waiting).
    const int64_t per_cu = (n_segs + cus - 1) / cus;
    // (segments of very different lengths — dealt out striped, order_mode 2 — get two waves per SIMD instead of three: the launch
    // lasts as long as its longest segment's chain, which runs faster with fewer waves beside it; 6.27 -> 6.04 ms on bench.py's
    // mixed content, where segments of one kind lose 2 % with eight)
    const int nw_cap = order_mode == 2 ? 8 : 12;
    int nw = env_nw > 0 ? env_nw : (int)std::min<int64_t>(nw_cap, std::max<int64_t>(1, (per_cu + 15) / 16));
    // 128 bytes of stream window per lane when every segment then has its lane at once, else 64 (a third more lanes per CU:
    // a second round of workgroups would take as long again as the first)
    int ring = 128, fit = 0, lpw = 0;
    int64_t blocks = 0;
    for (;; ring = 64) {
        auto fit_of = [&](int w) { int f = 64; while (f > 1 && lds13(n_ac, n_dc, w, f, ring) > 160 * 1024) --f; return f; };
        fit = fit_of(nw);
        const int64_t rounds = (per_cu + (int64_t)nw * fit - 1) / ((int64_t)nw * fit);
        const int64_t per_wg = (per_cu + rounds - 1) / rounds;
        lpw = env_lpw > 0 ? std::min(env_lpw, fit) : (int)std::min<int64_t>(fit, std::max<int64_t>(1, (per_wg + nw - 1) / nw));
        blocks = (n_segs + (int64_t)nw * lpw - 1) / ((int64_t)nw * lpw);
        if (rounds == 1 || ring == 64) break;
    }
    if (const char *e = opt("MJ_LANES_RING")) { const int v = atoi(e); if ((v == 64 || v == 128) && lds13(n_ac, n_dc, nw, lpw, v) <= 160 * 1024) ring = v; }
    const size_t lds = lds13(n_ac, n_dc, nw, lpw, ring);
    static OncePerDevice attr_once;
    attr_once.run([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_huffman_lanes13), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    lanes13::Args A{dstream, seg_bits, segs, n_segs, images, huff, lut11, lut13, n_ac, n_dc, ac_slot_pk, dc_slot_pk, dc_tab_pk,
                    coef, status, lpw, transposed, vsegs, by_length, order_mode, ring, n_ac * kLanes13SlotBytes, kLaneLutBits, {0, 0, 0, 0}, {13, 13, 13, 13}, 0, nullptr};
    hipLaunchKernelGGL(k_huffman_lanes13, dim3((unsigned)blocks), dim3(64 * nw), lds, stream, A);
#ifdef MJ_X_STAMP
    if (getenv("MJ_X_REPORT")) {
        (void)hipStreamSynchronize(stream);
        unsigned long long h[16], z[16] = {0};
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg13), sizeof(h));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg13), z, sizeof(z));
        const double w = (double)h[5], turns = (double)(h[1] ? h[1] : 1);
        fprintf(stderr, "[lanes13 nw=%d lpw=%d] waves %llu: per wave: total %.0f cyc, DC etc. %.0f, AC loop %.0f of which in the asm %.0f, flush+clear %.0f; %.0f turns of the loop in %.0f asm runs; "
                        "per turn: loop back %.0f, look for open entries + window %.0f, refill %.0f, look1 %.0f, core1+look2 %.0f, core2+write %.0f, second round %.0f; of the window: up to the wait %.0f, the wait %.0f\n",
                nw, lpw, h[5], h[4] / w, ((double)h[4] - h[2] - h[3]) / w, h[2] / w, h[6] / w, h[3] / w, h[1] / w, h[8] / w,
                h[9] / turns, h[10] / turns, h[11] / turns, h[12] / turns, h[13] / turns, h[14] / turns, h[15] / turns, h[0] / turns, h[7] / turns);
    }
#endif
    return hipGetLastError();
}

}  // namespace mj
