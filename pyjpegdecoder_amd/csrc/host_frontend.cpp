// Host front end of libmijpeg.so: header parse + batch assembly on host threads (no GPU work, no HIP calls).
//
// What the Python host code does per file — the marker loop of the reference's constructor (jpeg_decoder.py:78-110)
// with its handlers for SOF0 (:112-247), DHT (:249-390), DQT (:392-472), DRI (:474-478) and SOS (:505-650), stopped
// at the SOS (`_parse.parse_jpeg(headers_only=True)`) — and what `batch.prepare_batch` then does per batch, for the
// files a decode service sees all day: baseline, 8 bit, 1 or 3 components, one scan that names every component.
// 512 x 1080p files cost the Python path ~60 ms, most of a decode call whose GPU part is 10 ms; here every file
// is an independent job for a thread and the batch-wide table numbering is one short serial pass.
//
// Anything else — progressive, several scans, DNL, a marker this loop does not know, truncated or inconsistent
// headers — is DECLINED, not diagnosed: the caller then runs the full Python marker loop over the batch, which
// raises exactly what the reference raises.  Declining is always safe; accepting is only done when the result is
// what `_parse.py` + `batch.py` would have produced (tests/test_host_frontend.py compares the arrays byte for byte).
// With mj_host_job.skip the declined files are marked and left out instead of ending the call: a service's batch with a
// few progressive files in it keeps the front end for the rest.
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/mijpeg.h"

namespace {

struct TableRef {                 // one DHT table as it lies in the file
    const uint8_t *bits = nullptr;    // 16 counts
    const uint8_t *vals = nullptr;    // the values, contiguous
    int n_vals = 0;
};

struct FileHeader {
    int32_t width = 0, height = 0, ncomp = 0;
    int32_t hs[3] = {0, 0, 0}, vs[3] = {0, 0, 0};
    int32_t restart_interval = 0;
    int64_t entropy_start = 0;
    const uint8_t *qt[3] = {nullptr, nullptr, nullptr};      // 64 bytes each, zig-zag (file) order
    TableRef dc[3], ac[3];
};

inline int be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

// The marker loop (:78-110) up to the first SOS.  false = declined.
bool parse_headers(const uint8_t *raw, int64_t n, FileHeader &o) {
    if (n < 3 || raw[0] != 0xFF || raw[1] != 0xD8 || raw[2] != 0xFF) return false;      // NotJpeg (:45-47)
    TableRef huff[256];               // by destination byte: Tc << 4 | Th (:305)
    const uint8_t *qt[256] = {};      // by destination byte (:442: the whole byte is the key)
    bool have_sof = false;
    int comp_id[3] = {-1, -1, -1}, comp_qt[3] = {0, 0, 0};
    int64_t pos = 2;
    for (;;) {
        if (pos + 1 >= n) return false;
        if (raw[pos] != 0xFF) { ++pos; continue; }                     // bytes between segments are skipped (:84-86)
        const int m = raw[pos + 1];
        pos += 2;
        if (m == 0x00 || (m >= 0xD0 && m <= 0xD7)) continue;           // (:90-91)
        if (pos + 2 > n) return false;
        const int64_t size = be16(raw + pos) - 2;                      // (:94)
        pos += 2;
        if (m == 0xD9 || size < 0 || pos + size > n) return false;     // EOI before any scan / short segment: not for this loop
        const uint8_t *data = raw + pos;
        if (m == 0xC0) {                                               // start_of_frame (:112-247)
            if (have_sof || size < 6 || data[0] != 8) return false;
            o.height = be16(data + 1);
            o.width = be16(data + 3);
            o.ncomp = data[5];
            if (o.width == 0 || o.height == 0) return false;           // zero height = DNL (:575-581): host path
            if ((o.ncomp != 1 && o.ncomp != 3) || size < 6 + 3 * o.ncomp) return false;
            for (int c = 0; c < o.ncomp; ++c) {
                comp_id[c] = data[6 + 3 * c];
                o.hs[c] = data[7 + 3 * c] >> 4;
                o.vs[c] = data[7 + 3 * c] & 15;
                comp_qt[c] = data[8 + 3 * c];
                if (o.hs[c] == 0 || o.vs[c] == 0) return false;
                for (int d = 0; d < c; ++d)
                    if (comp_id[d] == comp_id[c]) return false;
            }
            have_sof = true;
            pos += size;
        } else if (m == 0xC4) {                                        // define_huffman_table (:249-390)
            int64_t q = 0;
            while (q < size) {
                if (q + 17 > size) return false;
                TableRef t;
                t.bits = data + q + 1;
                int total = 0;
                long kraft = 0;                                        // codes used so far, in units of 2^-16
                for (int l = 0; l < 16; ++l) { total += t.bits[l]; kraft += (long)t.bits[l] << (15 - l); }
                // (an over-subscribed table: the Python parser raises CorruptedJpeg for it — declined here, so that both
                // front ends give such a file the same answer)
                if (total > 256 || kraft > (1L << 16) || q + 17 + total > size) return false;
                t.vals = data + q + 17;
                t.n_vals = total;
                huff[data[q]] = t;
                q += 17 + total;
            }
            pos += size;
        } else if (m == 0xDB) {                                        // define_quantization_table (:392-472)
            int64_t q = 0;
            while (q < size) {
                if (q + 65 > size) return false;
                qt[data[q]] = data + q + 1;
                q += 65;
            }
            pos += size;
        } else if (m == 0xDD) {                                        // define_restart_interval (:474-478)
            if (size < 2) return false;
            o.restart_interval = be16(data);
            pos += 2;                                                  // the reference advances by 2, not by the length
        } else if (m == 0xDA) {                                        // start_of_scan (:505-650)
            if (!have_sof || size < 1) return false;
            const int ns = data[0];
            if (ns != o.ncomp || size < 1 + 2 * ns) return false;      // a scan of some of the components: host path
            for (int c = 0; c < ns; ++c) {
                if (data[1 + 2 * c] != comp_id[c]) return false;       // frame order only
                const int t = data[2 + 2 * c];
                o.dc[c] = huff[t >> 4];                                // (:543-544)
                o.ac[c] = huff[(t & 15) | 0x10];
                o.qt[c] = qt[comp_qt[c]];
                if (!o.dc[c].bits || !o.ac[c].bits || !o.qt[c]) return false;
            }
            o.entropy_start = pos + size;                              // (:572)
            return true;
        } else if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
            pos += size;                                               // APPn / COM: skipped (:104-106)
        } else {
            return false;
        }
    }
}

std::string huff_key(const TableRef &t) {
    std::string k(272, '\0');
    memcpy(&k[0], t.bits, 16);
    memcpy(&k[16], t.vals, (size_t)t.n_vals);
    return k;
}

}  // namespace

extern "C" int mj_host_assemble(mj_host_job *job) {
    if (!job || job->n_files < 1 || !job->files || !job->sizes || !job->file_off || !job->blob || !job->images ||
        !job->seg_begin || !job->seg_end || !job->huff || !job->qt || job->huff_cap < 1 || job->qt_cap < 1)
        return MJ_ERR_INVALID;
    const int n = job->n_files;
    job->n_huff = job->n_qt = 0;
    job->declined_file = -1;
    job->n_accepted = 0;
    uint8_t *skip = job->skip;              // non-null: declined files are marked and left out, the rest is assembled
    if (skip) memset(skip, 0, (size_t)n);
    for (int i = 0; i < n; ++i) {
        const int64_t next = i + 1 < n ? job->file_off[i + 1] : job->blob_len;
        if (job->sizes[i] < 0 || job->file_off[i] < 0 || (job->file_off[i] & 3) || job->file_off[i] + job->sizes[i] > next)
            return MJ_ERR_INVALID;
    }

    std::vector<FileHeader> hdr((size_t)n);
    std::atomic<int> next_file{0};
    std::atomic<int> declined{n};           // lowest declined index
    auto worker = [&]() {
        for (;;) {
            const int i = next_file.fetch_add(1);
            if (i >= n || (!skip && declined.load(std::memory_order_relaxed) < n)) return;
            const uint8_t *raw = job->files[i];
            const int64_t sz = job->sizes[i];
            if (!raw || !parse_headers(raw, sz, hdr[(size_t)i])) {
                int cur = declined.load();
                while (i < cur && !declined.compare_exchange_weak(cur, i)) {}
                if (!skip) return;
                skip[i] = 1;                                            // its slot in the blob stays empty (zeroed)
                const int64_t end = i + 1 < n ? job->file_off[i + 1] : job->blob_len;
                memset(job->blob + job->file_off[i], 0, (size_t)(end - job->file_off[i]));
                continue;
            }
            // the file into the blob, the gap up to the next file (alignment / the read-ahead slack) zeroed
            uint8_t *dst = job->blob + job->file_off[i];
            memcpy(dst, raw, (size_t)sz);
            const int64_t end = i + 1 < n ? job->file_off[i + 1] : job->blob_len;
            memset(dst + sz, 0, (size_t)(end - job->file_off[i] - sz));
        }
    };
    int nt = job->n_threads < 1 ? 1 : std::min(job->n_threads, 64);
    nt = std::min(nt, n);
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (declined.load() < n) {
        job->declined_file = declined.load();
        if (!skip) return MJ_HOST_DECLINED;
    }
    if (job->file_off[0] > 0) memset(job->blob, 0, (size_t)job->file_off[0]);

    // tables numbered in order of first use: per image, per component, quantisation table / DC table / AC table
    // (the order batch.prepare_batch numbers them in)
    std::unordered_map<std::string, int> huff_ids, qt_ids;
    auto huff_id = [&](const TableRef &t) -> int {
        std::string k = huff_key(t);
        auto it = huff_ids.find(k);
        if (it != huff_ids.end()) return it->second;
        const int id = (int)huff_ids.size();
        if (id >= job->huff_cap) return -1;
        memcpy(job->huff[id].bits, k.data(), 16);
        memcpy(job->huff[id].vals, k.data() + 16, 256);
        huff_ids.emplace(std::move(k), id);
        return id;
    };
    auto qt_id = [&](const uint8_t *q) -> int {
        std::string k(reinterpret_cast<const char *>(q), 64);
        auto it = qt_ids.find(k);
        if (it != qt_ids.end()) return it->second;
        const int id = (int)qt_ids.size();
        if (id >= job->qt_cap) return -1;
        for (int j = 0; j < 64; ++j) job->qt[(size_t)id * 64 + j] = q[j];
        qt_ids.emplace(std::move(k), id);
        return id;
    };
    int k = 0;                               // accepted files so far = index into the output arrays
    for (int i = 0; i < n; ++i) {
        if (skip && skip[i]) continue;
        const FileHeader &h = hdr[(size_t)i];
        mj_image_desc d;
        memset(&d, 0, sizeof d);
        d.width = h.width; d.height = h.height; d.ncomp = h.ncomp;
        int hmax = 1, vmax = 1;
        for (int c = 0; c < h.ncomp; ++c) {
            d.hs[c] = h.hs[c]; d.vs[c] = h.vs[c];
            hmax = std::max(hmax, h.hs[c]); vmax = std::max(vmax, h.vs[c]);
            d.qt_sel[c] = qt_id(h.qt[c]);
            d.dc_sel[c] = huff_id(h.dc[c]);
            d.ac_sel[c] = huff_id(h.ac[c]);
            if (d.qt_sel[c] < 0 || d.dc_sel[c] < 0 || d.ac_sel[c] < 0) return MJ_ERR_INVALID;     // capacities too small
        }
        if (h.ncomp == 1) { d.hs[0] = d.vs[0] = 1; hmax = vmax = 1; }     // a single-component scan has 8x8 MCUs (:595-598, :612-619)
        d.restart_interval = h.restart_interval;
        d.mcu_count_h = (h.width + 8 * hmax - 1) / (8 * hmax);             // (:609-611)
        d.mcu_count_v = (h.height + 8 * vmax - 1) / (8 * vmax);
        d.n_segments = 1;                                                  // MJ_FLAG_GPU_SEGMENT: one byte range per image
        d.first_segment = k;
        job->images[k] = d;
        job->seg_begin[k] = job->file_off[i] + h.entropy_start;
        job->seg_end[k] = job->file_off[i] + job->sizes[i];
        ++k;
    }
    job->n_accepted = k;
    job->n_huff = (int32_t)huff_ids.size();
    job->n_qt = (int32_t)qt_ids.size();
    return k > 0 ? MJ_OK : MJ_HOST_DECLINED;
}
