// Test and tuning switches of libmijpeg.so (mj_set_option / mj_get_option, include/mijpeg.h): process-wide, set through the API
// only — the product library does not look at the environment for them.
#include <string.h>
#include <stdlib.h>

#include <mutex>
#include <string>

#include "mijpeg_internal.h"

namespace mj {
namespace {
// name, and what a value must look like: one of `words` (separated by '|'), or an integer in [lo, hi] (a multiple of `step`)
struct OptionRule { const char *name; const char *words; int lo, hi, step; };
const OptionRule kOptionRules[] = {
    {"MJ_HUFFMAN", "wave|lanes|lanes11|sync", 0, 0, 1}, {"MJ_SEG_ORDER", "blob|binned|striped", 0, 0, 1},
    {"MJ_SYNC_ROUNDS", nullptr, 0, 64, 1},   {"MJ_SYNC_CHUNK", nullptr, 256, 65536, 4}, {"MJ_SYNC_WARM", nullptr, 0, 65536, 1},
    {"MJ_PROG_BANDS", nullptr, 0, 1, 1},     {"MJ_PROG_ROWS", nullptr, 1, 4096, 1},     {"MJ_PROG_FAST", nullptr, 0, 1, 1},
    {"MJ_LANES_WAVES", nullptr, 1, 16, 1},   {"MJ_LANES_PER_WAVE", nullptr, 1, 64, 1},  {"MJ_LANES_RING", "64|128", 0, 0, 1},
    {"MJ_STAGE2_CHUNK", nullptr, 1, 4096, 1}, {"MJ_PROG_SPLIT", nullptr, 0, 3, 1},      {"MJ_PROG_PARTS", nullptr, 1, kProgSub, 1},
    {"MJ_FUSED", nullptr, 0, 1, 1},          {"MJ_FUSED_CONSUMERS", nullptr, 0, 15, 1},  {"MJ_FUSED_PATIENCE", nullptr, 0, 1000000, 1},
    {"MJ_FUSED_LUMA13", nullptr, 0, 1, 1},   {"MJ_SYNC_COUNT", "classic|resolved", 0, 0, 1}, {"MJ_SYNC_BITS", nullptr, 10, 13, 1},
    {"MJ_FUSED_ACBITS", nullptr, 10, 13, 1}, {"MJ_FUSED_PRODUCERS", nullptr, 1, 8, 1}, {"MJ_FUSED_SIMD_SPLIT", nullptr, 0, 1, 1}, {"MJ_FUSED_ORDER", "ticket|newest", 0, 0, 1},
    {"MJ_FUSED_PIECE", nullptr, 1, 4096, 1},
    {"MJ_PROG_CHUNKS", nullptr, 0, 2, 1},    {"MJ_PROG_CHUNK", nullptr, 128, 65536, 4},
};
constexpr int kNumOptions = (int)(sizeof(kOptionRules) / sizeof(kOptionRules[0]));
struct OptionTable {
    std::mutex mu;
    struct Entry { std::string value; bool set = false; };
    Entry e[kNumOptions];
};
OptionTable g_options;
bool option_value_ok(const OptionRule &r, const char *v) {
    if (r.words) {
        const size_t n = strlen(v);
        for (const char *w = r.words; *w;) {
            const char *bar = strchr(w, '|');
            const size_t len = bar ? (size_t)(bar - w) : strlen(w);
            if (len == n && !strncmp(w, v, n)) return true;
            w += len + (bar ? 1 : 0);
        }
        return false;
    }
    char *end = nullptr;
    const long x = strtol(v, &end, 10);
    return end != v && *end == 0 && x >= r.lo && x <= r.hi && x % r.step == 0;
}
}  // namespace
// The value is COPIED under the lock (another thread may set the option while this one parses it) into a small per-thread
// ring: the pointer stays good for this thread's next seven opt() calls — every caller consumes it on the spot.
const char *opt(const char *name) {
    thread_local std::string ring[8];
    thread_local unsigned turn = 0;
    {
        std::lock_guard<std::mutex> lk(g_options.mu);
        for (int i = 0; i < kNumOptions; ++i)
            if (!strcmp(kOptionRules[i].name, name)) {
                if (!g_options.e[i].set) break;
                std::string &slot = ring[turn++ & 7u];
                slot = g_options.e[i].value;
                return slot.c_str();
            }
    }
#ifdef MJ_DIAGNOSTIC
    return getenv(name);
#else
    return nullptr;
#endif
}
int get_opt(const char *name, char *out, int cap) {
    std::lock_guard<std::mutex> lk(g_options.mu);
    for (int i = 0; i < kNumOptions; ++i)
        if (!strcmp(kOptionRules[i].name, name)) {
            if (cap > 0) { strncpy(out, g_options.e[i].set ? g_options.e[i].value.c_str() : "", (size_t)cap - 1); out[cap - 1] = 0; }
            return MJ_OK;
        }
    return MJ_ERR_INVALID;
}
int set_opt(const char *name, const char *value) {
    std::lock_guard<std::mutex> lk(g_options.mu);
    for (int i = 0; i < kNumOptions; ++i)
        if (!strcmp(kOptionRules[i].name, name)) {
            const bool set = value != nullptr && value[0] != 0;
            if (set && !option_value_ok(kOptionRules[i], value)) return MJ_ERR_INVALID;     // a sweep must not time the default under another label
            g_options.e[i].set = set;
            g_options.e[i].value = set ? value : "";
            return MJ_OK;
        }
    return MJ_ERR_INVALID;
}
}  // namespace mj

extern "C" {

int mj_set_option(const char *name, const char *value) {
    if (!name) return MJ_ERR_INVALID;
    return mj::set_opt(name, value);
}

int mj_get_option(const char *name, char *value_out, int32_t cap) {
    if (!name || (cap > 0 && !value_out)) return MJ_ERR_INVALID;
    return mj::get_opt(name, value_out, cap);
}

}  // extern "C"
