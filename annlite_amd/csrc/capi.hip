// capi.hip -- library-level C ABI: version, error reporting, device queries.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "common.h"

namespace annlite {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return ANNLITE_ERR_HIP;
}

int device_cu_count() {
    static thread_local int cached_dev = -1;
    static thread_local int cached_cu = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
        cached_cu = cu;
        cached_dev = dev;
    }
    return cached_cu;
}

static int env_int(const char *name, int unset) {
    const char *e = getenv(name);
    return e ? atoi(e) : unset;
}
static int64_t env_i64(const char *name, int64_t unset) {
    const char *e = getenv(name);
    return e ? (int64_t)atoll(e) : unset;
}

static Knobs *parse_knobs() {
    Knobs *k = new Knobs();
    k->scan_variant = env_int("ANNLITE_SCAN_VARIANT", -1);
    k->no_fast_code16 = getenv("ANNLITE_NO_FAST_CODE16") != nullptr;
    k->scan_slices = env_i64("ANNLITE_SCAN_SLICES", 0);
    k->debug_skip = env_int("ANNLITE_DEBUG_SKIP", 0);
    k->debug_counters = env_int("ANNLITE_DEBUG_COUNTERS", 0);
    if (getenv("ANNLITE_DEBUG_COUNTERS") && k->debug_counters == 0) k->debug_counters = 1;  // (set to anything: on)
    k->q8_map = env_int("ANNLITE_Q8_MAP", -1);
    k->q8_ilv = env_int("ANNLITE_Q8_ILV", -1);
    k->q8_rebuild = env_int("ANNLITE_Q8_REBUILD", -1);
    k->q8_target = env_int("ANNLITE_Q8_TARGET", -1);
    k->q8_tune_set = k->q8_tune_ok = false;
    if (const char *e = getenv("ANNLITE_Q8_TUNE")) {
        k->q8_tune_set = true;
        int v[4];
        // (a ring limit below 192 can deadlock: the consumer waits for entries of a producer the limit holds back)
        if (sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] >= 0 && v[1] >= 2 && v[2] >= 192 && v[2] <= 448 && v[3] >= 0) {
            k->q8_tune_ok = true;
            for (int i = 0; i < 4; ++i) k->q8_tune[i] = v[i];
        }
    }
    k->guard_base = env_i64("ANNLITE_GUARD_BASE", -1);
    k->q8_pos_ok = false;
    if (const char *e = getenv("ANNLITE_Q8_POS")) {
        double g[4];
        if (sscanf(e, "%lf,%lf,%lf,%lf", &g[0], &g[1], &g[2], &g[3]) == 4 && g[0] > 0 && g[0] < g[1] && g[1] < g[2] && g[2] < g[3] && g[3] >= 1.0) {
            k->q8_pos_ok = true;
            for (int i = 0; i < 4; ++i) k->q8_pos[i] = g[i];
        }
    }
    k->flush_mask = env_int("ANNLITE_FLUSH_MASK", -1);
    k->seed_rows_set = getenv("ANNLITE_SEED_ROWS") != nullptr;
    k->seed_rows = env_i64("ANNLITE_SEED_ROWS", 0);
    k->seed_contiguous = getenv("ANNLITE_SEED_CONTIGUOUS") != nullptr;
    {
        const int t = env_int("ANNLITE_SEED_CHUNK_LOG", 3);
        k->seed_chunk_log = t < 0 ? 0 : t > 6 ? 6 : t;
    }
    k->no_fused_seed = getenv("ANNLITE_NO_FUSED_SEED") != nullptr;
    k->no_prebuilt_tables = getenv("ANNLITE_NO_PREBUILT_TABLES") != nullptr;
    k->no_early_merge = getenv("ANNLITE_NO_EARLY_MERGE") != nullptr;
    k->early_merge_patience = env_i64("ANNLITE_EARLY_MERGE_PATIENCE", -1);
    k->no_inkernel_merge = getenv("ANNLITE_NO_INKERNEL_MERGE") != nullptr;
    k->no_fused_lut = getenv("ANNLITE_NO_FUSED_LUT") != nullptr;
    k->mfma_seed = getenv("ANNLITE_MFMA_SEED") != nullptr;
    k->no_cand_seed = getenv("ANNLITE_NO_CAND_SEED") != nullptr;
    k->graph_hash_bits = env_int("ANNLITE_GRAPH_HASH_BITS", -1);
    k->graph_seq_insert = getenv("ANNLITE_GRAPH_SEQ_INSERT") != nullptr;
    k->ivf_first = env_int("ANNLITE_IVF_FIRST", -1);
    k->ivf_static_tiles = getenv("ANNLITE_IVF_STATIC_TILES") != nullptr;
    return k;
}

// (parsed when the library is loaded -- the static initialiser -- and again only on annlite_knobs_reload(); superseded blocks are
// leaked on purpose: a reader may still hold one, and a reload is a test / measurement event, 200 bytes each)
static std::atomic<const Knobs *> g_knobs{parse_knobs()};

const Knobs &knobs() { return *g_knobs.load(std::memory_order_acquire); }

}  // namespace annlite

using namespace annlite;

extern "C" int annlite_knobs_reload(void) {
    g_knobs.store(parse_knobs(), std::memory_order_release);
    return ANNLITE_OK;
}

extern "C" int annlite_hip_abi_version(void) { return ANNLITE_HIP_ABI_VERSION; }

extern "C" const char *annlite_hip_last_error(void) { return g_err; }

extern "C" int annlite_hip_device_count(int *count) {
    ANNLITE_REQUIRE(count != nullptr, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        (void)hipGetLastError();
        return hip_fail(e, "hipGetDeviceCount");
    }
    *count = n;
    return ANNLITE_OK;
}

extern "C" int annlite_hip_device_arch(int dev, char *buf, size_t buf_len) {
    ANNLITE_REQUIRE(buf != nullptr && buf_len > 0, "buf is NULL");
    hipDeviceProp_t p;
    ANNLITE_HIP_TRY(hipGetDeviceProperties(&p, dev));
    strncpy(buf, p.gcnArchName, buf_len - 1);
    buf[buf_len - 1] = 0;
    return ANNLITE_OK;
}
