// capi.hip -- the extern "C" front end of libsetk_hip.so (include/setk_hip.h).
// Host-side orchestration only: argument checking, staging of host buffers,
// work-list construction and kernel launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/setk_hip.h"
#include "common.h"
#include "mcdft_tables.h"

using namespace setk;

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr int kKindPevd = 100;

struct Block {
    char* ptr = nullptr;
    size_t cap = 0;
    size_t off = 0;
};

}  // namespace

struct setk_context {
    int device = 0;
    std::string err;
    // STFT plan
    bool planned = false;
    int frame_len = 0, hop = 0, n_fft = 0, center = 0;
    float* d_window = nullptr;  // [n_fft] padded analysis/synthesis window, scaled by 0.5
                                // (the rfft split / irfft merge omit their 1/2)
    float* d_winsq = nullptr;   // [n_fft]
    float2* d_tw256 = nullptr;  // [256]
    float2* d_tw512 = nullptr;  // [129]
    // matrix-core transforms of the fused path (n_fft = 512; mcdft.h)
    unsigned* d_mc_tab = nullptr;  // [mc::kTabWords][64] operand tiles (once per handle)
    float* d_window_pcm = nullptr; // d_window x 2^-15: pass 1 on 16-bit PCM (SETK_FLAG_IN_PCM16)
    float* d_mc_win = nullptr;     // [8][64] analysis window rows x mc_scale
    float* d_mc_syn = nullptr;     // [8][64] synthesis window rows / 512 / sum(window^2) (hop = n_fft / 2)
    float* d_mc_edge = nullptr;    // [8][64] corrections of the single-contribution blocks
    int mc_cus = 256;
    int mc_p2_items = 0;           // SETK_MC_P2_ITEMS: resident workgroup slots of pass2_mc (0: from the kernel)
    double mc_peak = 1.0;          // |audio| <= mc_peak (a power of two)
    bool mc_enabled = true;        // SETK_LEGACY_FFT=1: the fp32 butterfly kernels
    float2* d_twn = nullptr;    // [n_fft / 2] exp(-2 pi i k / n_fft), generic kernels
                                // (Bluestein plans: [M / 2] exp(-2 pi i k / M))
    // n_fft that is not a power of two: Bluestein tables (modular.hip)
    int blu_M = 0;
    float2* d_chirp = nullptr;  // [n_fft]
    float2* d_bhat = nullptr;   // [M], bit-reversed order
    // device arena (bump allocated per call, blocks reused across calls)
    std::vector<Block> blocks;
    // descriptor cache of the fused path
    std::vector<char> desc_cache;
    char* d_desc = nullptr;
    size_t d_desc_cap = 0;
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;   // free events
    std::vector<hipEvent_t> ev_used;   // 5 per profiled call, in call order
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    // stream of the most recent call (arena_reset)
    hipStream_t last_stream = nullptr;
    bool have_last_stream = false;
    // page-locked staging of the small host tables (h2d_small)
    char* pin_base = nullptr;
    size_t pin_head = 0;
    bool pin_failed = false;
    std::vector<hipEvent_t> pin_live;  // one per copy issued out of the buffer since the last lap
    std::vector<hipEvent_t> pin_free;
    // tunables
    int p1_items = 1024;
    int p2_items = 1024;
};

namespace {

int fail(setk_handle_t h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                              \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess)                                                         \
            return fail(h, SETK_ERR_HIP,                                              \
                        std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

// Every entry point starts by recycling the per-handle arena (and may rewrite
// the cached descriptor block).  Work of the previous call may still be in
// flight on ITS stream: calls on the same stream are ordered behind it, a call
// on a different stream first drains the previous one.
void arena_reset(setk_handle_t h, hipStream_t s) {
    if (h->have_last_stream && h->last_stream != s) (void)hipStreamSynchronize(h->last_stream);
    h->last_stream = s;
    h->have_last_stream = true;
    for (auto& b : h->blocks) b.off = 0;
}

void* arena_alloc(setk_handle_t h, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    for (auto& b : h->blocks) {
        if (b.cap - b.off >= bytes) {
            void* p = b.ptr + b.off;
            b.off += bytes;
            return p;
        }
    }
    size_t cap = std::max(bytes, (size_t)64 << 20);
    if (!h->blocks.empty()) cap = std::max(cap, h->blocks.back().cap);
    Block nb;
    if (hipMalloc(reinterpret_cast<void**>(&nb.ptr), cap) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    nb.cap = cap;
    nb.off = bytes;
    h->blocks.push_back(nb);
    return nb.ptr;
}

// Small host tables (descriptors, pointer lists, argument blocks) go to the device through a
// page-locked buffer of the handle.  hipMemcpyAsync from PAGEABLE memory does not return
// before the copy has run, i.e. before everything queued on the stream ahead of it has -- in
// the streaming pipeline that is the 300 MB slab transfer the compute stream is waiting for,
// ~6 ms per batch during which the launching thread could not queue the next batch.  From the
// page-locked buffer the copy is queued and the call returns.  The buffer is used linearly;
// when it is full, every copy issued out of it is waited for (an event each) and it starts
// over.  Tables larger than a quarter of it, or a failed allocation, take the pageable path.
size_t pin_cap() {
    // 8 MB; SETK_PIN_CAP_KB shrinks it so that tests see the buffer wrap
    static const size_t cap = [] {
        const char* e = getenv("SETK_PIN_CAP_KB");
        const long kb = e ? atol(e) : 0;
        return kb >= 4 ? (size_t)kb << 10 : (size_t)8 << 20;
    }();
    return cap;
}

hipError_t h2d_small(setk_handle_t h, void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (!bytes) return hipSuccess;
    const size_t kPinCap = pin_cap();
    if (bytes > kPinCap / 4 || h->pin_failed)
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
    if (!h->pin_base) {
        void* p = nullptr;
        if (hipHostMalloc(&p, kPinCap, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            h->pin_failed = true;
            return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
        }
        h->pin_base = static_cast<char*>(p);
    }
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (h->pin_head + need > kPinCap || h->pin_live.size() >= 4096) {
        for (hipEvent_t e : h->pin_live) {
            (void)hipEventSynchronize(e);
            h->pin_free.push_back(e);
        }
        h->pin_live.clear();
        h->pin_head = 0;
    }
    char* p = h->pin_base + h->pin_head;
    h->pin_head += need;
    memcpy(p, src, bytes);
    hipError_t e = hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    hipEvent_t ev = nullptr;
    if (!h->pin_free.empty()) {
        ev = h->pin_free.back();
        h->pin_free.pop_back();
    } else {
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    e = hipEventRecord(ev, s);
    h->pin_live.push_back(ev);
    return e;
}

// Stage an input: device pointers pass through, host data is copied.
template <typename T>
int stage_in(setk_handle_t h, const T* src, size_t count, hipStream_t s, const T** out) {
    if (is_device_ptr(src)) {
        *out = src;
        return SETK_OK;
    }
    void* d = arena_alloc(h, count * sizeof(T));
    if (!d) return fail(h, SETK_ERR_NOMEM, "device arena allocation failed");
    HIP_TRY(h, hipMemcpyAsync(d, src, count * sizeof(T), hipMemcpyHostToDevice, s));
    *out = static_cast<const T*>(d);
    return SETK_OK;
}

// Prepare an output: device pointers are written in place, host outputs get a
// device twin that copy_back() drains.
struct OutBuf {
    void* user = nullptr;
    void* dev = nullptr;
    size_t bytes = 0;
    bool host = false;
};

int stage_out(setk_handle_t h, void* dst, size_t bytes, OutBuf* ob) {
    ob->user = dst;
    ob->bytes = bytes;
    if (is_device_ptr(dst)) {
        ob->dev = dst;
        ob->host = false;
        return SETK_OK;
    }
    ob->dev = arena_alloc(h, bytes);
    ob->host = true;
    if (!ob->dev) return fail(h, SETK_ERR_NOMEM, "device arena allocation failed");
    return SETK_OK;
}

int copy_back(setk_handle_t h, const OutBuf& ob, hipStream_t s) {
    if (ob.host && ob.bytes)
        HIP_TRY(h, hipMemcpyAsync(ob.user, ob.dev, ob.bytes, hipMemcpyDeviceToHost, s));
    return SETK_OK;
}

// Descriptor tables are built in ordinary host vectors that die when the entry point
// returns: h2d_small copies them into the handle's page-locked buffer first (or, for a large
// table, lets the runtime stage the pageable source before hipMemcpyAsync returns), so the
// source may be released either way.
int upload(setk_handle_t h, const void* src, size_t bytes, hipStream_t s, void** out) {
    void* d = arena_alloc(h, bytes);
    if (!d) return fail(h, SETK_ERR_NOMEM, "device arena allocation failed");
    HIP_TRY(h, h2d_small(h, d, src, bytes, s));
    *out = d;
    return SETK_OK;
}

int pitch_of(int F) { return (F == kBins) ? kBinsPad : ((F + 7) / 8) * 8; }

StftGeom geom_of(setk_handle_t h) {
    StftGeom g;
    g.hop = h->hop;
    g.center = h->center;
    g.pad = h->center ? h->n_fft / 2 : 0;
    g.keep = (h->n_fft + h->hop - 1) / h->hop - 1;
    return g;
}

int require_plan512(setk_handle_t h) {
    if (!h->planned) return fail(h, SETK_ERR_INVALID, "setk_stft_plan has not been called");
    if (h->n_fft != kNfft)
        return fail(h, SETK_ERR_UNSUPPORTED,
                    "only the n_fft = 512 kernels are built (n_fft = " + std::to_string(h->n_fft) +
                        ")");
    if (h->hop > h->n_fft) return fail(h, SETK_ERR_UNSUPPORTED, "frame_hop > n_fft");
    if (geom_of(h).keep > kMaxKeep)
        return fail(h, SETK_ERR_UNSUPPORTED, "frame_hop < 64 is not supported");
    return SETK_OK;
}

// split [0, T) into ranges of about `target` frames, multiples of `quant`
void split_frames(int T, int target, int quant, std::vector<std::pair<int, int>>* out) {
    int nparts = std::max(1, (T + target - 1) / target);
    int fw = (T + nparts - 1) / nparts;
    fw = ((fw + quant - 1) / quant) * quant;
    for (int t = 0; t < T; t += fw) out->push_back({t, std::min(T, t + fw)});
}

// Frames per work item such that the work list fills whole "waves" of resident
// workgroups: among the splits of the longest utterance into 1..32 ranges pick
// the one minimising  ceil(items / slots) * frames_per_item  (makespan in
// frames); utterances shorter than the target become single items.
int choose_target(const std::vector<int>& frames, int slots, int quant, int min_frames) {
    int tmax = 1;
    for (int t : frames) tmax = std::max(tmax, t);
    long best_cost = -1;
    int best = tmax;
    for (int parts = 1; parts <= 32; ++parts) {
        int target = (tmax + parts - 1) / parts;
        target = ((target + quant - 1) / quant) * quant;
        if (target < min_frames && parts > 1) break;
        long items = 0;
        for (int t : frames) items += (t + target - 1) / target;
        const long waves = (items + slots - 1) / slots;
        const long cost = waves * (long)target + 8 * waves;  // + per-wave fixed cost
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = target;
        }
    }
    return best;
}

}  // namespace

extern "C" {

int setk_abi_version(void) { return SETK_ABI_VERSION; }

int setk_create(setk_handle_t* out, int device_ordinal) {
    if (!out) return SETK_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return SETK_ERR_HIP;
    }
    if (device_ordinal < 0 || device_ordinal >= n) return SETK_ERR_INVALID;
    if (hipSetDevice(device_ordinal) != hipSuccess) return SETK_ERR_HIP;
    setk_context* h = new setk_context();
    h->device = device_ordinal;
    // resident workgroup slots: pass 1 runs 1 workgroup per CU, pass 2 two; the
    // work lists are cut to fill whole waves of those slots (choose_target)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0) {
        h->p1_items = prop.multiProcessorCount;
        h->p2_items = 2 * prop.multiProcessorCount;
        h->mc_cus = prop.multiProcessorCount;
    }
    if (const char* e = getenv("SETK_MC_P2_ITEMS")) h->mc_p2_items = std::max(1, atoi(e));
    if (const char* e = getenv("SETK_P1_ITEMS")) h->p1_items = std::max(1, atoi(e));
    if (const char* e = getenv("SETK_P2_ITEMS")) h->p2_items = std::max(1, atoi(e));
    *out = h;
    return SETK_OK;
}

int setk_destroy(setk_handle_t h) {
    if (!h) return SETK_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (auto& b : h->blocks) (void)hipFree(b.ptr);
    if (h->d_window) (void)hipFree(h->d_window);
    if (h->d_winsq) (void)hipFree(h->d_winsq);
    if (h->d_tw256) (void)hipFree(h->d_tw256);
    if (h->d_tw512) (void)hipFree(h->d_tw512);
    if (h->d_mc_tab) (void)hipFree(h->d_mc_tab);
    if (h->d_mc_win) (void)hipFree(h->d_mc_win);
    if (h->d_window_pcm) (void)hipFree(h->d_window_pcm);
    if (h->d_mc_syn) (void)hipFree(h->d_mc_syn);
    if (h->d_mc_edge) (void)hipFree(h->d_mc_edge);
    if (h->d_twn) (void)hipFree(h->d_twn);
    if (h->d_chirp) (void)hipFree(h->d_chirp);
    if (h->d_bhat) (void)hipFree(h->d_bhat);
    if (h->d_desc) (void)hipFree(h->d_desc);
    for (auto& e : h->ev_pool) (void)hipEventDestroy(e);
    for (auto& e : h->ev_used) (void)hipEventDestroy(e);
    for (auto& e : h->pin_live) {
        (void)hipEventSynchronize(e);
        (void)hipEventDestroy(e);
    }
    for (auto& e : h->pin_free) (void)hipEventDestroy(e);
    if (h->pin_base) (void)hipHostFree(h->pin_base);
    delete h;
    return SETK_OK;
}

const char* setk_last_error(setk_handle_t h) { return h ? h->err.c_str() : "null handle"; }

int setk_device_pci_bus_id(setk_handle_t h, char* out, int len) {
    if (!h || !out || len < 13) return SETK_ERR_INVALID;
    HIP_TRY(h, hipDeviceGetPCIBusId(out, len, h->device));
    return SETK_OK;
}

int setk_set_profiling(setk_handle_t h, int enable) {
    if (!h) return SETK_ERR_INVALID;
    h->profiling = enable != 0;
    return SETK_OK;
}

// mean stage times (ms) over the profiled setk_enhance_batch calls since the
// last query; the recorded events are recycled.
int setk_last_stage_ms(setk_handle_t h, float out[4]) {
    if (!h || !out) return SETK_ERR_INVALID;
    const size_t calls = h->ev_used.size() / 5;
    if (calls == 0) return fail(h, SETK_ERR_INVALID, "no profiled run available");
    double acc[4] = {0, 0, 0, 0};
    for (size_t c = 0; c < calls; ++c) {
        hipEvent_t* e = &h->ev_used[c * 5];
        HIP_TRY(h, hipEventSynchronize(e[4]));
        for (int i = 0; i < 4; ++i) {
            float ms = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&ms, e[i], e[i + 1]));
            acc[i] += ms;
        }
    }
    for (int i = 0; i < 4; ++i) out[i] = (float)(acc[i] / (double)calls);
    for (auto& e : h->ev_used) h->ev_pool.push_back(e);
    h->ev_used.clear();
    return SETK_OK;
}

// ---- host-memory plumbing of the streaming pipeline (setk_amd/pipeline.py) ----
// A wave or mask file that sits in the page cache can be DMA'd from where it is:
// mmap it, pin the mapping, copy from it.  Measured (profiles/r02j_*): pinning a
// 7.7 MB mapping costs 0.23 ms and the copy then runs at 52 GB/s, against 0.9 ms
// for reading the same bytes into a staging buffer first.
int setk_host_register(setk_handle_t h, void* ptr, size_t bytes) {
    if (!h || !ptr || !bytes) return SETK_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(h, SETK_ERR_HIP, std::string("hipHostRegister: ") + hipGetErrorString(e));
    }
    return SETK_OK;
}

int setk_host_unregister(setk_handle_t h, void* ptr) {
    if (!h || !ptr) return SETK_ERR_INVALID;
    hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(h, SETK_ERR_HIP, std::string("hipHostUnregister: ") + hipGetErrorString(e));
    }
    return SETK_OK;
}

int setk_memcpy_h2d_async(setk_handle_t h, void* dst, const void* src, size_t bytes, void* stream) {
    if (!h || !dst || !src) return SETK_ERR_INVALID;
    HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice,
                              static_cast<hipStream_t>(stream)));
    return SETK_OK;
}

int setk_memcpy_d2h_async(setk_handle_t h, void* dst, const void* src, size_t bytes, void* stream) {
    if (!h || !dst || !src) return SETK_ERR_INVALID;
    HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost,
                              static_cast<hipStream_t>(stream)));
    return SETK_OK;
}

// ---- buffers, streams and events for a host pipeline that brings no runtime of its own ----
int setk_device_alloc(setk_handle_t h, size_t bytes, void** out) {
    if (!h || !out || !bytes) return SETK_ERR_INVALID;
    *out = nullptr;
    HIP_TRY(h, hipSetDevice(h->device));
    if (hipMalloc(out, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return fail(h, SETK_ERR_NOMEM, "hipMalloc");
    }
    return SETK_OK;
}

int setk_device_free(setk_handle_t h, void* ptr) {
    if (!h) return SETK_ERR_INVALID;
    if (ptr) HIP_TRY(h, hipFree(ptr));
    return SETK_OK;
}

int setk_host_alloc(setk_handle_t h, size_t bytes, void** out) {
    if (!h || !out || !bytes) return SETK_ERR_INVALID;
    *out = nullptr;
    HIP_TRY(h, hipSetDevice(h->device));
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return fail(h, SETK_ERR_NOMEM, "hipHostMalloc");
    }
    return SETK_OK;
}

int setk_host_free(setk_handle_t h, void* ptr) {
    if (!h) return SETK_ERR_INVALID;
    if (ptr) HIP_TRY(h, hipHostFree(ptr));
    return SETK_OK;
}

int setk_stream_create(setk_handle_t h, void** out) {
    if (!h || !out) return SETK_ERR_INVALID;
    hipStream_t s = nullptr;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = s;
    return SETK_OK;
}

int setk_stream_destroy(setk_handle_t h, void* stream) {
    if (!h) return SETK_ERR_INVALID;
    if (stream) HIP_TRY(h, hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return SETK_OK;
}

int setk_stream_synchronize(setk_handle_t h, void* stream) {
    if (!h) return SETK_ERR_INVALID;
    HIP_TRY(h, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return SETK_OK;
}

int setk_stream_wait_event(setk_handle_t h, void* stream, void* event) {
    if (!h || !event) return SETK_ERR_INVALID;
    HIP_TRY(h, hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
    return SETK_OK;
}

int setk_event_create(setk_handle_t h, void** out) {
    if (!h || !out) return SETK_ERR_INVALID;
    hipEvent_t e = nullptr;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *out = e;
    return SETK_OK;
}

int setk_event_destroy(setk_handle_t h, void* event) {
    if (!h) return SETK_ERR_INVALID;
    if (event) HIP_TRY(h, hipEventDestroy(static_cast<hipEvent_t>(event)));
    return SETK_OK;
}

int setk_event_record(setk_handle_t h, void* event, void* stream) {
    if (!h || !event) return SETK_ERR_INVALID;
    HIP_TRY(h, hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
    return SETK_OK;
}

int setk_event_synchronize(setk_handle_t h, void* event) {
    if (!h || !event) return SETK_ERR_INVALID;
    HIP_TRY(h, hipEventSynchronize(static_cast<hipEvent_t>(event)));
    return SETK_OK;
}

int setk_stft_plan(setk_handle_t h, int frame_len, int frame_hop, int n_fft, int center,
                   const float* window) {
    if (!h) return SETK_ERR_INVALID;
    const bool pow2 = (n_fft & (n_fft - 1)) == 0;
    if (frame_len <= 0 || frame_hop <= 0 || n_fft < 16 || n_fft > 4096 || (n_fft & 1) ||
        (pow2 && n_fft < 64))
        return fail(h, SETK_ERR_INVALID,
                    "n_fft must be even, in [16, 4096] (powers of two: [64, 4096])");
    if (frame_len > n_fft) return fail(h, SETK_ERR_INVALID, "frame_len > n_fft");
    HIP_TRY(h, hipSetDevice(h->device));
    std::vector<float> w(n_fft, 0.f), w2(n_fft, 0.f);
    const int lpad = (n_fft - frame_len) / 2;
    for (int i = 0; i < frame_len; ++i) {
        double v = window ? (double)window[i] : 0.5 - 0.5 * std::cos(2.0 * kPi * i / frame_len);
        w[lpad + i] = 0.5f * (float)v;
        w2[lpad + i] = (float)(v * v);
    }
    std::vector<float2> t256(256), t512(129);
    for (int q = 0; q < 16; ++q)
        for (int la = 0; la < 16; ++la) {
            const double ang = -2.0 * kPi * (double)(la * q) / 256.0;
            t256[q * 16 + la] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
    for (int k = 0; k <= 128; ++k) {
        const double ang = -2.0 * kPi * (double)k / 512.0;
        t512[k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    HIP_TRY(h, hipDeviceSynchronize());
    if (h->d_window) (void)hipFree(h->d_window);
    if (h->d_winsq) (void)hipFree(h->d_winsq);
    h->d_window = h->d_winsq = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_window), n_fft * sizeof(float)));
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_winsq), n_fft * sizeof(float)));
    if (!h->d_tw256) HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_tw256), 256 * sizeof(float2)));
    if (!h->d_tw512) HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_tw512), 129 * sizeof(float2)));
    HIP_TRY(h, hipMemcpy(h->d_window, w.data(), n_fft * sizeof(float), hipMemcpyHostToDevice));
    {
        // the same window with read_wav's 1 / 32768 folded in (exact: a power of two)
        std::vector<float> wp(n_fft);
        for (int i = 0; i < n_fft; ++i) wp[i] = w[i] * 3.0517578125e-05f;
        if (h->d_window_pcm) (void)hipFree(h->d_window_pcm);
        h->d_window_pcm = nullptr;
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_window_pcm), n_fft * sizeof(float)));
        HIP_TRY(h, hipMemcpy(h->d_window_pcm, wp.data(), n_fft * sizeof(float), hipMemcpyHostToDevice));
    }
    HIP_TRY(h, hipMemcpy(h->d_winsq, w2.data(), n_fft * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_tw256, t256.data(), 256 * sizeof(float2), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_tw512, t512.data(), 129 * sizeof(float2), hipMemcpyHostToDevice));
    h->blu_M = 0;
    if (h->d_chirp) (void)hipFree(h->d_chirp);
    if (h->d_bhat) (void)hipFree(h->d_bhat);
    h->d_chirp = h->d_bhat = nullptr;
    int tw_n = n_fft;
    if (!pow2) {
        // Bluestein: chirp c[k] = exp(-i pi k^2 / n) (k^2 reduced mod 2n in integers),
        // and the length-M spectrum of the wrapped conj(c), in bit-reversed order
        int M = 1;
        while (M < 2 * n_fft - 1) M <<= 1;
        int logM = 0;
        while ((1 << logM) < M) ++logM;
        std::vector<double> cr(n_fft), ci(n_fft);
        for (int k = 0; k < n_fft; ++k) {
            const long k2 = ((long)k * k) % (2L * n_fft);
            const double ang = -kPi * (double)k2 / (double)n_fft;
            cr[k] = std::cos(ang);
            ci[k] = std::sin(ang);
        }
        std::vector<double> br(M, 0.0), bi(M, 0.0);
        for (int k = 0; k < n_fft; ++k) {
            br[k] = cr[k];
            bi[k] = -ci[k];
            if (k) {
                br[M - k] = cr[k];
                bi[M - k] = -ci[k];
            }
        }
        // iterative radix-2 DIT in double (bit reversal first)
        for (int i = 0; i < M; ++i) {
            int r = 0;
            for (int b = 0; b < logM; ++b) r |= ((i >> b) & 1) << (logM - 1 - b);
            if (r > i) {
                std::swap(br[i], br[r]);
                std::swap(bi[i], bi[r]);
            }
        }
        for (int len = 2; len <= M; len <<= 1) {
            const double a0 = -2.0 * kPi / (double)len;
            for (int i0 = 0; i0 < M; i0 += len)
                for (int k = 0; k < len / 2; ++k) {
                    const double wr = std::cos(a0 * k), wi = std::sin(a0 * k);
                    const int a = i0 + k, b = a + len / 2;
                    const double tr = br[b] * wr - bi[b] * wi, ti = br[b] * wi + bi[b] * wr;
                    br[b] = br[a] - tr;
                    bi[b] = bi[a] - ti;
                    br[a] += tr;
                    bi[a] += ti;
                }
        }
        std::vector<float2> chirp(n_fft), bhat(M);
        for (int k = 0; k < n_fft; ++k) chirp[k] = make_float2((float)cr[k], (float)ci[k]);
        for (int i = 0; i < M; ++i) {
            int r = 0;
            for (int b = 0; b < logM; ++b) r |= ((i >> b) & 1) << (logM - 1 - b);
            bhat[r] = make_float2((float)br[i], (float)bi[i]);
        }
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_chirp), n_fft * sizeof(float2)));
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_bhat), M * sizeof(float2)));
        HIP_TRY(h, hipMemcpy(h->d_chirp, chirp.data(), n_fft * sizeof(float2), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_bhat, bhat.data(), M * sizeof(float2), hipMemcpyHostToDevice));
        h->blu_M = M;
        tw_n = M;
    }
    {
        std::vector<float2> tn(tw_n / 2);
        for (int k = 0; k < tw_n / 2; ++k) {
            const double ang = -2.0 * kPi * (double)k / (double)tw_n;
            tn[k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
        if (h->d_twn) (void)hipFree(h->d_twn);
        h->d_twn = nullptr;
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_twn), tn.size() * sizeof(float2)));
        HIP_TRY(h, hipMemcpy(h->d_twn, tn.data(), tn.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    if (n_fft == kNfft) {
        // matrix-core DFT-512: operand tiles once per handle, window rows per plan
        if (!h->d_mc_tab) {
            const std::vector<uint32_t> tab = mc::build_table();
            HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_mc_tab), tab.size() * sizeof(uint32_t)));
            HIP_TRY(h, hipMemcpy(h->d_mc_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        if (!h->d_mc_win) HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_mc_win), 8 * 64 * sizeof(float)));
        if (!h->d_mc_syn) HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_mc_syn), 8 * 64 * sizeof(float)));
        if (!h->d_mc_edge) HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_mc_edge), 8 * 64 * sizeof(float)));
        std::vector<float> wt(n_fft);
        for (int i = 0; i < n_fft; ++i) wt[i] = 2.f * w[i];  // w holds 0.5 x window (exact)
        const std::vector<float> wr = mc::build_window_rows(wt.data(), 1024.0 / h->mc_peak);
        // pass2_mc (hop = n_fft / 2): a block of hop samples is first half of frame t + second
        // half of frame t - 1, both over the same sum(window^2) -- folded into the rows
        // (librosa.istft: divided only where it exceeds tiny); blocks with one contribution
        // (first / last of an utterance, center = False) take the ratio as a correction
        const double tiny = 1.17549435e-38;
        std::vector<float> syn(n_fft), edge(n_fft);
        for (int m = 0; m < n_fft / 2; ++m) {
            const double a2 = (double)w2[m], b2 = (double)w2[m + n_fft / 2];
            const double mid = (a2 + b2 > tiny) ? a2 + b2 : 1.0;
            syn[m] = (float)((double)wt[m] * h->mc_peak / 1024.0 / 512.0 / mid);
            syn[m + n_fft / 2] = (float)((double)wt[m + n_fft / 2] * h->mc_peak / 1024.0 / 512.0 / mid);
            edge[m] = (float)(mid / (a2 > tiny ? a2 : 1.0));
            edge[m + n_fft / 2] = (float)(mid / (b2 > tiny ? b2 : 1.0));
        }
        const std::vector<float> sr = mc::build_synth_rows(syn.data(), 1.0);
        const std::vector<float> er = mc::build_synth_rows(edge.data(), 1.0);
        HIP_TRY(h, hipMemcpy(h->d_mc_win, wr.data(), wr.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_mc_syn, sr.data(), sr.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_mc_edge, er.data(), er.size() * sizeof(float), hipMemcpyHostToDevice));
        h->mc_enabled = !(getenv("SETK_LEGACY_FFT") && atoi(getenv("SETK_LEGACY_FFT")) != 0);
    }
    h->frame_len = frame_len;
    h->hop = frame_hop;
    h->n_fft = n_fft;
    h->center = center ? 1 : 0;
    h->planned = true;
    h->desc_cache.clear();
    return SETK_OK;
}

int setk_stft_num_frames(setk_handle_t h, int num_samples) {
    if (!h || !h->planned) return SETK_ERR_INVALID;
    if (h->center) {
        if (num_samples < h->n_fft / 2 + 1)
            return fail(h, SETK_ERR_INVALID, "signal shorter than n_fft/2+1 (reflect padding)");
        return 1 + num_samples / h->hop;
    }
    if (num_samples < h->n_fft) return fail(h, SETK_ERR_INVALID, "signal shorter than n_fft");
    return 1 + (num_samples - h->n_fft) / h->hop;
}

int setk_istft_num_samples(setk_handle_t h, int num_frames, int nsamps) {
    if (!h || !h->planned || num_frames <= 0) return SETK_ERR_INVALID;
    if (nsamps >= 0) return nsamps;
    return h->center ? h->hop * (num_frames - 1) : h->n_fft + h->hop * (num_frames - 1);
}

// n_fft != 512: generic LDS radix-2 kernels (modular.hip)
static int stft_generic(setk_handle_t h, const float* audio, int C, int N, float* spec,
                        hipStream_t s) {
    const int T = setk_stft_num_frames(h, N);
    if (T < 0) return T;
    const int F = h->n_fft / 2 + 1;
    const BluesteinPlan bp = {h->blu_M, reinterpret_cast<const float*>(h->d_chirp),
                              reinterpret_cast<const float*>(h->d_bhat)};
    arena_reset(h, s);
    const float* d_audio;
    int rc = stage_in(h, audio, (size_t)C * N, s, &d_audio);
    if (rc) return rc;
    OutBuf ob;
    rc = stage_out(h, spec, (size_t)C * T * F * sizeof(float2), &ob);
    if (rc) return rc;
    HIP_TRY(h, launch_stft_generic(d_audio, C, N, T, h->n_fft, h->hop, h->center ? h->n_fft / 2 : 0,
                                   h->d_window, reinterpret_cast<const float*>(h->d_twn),
                                   static_cast<float*>(ob.dev), &bp, s));
    rc = copy_back(h, ob, s);
    if (rc) return rc;
    if (ob.host) HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

static int istft_generic(setk_handle_t h, const float* spec, int B, int T, int nsamps,
                         const float* norm, float* wave, hipStream_t s) {
    const int F = h->n_fft / 2 + 1;
    const BluesteinPlan bp = {h->blu_M, reinterpret_cast<const float*>(h->d_chirp),
                              reinterpret_cast<const float*>(h->d_bhat)};
    const int L = setk_istft_num_samples(h, T, nsamps);
    int T_eff = T;
    if (nsamps >= 0) {
        const long padded = (long)nsamps + (h->center ? h->n_fft : 0);
        T_eff = (int)std::max<long>(1, std::min<long>(T, (padded + h->hop - 1) / h->hop));
    }
    arena_reset(h, s);
    const float* d_spec;
    int rc = stage_in(h, spec, (size_t)B * T * F * 2, s, &d_spec);
    if (rc) return rc;
    OutBuf ob;
    rc = stage_out(h, wave, (size_t)B * L * sizeof(float), &ob);
    if (rc) return rc;
    std::vector<float> hn(B, -1.f);
    if (norm) {
        if (is_device_ptr(norm)) {
            HIP_TRY(h, hipMemcpyAsync(hn.data(), norm, B * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipStreamSynchronize(s));
        } else
            memcpy(hn.data(), norm, B * sizeof(float));
    }
    void* d_norm;
    rc = upload(h, hn.data(), B * sizeof(float), s, &d_norm);
    if (rc) return rc;
    float* d_frames = static_cast<float*>(arena_alloc(h, (size_t)B * T * h->n_fft * 4));
    unsigned* d_omax = static_cast<unsigned*>(arena_alloc(h, (size_t)B * 4));
    if (!d_frames || !d_omax) return fail(h, SETK_ERR_NOMEM, "arena");
    HIP_TRY(h, hipMemsetAsync(d_omax, 0, (size_t)B * 4, s));
    // the frames kernel indexes spec with the caller's T; only T_eff frames are overlap-added
    HIP_TRY(h, launch_istft_generic(d_spec, B, T, h->n_fft, h->hop, h->center ? h->n_fft / 2 : 0, L,
                                    h->d_window, h->d_winsq,
                                    reinterpret_cast<const float*>(h->d_twn), d_frames,
                                    static_cast<float*>(ob.dev), d_omax,
                                    norm ? static_cast<const float*>(d_norm) : nullptr, T_eff, &bp,
                                    s));
    rc = copy_back(h, ob, s);
    if (rc) return rc;
    if (ob.host) HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

int setk_stft(setk_handle_t h, const float* audio, int num_channels, int num_samples,
              float* spec, void* stream) {
    if (!h || !audio || !spec || num_channels <= 0) return fail(h, SETK_ERR_INVALID, "bad args");
    if (h->planned && h->n_fft != kNfft) {
        HIP_TRY(h, hipSetDevice(h->device));
        return stft_generic(h, audio, num_channels, num_samples, spec,
                            static_cast<hipStream_t>(stream));
    }
    int rc = require_plan512(h);
    if (rc) return rc;
    const int T = setk_stft_num_frames(h, num_samples);
    if (T < 0) return T;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    const float* d_audio;
    rc = stage_in(h, audio, (size_t)num_channels * num_samples, s, &d_audio);
    if (rc) return rc;
    OutBuf ob;
    rc = stage_out(h, spec, (size_t)num_channels * T * kBins * sizeof(float2), &ob);
    if (rc) return rc;
    for (int c0 = 0; c0 < num_channels; c0 += kMaxChannels) {
        const int C = std::min(kMaxChannels, num_channels - c0);
        UttDesc ud;
        memset(&ud, 0, sizeof(ud));
        ud.audio = d_audio + (size_t)c0 * num_samples;
        ud.num_samples = num_samples;
        ud.num_frames = T;
        std::vector<std::pair<int, int>> ranges;
        split_frames(T, 64, 32, &ranges);
        std::vector<WorkItem> items;
        for (auto& r : ranges) items.push_back({0, r.first, r.second, 0, r.second == T});
        void *d_ud, *d_items;
        rc = upload(h, &ud, sizeof(ud), s, &d_ud);
        if (rc) return rc;
        rc = upload(h, items.data(), items.size() * sizeof(WorkItem), s, &d_items);
        if (rc) return rc;
        Pass1Args a;
        memset(&a, 0, sizeof(a));
        a.utts = static_cast<const UttDesc*>(d_ud);
        a.items = static_cast<const WorkItem*>(d_items);
        a.window = h->d_window;
        a.tw256 = h->d_tw256;
        a.tw512 = h->d_tw512;
        a.spec_dump = static_cast<float*>(ob.dev) + (size_t)c0 * T * kBins * 2;
        a.g = geom_of(h);
        HIP_TRY(h, launch_pass1(C, true, a, (int)items.size(), s));
    }
    rc = copy_back(h, ob, s);
    if (rc) return rc;
    if (ob.host) HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

int setk_stft_batch(setk_handle_t h, int n_utts, int num_channels, const float* const* audio,
                    const int* num_samples, float* const* spec, int spec_pitch, void* stream) {
    if (spec_pitch != 0 && spec_pitch < kBins) return fail(h, SETK_ERR_INVALID, "spec_pitch < F");
    if (!h || n_utts <= 0 || !audio || !num_samples || !spec)
        return fail(h, SETK_ERR_INVALID, "bad args");
    int rc = require_plan512(h);
    if (rc) return rc;
    const int C = num_channels;
    if (C < 1 || C > kMaxChannels) return fail(h, SETK_ERR_UNSUPPORTED, "1 <= num_channels <= 8");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    std::vector<UttDesc> uds(n_utts);
    std::vector<WorkItem> items;
    for (int u = 0; u < n_utts; ++u) {
        const int T = setk_stft_num_frames(h, num_samples[u]);
        if (T < 0) return T;
        if (!audio[u] || !spec[u]) return fail(h, SETK_ERR_INVALID, "null utterance pointer");
        UttDesc& ud = uds[u];
        memset(&ud, 0, sizeof(ud));
        ud.audio = audio[u];
        ud.num_samples = num_samples[u];
        ud.num_frames = T;
        ud.wave_out = spec[u];
        std::vector<std::pair<int, int>> ranges;
        split_frames(T, 128, 32, &ranges);
        for (auto& r : ranges) items.push_back({u, r.first, r.second, 0, r.second == T});
    }
    void *d_ud, *d_items;
    rc = upload(h, uds.data(), uds.size() * sizeof(UttDesc), s, &d_ud);
    if (rc) return rc;
    rc = upload(h, items.data(), items.size() * sizeof(WorkItem), s, &d_items);
    if (rc) return rc;
    Pass1Args a;
    memset(&a, 0, sizeof(a));
    a.utts = static_cast<const UttDesc*>(d_ud);
    a.items = static_cast<const WorkItem*>(d_items);
    a.window = h->d_window;
    a.tw256 = h->d_tw256;
    a.tw512 = h->d_tw512;
    a.spec_dump = nullptr;  // per-utterance outputs: UttDesc::wave_out
    a.dump_pitch = spec_pitch;
    a.g = geom_of(h);
    HIP_TRY(h, launch_pass1(C, true, a, (int)items.size(), s));
    return SETK_OK;
}

int setk_istft(setk_handle_t h, const float* spec, int batch, int num_frames, int nsamps,
               const float* norm, float* wave, void* stream) {
    if (!h || !spec || !wave || batch <= 0 || num_frames <= 0)
        return fail(h, SETK_ERR_INVALID, "bad args");
    if (h->planned && h->n_fft != kNfft) {
        HIP_TRY(h, hipSetDevice(h->device));
        return istft_generic(h, spec, batch, num_frames, nsamps, norm, wave,
                             static_cast<hipStream_t>(stream));
    }
    int rc = require_plan512(h);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    const int T = num_frames, F = kBins;
    const int L = setk_istft_num_samples(h, T, nsamps);
    int T_eff = T;
    if (nsamps >= 0) {
        const long padded = (long)nsamps + (h->center ? h->n_fft : 0);
        T_eff = (int)std::min<long>(T, (padded + h->hop - 1) / h->hop);
        if (T_eff < 1) T_eff = 1;
    }
    const float* d_spec;
    rc = stage_in(h, spec, (size_t)batch * T * F * 2, s, &d_spec);
    if (rc) return rc;
    OutBuf ob;
    rc = stage_out(h, wave, (size_t)batch * L * sizeof(float), &ob);
    if (rc) return rc;
    if (L > 0) HIP_TRY(h, hipMemsetAsync(ob.dev, 0, (size_t)batch * L * sizeof(float), s));
    std::vector<float> hn(batch, -1.f);
    if (norm) {
        if (is_device_ptr(norm)) {
            HIP_TRY(h, hipMemcpyAsync(hn.data(), norm, batch * sizeof(float), hipMemcpyDeviceToHost,
                                      s));
            HIP_TRY(h, hipStreamSynchronize(s));
        } else
            memcpy(hn.data(), norm, batch * sizeof(float));
    }
    void* d_norm;
    rc = upload(h, hn.data(), batch * sizeof(float), s, &d_norm);
    if (rc) return rc;
    std::vector<UttDesc> uds(batch);
    std::vector<WorkItem> items;
    for (int b = 0; b < batch; ++b) {
        UttDesc& ud = uds[b];
        memset(&ud, 0, sizeof(ud));
        ud.audio = d_spec + (size_t)b * T * F * 2;  // ISTFT mode: per-item spectrogram
        ud.num_frames = T_eff;
        ud.out_len = L;
        ud.wave_f32 = static_cast<float*>(ob.dev) + (size_t)b * L;
        ud.wave_out = ud.wave_f32;
        std::vector<std::pair<int, int>> ranges;
        split_frames(T_eff, 128, kSuperTile, &ranges);
        for (auto& r : ranges) items.push_back({b, r.first, r.second, 0, r.second == T_eff});
    }
    void *d_ud, *d_items, *d_omax;
    rc = upload(h, uds.data(), uds.size() * sizeof(UttDesc), s, &d_ud);
    if (rc) return rc;
    rc = upload(h, items.data(), items.size() * sizeof(WorkItem), s, &d_items);
    if (rc) return rc;
    d_omax = arena_alloc(h, batch * sizeof(unsigned));
    if (!d_omax) return fail(h, SETK_ERR_NOMEM, "arena");
    HIP_TRY(h, hipMemsetAsync(d_omax, 0, batch * sizeof(unsigned), s));
    Pass2Args a;
    memset(&a, 0, sizeof(a));
    a.utts = static_cast<const UttDesc*>(d_ud);
    a.items = static_cast<const WorkItem*>(d_items);
    a.window = h->d_window;
    a.synwin = h->d_window;
    a.winsq = h->d_winsq;
    a.tw256 = h->d_tw256;
    a.tw512 = h->d_tw512;
    a.outmax_bits = static_cast<unsigned*>(d_omax);
    a.g = geom_of(h);
    HIP_TRY(h, launch_pass2(1, true, a, (int)items.size(), s));
    if (norm) {
        ScaleArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.utts = a.utts;
        sa.outmax_bits = a.outmax_bits;
        sa.norm_override = static_cast<const float*>(d_norm);
        HIP_TRY(h, launch_scale(sa, batch, L, s));
    }
    rc = copy_back(h, ob, s);
    if (rc) return rc;
    if (ob.host) HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

int setk_covar(setk_handle_t h, const float* spec, const float* mask, int num_channels,
               int num_frames, int num_bins, float* covar, void* stream) {
    if (!h || !spec || !mask || !covar || num_frames <= 0 || num_bins <= 0)
        return fail(h, SETK_ERR_INVALID, "bad args");
    if (num_channels < 1 || num_channels > kMaxChannels16)
        return fail(h, SETK_ERR_UNSUPPORTED, "1 <= num_channels <= 16");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    const int C = num_channels, T = num_frames, F = num_bins;
    const float *d_spec, *d_mask;
    int rc = stage_in(h, spec, (size_t)C * T * F * 2, s, &d_spec);
    if (rc) return rc;
    rc = stage_in(h, mask, (size_t)T * F, s, &d_mask);
    if (rc) return rc;
    OutBuf ob;
    rc = stage_out(h, covar, (size_t)F * C * C * sizeof(float2), &ob);
    if (rc) return rc;
    const int split = std::max(1, std::min(64, (T + 31) / 32));
    const int pitch = ((F + 7) / 8) * 8;
    float* d_part =
        static_cast<float*>(arena_alloc(h, (size_t)split * (2 * npairs(C) + 1) * pitch * 4));
    if (!d_part) return fail(h, SETK_ERR_NOMEM, "arena");
    const int per = (T + split - 1) / split;
    const int used = (T + per - 1) / per;
    HIP_TRY(h, launch_covar_spec(C, d_spec, d_mask, T, F, d_part, split, s));
    HIP_TRY(h, launch_covar_spec_finalize(C, d_part, used, F, static_cast<float*>(ob.dev), s));
    rc = copy_back(h, ob, s);
    if (rc) return rc;
    if (ob.host) HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

static int run_weights(setk_handle_t h, const setk_bf_opts& o, int kind, const float* Rs,
                       const float* Rn, const float* Ry, int F, int C_in, float* weight,
                       int* status, int* ref_out, hipStream_t s) {
    // 8 < C <= 16: embedded in 16 x 16 problems, blkdiag(Rs, 0) / blkdiag(Rn, I): same
    // solution in the first C components, 16 lanes per problem (solve.hip)
    const int C = C_in > kMaxChannels ? kMaxChannels16 : C_in;
    const int NP = npairs(C);
    const int pitch = pitch_of(F);
    const bool mpdr = (kind == SETK_BF_MPDR || kind == SETK_BF_MPDR_WHITEN);
    const int planes = mpdr ? 6 * NP : (Rn ? 4 * NP : 2 * NP);
    arena_reset(h, s);
    const float *d_Rs, *d_Rn = nullptr, *d_Ry = nullptr;
    const size_t nmat = (size_t)F * C_in * C_in * 2;
    int rc = stage_in(h, Rs, nmat, s, &d_Rs);
    if (rc) return rc;
    if (Rn) {
        rc = stage_in(h, Rn, nmat, s, &d_Rn);
        if (rc) return rc;
    }
    if (Ry) {
        rc = stage_in(h, Ry, nmat, s, &d_Ry);
        if (rc) return rc;
    }
    float* d_planes = static_cast<float*>(arena_alloc(h, (size_t)planes * pitch * 4));
    float* d_w = static_cast<float*>(arena_alloc(h, (size_t)C * pitch * 8));
    int* d_bin = static_cast<int*>(arena_alloc(h, (size_t)F * 4));
    if (!d_planes || !d_w || !d_bin) return fail(h, SETK_ERR_NOMEM, "arena");
    HIP_TRY(h, hipMemsetAsync(d_planes, 0, (size_t)planes * pitch * 4, s));
    HIP_TRY(h, launch_pack_covar(d_Rs, F, C_in, C, 0.f, d_planes, 0, s));
    if (d_Rn) HIP_TRY(h, launch_pack_covar(d_Rn, F, C_in, C, 1.f, d_planes, 2 * NP, s));
    if (d_Ry) HIP_TRY(h, launch_pack_covar(d_Ry, F, C_in, C, 1.f, d_planes, 4 * NP, s));
    OutBuf ob;
    rc = stage_out(h, weight, (size_t)F * C_in * sizeof(float2), &ob);
    if (rc) return rc;
    SolveArgs a;
    memset(&a, 0, sizeof(a));
    a.covar = d_planes;
    a.weight = d_w;
    a.bin_status = d_bin;
    a.n_utts = 1;
    a.num_bins = F;
    a.num_channels = C;
    a.planes = planes;
    a.kind = kind;
    a.flags = o.flags;
    a.rank1 = o.rank1;
    a.pmwf_ref = o.pmwf_ref;
    a.pmwf_beta = o.pmwf_beta;
    int* d_ref = nullptr;
    if (kind == SETK_BF_PMWF && o.pmwf_ref < 0) {
        a.snr_acc = static_cast<double*>(arena_alloc(h, (size_t)F * C * 2 * sizeof(double)));
        a.wmat = static_cast<float*>(arena_alloc(h, (size_t)F * C * C * 8));
        d_ref = static_cast<int*>(arena_alloc(h, sizeof(int)));
        if (!a.snr_acc || !a.wmat || !d_ref) return fail(h, SETK_ERR_NOMEM, "arena");
    }
    HIP_TRY(h, launch_solve(a, s));
    if (kind == SETK_BF_PMWF && o.pmwf_ref < 0) HIP_TRY(h, launch_pmwf_select(a, d_ref, s));
    HIP_TRY(h, launch_unpack_weight(d_w, F, C_in, static_cast<float*>(ob.dev), s));
    rc = copy_back(h, ob, s);
    if (rc) return rc;
    if (status) {
        if (is_device_ptr(status))
            HIP_TRY(h, hipMemcpyAsync(status, d_bin, F * 4, hipMemcpyDeviceToDevice, s));
        else
            HIP_TRY(h, hipMemcpyAsync(status, d_bin, F * 4, hipMemcpyDeviceToHost, s));
    }
    if (ref_out) {
        if (d_ref)
            HIP_TRY(h, hipMemcpyAsync(ref_out, d_ref, 4, hipMemcpyDeviceToHost, s));
        else
            *ref_out = o.pmwf_ref;
    }
    if (ob.host || (status && !is_device_ptr(status)) || ref_out)
        HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

int setk_pevd(setk_handle_t h, const float* Rs, const float* Rn, int num_bins, int num_channels,
              int flags, float* pvec, int* status, void* stream) {
    if (!h || !Rs || !pvec || num_bins <= 0) return fail(h, SETK_ERR_INVALID, "bad args");
    if (num_channels < 1 || num_channels > kMaxChannels16)
        return fail(h, SETK_ERR_UNSUPPORTED, "1 <= num_channels <= 16");
    HIP_TRY(h, hipSetDevice(h->device));
    setk_bf_opts o;
    memset(&o, 0, sizeof(o));
    o.flags = flags & SETK_FLAG_NO_GAUGE;
    return run_weights(h, o, kKindPevd, Rs, Rn, nullptr, num_bins, num_channels, pvec, status,
                       nullptr, static_cast<hipStream_t>(stream));
}

int setk_weights(setk_handle_t h, const setk_bf_opts* opts, const float* Rs, const float* Rn,
                 const float* Ry, int num_bins, int num_channels, float* weight, int* status,
                 int* ref_out, void* stream) {
    if (!h || !opts || !Rs || !weight || num_bins <= 0)
        return fail(h, SETK_ERR_INVALID, "bad args");
    if (num_channels < 1 || num_channels > kMaxChannels16)
        return fail(h, SETK_ERR_UNSUPPORTED, "1 <= num_channels <= 16");
    const int kind = opts->kind;
    if (kind < SETK_BF_MVDR || kind > SETK_BF_MPDR_WHITEN)
        return fail(h, SETK_ERR_INVALID, "unknown beamformer kind");
    const bool mpdr = (kind == SETK_BF_MPDR || kind == SETK_BF_MPDR_WHITEN);
    if (mpdr && !Ry) return fail(h, SETK_ERR_INVALID, "MPDR needs Ry");
    if (kind != SETK_BF_MPDR && !Rn) return fail(h, SETK_ERR_INVALID, "Rn is required");
    if (kind == SETK_BF_MPDR && (opts->flags & SETK_FLAG_BAN))
        return fail(h, SETK_ERR_INVALID, "BAN needs a noise covariance (mpdr without whiten)");
    if (kind == SETK_BF_PMWF && opts->pmwf_ref >= num_channels)
        return fail(h, SETK_ERR_INVALID, "Reference channel ID exceeds total channels");
    HIP_TRY(h, hipSetDevice(h->device));
    return run_weights(h, *opts, kind, Rs, Rn, Ry, num_bins, num_channels, weight, status,
                       ref_out, static_cast<hipStream_t>(stream));
}

int setk_ban(setk_handle_t h, const float* weight, const float* Rn, int num_bins,
             int num_channels, float* out, void* stream) {
    if (!h || !weight || !Rn || !out || num_bins <= 0 || num_channels <= 0)
        return fail(h, SETK_ERR_INVALID, "bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    const int F = num_bins, C = num_channels;
    const float *d_w, *d_Rn;
    int rc = stage_in(h, weight, (size_t)F * C * 2, s, &d_w);
    if (rc) return rc;
    rc = stage_in(h, Rn, (size_t)F * C * C * 2, s, &d_Rn);
    if (rc) return rc;
    OutBuf ob;
    rc = stage_out(h, out, (size_t)F * C * sizeof(float2), &ob);
    if (rc) return rc;
    HIP_TRY(h, launch_ban(d_w, d_Rn, F, C, static_cast<float*>(ob.dev), s));
    rc = copy_back(h, ob, s);
    if (rc) return rc;
    if (ob.host) HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

int setk_pcm16_to_float(setk_handle_t h, const int16_t* pcm, int num_channels, int num_samples,
                        float* audio, void* stream) {
    if (!h || !pcm || !audio || num_channels <= 0 || num_samples <= 0)
        return fail(h, SETK_ERR_INVALID, "bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    const size_t n16 = (size_t)num_channels * num_samples;
    const int16_t* d_pcm;
    int rc = stage_in(h, pcm, n16, s, &d_pcm);
    if (rc) return rc;
    OutBuf ob;
    rc = stage_out(h, audio, n16 * sizeof(float), &ob);
    if (rc) return rc;
    HIP_TRY(h, launch_pcm16_to_float(d_pcm, num_channels, num_samples,
                                     static_cast<float*>(ob.dev), s));
    rc = copy_back(h, ob, s);
    if (rc) return rc;
    if (ob.host) HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

int setk_float_to_pcm16(setk_handle_t h, const float* audio, int num_channels, int num_samples,
                        int16_t* pcm, void* stream) {
    if (!h || !pcm || !audio || num_channels <= 0 || num_samples <= 0)
        return fail(h, SETK_ERR_INVALID, "bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    const size_t n = (size_t)num_channels * num_samples;
    const float* d_in;
    int rc = stage_in(h, audio, n, s, &d_in);
    if (rc) return rc;
    OutBuf ob;
    rc = stage_out(h, pcm, n * sizeof(int16_t), &ob);
    if (rc) return rc;
    HIP_TRY(h, launch_float_to_pcm16(d_in, num_channels, num_samples, static_cast<int16_t*>(ob.dev), s));
    rc = copy_back(h, ob, s);
    if (rc) return rc;
    if (ob.host) HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

int setk_pcm16_to_float_batch(setk_handle_t h, int n_utts, int num_channels,
                              const int16_t* const* pcm, const int* num_samples,
                              float* const* audio, double* power0, void* stream) {
    if (!h || n_utts <= 0 || num_channels <= 0 || !pcm || !num_samples || !audio)
        return fail(h, SETK_ERR_INVALID, "bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    std::vector<char> tbl(pcm_item_bytes() * n_utts);
    int max_n = 0;
    for (int u = 0; u < n_utts; ++u) {
        if (!pcm[u] || !audio[u] || num_samples[u] <= 0)
            return fail(h, SETK_ERR_INVALID, "null utterance pointer");
        pcm_item_fill(tbl.data(), u, pcm[u], audio[u], num_samples[u]);
        max_n = std::max(max_n, num_samples[u]);
    }
    void* d_tbl;
    int rc = upload(h, tbl.data(), tbl.size(), s, &d_tbl);
    if (rc) return rc;
    if (power0) HIP_TRY(h, hipMemsetAsync(power0, 0, (size_t)n_utts * sizeof(double), s));
    HIP_TRY(h, launch_pcm16_to_float_batch(d_tbl, n_utts, num_channels, max_n, power0, s));
    return SETK_OK;
}

int setk_pcm16_channel_stride(int num_samples) { return num_samples <= 0 ? 0 : ((num_samples + 7) & ~7); }

int setk_pcm16_deinterleave_batch(setk_handle_t h, int n_utts, int num_channels,
                                  const int16_t* const* pcm, const int* num_samples,
                                  int16_t* const* out, double* power0, void* stream) {
    if (!h || n_utts <= 0 || num_channels <= 0 || !pcm || !num_samples || !out)
        return fail(h, SETK_ERR_INVALID, "bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    std::vector<char> tbl(pcm_item_bytes() * n_utts);
    int max_n = 0;
    for (int u = 0; u < n_utts; ++u) {
        if (!pcm[u] || !out[u] || num_samples[u] <= 0)
            return fail(h, SETK_ERR_INVALID, "null utterance pointer");
        pcm_item_fill_planar(tbl.data(), u, pcm[u], out[u], num_samples[u],
                             setk_pcm16_channel_stride(num_samples[u]));
        max_n = std::max(max_n, num_samples[u]);
    }
    void* d_tbl;
    int rc = upload(h, tbl.data(), tbl.size(), s, &d_tbl);
    if (rc) return rc;
    if (power0) HIP_TRY(h, hipMemsetAsync(power0, 0, (size_t)n_utts * sizeof(double), s));
    HIP_TRY(h, launch_pcm16_deinterleave_batch(d_tbl, n_utts, num_channels, max_n, power0, s));
    return SETK_OK;
}

int setk_kaldi_cm_decode_batch(setk_handle_t h, int n, const int* kinds, const float* vmin, const float* vrange,
                               const int* rows, const int* cols, const int* transpose,
                               const void* const* src, float* const* dst, void* stream) {
    if (!h || n <= 0 || !kinds || !vmin || !vrange || !rows || !cols || !src || !dst)
        return fail(h, SETK_ERR_INVALID, "bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    std::vector<char> tbl(cm_item_bytes() * n);
    long max_elems = 0;
    for (int i = 0; i < n; ++i) {
        if (!src[i] || !dst[i] || rows[i] <= 0 || cols[i] <= 0) return fail(h, SETK_ERR_INVALID, "null or empty matrix");
        if (kinds[i] < SETK_KALDI_CM || kinds[i] > SETK_KALDI_CM3)
            return fail(h, SETK_ERR_UNSUPPORTED, "compressed matrix kind: 1 (CM), 2 (CM2) or 3 (CM3)");
        if (kinds[i] != SETK_KALDI_CM3 && (reinterpret_cast<uintptr_t>(src[i]) & 1))
            return fail(h, SETK_ERR_INVALID, "CM / CM2 bodies must be 2-byte aligned");
        cm_item_fill(tbl.data(), i, src[i], dst[i], vmin[i], vrange[i], rows[i], cols[i], kinds[i],
                     transpose ? transpose[i] : 0);
        max_elems = std::max(max_elems, (long)rows[i] * cols[i]);
    }
    void* d_tbl;
    int rc = upload(h, tbl.data(), tbl.size(), s, &d_tbl);
    if (rc) return rc;
    HIP_TRY(h, launch_kaldi_cm_decode_batch(d_tbl, n, max_elems, s));
    return SETK_OK;
}

int setk_rank1(setk_handle_t h, const float* Rs, const float* Rn, int num_bins,
               int num_channels, float* out, int* status, void* stream) {
    if (!h || !Rs || !out || num_bins <= 0) return fail(h, SETK_ERR_INVALID, "bad args");
    if (num_channels < 1 || num_channels > kMaxChannels16)
        return fail(h, SETK_ERR_UNSUPPORTED, "1 <= num_channels <= 16");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    const int F = num_bins, C = num_channels;
    // principal vectors first (device resident), then the rebuild kernel
    float* d_pv = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&d_pv), (size_t)F * C * sizeof(float2)));
    setk_bf_opts o;
    memset(&o, 0, sizeof(o));
    o.flags = SETK_FLAG_NO_GAUGE;
    int rc = run_weights(h, o, kKindPevd, Rs, Rn, nullptr, F, C, d_pv, status, nullptr, s);
    if (rc == SETK_OK) {
        arena_reset(h, s);
        const float *d_Rs, *d_Rn = nullptr;
        rc = stage_in(h, Rs, (size_t)F * C * C * 2, s, &d_Rs);
        if (rc == SETK_OK && Rn) rc = stage_in(h, Rn, (size_t)F * C * C * 2, s, &d_Rn);
        OutBuf ob;
        if (rc == SETK_OK) rc = stage_out(h, out, (size_t)F * C * C * sizeof(float2), &ob);
        if (rc == SETK_OK) {
            hipError_t e = launch_rank1(d_pv, d_Rs, d_Rn, F, C, static_cast<float*>(ob.dev), s);
            if (e != hipSuccess) rc = fail(h, SETK_ERR_HIP, hipGetErrorString(e));
        }
        if (rc == SETK_OK) rc = copy_back(h, ob, s);
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(d_pv);
    return rc;
}

int setk_beamform(setk_handle_t h, const float* weight, const float* spec, int num_channels,
                  int num_frames, int num_bins, float* out, void* stream) {
    if (!h || !weight || !spec || !out || num_channels <= 0 || num_frames <= 0 || num_bins <= 0)
        return fail(h, SETK_ERR_INVALID, "bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    const int C = num_channels, T = num_frames, F = num_bins;
    const float *d_w, *d_spec;
    int rc = stage_in(h, weight, (size_t)F * C * 2, s, &d_w);
    if (rc) return rc;
    rc = stage_in(h, spec, (size_t)C * T * F * 2, s, &d_spec);
    if (rc) return rc;
    OutBuf ob;
    rc = stage_out(h, out, (size_t)T * F * sizeof(float2), &ob);
    if (rc) return rc;
    HIP_TRY(h, launch_beamform_spec(d_w, d_spec, C, T, F, static_cast<float*>(ob.dev), s));
    rc = copy_back(h, ob, s);
    if (rc) return rc;
    if (ob.host) HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

namespace {
// The bin-resident EM (cgmm_bin.hip) when one bin of the longest utterance fits a CU,
// otherwise (or with SETK_CGMM_STREAMING=1) the streaming kernels of cgmm.hip.
bool cgmm_use_bin(int C, int max_frames) {
    static const bool forced_off = [] {
        const char* e = getenv("SETK_CGMM_STREAMING");
        return e && *e && *e != '0';
    }();
    return !forced_off && cgmm_bin_threads(C, max_frames) != 0;
}

// spec / init / mask / gamma: device pointers per utterance (gamma entries may be null)
// spec == NULL: `audio` / `num_samples` are given instead and the spectrograms are computed
// straight into the bin-major layout (stft_binmajor_kernel, n_fft = 512 plan)
int run_cgmm_bin(setk_handle_t h, int C, int n_utts, const float* const* spec, const int* frames,
                 int F, int num_iters, const float* const* init, float* const* mask,
                 float* const* gamma, int flags, int spec_pitch, hipStream_t s,
                 const float* const* audio = nullptr, const int* num_samples = nullptr) {
    const size_t ab = cgmm_bin_args_bytes();
    std::vector<char> tbl((size_t)n_utts * ab);
    std::vector<const float*> sp(n_utts);
    std::vector<float*> mp(n_utts), gp(n_utts), xbs;
    int max_frames = 0;
    const int nout = gamma ? 2 : 1;
    // diagnostic: SETK_CGMM_TIMING=<file> dumps the per-bin cycle counters of utterance 0
    const char* timing_path = getenv("SETK_CGMM_TIMING");
    void* d_timing = nullptr;
    if (timing_path && *timing_path) {
        d_timing = arena_alloc(h, (size_t)F * cgmm_bin_timing_slots() * sizeof(long long));
        if (d_timing) HIP_TRY(h, hipMemsetAsync(d_timing, 0, (size_t)F * cgmm_bin_timing_slots() * sizeof(long long), s));
    }
    for (int u = 0; u < n_utts; ++u) {
        const int T = frames[u], Tp = cgmm_bin_pitch(T);
        max_frames = std::max(max_frames, T);
        float* xb = static_cast<float*>(arena_alloc(h, (size_t)F * C * Tp * sizeof(float2)));
        float* gb = static_cast<float*>(arena_alloc(h, (size_t)nout * F * Tp * sizeof(float)));
        if (!xb || !gb) return fail(h, SETK_ERR_NOMEM, "arena");
        cgmm_bin_fill_args(tbl.data() + (size_t)u * ab, xb, init ? init[u] : nullptr, gb, T, F,
                           (flags & SETK_CGMM_UPDATE_ALPHA) ? 1 : 0, nout, u == 0 ? d_timing : nullptr);
        sp[u] = spec ? spec[u] : nullptr;
        xbs.push_back(xb);
        mp[u] = mask[u];
        gp[u] = gamma ? gamma[u] : nullptr;
    }
    void *d_tbl, *d_sp = nullptr, *d_mp, *d_gp = nullptr;
    int rc = upload(h, tbl.data(), tbl.size(), s, &d_tbl);
    if (rc) return rc;
    if (spec) {
        rc = upload(h, sp.data(), sp.size() * sizeof(void*), s, &d_sp);
        if (rc) return rc;
    } else {
        // STFT of every utterance into its bin-major array: 64-frame blocks
        std::vector<UttDesc> uds(n_utts);
        std::vector<WorkItem> items;
        for (int u = 0; u < n_utts; ++u) {
            UttDesc& ud = uds[u];
            memset(&ud, 0, sizeof(ud));
            ud.audio = audio[u];
            ud.num_samples = num_samples[u];
            ud.num_frames = frames[u];
            ud.wave_out = xbs[u];
            for (int t0 = 0; t0 < frames[u]; t0 += 64)
                items.push_back({u, t0, std::min(t0 + 64, frames[u]), 0, t0 + 64 >= frames[u]});
        }
        void *d_ud, *d_items;
        rc = upload(h, uds.data(), uds.size() * sizeof(UttDesc), s, &d_ud);
        if (rc) return rc;
        rc = upload(h, items.data(), items.size() * sizeof(WorkItem), s, &d_items);
        if (rc) return rc;
        Pass1Args a;
        memset(&a, 0, sizeof(a));
        a.utts = static_cast<const UttDesc*>(d_ud);
        a.items = static_cast<const WorkItem*>(d_items);
        a.window = h->d_window;
        a.tw256 = h->d_tw256;
        a.tw512 = h->d_tw512;
        a.g = geom_of(h);
        HIP_TRY(h, launch_stft_binmajor(C, a, (int)items.size(), s));
    }
    rc = upload(h, mp.data(), mp.size() * sizeof(void*), s, &d_mp);
    if (rc) return rc;
    if (gamma) {
        rc = upload(h, gp.data(), gp.size() * sizeof(void*), s, &d_gp);
        if (rc) return rc;
    }
    HIP_TRY(h, launch_cgmm_bin(C, d_tbl, static_cast<const float* const*>(d_sp),
                               spec_pitch > 0 ? spec_pitch : F, static_cast<float* const*>(d_mp),
                               static_cast<float* const*>(d_gp), n_utts, F, max_frames, num_iters,
                               nout, s));
    if (d_timing) {
        const int ns = cgmm_bin_timing_slots();
        std::vector<long long> tm((size_t)F * ns);
        HIP_TRY(h, hipMemcpyAsync(tm.data(), d_timing, tm.size() * sizeof(long long),
                                  hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        if (FILE* fp = fopen(timing_path, "w")) {
            fprintf(fp, "# bin | 0 frames 1 - 2 barrier 3 solve 4 tail 5 passes 6 nfast0 7 nfast1 | -DSETK_CGMM_PHASES: 8 E0 9 E1 10 P "
                        "11 R-acc 12 R-sum 13 I-acc 14 I-sum | 16 solve:sums 17 scale 18 chol 19 bound 20 logdet 21 exact-path "
                        "(shader cycles of wave 0, summed over passes)\n");
            for (int f = 0; f < F; ++f) {
                fprintf(fp, "%d", f);
                for (int k = 0; k < ns; ++k) fprintf(fp, " %lld", tm[(size_t)f * ns + k]);
                fprintf(fp, "\n");
            }
            fclose(fp);
        }
    }
    return SETK_OK;
}
}  // namespace

int setk_cgmm_masks_batch(setk_handle_t h, int n_utts, int num_channels,
                          const float* const* spec, const int* num_frames, int num_bins,
                          int num_iters, const float* const* init_mask, float* const* mask_out,
                          int flags, int spec_pitch, void* stream) {
    if (spec_pitch != 0 && spec_pitch < num_bins) return fail(h, SETK_ERR_INVALID, "spec_pitch < F");
    if (!h || n_utts <= 0 || !spec || !num_frames || !mask_out || num_bins <= 0 || num_iters < 0)
        return fail(h, SETK_ERR_INVALID, "bad args");
    if (num_channels < 1 || num_channels > kMaxChannels)
        return fail(h, SETK_ERR_UNSUPPORTED, "1 <= num_channels <= 8");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    const int C = num_channels, F = num_bins;
    const size_t ab = cgmm_args_bytes();
    std::vector<char> tbl((size_t)n_utts * ab);
    int max_frames = 0;
    for (int u = 0; u < n_utts; ++u) {
        if (!spec[u] || !mask_out[u] || num_frames[u] <= 0)
            return fail(h, SETK_ERR_INVALID, "null utterance pointer");
        if (!is_device_ptr(spec[u]) || !is_device_ptr(mask_out[u]) ||
            (init_mask && init_mask[u] && !is_device_ptr(init_mask[u])))
            return fail(h, SETK_ERR_INVALID, "setk_cgmm_masks_batch takes device pointers");
        max_frames = std::max(max_frames, num_frames[u]);
    }
    if (cgmm_use_bin(C, max_frames))
        return run_cgmm_bin(h, C, n_utts, spec, num_frames, F, num_iters, init_mask, mask_out,
                            nullptr, flags, spec_pitch, s);
    for (int u = 0; u < n_utts; ++u) {
        const int T = num_frames[u];
        void* scr = arena_alloc(h, cgmm_scratch_bytes(C, T, F));
        if (!scr) return fail(h, SETK_ERR_NOMEM, "arena");
        cgmm_fill_args(tbl.data() + (size_t)u * ab, C, spec[u], T, F,
                       init_mask ? init_mask[u] : nullptr, nullptr, mask_out[u], scr,
                       (flags & SETK_CGMM_UPDATE_ALPHA) ? 1 : 0, spec_pitch);
    }
    void* d_tbl;
    int rc = upload(h, tbl.data(), tbl.size(), s, &d_tbl);
    if (rc) return rc;
    HIP_TRY(h, launch_cgmm_batch(C, d_tbl, n_utts, F, max_frames, num_iters, s));
    return SETK_OK;
}

int setk_cgmm_estimate_batch(setk_handle_t h, int n_utts, int num_channels,
                             const float* const* audio, const int* num_samples, int num_iters,
                             const float* const* init_mask, float* const* mask_out, int flags,
                             void* stream) {
    if (!h || n_utts <= 0 || !audio || !num_samples || !mask_out || num_iters < 0)
        return fail(h, SETK_ERR_INVALID, "bad args");
    int rc = require_plan512(h);
    if (rc) return rc;
    const int C = num_channels;
    if (C < 1 || C > kMaxChannels) return fail(h, SETK_ERR_UNSUPPORTED, "1 <= num_channels <= 8");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    std::vector<int> frames(n_utts);
    int max_frames = 0;
    for (int u = 0; u < n_utts; ++u) {
        if (!audio[u] || !mask_out[u]) return fail(h, SETK_ERR_INVALID, "null utterance pointer");
        if (!is_device_ptr(audio[u]) || !is_device_ptr(mask_out[u]) ||
            (init_mask && init_mask[u] && !is_device_ptr(init_mask[u])))
            return fail(h, SETK_ERR_INVALID, "setk_cgmm_estimate_batch takes device pointers");
        frames[u] = setk_stft_num_frames(h, num_samples[u]);
        if (frames[u] < 0) return frames[u];
        max_frames = std::max(max_frames, frames[u]);
    }
    if (!cgmm_use_bin(C, max_frames))
        return fail(h, SETK_ERR_UNSUPPORTED,
                    "one bin of the longest utterance does not fit a CU: use setk_stft_batch + "
                    "setk_cgmm_masks_batch (streaming kernels)");
    return run_cgmm_bin(h, C, n_utts, nullptr, frames.data(), kBins, num_iters, init_mask, mask_out,
                        nullptr, flags, 0, s, audio, num_samples);
}

int setk_directional_feats(setk_handle_t h, const float* spec, const float* steer_vector,
                           const int* pairs, int n_pairs, int num_channels, int num_frames,
                           int num_bins, float* out, void* stream) {
    if (!h || !spec || !steer_vector || !pairs || !out || n_pairs <= 0 || num_channels <= 0 ||
        num_frames <= 0 || num_bins <= 0)
        return fail(h, SETK_ERR_INVALID, "bad args");
    for (int p = 0; p < 2 * n_pairs; ++p)
        if (pairs[p] < 0 || pairs[p] >= num_channels)
            return fail(h, SETK_ERR_INVALID, "microphone pair out of range");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    const int C = num_channels, T = num_frames, F = num_bins;
    const float *d_spec, *d_sv;
    int rc = stage_in(h, spec, (size_t)C * T * F * 2, s, &d_spec);
    if (rc) return rc;
    rc = stage_in(h, steer_vector, (size_t)F * C * 2, s, &d_sv);
    if (rc) return rc;
    void* d_pairs;
    rc = upload(h, pairs, (size_t)2 * n_pairs * sizeof(int), s, &d_pairs);
    if (rc) return rc;
    OutBuf ob;
    rc = stage_out(h, out, (size_t)T * F * sizeof(float), &ob);
    if (rc) return rc;
    HIP_TRY(h, launch_directional_feats(d_spec, d_sv, static_cast<const int*>(d_pairs), n_pairs, C, T,
                                        F, static_cast<float*>(ob.dev), s));
    rc = copy_back(h, ob, s);
    if (rc) return rc;
    // (`pairs` went through the handle's page-locked buffer: upload() has copied it)  Host
    // operands are pageable copies queued on the stream: drained before they may be released;
    // with device operands the call is asynchronous like the other stand-alone operators
    if (ob.host || !is_device_ptr(spec) || !is_device_ptr(steer_vector)) HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

int setk_cgmm_masks(setk_handle_t h, const float* spec, int num_channels, int num_frames,
                    int num_bins, int num_iters, const float* init_mask, float* gamma_out,
                    float* mask_out, int flags, void* stream) {
    if (!h || !spec || !mask_out || num_frames <= 0 || num_bins <= 0 || num_iters < 0)
        return fail(h, SETK_ERR_INVALID, "bad args");
    if (num_channels < 1 || num_channels > kMaxChannels)
        return fail(h, SETK_ERR_UNSUPPORTED, "1 <= num_channels <= 8");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    const int C = num_channels, T = num_frames, F = num_bins;
    const float *d_spec, *d_init = nullptr;
    int rc = stage_in(h, spec, (size_t)C * T * F * 2, s, &d_spec);
    if (rc) return rc;
    if (init_mask) {
        rc = stage_in(h, init_mask, (size_t)T * F, s, &d_init);
        if (rc) return rc;
    }
    OutBuf om, og;
    rc = stage_out(h, mask_out, (size_t)T * F * 4, &om);
    if (rc) return rc;
    float* d_gamma = nullptr;
    if (gamma_out) {
        rc = stage_out(h, gamma_out, (size_t)2 * T * F * 4, &og);
        if (rc) return rc;
        d_gamma = static_cast<float*>(og.dev);
    }
    if (cgmm_use_bin(C, T)) {
        float* mo = static_cast<float*>(om.dev);
        rc = run_cgmm_bin(h, C, 1, &d_spec, &T, F, num_iters, d_init ? &d_init : nullptr, &mo,
                          d_gamma ? &d_gamma : nullptr, flags, 0, s);
        if (rc) return rc;
    } else {
        void* d_scr = arena_alloc(h, cgmm_scratch_bytes(C, T, F));
        if (!d_scr) return fail(h, SETK_ERR_NOMEM, "arena");
        std::vector<char> tbl(cgmm_args_bytes());
        cgmm_fill_args(tbl.data(), C, d_spec, T, F, d_init, d_gamma, static_cast<float*>(om.dev),
                       d_scr, (flags & SETK_CGMM_UPDATE_ALPHA) ? 1 : 0, 0);
        void* d_tbl;
        rc = upload(h, tbl.data(), tbl.size(), s, &d_tbl);
        if (rc) return rc;
        HIP_TRY(h, launch_cgmm_batch(C, d_tbl, 1, F, T, num_iters, s));
    }
    rc = copy_back(h, om, s);
    if (rc) return rc;
    if (gamma_out) {
        rc = copy_back(h, og, s);
        if (rc) return rc;
    }
    if (om.host || (gamma_out && og.host)) HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

int setk_cgmm_masks_k(setk_handle_t h, const float* spec, int num_channels, int num_frames,
                      int num_bins, int num_classes, int num_iters, const double* gamma0,
                      const float* init_mask, float* gamma_out, int flags, void* stream) {
    return setk_cgmm_masks_k_status(h, spec, num_channels, num_frames, num_bins, num_classes, num_iters,
                                    gamma0, init_mask, gamma_out, flags, nullptr, stream);
}

int setk_cgmm_masks_k_status(setk_handle_t h, const float* spec, int num_channels, int num_frames,
                             int num_bins, int num_classes, int num_iters, const double* gamma0,
                             const float* init_mask, float* gamma_out, int flags, int* status,
                             void* stream) {
    if (!h || !spec || !gamma_out || num_frames <= 0 || num_bins <= 0 || num_iters < 0)
        return fail(h, SETK_ERR_INVALID, "bad args");
    const int C = num_channels, T = num_frames, F = num_bins, K = num_classes;
    if (!cgmm_k_supported(C, K))
        return fail(h, SETK_ERR_UNSUPPORTED, "general CGMM: 1 <= num_channels <= 16, 2 <= num_classes <= 4");
    if (K != 2 && !gamma0) return fail(h, SETK_ERR_INVALID, "num_classes > 2 needs the start gamma0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    const float *d_spec, *d_init = nullptr, *d_g0f = nullptr;
    int rc = stage_in(h, spec, (size_t)C * T * F * 2, s, &d_spec);
    if (rc) return rc;
    if (gamma0) {  // (stage_in counts floats: a double is two)
        rc = stage_in(h, reinterpret_cast<const float*>(gamma0), (size_t)K * F * T * 2, s, &d_g0f);
        if (rc) return rc;
    } else if (init_mask) {
        rc = stage_in(h, init_mask, (size_t)T * F, s, &d_init);
        if (rc) return rc;
    }
    OutBuf og;
    rc = stage_out(h, gamma_out, (size_t)K * T * F * 4, &og);
    if (rc) return rc;
    double* d_work = static_cast<double*>(arena_alloc(h, cgmm_k_work_bytes(K, T, F)));
    if (!d_work) return fail(h, SETK_ERR_NOMEM, "arena");
    OutBuf os;
    if (status) {
        rc = stage_out(h, status, (size_t)F * sizeof(int), &os);
        if (rc) return rc;
        HIP_TRY(h, hipMemsetAsync(os.dev, 0, (size_t)F * sizeof(int), s));
    }
    HIP_TRY(h, launch_cgmm_k(d_spec, reinterpret_cast<const double*>(d_g0f), d_init, static_cast<float*>(og.dev),
                             d_work, status ? static_cast<int*>(os.dev) : nullptr, C, T, F, K, num_iters,
                             (flags & SETK_CGMM_UPDATE_ALPHA) ? 1 : 0, s));
    rc = copy_back(h, og, s);
    if (rc) return rc;
    if (status) {
        rc = copy_back(h, os, s);
        if (rc) return rc;
    }
    if (og.host || (status && os.host)) HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

int setk_apply_weights_batch(setk_handle_t h, int n_utts, int num_channels,
                             const float* const* audio, const int* num_samples,
                             const float* weights, int n_sets, const int* weight_index,
                             void* const* wave, int flags, void* stream) {
    if (!h || n_utts <= 0 || !audio || !num_samples || !weights || n_sets <= 0 || !wave)
        return fail(h, SETK_ERR_INVALID, "bad args");
    int rc = require_plan512(h);
    if (rc) return rc;
    const int C = num_channels;
    if (C < 1 || C > kMaxChannels) return fail(h, SETK_ERR_UNSUPPORTED, "1 <= num_channels <= 8");
    if (weight_index)
        for (int u = 0; u < n_utts; ++u)
            if (weight_index[u] < 0 || weight_index[u] >= n_sets)
                return fail(h, SETK_ERR_INVALID, "weight index out of range");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    const bool pcm16 = (flags & SETK_FLAG_OUT_PCM16) != 0;
    const StftGeom g = geom_of(h);

    std::vector<UttDesc> uds(n_utts);
    std::vector<WorkItem> items;
    std::vector<int> all_frames(n_utts);
    for (int u = 0; u < n_utts; ++u) {
        all_frames[u] = setk_stft_num_frames(h, num_samples[u]);
        if (all_frames[u] < 0) return all_frames[u];
    }
    const int target = choose_target(all_frames, h->p2_items, kSuperTile, kSuperTile * 4);
    int max_len = 0, max_samples = 0;
    size_t f32_scratch = 0;
    for (int u = 0; u < n_utts; ++u) {
        UttDesc& ud = uds[u];
        memset(&ud, 0, sizeof(ud));
        if (!audio[u] || !wave[u]) return fail(h, SETK_ERR_INVALID, "null utterance pointer");
        ud.audio = audio[u];
        ud.num_samples = num_samples[u];
        ud.num_frames = all_frames[u];
        ud.out_len = setk_istft_num_samples(h, ud.num_frames, -1);
        ud.wave_out = wave[u];
        max_len = std::max(max_len, ud.out_len);
        max_samples = std::max(max_samples, ud.num_samples);
        std::vector<std::pair<int, int>> r;
        split_frames(ud.num_frames, target, kSuperTile, &r);
        for (auto& q : r) items.push_back({u, q.first, q.second, 0, q.second == ud.num_frames});
        f32_scratch += ((size_t)ud.out_len * 4 + 255) & ~(size_t)255;
    }
    float* d_f32 = nullptr;
    if (pcm16) {
        d_f32 = static_cast<float*>(arena_alloc(h, f32_scratch));
        if (!d_f32) return fail(h, SETK_ERR_NOMEM, "arena");
    }
    size_t off = 0;
    for (int u = 0; u < n_utts; ++u) {
        if (pcm16) {
            uds[u].wave_f32 = reinterpret_cast<float*>(reinterpret_cast<char*>(d_f32) + off);
            off += ((size_t)uds[u].out_len * 4 + 255) & ~(size_t)255;
        } else {
            uds[u].wave_f32 = static_cast<float*>(wave[u]);
        }
    }
    void *d_uds_v, *d_items_v, *d_idx_v = nullptr;
    rc = upload(h, uds.data(), uds.size() * sizeof(UttDesc), s, &d_uds_v);
    if (rc) return rc;
    rc = upload(h, items.data(), items.size() * sizeof(WorkItem), s, &d_items_v);
    if (rc) return rc;
    if (weight_index) {
        rc = upload(h, weight_index, (size_t)n_utts * sizeof(int), s, &d_idx_v);
        if (rc) return rc;
    }
    const float* d_sets;
    rc = stage_in(h, weights, (size_t)n_sets * kBins * C * 2, s, &d_sets);
    if (rc) return rc;
    float* d_w = static_cast<float*>(arena_alloc(h, (size_t)n_utts * C * kBinsPad * sizeof(float2)));
    unsigned* d_norm = static_cast<unsigned*>(arena_alloc(h, (size_t)2 * n_utts * sizeof(unsigned)));
    if (!d_w || !d_norm) return fail(h, SETK_ERR_NOMEM, "arena");
    unsigned* d_omax = d_norm + n_utts;
    HIP_TRY(h, hipMemsetAsync(d_norm, 0, (size_t)2 * n_utts * sizeof(unsigned), s));
    const UttDesc* d_uds = static_cast<const UttDesc*>(d_uds_v);
    // norm stays 0 with SETK_FLAG_NO_RENORM: scale_kernel then only converts the sample type
    if (!(flags & SETK_FLAG_NO_RENORM))
        HIP_TRY(h, launch_maxabs(d_uds, C, d_norm, n_utts, max_samples, s));
    HIP_TRY(h, launch_pack_fixed_weights(d_sets, static_cast<const int*>(d_idx_v), n_utts, C, d_w, s));

    Pass2Args p2;
    memset(&p2, 0, sizeof(p2));
    p2.utts = d_uds;
    p2.items = static_cast<const WorkItem*>(d_items_v);
    p2.weight = d_w;
    p2.window = h->d_window;
    p2.synwin = h->d_window;
    p2.winsq = h->d_winsq;
    p2.tw256 = h->d_tw256;
    p2.tw512 = h->d_tw512;
    p2.outmax_bits = d_omax;
    p2.g = g;
    p2.flags = 0;
    HIP_TRY(h, launch_pass2(C, false, p2, (int)items.size(), s));
    ScaleArgs sc;
    memset(&sc, 0, sizeof(sc));
    sc.utts = d_uds;
    sc.norm_bits = d_norm;
    sc.outmax_bits = d_omax;
    sc.pcm16 = pcm16 ? 1 : 0;
    HIP_TRY(h, launch_scale(sc, n_utts, max_len, s));
    // the uploaded descriptors live in the arena: the next call on this handle may
    // reuse it, so this one must have drained
    HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

// wpe_step (libs/wpe.py:58-81) `num_iters` times.  lambda of iteration 0 comes from
// `lambda_enh` (facted_wpd: |previous enhanced|^2) when given, else from
// compute_lambda(spec); later iterations use compute_lambda(dereverb).
namespace {
// n_utts utterances of the same channel count per call: one wpe_step launch per
// iteration covers every (bin, utterance).  lambda_enh / inv_lambda_out: per-utterance arrays
// (facted_wpd) or NULL; lambda_ft only with n_utts == 1 (wpe_step).  status: [n_utts][F] (host
// or device) or NULL.
// fnt: spec / out are in the reference's own layout, F x N x T (libs/wpe.py:84-110) -- which is
// the layout the step kernel works in, so the two transposes fall away
int wpe_batch_impl(setk_handle_t h, int n_utts, const float* const* spec, int num_channels,
                   const int* num_frames, int num_bins, int taps, int delay, int context,
                   int num_iters, const float* const* lambda_enh, const double* lambda_ft,
                   float* const* out, float* const* inv_lambda_out, int* status, void* stream,
                   bool fnt = false) {
    if (!h || n_utts <= 0 || !spec || !out || !num_frames || num_bins <= 0 || num_iters <= 0 ||
        delay < 0 || context < 0)
        return fail(h, SETK_ERR_INVALID, "bad args");
    if (lambda_ft && n_utts != 1)
        return fail(h, SETK_ERR_INVALID, "caller-supplied variances (wpe_step) need n_utts == 1");
    const int C = num_channels, F = num_bins;
    if (!wpe_supported(C, taps))
        return fail(h, SETK_ERR_UNSUPPORTED, wpe_limit_message(C, taps));
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    struct Utt {
        const float* d_spec;
        OutBuf ob;
        float *x_fct, *bufs[2];
        double* lam;
        const float* d_enh;
        OutBuf ob_il;
        int T;
    };
    std::vector<Utt> us(n_utts);
    int rc;
    for (int u = 0; u < n_utts; ++u) {
        Utt& q = us[u];
        q.T = num_frames[u];
        if (!spec[u] || !out[u] || q.T <= 0) return fail(h, SETK_ERR_INVALID, "null utterance");
        const size_t n = (size_t)C * q.T * F;
        rc = stage_in(h, spec[u], n * 2, s, &q.d_spec);
        if (rc) return rc;
        rc = stage_out(h, out[u], n * sizeof(float2), &q.ob);
        if (rc) return rc;
        q.x_fct = fnt ? const_cast<float*>(q.d_spec) : static_cast<float*>(arena_alloc(h, n * sizeof(float2)));
        q.bufs[0] = static_cast<float*>(arena_alloc(h, n * sizeof(float2)));
        q.bufs[1] = static_cast<float*>(arena_alloc(h, n * sizeof(float2)));
        q.lam = static_cast<double*>(arena_alloc(h, (size_t)q.T * F * sizeof(double)));
        if (!q.x_fct || !q.bufs[0] || !q.bufs[1] || !q.lam) return fail(h, SETK_ERR_NOMEM, "arena");
        // F x N x T in and out: the last iteration writes the caller's (or its staged) output
        if (fnt) q.bufs[(num_iters - 1) & 1] = static_cast<float*>(q.ob.dev);
        q.d_enh = nullptr;
        if (lambda_enh && lambda_enh[u]) {
            rc = stage_in(h, lambda_enh[u], (size_t)q.T * F * 2, s, &q.d_enh);
            if (rc) return rc;
        }
        if (inv_lambda_out) {
            if (!inv_lambda_out[u]) return fail(h, SETK_ERR_INVALID, "null inv_lambda_out entry");
            rc = stage_out(h, inv_lambda_out[u], (size_t)q.T * F * sizeof(float), &q.ob_il);
            if (rc) return rc;
        }
    }
    const double* d_lam_in = nullptr;
    if (lambda_ft) {
        rc = stage_in(h, lambda_ft, (size_t)us[0].T * F, s, &d_lam_in);
        if (rc) return rc;
    }
    int* d_st = static_cast<int*>(arena_alloc(h, (size_t)n_utts * F * sizeof(int) * (size_t)num_iters));
    if (!d_st) return fail(h, SETK_ERR_NOMEM, "arena");
    if (!fnt)
        for (int u = 0; u < n_utts; ++u)
            HIP_TRY(h, launch_wpe_transpose(us[u].d_spec, C, us[u].T, F, us[u].x_fct, true, s));
    const size_t ab = wpe_args_bytes();
    std::vector<char> tbl((size_t)n_utts * ab);
    // SETK_WPE_TIMING=<file>: in-kernel cycle counters of the LAST iteration, [n_utts][F][4]
    // int64 (correlation, factorisation, back substitution, filter)
    const char* timing_path = getenv("SETK_WPE_TIMING");
    long long* d_timing = nullptr;
    if (timing_path && *timing_path) {
        d_timing = static_cast<long long*>(arena_alloc(h, (size_t)n_utts * F * 4 * sizeof(long long)));
        if (!d_timing) return fail(h, SETK_ERR_NOMEM, "arena");
    }
    // channels x taps beyond LDS: R in global scratch, [F][NK][NK] complex128 per utterance of a
    // launch; the launches of an iteration then cover as many utterances as ~2 GB of it hold
    const size_t wide_utt = wpe_wide_bytes_per_bin(C, taps) * (size_t)F;
    int per_launch = n_utts;
    char* d_rwork = nullptr;
    if (wide_utt) {
        per_launch = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_utts, ((size_t)2 << 30) / wide_utt));
        d_rwork = static_cast<char*>(arena_alloc(h, wide_utt * per_launch));
        if (!d_rwork) return fail(h, SETK_ERR_NOMEM, "arena");
    }
    for (int it = 0; it < num_iters; ++it) {
        for (int u = 0; u < n_utts; ++u) {
            Utt& q = us[u];
            const float* cur = it == 0 ? q.x_fct : q.bufs[(it - 1) & 1];
            if (it == 0 && d_lam_in)
                // wpe_step (libs/wpe.py:58-81): the caller's variances as given, F x T float64
                HIP_TRY(h, hipMemcpyAsync(q.lam, d_lam_in, (size_t)q.T * F * sizeof(double),
                                          hipMemcpyDeviceToDevice, s));
            else if (it == 0 && q.d_enh)
                HIP_TRY(h, launch_wpe_lambda_from_enh(q.d_enh, q.T, F, q.lam, s));
            else
                HIP_TRY(h, launch_wpe_lambda(cur, C, q.T, F, context, q.lam, s));
            wpe_fill_args(tbl.data() + (size_t)u * ab, q.x_fct, q.lam, q.bufs[it & 1],
                          d_st + ((size_t)it * n_utts + u) * F, C, q.T, taps, delay,
                          d_timing ? d_timing + (size_t)u * F * 4 : nullptr,
                          d_rwork ? d_rwork + wide_utt * (size_t)(u % per_launch) : nullptr);
        }
        void* d_tbl;
        rc = upload(h, tbl.data(), tbl.size(), s, &d_tbl);
        if (rc) return rc;
        for (int u0 = 0; u0 < n_utts; u0 += per_launch)
            HIP_TRY(h, launch_wpe_step_batch(static_cast<char*>(d_tbl) + (size_t)u0 * ab,
                                             std::min(per_launch, n_utts - u0), C, F, taps, s));
    }
    for (int u = 0; u < n_utts; ++u) {
        Utt& q = us[u];
        if (!fnt)
            HIP_TRY(h, launch_wpe_transpose(q.bufs[(num_iters - 1) & 1], C, q.T, F,
                                            static_cast<float*>(q.ob.dev), false, s));
        rc = copy_back(h, q.ob, s);
        if (rc) return rc;
        if (inv_lambda_out) {
            HIP_TRY(h, launch_wpe_inv_lambda(q.lam, q.T, F, static_cast<float*>(q.ob_il.dev), s));
            rc = copy_back(h, q.ob_il, s);
            if (rc) return rc;
        }
    }
    if (status) {
        // worst status over the iterations, per utterance and bin: an error code (1..3) wins
        // over the SETK_NUM_RANKDEF note (4)
        std::vector<int> st((size_t)n_utts * F * num_iters);
        HIP_TRY(h, hipMemcpyAsync(st.data(), d_st, st.size() * sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        std::vector<int> worst((size_t)n_utts * F, 0);
        for (int it = 0; it < num_iters; ++it)
            for (size_t i = 0; i < worst.size(); ++i) {
                const int v = st[(size_t)it * n_utts * F + i], w0 = worst[i];
                const bool ev = v > 0 && v != SETK_NUM_RANKDEF, ew = w0 > 0 && w0 != SETK_NUM_RANKDEF;
                worst[i] = (ev && ew) ? std::max(v, w0) : ev ? v : ew ? w0 : std::max(v, w0);
            }
        if (is_device_ptr(status))
            HIP_TRY(h, hipMemcpy(status, worst.data(), worst.size() * sizeof(int),
                                 hipMemcpyHostToDevice));
        else
            memcpy(status, worst.data(), worst.size() * sizeof(int));
    }
    // descriptors and staged buffers live in the arena: drained before the next call reuses it
    HIP_TRY(h, hipStreamSynchronize(s));
    if (d_timing) {
        std::vector<long long> tm((size_t)n_utts * F * 4);
        HIP_TRY(h, hipMemcpy(tm.data(), d_timing, tm.size() * sizeof(long long), hipMemcpyDeviceToHost));
        if (FILE* fp = fopen(timing_path, "wb")) {
            fwrite(tm.data(), sizeof(long long), tm.size(), fp);
            fclose(fp);
        }
    }
    return SETK_OK;
}
}  // namespace

int setk_wpe(setk_handle_t h, const float* spec, int num_channels, int num_frames, int num_bins,
             int taps, int delay, int context, int num_iters, const float* lambda_enh,
             float* out, float* inv_lambda_out, int* status, void* stream) {
    return wpe_batch_impl(h, 1, &spec, num_channels, &num_frames, num_bins, taps, delay, context,
                          num_iters, lambda_enh ? &lambda_enh : nullptr, nullptr, &out,
                          inv_lambda_out ? &inv_lambda_out : nullptr, status, stream);
}

int setk_wpe_batch_var(setk_handle_t h, int n_utts, const float* const* spec, int num_channels,
                       const int* num_frames, int num_bins, int taps, int delay, int context,
                       int num_iters, const float* const* lambda_enh, float* const* out,
                       float* const* inv_lambda_out, int* status, void* stream) {
    return wpe_batch_impl(h, n_utts, spec, num_channels, num_frames, num_bins, taps, delay, context,
                          num_iters, lambda_enh, nullptr, out, inv_lambda_out, status, stream);
}

int setk_wpe_step(setk_handle_t h, const float* spec, int num_channels, int num_frames,
                  int num_bins, int taps, int delay, const double* lambda_ft, float* out,
                  int* status, void* stream) {
    if (!lambda_ft) return fail(h, SETK_ERR_INVALID, "setk_wpe_step needs lambda");
    return wpe_batch_impl(h, 1, &spec, num_channels, &num_frames, num_bins, taps, delay, 0, 1,
                          nullptr, lambda_ft, &out, nullptr, status, stream);
}

int setk_wpe_batch(setk_handle_t h, int n_utts, const float* const* spec, int num_channels,
                   const int* num_frames, int num_bins, int taps, int delay, int context,
                   int num_iters, float* const* out, int* status, void* stream) {
    return wpe_batch_impl(h, n_utts, spec, num_channels, num_frames, num_bins, taps, delay, context,
                          num_iters, nullptr, nullptr, out, nullptr, status, stream);
}

int setk_wpe_batch_fnt(setk_handle_t h, int n_utts, const float* const* spec, int num_channels,
                       const int* num_frames, int num_bins, int taps, int delay, int context,
                       int num_iters, float* const* out, int* status, void* stream) {
    return wpe_batch_impl(h, n_utts, spec, num_channels, num_frames, num_bins, taps, delay, context,
                          num_iters, nullptr, nullptr, out, nullptr, status, stream, true);
}

int setk_enhance_batch(setk_handle_t h, const setk_bf_opts* opts, int n_utts, int num_channels,
                       const float* const* audio, const int* num_samples,
                       const float* const* mask_s, const float* const* mask_n,
                       void* const* wave, int* status, void* stream) {
    return setk_enhance_batch_taps(h, opts, n_utts, num_channels, audio, num_samples, mask_s,
                                   mask_n, wave, status, nullptr, stream);
}

int setk_enhance_batch_taps(setk_handle_t h, const setk_bf_opts* opts, int n_utts,
                            int num_channels, const float* const* audio, const int* num_samples,
                            const float* const* mask_s, const float* const* mask_n,
                            void* const* wave, int* status, const setk_batch_taps* taps,
                            void* stream) {
    if (!h || !opts || n_utts <= 0 || !audio || !num_samples || !mask_s || !wave)
        return fail(h, SETK_ERR_INVALID, "bad args");
    int rc = require_plan512(h);
    if (rc) return rc;
    const int C = num_channels;
    if (C < 1 || C > kMaxChannels) return fail(h, SETK_ERR_UNSUPPORTED, "1 <= num_channels <= 8");
    const int kind = opts->kind;
    if (kind < SETK_BF_MVDR || kind > SETK_BF_MPDR_WHITEN)
        return fail(h, SETK_ERR_INVALID, "unknown beamformer kind");
    const bool mpdr = (kind == SETK_BF_MPDR || kind == SETK_BF_MPDR_WHITEN);
    if (mpdr && mask_n)
        return fail(h, SETK_ERR_UNSUPPORTED,
                    "fused MPDR derives Ry from mask_s + (1 - mask_s); use the modular API "
                    "with an interferer mask");
    if (kind == SETK_BF_MPDR && (opts->flags & SETK_FLAG_BAN))
        return fail(h, SETK_ERR_INVALID, "BAN needs a noise covariance (mpdr without whiten)");
    if (kind == SETK_BF_PMWF && opts->pmwf_ref >= C)
        return fail(h, SETK_ERR_INVALID, "Reference channel ID exceeds total channels");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    arena_reset(h, s);
    const bool pcm16 = (opts->flags & SETK_FLAG_OUT_PCM16) != 0;
    const bool in_pcm = (opts->flags & SETK_FLAG_IN_PCM16) != 0;
    const int NP = npairs(C);
    const StftGeom g = geom_of(h);

    // ---- descriptors and work lists ----
    std::vector<UttDesc> uds(n_utts);
    std::vector<WorkItem> items1, items2;
    long total_frames = 0;
    int max_len = 0;
    size_t f32_scratch = 0;
    for (int u = 0; u < n_utts; ++u) {
        const int T = setk_stft_num_frames(h, num_samples[u]);
        if (T < 0) return T;
        total_frames += T;
    }
    std::vector<int> all_frames(n_utts);
    for (int u = 0; u < n_utts; ++u) all_frames[u] = setk_stft_num_frames(h, num_samples[u]);
    const int target1 = choose_target(all_frames, h->p1_items, pass1_tile_frames(C), pass1_tile_frames(C) * 8);
    // pass 2 on the matrix cores: hop = n_fft / 2 only (wave-resident overlap-add, pass2_mc.hip)
    const bool mc2 = h->mc_enabled && 2 * g.hop == kNfft && g.keep == 1 &&
                     !(getenv("SETK_MC_PASS2") && atoi(getenv("SETK_MC_PASS2")) == 0);
    if (in_pcm && !mc2)
        return fail(h, SETK_ERR_UNSUPPORTED,
                    "16-bit PCM input (SETK_FLAG_IN_PCM16) needs hop = n_fft / 2 and the matrix-core "
                    "pass 2; convert with setk_pcm16_to_float_batch");
    const int quant2 = mc2 ? 8 : kSuperTile;
    const int target2 = mc2 ? choose_target(all_frames, h->mc_p2_items > 0 ? h->mc_p2_items : h->mc_cus * pass2_mc_wgs_per_cu(C, in_pcm), quant2, 64)
                            : choose_target(all_frames, h->p2_items, kSuperTile, kSuperTile * 4);
    int nparts_total = 0, max_parts = 0;
    for (int u = 0; u < n_utts; ++u) {
        UttDesc& ud = uds[u];
        memset(&ud, 0, sizeof(ud));
        if (!audio[u] || !mask_s[u] || !wave[u] || (mask_n && !mask_n[u]))
            return fail(h, SETK_ERR_INVALID, "null utterance pointer");
        ud.audio = audio[u];
        ud.audio_fmt = in_pcm ? kAudioPcm16 : kAudioF32;
        ud.ch_stride = in_pcm ? setk_pcm16_channel_stride(num_samples[u]) : num_samples[u];
        if (in_pcm && (reinterpret_cast<uintptr_t>(audio[u]) & 3))
            return fail(h, SETK_ERR_INVALID, "16-bit PCM input must be 4-byte aligned");
        ud.mask_s = mask_s[u];
        ud.mask_n = mask_n ? mask_n[u] : nullptr;
        ud.num_samples = num_samples[u];
        ud.num_frames = setk_stft_num_frames(h, num_samples[u]);
        ud.out_len = setk_istft_num_samples(h, ud.num_frames, -1);
        ud.wave_out = wave[u];
        max_len = std::max(max_len, ud.out_len);
        std::vector<std::pair<int, int>> r1, r2;
        split_frames(ud.num_frames, target1, pass1_tile_frames(C), &r1);
        split_frames(ud.num_frames, target2, quant2, &r2);
        ud.part0 = nparts_total;
        ud.nparts = (int)r1.size();
        max_parts = std::max(max_parts, ud.nparts);
        for (auto& r : r1)
            items1.push_back({u, r.first, r.second, nparts_total++, r.second == ud.num_frames});
        for (auto& r : r2) items2.push_back({u, r.first, r.second, 0, r.second == ud.num_frames});
        f32_scratch += ((size_t)ud.out_len * 4 + 255) & ~(size_t)255;
    }
    float* d_f32 = nullptr;
    if (pcm16) {
        d_f32 = static_cast<float*>(arena_alloc(h, f32_scratch));
        if (!d_f32) return fail(h, SETK_ERR_NOMEM, "arena");
    }
    {
        size_t off = 0;
        for (int u = 0; u < n_utts; ++u) {
            if (pcm16) {
                uds[u].wave_f32 = reinterpret_cast<float*>(reinterpret_cast<char*>(d_f32) + off);
                off += ((size_t)uds[u].out_len * 4 + 255) & ~(size_t)255;
            } else {
                uds[u].wave_f32 = static_cast<float*>(wave[u]);
            }
        }
    }
    // descriptors: [uds | items1 | items2], cached on the device while unchanged
    const size_t b_ud = uds.size() * sizeof(UttDesc);
    const size_t b_i1 = items1.size() * sizeof(WorkItem);
    const size_t b_i2 = items2.size() * sizeof(WorkItem);
    std::vector<char> blob(b_ud + b_i1 + b_i2);
    memcpy(blob.data(), uds.data(), b_ud);
    memcpy(blob.data() + b_ud, items1.data(), b_i1);
    memcpy(blob.data() + b_ud + b_i1, items2.data(), b_i2);
    if (blob != h->desc_cache) {
        if (blob.size() > h->d_desc_cap) {
            HIP_TRY(h, hipStreamSynchronize(s));
            if (h->d_desc) (void)hipFree(h->d_desc);
            h->d_desc = nullptr;
            h->d_desc_cap = 0;
            HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_desc), blob.size() * 2));
            h->d_desc_cap = blob.size() * 2;
        }
        // ordered on the stream behind the previous call's kernels (which read the old
        // contents) and ahead of this call's; a previous call on ANOTHER stream was drained
        // by arena_reset.  No host synchronisation: the launching thread runs ahead.
        HIP_TRY(h, h2d_small(h, h->d_desc, blob.data(), blob.size(), s));
        h->desc_cache.swap(blob);
    }
    const UttDesc* d_uds = reinterpret_cast<const UttDesc*>(h->d_desc);
    const WorkItem* d_items1 = reinterpret_cast<const WorkItem*>(h->d_desc + b_ud);
    const WorkItem* d_items2 = reinterpret_cast<const WorkItem*>(h->d_desc + b_ud + b_i1);

    // ---- scratch ----
    const int planes_out = mpdr ? 6 * NP : 4 * NP;
    float* d_part = static_cast<float*>(
        arena_alloc(h, (size_t)nparts_total * nplanes_partial(C) * kBinsPad * 4));
    float* d_covar = static_cast<float*>(arena_alloc(h, (size_t)n_utts * planes_out * kBinsPad * 4));
    float* d_w = static_cast<float*>(arena_alloc(h, (size_t)n_utts * C * kBinsPad * 8));
    unsigned* d_small = static_cast<unsigned*>(arena_alloc(h, (size_t)n_utts * 3 * 4));
    if (!d_part || !d_covar || !d_w || !d_small) return fail(h, SETK_ERR_NOMEM, "arena");
    unsigned* d_norm = d_small;
    unsigned* d_omax = d_small + n_utts;
    int* d_status = reinterpret_cast<int*>(d_small + 2 * n_utts);
    HIP_TRY(h, hipMemsetAsync(d_small, 0, (size_t)n_utts * 3 * 4, s));

    const bool prof = h->profiling;
    if (prof) {
        for (int i = 0; i < 5; ++i) {
            hipEvent_t e;
            if (!h->ev_pool.empty()) {
                e = h->ev_pool.back();
                h->ev_pool.pop_back();
            } else {
                HIP_TRY(h, hipEventCreate(&e));
            }
            h->ev[i] = e;
            h->ev_used.push_back(e);
        }
        HIP_TRY(h, hipEventRecord(h->ev[0], s));
    }

    // ---- stage 1: STFT + covariance partials (timed alone), then finalize ----
    Pass1Args p1;
    memset(&p1, 0, sizeof(p1));
    p1.utts = d_uds;
    p1.items = d_items1;
    p1.partials = d_part;
    p1.window = h->d_window;
    p1.tw256 = h->d_tw256;
    p1.tw512 = h->d_tw512;
    p1.norm_bits = d_norm;
    p1.g = g;
    p1.flags = opts->flags;
    p1.mc_tab = h->d_mc_tab;
    p1.mc_win = h->d_mc_win;
    // pass 1 on the matrix cores is opt-in (SETK_MC_PASS1=1): parity-green, but its transform
    // waves are the long pole of the tile pipeline (0.95 ms against 0.88)
    const bool mc1 = !in_pcm && h->mc_enabled && pass1_mc_supported(C, g.hop) && getenv("SETK_MC_PASS1") &&
                     atoi(getenv("SETK_MC_PASS1")) != 0;
    if (in_pcm) p1.window = h->d_window_pcm;
    if (mc1)
        HIP_TRY(h, launch_pass1_mc(C, p1, (int)items1.size(), s));
    else
        HIP_TRY(h, launch_pass1(C, false, p1, (int)items1.size(), s, in_pcm));
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[1], s));
    FinalizeArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.utts = d_uds;
    fa.partials = d_part;
    fa.covar = d_covar;
    fa.num_channels = C;
    fa.with_ry = mpdr ? 1 : 0;
    fa.num_scale = mc1 ? (float)((h->mc_peak / 1024.0) * (h->mc_peak / 1024.0)) : 1.f;
    // With a few slabs per utterance (the shard of the bench: two) the solve sums them itself and
    // this launch -- 26 us of an 87 us stage, mostly launch and tail -- falls away.  Not when the
    // covariances are tapped, not for PMWF's reference search (pmwf_select_kernel reads Rn back
    // for BAN), not for long utterances (32 slabs: the parallel reduction is the better one).
    const bool fuse_reduce = !(taps && (taps->Rs || taps->Rn)) && max_parts <= 4 &&
                             !(kind == SETK_BF_PMWF && opts->pmwf_ref < 0) &&
                             !(getenv("SETK_FUSED_REDUCE") && atoi(getenv("SETK_FUSED_REDUCE")) == 0);
    if (!fuse_reduce) HIP_TRY(h, launch_finalize(fa, n_utts, s));
    OutBuf tap_rs, tap_rn, tap_w;
    if (taps && taps->Rs) {
        rc = stage_out(h, taps->Rs, (size_t)n_utts * kBins * C * C * sizeof(float2), &tap_rs);
        if (rc) return rc;
        HIP_TRY(h, launch_unpack_covar(d_covar, n_utts, planes_out, 0, kBins, C,
                                       static_cast<float*>(tap_rs.dev), s));
        rc = copy_back(h, tap_rs, s);
        if (rc) return rc;
    }
    if (taps && taps->Rn) {
        rc = stage_out(h, taps->Rn, (size_t)n_utts * kBins * C * C * sizeof(float2), &tap_rn);
        if (rc) return rc;
        HIP_TRY(h, launch_unpack_covar(d_covar, n_utts, planes_out, 2 * NP, kBins, C,
                                       static_cast<float*>(tap_rn.dev), s));
        rc = copy_back(h, tap_rn, s);
        if (rc) return rc;
    }

    // ---- stage 2: weights ----
    SolveArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.covar = d_covar;
    sa.weight = d_w;
    sa.status = d_status;
    sa.n_utts = n_utts;
    sa.num_bins = kBins;
    sa.num_channels = C;
    sa.planes = planes_out;
    sa.kind = kind;
    sa.flags = opts->flags;
    sa.rank1 = opts->rank1;
    sa.pmwf_ref = opts->pmwf_ref;
    sa.pmwf_beta = opts->pmwf_beta;
    sa.partials = fuse_reduce ? d_part : nullptr;
    sa.utts = d_uds;
    sa.num_scale = fa.num_scale;
    if (kind == SETK_BF_PMWF && opts->pmwf_ref < 0) {
        sa.snr_acc =
            static_cast<double*>(arena_alloc(h, (size_t)n_utts * kBins * C * 2 * sizeof(double)));
        sa.wmat = static_cast<float*>(arena_alloc(h, (size_t)n_utts * kBins * C * C * 8));
        if (!sa.snr_acc || !sa.wmat) return fail(h, SETK_ERR_NOMEM, "arena");
    }
    HIP_TRY(h, launch_solve(sa, s));
    if (kind == SETK_BF_PMWF && opts->pmwf_ref < 0) HIP_TRY(h, launch_pmwf_select(sa, nullptr, s));
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[2], s));
    if (taps && taps->weight) {
        rc = stage_out(h, taps->weight, (size_t)n_utts * kBins * C * sizeof(float2), &tap_w);
        if (rc) return rc;
        HIP_TRY(h, launch_unpack_weight_batch(d_w, n_utts, kBins, C,
                                              static_cast<float*>(tap_w.dev), s));
        rc = copy_back(h, tap_w, s);
        if (rc) return rc;
    }

    // ---- stage 3: beamform + iSTFT ----
    Pass2Args p2;
    memset(&p2, 0, sizeof(p2));
    p2.utts = d_uds;
    p2.items = d_items2;
    p2.weight = d_w;
    p2.window = h->d_window;
    p2.synwin = h->d_window;
    p2.winsq = h->d_winsq;
    p2.tw256 = h->d_tw256;
    p2.tw512 = h->d_tw512;
    p2.outmax_bits = d_omax;
    p2.norm_bits = d_norm;
    p2.g = g;
    p2.flags = opts->flags;
    p2.mc_tab = h->d_mc_tab;
    p2.mc_win = h->d_mc_win;
    p2.mc_syn = h->d_mc_syn;
    p2.mc_edge = h->d_mc_edge;
    if (mc2)
        HIP_TRY(h, launch_pass2_mc(C, p2, (int)items2.size(), s, in_pcm));
    else
        HIP_TRY(h, launch_pass2(C, false, p2, (int)items2.size(), s));
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[3], s));

    // ---- stage 4: renorm ----
    ScaleArgs sc;
    memset(&sc, 0, sizeof(sc));
    sc.utts = d_uds;
    sc.norm_bits = d_norm;
    sc.outmax_bits = d_omax;
    sc.pcm16 = pcm16 ? 1 : 0;
    HIP_TRY(h, launch_scale(sc, n_utts, max_len, s));
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[4], s));
    if (taps && taps->maxabs) {
        // max |audio| per utterance (WaveReader.maxabs): the float bit patterns
        if (is_device_ptr(taps->maxabs))
            HIP_TRY(h, hipMemcpyAsync(taps->maxabs, d_norm, (size_t)n_utts * 4,
                                      hipMemcpyDeviceToDevice, s));
        else
            HIP_TRY(h, hipMemcpyAsync(taps->maxabs, d_norm, (size_t)n_utts * 4,
                                      hipMemcpyDeviceToHost, s));
    }
    const bool status_dev = status && is_device_ptr(status);
    if (status)
        HIP_TRY(h, hipMemcpyAsync(status, d_status, (size_t)n_utts * 4,
                                  status_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    if ((status && !status_dev) || tap_rs.host || tap_rn.host || tap_w.host ||
        (taps && taps->maxabs && !is_device_ptr(taps->maxabs)))
        HIP_TRY(h, hipStreamSynchronize(s));
    return SETK_OK;
}

}  // extern "C"
