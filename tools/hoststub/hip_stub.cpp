// A host-memory stand-in for the 34 HIP runtime entry points libsetk_hip.so imports, for the
// sanitizer test of the library's HOST side (tools/hoststub/build.sh, tests/test_host_asan.py).
// Test infrastructure only: nothing in the product links or loads it.
//
//   * "device" memory is calloc'ed host memory kept POISONED for AddressSanitizer, so host code
//     that dereferences a device pointer is reported; the copy / memset entry points check
//     that the device side of every transfer lies inside ONE live allocation (the bound a
//     real GPU would not check) and the host side through ASAN's own memcpy interceptor;
//   * kernel launches do nothing, but their configuration is validated against the gfx950
//     limits (block <= 1024 threads, grid.y / grid.z <= 65535, non-empty grid, dynamic LDS
//     <= 160 KB and <= what hipFuncSetAttribute granted for that kernel);
//   * every launch and every violation is counted; hoststub_report() hands the counters to
//     the test.
#include <hip/hip_runtime_api.h>
#include <sanitizer/asan_interface.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
std::mutex g_mu;
std::map<char*, size_t> g_allocs;          // live "device" allocations
std::map<const void*, int> g_lds_limit;    // kernel -> dynamic LDS granted
// kernel host stub -> device symbol; filled by the library's static constructors, which may
// run before this file's, hence constructed on first use
std::map<const void*, std::string>& names() {
    static auto* m = new std::map<const void*, std::string>();
    return *m;
}
const char* kname(const void* f) {
    auto it = names().find(f);
    return it == names().end() ? "?" : it->second.c_str();
}
long g_launches = 0, g_violations = 0, g_copies = 0;
char g_first[512] = "";

struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local std::vector<CallCfg> t_cfg;

void violation(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void violation(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    fprintf(stderr, "hoststub: VIOLATION: %s\n", buf);
    if (!g_violations) snprintf(g_first, sizeof g_first, "%s", buf);
    ++g_violations;
}

// the allocation holding [p, p+n), or nullptr
const std::pair<char* const, size_t>* owner(const void* p, size_t n) {
    char* c = const_cast<char*>(static_cast<const char*>(p));
    auto it = g_allocs.upper_bound(c);
    if (it == g_allocs.begin()) return nullptr;
    --it;
    if (c >= it->first && c + n <= it->first + it->second) return &*it;
    return nullptr;
}
bool is_device(const void* p) {
    char* c = const_cast<char*>(static_cast<const char*>(p));
    auto it = g_allocs.upper_bound(c);
    if (it == g_allocs.begin()) return false;
    --it;
    return c >= it->first && c < it->first + it->second;
}

hipError_t do_copy(void* dst, const void* src, size_t n, const char* what) {
    if (!n) return hipSuccess;
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_copies;
    bool dd = is_device(dst), sd = is_device(src);
    if (dd && !owner(dst, n)) violation("%s: destination [%p, +%zu) leaves its device allocation", what, dst, n);
    if (sd && !owner(src, n)) violation("%s: source [%p, +%zu) leaves its device allocation", what, src, n);
    if ((dd && !owner(dst, n)) || (sd && !owner(src, n))) return hipErrorInvalidValue;
    if (dd) ASAN_UNPOISON_MEMORY_REGION(dst, n);
    if (sd) ASAN_UNPOISON_MEMORY_REGION(src, n);
    memmove(dst, src, n);   // the host side is checked by ASAN's interceptor
    if (dd) ASAN_POISON_MEMORY_REGION(dst, n);
    if (sd) ASAN_POISON_MEMORY_REGION(src, n);
    return hipSuccess;
}
}  // namespace

extern "C" {

void hoststub_report(long* launches, long* violations, long* copies, long* live, char* first, int cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    *launches = g_launches;
    *violations = g_violations;
    *copies = g_copies;
    *live = static_cast<long>(g_allocs.size());
    if (first && cap > 0) snprintf(first, cap, "%s", g_first);
}

// HOSTSTUB_DEVICES: how many devices to pretend to have (default 1; the two-rank CLI test: 2)
static int device_count() {
    const char* e = getenv("HOSTSTUB_DEVICES");
    const int n = e ? atoi(e) : 1;
    return n > 0 ? n : 1;
}
hipError_t hipGetDeviceCount(int* n) { *n = device_count(); return hipSuccess; }
hipError_t hipSetDevice(int d) { return (d >= 0 && d < device_count()) ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hoststub error"; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

// a bus id no sysfs entry answers to: setk_amd/numa.py then reports "node unknown" and binds nothing
hipError_t hipDeviceGetPCIBusId(char* out, int len, int d) {
    if (len < 13 || d < 0 || d >= device_count()) return hipErrorInvalidValue;
    snprintf(out, (size_t)len, "ffff:%02x:00.0", d & 0xff);
    return hipSuccess;
}

hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "hoststub gfx950");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = 288ull << 30;
    p->sharedMemPerBlock = 160 << 10;
    p->maxThreadsPerBlock = 1024;
    p->warpSize = 64;
    return hipSuccess;
}

hipError_t hipMalloc(void** p, size_t n) {
    if (!n) { *p = nullptr; return hipSuccess; }
    char* c = static_cast<char*>(calloc(1, n));
    if (!c) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lk(g_mu);
    g_allocs[c] = n;
    ASAN_POISON_MEMORY_REGION(c, n);
    *p = c;
    return hipSuccess;
}

hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_allocs.find(static_cast<char*>(p));
    if (it == g_allocs.end()) {
        violation("hipFree(%p): not a live device allocation", p);
        return hipErrorInvalidValue;
    }
    ASAN_UNPOISON_MEMORY_REGION(p, it->second);
    g_allocs.erase(it);
    free(p);
    return hipSuccess;
}

hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { return do_copy(d, s, n, "hipMemcpy"); }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    return do_copy(d, s, n, "hipMemcpyAsync");
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
    if (!n) return hipSuccess;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!owner(d, n)) {
        violation("hipMemsetAsync: [%p, +%zu) is not inside one device allocation", d, n);
        return hipErrorInvalidValue;
    }
    ASAN_UNPOISON_MEMORY_REGION(d, n);
    memset(d, v, n);
    ASAN_POISON_MEMORY_REGION(d, n);
    return hipSuccess;
}

hipError_t hipPointerGetAttributes(hipPointerAttribute_t* at, const void* p) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!is_device(p)) return hipErrorInvalidValue;   // what the runtime says of plain host memory
    memset(at, 0, sizeof *at);
    at->type = hipMemoryTypeDevice;
    at->devicePointer = const_cast<void*>(p);
    return hipSuccess;
}

// page-locked host memory is ordinary (ASAN-tracked) heap here
hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
    *p = calloc(1, n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p) {
    free(p);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    *s = reinterpret_cast<hipStream_t>(new int(0));
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    delete reinterpret_cast<int*>(s);
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return e ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
    *e = reinterpret_cast<hipEvent_t>(new int(0));
    return hipSuccess;
}

hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }

hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(new int(0)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<int*>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute a, int v) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (a == hipFuncAttributeMaxDynamicSharedMemorySize) {
        if (v > (160 << 10)) violation("hipFuncSetAttribute: %d bytes of dynamic LDS requested (> 160 KB)", v);
        g_lds_limit[f] = v;
    }
    return hipSuccess;
}
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 2; return hipSuccess; }

hipError_t hipLaunchKernel(const void* f, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t) {
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_launches;
    size_t threads = size_t(block.x) * block.y * block.z;
    if (!grid.x || !grid.y || !grid.z) violation("launch %s: empty grid (%u, %u, %u)", kname(f), grid.x, grid.y, grid.z);
    if (grid.y > 65535u || grid.z > 65535u) violation("launch %s: grid (%u, %u, %u) over 65535 in y/z", kname(f), grid.x, grid.y, grid.z);
    if (!threads || threads > 1024) violation("launch %s: %zu threads per workgroup", kname(f), threads);
    if (uint64_t(grid.x) * block.x > 0xffffffffull) violation("launch %s: grid.x * block.x overflows 32 bits", kname(f));
    auto it = g_lds_limit.find(f);
    size_t cap = it == g_lds_limit.end() ? (64u << 10) : size_t(it->second);
    if (shmem > cap) violation("launch %s: %zu bytes of dynamic LDS, %zu granted", kname(f), shmem, cap);
    if (!args) violation("launch %s: null argument array", kname(f));
    return hipSuccess;
}

hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t s) {
    t_cfg.push_back({grid, block, shmem, s});
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* s) {
    if (t_cfg.empty()) return hipErrorInvalidValue;
    CallCfg c = t_cfg.back();
    t_cfg.pop_back();
    *grid = c.grid; *block = c.block; *shmem = c.shmem; *s = c.stream;
    return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipRegisterFunction(void**, const void* host_fn, char*, const char* device_name, unsigned, void*, void*, void*,
                           void*, int*) {
    names()[host_fn] = device_name ? device_name : "?";   // static constructors: single-threaded
}
void __hipUnregisterFatBinary(void**) {}

}  // extern "C"
