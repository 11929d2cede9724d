// csrc/ssd_aql.hip -- the library's own dispatch path: AQL packets written straight into HSA queues it owns.
//
// Why.  A rollout at the named batch (4096 envs) is a chain of dependent ~4 us kernels: hipLaunchKernel costs 2.3-3 us of host
// time per launch behind a runtime lock (two launches per step), a short ssd_rollout_random call (rollout.py:58-70 called per
// training iteration) paid 50-250 us of runtime bookkeeping on top, and nothing about the packets could be chosen.  An AQL
// kernel-dispatch packet is 64 bytes; writing one and ringing the doorbell takes ~0.15 us.  So ssd_rollout_random writes its step
// launches itself (ssd_capi.hip: rollout_aql); this file is the mechanism:
//   * the HSA runtime is the one the process already runs on (the HIP runtime's dependency, found with RTLD_NOLOAD);
//   * queues: a pool of at most kPoolQueues HSA user-mode queues per DEVICE, shared by all handles (queues are scarce, see
//     queue_create), one per chain of a rollout;
//   * the code object is the one hipcc built for the HIP path (the offload bundle of ssd_kernels.o, embedded a second time as
//     plain data by ssd_codeobj.S), loaded through the HSA executable API; a kernel is looked up by the name HIP reports for the
//     same __global__ stub (hipKernelNameRefByPtr), so both paths run the very same instantiation;
//   * kernel arguments are the caller's business (ssd_capi.hip keeps them static, in device memory): dispatch() takes a pointer;
//   * ordering against the caller's HIP stream: FORK -- a one-wave HIP kernel on the stream zeroes an HSA signal that a
//     barrier-AND packet at the head of every chain waits on; JOIN -- each chain ends with a one-wave dispatch (barrier bit,
//     system-scope release) that bumps a counter in device memory, which a one-wave kernel on the stream sleeps on
//     (hipStreamWaitValue64 instead made the command processor poll host memory for the stream's queue, and that slowed the
//     dispatch queues sharing its micro-engine: 6.7 against 6.06 us per step); join_and_wait() is the synchronous form for tools
//     that serialise kernels.
// Packets within a chain carry the barrier bit; their fence scopes are the caller's choice (agent / agent = what the HIP runtime
// writes for its own kernel packets; none for the coherent kernel variants).
//
// Nothing here computes anything: if the HSA runtime, a queue or the code object cannot be set up, aql::available() is false
// and the rollout calls issue the same launches through hipLaunchKernel.  SSD_AQL=0 forces that.
// Queues are a scarce, process-wide resource (below, "the cliff"): every queue of the pool is PROBED when it is created -- a burst of
// one-wave dispatches on it and a burst of HIP launches on a stream, against the figures measured before the pool existed -- and a
// queue whose arrival slows either down is destroyed again; the pool then stays at the size that was fine.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/amd_hsa_signal.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "ssd_internal.hpp"
#include "ssd_aql.hpp"
#include "ssd_agent_match.hpp"

extern "C" const unsigned char ssd_kernels_bundle[];        // ssd_codeobj.S: the clang offload bundle of ssd_kernels.o
extern "C" const unsigned char ssd_kernels_bundle_end[];

namespace ssd {
namespace aql {

namespace {

#define SSD_HSA_FUNCS(X)                                                                                                  \
    X(hsa_init) X(hsa_iterate_agents) X(hsa_agent_get_info) X(hsa_amd_agent_iterate_memory_pools)                         \
    X(hsa_amd_memory_pool_get_info) X(hsa_amd_memory_pool_allocate) X(hsa_amd_memory_pool_free)                           \
    X(hsa_amd_agents_allow_access) X(hsa_queue_create) X(hsa_queue_destroy) X(hsa_queue_load_read_index_scacquire)         \
    X(hsa_queue_store_write_index_screlease) X(hsa_signal_create) X(hsa_signal_destroy) X(hsa_signal_store_screlease)      \
    X(hsa_signal_store_relaxed) X(hsa_signal_load_relaxed) X(hsa_code_object_reader_create_from_memory)                    \
    X(hsa_code_object_reader_destroy) X(hsa_executable_create_alt) X(hsa_executable_load_agent_code_object)                \
    X(hsa_executable_freeze) X(hsa_executable_get_symbol_by_name) X(hsa_executable_symbol_get_info) X(hsa_status_string)           \
    X(hsa_signal_wait_scacquire)

constexpr int kPoolQueues = 3;                              // what the product ever holds per device
constexpr int kPoolSlots = 8;                               // (test-hook build: SSD_AQL_POOL_MAX lifts the limit up to here, to find the cliff)
#ifndef SSD_ARCH
#define SSD_ARCH "gfx950"                                   // (the Makefile passes its ARCH)
#endif

struct Api {
#define X(f) decltype(&::f) f = nullptr;
    SSD_HSA_FUNCS(X)
#undef X
};
Api g_api;

struct DeviceCtx {
    bool tried = false, ok = false;
    int device = -1;
    hsa_agent_t gpu{}, cpu{};
    hsa_amd_memory_pool_t host_kernarg_pool{};
    hsa_executable_t exe{};
    std::unordered_map<const void *, Kernel> kernels;     // by host stub
    std::string why;                                      // why not ok
    // the device's dispatch queues, shared by every handle on it (queues are scarce: see queue_create), and what they share
    struct Queue *pool[kPoolSlots] = {};
    std::mutex enqueue_mu;                                // one rollout call writes packets at a time
    uint32_t *abort_host = nullptr;                       // host memory (device-mapped): a queue of the device reported an error
    void *abort_dev = nullptr;
    // the probe (probe_pool_queue): what a burst of HIP launches / of dispatches on a pool queue cost before the pool grew
    unsigned long long *probe_counter = nullptr;          // device memory the probe's one-wave dispatches bump
    void *probe_kernarg = nullptr;
    unsigned long long probe_count = 0;                   // value probe_counter reaches when every probe dispatch so far has run
    double hip_burst_base_us = 0, hip_burst_last_us = 0, q_burst_base_us = 0, q_burst_last_us = 0;
    int pool_cap = kPoolSlots;                            // shrinks when a new queue fails its probe
    int dropped = 0;                                      // queues destroyed again by the probe
    const char *matched_by = "";                          // how the HSA agent was matched to the HIP device
    std::string agent_pci;                                // ... and that agent's PCI address (domain:bus:device.function)
    long long *hip_scratch = nullptr;                     // device memory (THIS device's) the probe's HIP burst writes
};
std::mutex g_mu;
bool g_api_tried = false, g_api_ok = false;
DeviceCtx g_dev[64];
bool g_verbose = false;

void say(const std::string &m) {
    if (g_verbose) fprintf(stderr, "[ssd aql] %s\n", m.c_str());
}

bool load_api() {
    if (g_api_tried) return g_api_ok;
    g_api_tried = true;
    { const char *v = getenv("SSD_AQL_VERBOSE"); g_verbose = v && atoi(v) != 0; }
    // the HSA runtime the process already runs on (the HIP runtime's dependency): never a second copy
    void *h = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libhsa-runtime64.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) { say("HSA runtime not loaded in this process"); return false; }
    bool all = true;
#define X(f) g_api.f = reinterpret_cast<decltype(&::f)>(dlsym(h, #f)); if (!g_api.f) { all = false; say(std::string("missing ") + #f); }
    SSD_HSA_FUNCS(X)
#undef X
    if (!all) return false;
    if (g_api.hsa_init() != HSA_STATUS_SUCCESS) { say("hsa_init failed"); return false; }   // reference-counted: HIP holds one already
    g_api_ok = true;
    return true;
}

// Every HSA agent as a plain record (ssd_agent_match.hpp decides on those), with its handle beside it.
struct AgentList { std::vector<AgentRecord> rec; std::vector<hsa_agent_t> handle; };
hsa_status_t agent_cb(hsa_agent_t a, void *data) {
    auto *L = static_cast<AgentList *>(data);
    AgentRecord r;
    hsa_device_type_t type;
    if (g_api.hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    r.type = type == HSA_DEVICE_TYPE_CPU ? kAgentCpu : type == HSA_DEVICE_TYPE_GPU ? kAgentGpu : kAgentOther;
    if (r.type == kAgentGpu) {
        r.has_bdf = g_api.hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &r.bdf) == HSA_STATUS_SUCCESS &&
                    g_api.hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &r.domain) == HSA_STATUS_SUCCESS;
        char uuid[64] = {};
        if (g_api.hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_UUID, uuid) == HSA_STATUS_SUCCESS) std::memcpy(r.uuid, uuid, sizeof r.uuid);
        char name[64] = {};
        if (g_api.hsa_agent_get_info(a, HSA_AGENT_INFO_NAME, name) == HSA_STATUS_SUCCESS) { std::memcpy(r.name, name, sizeof r.name - 1); }
        (void)g_api.hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_COMPUTE_UNIT_COUNT, &r.cu_count);
    }
    L->rec.push_back(r); L->handle.push_back(a);
    return HSA_STATUS_SUCCESS;
}
hsa_status_t pool_cb(hsa_amd_memory_pool_t pool, void *data) {
    hsa_amd_segment_t seg;
    if (g_api.hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL)
        return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    g_api.hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    bool can_alloc = false;
    g_api.hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &can_alloc);
    if (can_alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT)) {
        *static_cast<hsa_amd_memory_pool_t *>(data) = pool;
        return HSA_STATUS_INFO_BREAK;
    }
    return HSA_STATUS_SUCCESS;
}

// the gfx950 code object inside the embedded clang offload bundle: magic, u64 count, then per entry u64 offset, u64 size,
// u64 triple length, triple
bool find_code_object(const unsigned char **co, size_t *size) {
    const unsigned char *b = ssd_kernels_bundle;
    const size_t n = (size_t)(ssd_kernels_bundle_end - ssd_kernels_bundle);
    static const char magic[] = "__CLANG_OFFLOAD_BUNDLE__";
    if (n < 32 || std::memcmp(b, magic, 24) != 0) return false;
    uint64_t cnt;
    std::memcpy(&cnt, b + 24, 8);
    size_t pos = 32;
    for (uint64_t i = 0; i < cnt; ++i) {
        if (pos + 24 > n) return false;
        uint64_t off, sz, tl;
        std::memcpy(&off, b + pos, 8); std::memcpy(&sz, b + pos + 8, 8); std::memcpy(&tl, b + pos + 16, 8);
        pos += 24;
        if (pos + tl > n) return false;
        const std::string triple(reinterpret_cast<const char *>(b + pos), (size_t)tl);
        pos += tl;
        if (triple.find("amdgcn-amd-amdhsa") != std::string::npos && triple.find(SSD_ARCH) != std::string::npos && off + sz <= n && sz > 0) {
            *co = b + off; *size = (size_t)sz;
            return true;
        }
    }
    return false;
}

DeviceCtx *device_ctx(int device) {
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx &c = g_dev[device];
    if (c.tried) return c.ok ? &c : nullptr;
    c.tried = true; c.device = device;
    static const bool off = SSD_KNOB("SSD_AQL", 1) == 0;
    if (off) { c.why = "SSD_AQL=0"; return nullptr; }
    if (!load_api()) { c.why = "HSA runtime unavailable"; return nullptr; }
    auto fail = [&](const std::string &m) -> DeviceCtx * { c.why = m; say(m); return nullptr; };
    // the HSA agent behind HIP device `device` (the rule and its tests: ssd_agent_match.hpp, tests/test_agent_match_cpu.py)
    DeviceRecord d;
    d.ordinal = device;
    {
        int dom = 0, bus = 0, dev = 0;
        d.has_pci = hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, device) == hipSuccess &&
                    hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device) == hipSuccess &&
                    hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, device) == hipSuccess;
        d.domain = (uint32_t)dom; d.bus = (uint32_t)bus; d.dev = (uint32_t)dev;
        hipUUID u{};                                  // (HIP's 16-byte device uuid is the hex digits of the HSA agent's "GPU-<hex>" string)
        if (hipDeviceGetUuid(&u, device) == hipSuccess && u.bytes[0]) {
            std::memcpy(d.uuid, "GPU-", 4);
            std::memcpy(d.uuid + 4, u.bytes, 16);
        }
        hipDeviceProp_t prop{};
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
            std::strncpy(d.arch, prop.gcnArchName, sizeof d.arch - 1);
            d.cu_count = (uint32_t)prop.multiProcessorCount;
        }
        (void)hipGetLastError();
        // (the n-th GPU agent is the HIP device n only while nothing filters or reorders the devices HIP shows)
        d.filtered = getenv("HIP_VISIBLE_DEVICES") || getenv("CUDA_VISIBLE_DEVICES") || getenv("ROCR_VISIBLE_DEVICES");
    }
    AgentList agents;
    g_api.hsa_iterate_agents(agent_cb, &agents);
    const AgentMatch m = match_agent(agents.rec, d);
    if (m.gpu < 0 || m.cpu < 0) return fail(m.why);
    c.matched_by = m.by;
    c.gpu = agents.handle[(size_t)m.gpu]; c.cpu = agents.handle[(size_t)m.cpu];
    {   // what bench.py prints per rank: the PCI address of the device the rank steps on, and the agent's
        char buf[96];
        std::snprintf(buf, sizeof buf, "%04x:%02x:%02x.%x", agents.rec[(size_t)m.gpu].domain, agents.rec[(size_t)m.gpu].bdf >> 8,
                      (agents.rec[(size_t)m.gpu].bdf >> 3) & 31u, agents.rec[(size_t)m.gpu].bdf & 7u);
        c.agent_pci = buf;
    }
    c.host_kernarg_pool.handle = 0;
    g_api.hsa_amd_agent_iterate_memory_pools(c.cpu, pool_cb, &c.host_kernarg_pool);
    if (!c.host_kernarg_pool.handle) return fail("no kernarg memory pool");
    const unsigned char *co = nullptr; size_t co_size = 0;
    if (!find_code_object(&co, &co_size)) return fail("embedded code object not found");
    hsa_code_object_reader_t reader;
    hsa_status_t st = g_api.hsa_code_object_reader_create_from_memory(co, co_size, &reader);
    if (st != HSA_STATUS_SUCCESS) return fail("code object reader failed");
    st = g_api.hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &c.exe);
    if (st != HSA_STATUS_SUCCESS) return fail("hsa_executable_create_alt failed");
    st = g_api.hsa_executable_load_agent_code_object(c.exe, c.gpu, reader, nullptr, nullptr);
    if (st != HSA_STATUS_SUCCESS) return fail("loading the code object failed");
    st = g_api.hsa_executable_freeze(c.exe, nullptr);
    if (st != HSA_STATUS_SUCCESS) return fail("hsa_executable_freeze failed");
    c.ok = true;
    say("device " + std::to_string(device) + ": own AQL dispatch path ready (HSA agent " + c.agent_pci + " matched by " + c.matched_by + ")");
    return &c;
}

}  // namespace

bool available(int device) { return device_ctx(device) != nullptr; }

// Is a profiling / tracing tool attached to this process?  Such tools may run kernels ONE AT A TIME (rocprofv3 --pmc does): a
// kernel that waits for another queue's kernel -- the stream-side wait of the join, the chains' wait for the fork -- would then
// never end.  With a tool attached the rollout calls therefore use host-side waits ("sync mode": the call itself waits for the
// stream before and for the chains after; no kernel waits for another).  SSD_AQL_SYNC=1 / 0 forces / forbids that.
// What counts as attached: the variables the ROCm tools are started with, or their libraries already loaded in the process.
int tool_attached_now() {
    static const char *const vars[] = {"ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROF_COUNTER_COLLECTION", "ROCPROFILER_METRICS_PATH",
                                       "ROCPROF_ATT_LIBRARY_PATH"};
    for (const char *n : vars) { const char *v = getenv(n); if (v && *v) return 1; }
    if (const char *pl = getenv("LD_PRELOAD")) { if (std::strstr(pl, "rocprof") || std::strstr(pl, "roctracer") || std::strstr(pl, "omnitrace") || std::strstr(pl, "rocsys")) return 1; }
    static const char *const libs[] = {"librocprofiler-sdk.so.1", "librocprofiler-sdk.so", "librocprofiler-sdk-tool.so", "librocprofiler64.so.2",
                                       "librocprofiler64.so.1", "librocprofiler64.so"};
    for (const char *l : libs)
        if (void *h = dlopen(l, RTLD_LAZY | RTLD_NOLOAD)) { dlclose(h); return 1; }
    return 0;
}
bool sync_mode() {
    static const bool v = [] {
        const char *e = getenv("SSD_AQL_SYNC");
        if (e && *e) return atoi(e) != 0;
        return tool_attached_now() != 0;
    }();
    return v;
}

const char *why_not(int device) {
    if (device < 0 || device >= 64) return "bad device";
    return g_dev[device].why.c_str();
}

bool lookup(int device, const void *host_fn, Kernel *out) {
    DeviceCtx *c = device_ctx(device);
    if (!c || !host_fn) return false;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = c->kernels.find(host_fn);
    if (it != c->kernels.end()) { *out = it->second; return out->object != 0; }
    Kernel k{};
    const char *name = hipKernelNameRefByPtr(host_fn, nullptr);
    if (name && *name) {
        const std::string sym = std::string(name) + ".kd";
        hsa_executable_symbol_t s;
        if (g_api.hsa_executable_get_symbol_by_name(c->exe, sym.c_str(), &c->gpu, &s) == HSA_STATUS_SUCCESS) {
            uint64_t obj = 0; uint32_t ka = 0, grp = 0, prv = 0;
            g_api.hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &obj);
            g_api.hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &ka);
            g_api.hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &grp);
            g_api.hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &prv);
            k.object = obj; k.kernarg_size = ka; k.group_static = grp; k.private_size = prv;
        } else {
            say("kernel symbol not found: " + sym);
        }
    } else {
        say("hipKernelNameRefByPtr gave no name");
    }
    c->kernels[host_fn] = k;
    *out = k;
    return k.object != 0;
}

// ------------------------------------------------------------------------------------------------------------------
struct Queue {
    DeviceCtx *ctx = nullptr;
    hsa_queue_t *q = nullptr;
    uint64_t widx = 0;                  // next packet index to write (this library is the queue's only producer)
    uint64_t rung = 0;                  // packets below this index have been handed to the doorbell
    uint32_t mask = 0;
    std::atomic<int> error{0};
    uint32_t *abort_flag = nullptr;             // host memory the waiting kernels look at: set on a queue error
    Kernel flag_kernel{};
    bool attach_signal = false;
    hsa_signal_t done_signal{};                 // completion signal of the join packet in synchronous mode (join_and_wait)
};

static void queue_error_cb(hsa_status_t status, hsa_queue_t *, void *data) {
    auto *Q = static_cast<Queue *>(data);
    Q->error.store((int)status ? (int)status : -1);
    if (Q->abort_flag) *reinterpret_cast<volatile uint32_t *>(Q->abort_flag) = 1;
    const char *msg = nullptr;
    if (g_api.hsa_status_string) g_api.hsa_status_string(status, &msg);
    fprintf(stderr, "[ssd aql] queue error: %s\n", msg ? msg : "?");
}

static void queue_destroy(Queue *Q) {
    if (!Q) return;
    if (Q->q) g_api.hsa_queue_destroy(Q->q);
    if (Q->done_signal.handle) g_api.hsa_signal_destroy(Q->done_signal);
    delete Q;
}

static Queue *queue_create(DeviceCtx *c) {
    Queue *Q = new Queue();
    Q->ctx = c;
    static const uint32_t qsize = [] { const int n = SSD_HOOK("SSD_AQL_QUEUE_SIZE", 4096); return (uint32_t)(n >= 64 ? n : 4096); }();
    uint32_t size = 64;
    while (size < qsize) size <<= 1;
    // (a process's total matters: with 4 queues of the library's beside the HIP runtime's -- even idle ones -- every launch of the
    // process slowed to ~30 us on this device, the hardware scheduler then time-slices its queue slots; 3 were fine.  Hence ONE
    // pool of at most kPoolQueues queues per device, shared by all handles; rollouts use 2, or 3 from 6144 envs)
    if (g_api.hsa_queue_create(c->gpu, size, HSA_QUEUE_TYPE_SINGLE, queue_error_cb, Q, UINT32_MAX, UINT32_MAX, &Q->q) != HSA_STATUS_SUCCESS) {
        say("hsa_queue_create failed");
        delete Q;
        return nullptr;
    }
    if (g_api.hsa_signal_create(0, 0, nullptr, &Q->done_signal) != HSA_STATUS_SUCCESS) Q->done_signal.handle = 0;
    Q->mask = Q->q->size - 1;
    Q->widx = Q->rung = g_api.hsa_queue_load_read_index_scacquire(Q->q);
    Q->abort_flag = c->abort_host;
    if (!lookup(c->device, flag_kernel_fn(), &Q->flag_kernel)) {
        say("flag kernel not found");
        queue_destroy(Q);
        return nullptr;
    }
    return Q;
}

// How many queues of its own the library may hold.  A process gets about FOUR hardware queues before the hardware scheduler starts
// time-slicing them (every dispatch then waits for its queue's turn: a 20-step rollout that follows an RCCL barrier 320 us instead of
// 130; measured: the HIP runtime's queues in use + the library's = 4 fine, 5 not -- with 4 queues of the library's beside the
// runtime's, idle ones included, EVERY launch of the process slowed to ~30 us).  The HIP runtime maps its streams onto at most
// GPU_MAX_HW_QUEUES queues (default 4, created as streams need them).  The rule (documented in include/ssd.h):
//   * SSD_AQL_QUEUES (1..3) sets the pool's size;
//   * else a process that sets GPU_MAX_HW_QUEUES -- bench.py does, to 2, as a rank of a process group -- has told us how many the
//     runtime takes: the pool gets 4 minus that, at least 1;
//   * else TWO: a host application with a few torch streams and RCCL's has room for them, not for three;
//   * and whatever the rule says, a queue that fails its probe when it is created (below) is destroyed again and the pool stays
//     at the size that was fine (pool_size(device)).
static int pool_limit() {
    static const int n = [] {
        int v = 2;
        if (const char *h = getenv("GPU_MAX_HW_QUEUES")) { const int hq = atoi(h); if (hq >= 1) v = 4 - hq; }
        if (const char *o = getenv("SSD_AQL_QUEUES")) v = atoi(o);
        int top = SSD_HOOK("SSD_AQL_POOL_MAX", kPoolQueues);
        top = top < 1 ? 1 : top > kPoolSlots ? kPoolSlots : top;
        return v < 1 ? 1 : v > top ? top : v;
    }();
    return n;
}
int pool_size(int device) {
    if (device < 0 || device >= 64) return 1;
    return g_dev[device].pool_cap < pool_limit() ? g_dev[device].pool_cap : pool_limit();      // (0: not even one queue fits)
}
bool over_the_cliff(int device) { return device >= 0 && device < 64 && g_dev[device].pool_cap == 0; }

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// A burst on ALL of `qs` at once -- n one-wave dispatches each (barrier bit: one after the other within a queue), interleaved
// so that every queue has work from the start -- joined the way a rollout call is joined: a one-wave kernel on the null stream
// polls the counter the dispatches bump.  Microseconds from the first packet to the stream's completion; < 0 on failure.
// This is the situation the cliff is about: the device runs about FOUR queues at a time (one per pipe of the command
// processor's compute micro-engine); a fifth ACTIVE queue is time-sliced against the others, and a rollout whose chains are so
// sliced takes twice as long (measured, 6144 envs: 3 chains 6.7 us per step, 4 chains 12.2 - 13.5, 5: 15, 8: 27 - 37 -- with the
// joining kernel's stream that is 4, 5, 6, 9 active queues; idle queues cost nothing, and a burst on ONE queue shows nothing).
static double concurrent_burst_us(DeviceCtx *c, Queue *const *qs, int nq, int n) {
    if (!c->probe_kernarg || !c->probe_counter) return -1;
    const double t0 = now_us();
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < nq; ++i) dispatch(qs[i], qs[i]->flag_kernel, 1, 64, 0, c->probe_kernarg, true, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE);
    for (int i = 0; i < nq; ++i) ring(qs[i]);
    c->probe_count += (unsigned long long)nq * (unsigned long long)n;
    launch_wait_counter_kernel(c->probe_counter, c->probe_count, static_cast<const uint32_t *>(c->abort_dev), 100000000ull /* 1 s */, nullptr, nullptr, nullptr);
    if (hipStreamSynchronize(nullptr) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return now_us() - t0;
}
// A burst of one-wave HIP launches on the null stream + synchronize; microseconds, < 0 on failure.
// (the word the burst's kernels write belongs to the context's own device -- ADVICE r03: one process-wide word made a second
// device's probe store into the first device's memory, a fault without peer access and a burst timed over xGMI with it)
static double hip_burst_us(DeviceCtx *c, int n) {
    if (!c->hip_scratch) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != c->device) { (void)hipGetLastError(); return -1; }   // (the caller is on the handle's device)
        void *ptr = nullptr;
        if (hipMalloc(&ptr, 8) != hipSuccess) { (void)hipGetLastError(); return -1; }
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, ptr) != hipSuccess || at.device != c->device) { (void)hipGetLastError(); (void)hipFree(ptr); return -1; }
        c->hip_scratch = static_cast<long long *>(ptr);
    }
    long long *scratch = c->hip_scratch;
    const double t0 = now_us();
    for (int i = 0; i < n; ++i) launch_signal_kernel(scratch, nullptr);
    if (hipStreamSynchronize(nullptr) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return now_us() - t0;
}
// THE PROBE.  Called with the device idle (the first rollout call of a handle synchronises anyway).  Before the pool's first queue
// exists it records what a burst of HIP launches costs.  After every new queue: (a) the concurrent burst above on the whole pool
// including the new queue, against the figure of the pool's first queue alone; (b) the HIP burst again.  The new queue must go
// when (a) exceeds 1.6 x the first queue's figure + 10 us (concurrent chains would be time-sliced: the chains it would carry are
// better off in fewer queues), or when (b) exceeds 2.5 x its base + 20 us (the process's own launches suffer); the first queue
// itself when its burst exceeds 100 us -- an MI355X calibration: 16 dependent one-wave dispatches + the join take 48 - 56 us on a
// queue that has a pipe to itself and 138 - 145 us on one that is time-sliced (gpurun_out/r03c/queue_probe3.txt) -- i.e. the
// process is past the cliff before the library came (a host application with four busy streams): then the library holds no queue
// at all and steps on the caller's own stream.
// Medians of 5 bursts of 16.  Returns false when the new queue must go.
static constexpr int kBurst = 16, kBurstReps = 5;
static bool probe_pool_queue(DeviceCtx *c, Queue *Q, int index) {
    static const bool probe_on = SSD_HOOK("SSD_AQL_PROBE", 1) != 0;
    if (!probe_on) return true;
    // (a profiling tool is attached: it may run kernels one at a time -- the burst's waiting kernel would sit out its time bound --
    // and its timings say nothing about the unprofiled process: no probe, the rule's pool size stands)
    // (whatever SSD_AQL_SYNC says about the waits: a tracer's per-dispatch overhead alone makes the first queue look time-sliced)
    if (sync_mode() || tool_attached_now()) { say("probe skipped: a profiling tool is attached"); return true; }
    Queue *qs[kPoolSlots];
    int nq = 0;
    for (int i = 0; i < index && i < kPoolSlots - 1; ++i) if (c->pool[i]) qs[nq++] = c->pool[i];
    qs[nq++] = Q;
    (void)concurrent_burst_us(c, qs, nq, 2);              // (first dispatches of a new queue: not timed)
    // (the verdict is permanent for the process, so one noisy moment -- another handle's rollout in flight, a co-tenant --
    // must not decide it: a queue that fails is measured once more after the device has drained, and only a second failure counts.
    // The first queue's bound stays the absolute MI355X figure: measured against the process's own HIP burst it would vanish in
    // exactly the case it is for -- a process past the cliff already has a slow HIP burst too)
    bool ok = true;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (attempt == 1 && hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
        double qv[kBurstReps], hv[kBurstReps];
        for (double &x : qv) x = concurrent_burst_us(c, qs, nq, kBurst);
        for (double &x : hv) x = hip_burst_us(c, kBurst);
        std::sort(qv, qv + kBurstReps); std::sort(hv, hv + kBurstReps);
        const double q_us = qv[0] < 0 ? -1 : qv[kBurstReps / 2], h_us = hv[0] < 0 ? -1 : hv[kBurstReps / 2];
        c->q_burst_last_us = q_us; c->hip_burst_last_us = h_us;
        if (index == 0) c->q_burst_base_us = q_us;
        char msg[360];
        snprintf(msg, sizeof msg, "probe%s: %d queue(s) at once, %d dispatches each: %.1f us (the first queue alone %.1f); %d HIP launches %.1f us (before the pool %.1f)",
                 attempt ? " (second look)" : "", nq, kBurst, q_us, c->q_burst_base_us, kBurst, h_us, c->hip_burst_base_us);
        say(msg);
        if (q_us < 0 || h_us < 0 || c->hip_burst_base_us <= 0) return true;       // (no figures: no verdict)
        const bool hip_slow = h_us > 2.5 * c->hip_burst_base_us + 20.0;
        const bool q_slow = index > 0 ? (c->q_burst_base_us > 0 && q_us > 1.6 * c->q_burst_base_us + 10.0) : (q_us > 100.0);
        ok = !(hip_slow || q_slow);
        if (ok) break;
    }
    return ok;
}

// Queue `index` (0 .. pool_size(device) - 1) of the device's pool, created -- and probed -- on first use; nullptr if that fails or
// the probe turns the queue down (the pool then stays smaller for the life of the process).
Queue *pool_queue(int device, int index) {
    DeviceCtx *c = device_ctx(device);
    if (!c || index < 0 || index >= pool_size(device)) return nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!c->abort_host) {
            void *h = nullptr, *d = nullptr;
            if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            c->abort_host = static_cast<uint32_t *>(h); c->abort_dev = d;
            *c->abort_host = 0;
        }
        if (c->pool[index]) return c->pool[index];
    }
    std::lock_guard<std::mutex> probe_lk(c->enqueue_mu);            // (one creation + probe at a time; nobody writes packets meanwhile)
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (c->pool[index]) return c->pool[index];
        if (index >= (c->pool_cap < pool_limit() ? c->pool_cap : pool_limit())) return nullptr;
    }
    if (!c->probe_counter) {
        void *ptr = nullptr;
        if (hipMalloc(&ptr, 8) == hipSuccess && hipMemset(ptr, 0, 8) == hipSuccess) {
            c->probe_counter = static_cast<unsigned long long *>(ptr);
            c->probe_kernarg = host_kernarg_alloc(device, 64);
            if (c->probe_kernarg) std::memcpy(c->probe_kernarg, &c->probe_counter, sizeof(void *));
        }
        (void)hipGetLastError();
        // what HIP launches cost in this process before the library holds any queue of its own
        (void)hip_burst_us(c, 2);
        double v[kBurstReps];
        for (double &x : v) x = hip_burst_us(c, kBurst);
        std::sort(v, v + kBurstReps);
        c->hip_burst_base_us = v[0] < 0 ? 0 : v[kBurstReps / 2];
    }
    Queue *Q = queue_create(c);                     // (takes g_mu itself, in lookup())
    if (Q && !probe_pool_queue(c, Q, index)) {
        say("queue " + std::to_string(index) + " slows the process down (hardware-queue cliff): destroyed, the pool stays at " + std::to_string(index));
        if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
        queue_destroy(Q);
        Q = nullptr;
        std::lock_guard<std::mutex> lk(g_mu);
        c->pool_cap = index;
        c->dropped++;
        // (not even one queue of the library's fits beside what the process already holds: no own dispatch path on this device,
        // the rollout calls go through hipLaunchKernel)
        if (index == 0) { c->ok = false; c->why = "the process is past the hardware-queue cliff already: no dispatch queue of the library's own"; }
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    c->pool[index] = Q;
    return Q;
}
// bits for ssd_rollout_path(): the size the pool settled on (<< 12) | "a queue failed its probe and was destroyed" (32) | how the
// HSA agent was matched to the HIP device (<< 16: 1 PCI address, 2 UUID, 3 ordinal; 0: no agent, no own dispatch path)
int pool_report(int device) {
    if (device < 0 || device >= 64) return 0;
    const char *by = g_dev[device].matched_by;
    const int how = !g_dev[device].ok ? 0 : by[0] == 'P' ? 1 : by[0] == 'U' ? 2 : by[0] == 'o' ? 3 : 0;
    return (pool_size(device) << 12) | (g_dev[device].dropped ? 32 : 0) | (how << 16);
}
void probe_figures(int device, double out[4]) {
    const DeviceCtx &c = g_dev[device];
    out[0] = c.hip_burst_base_us; out[1] = c.hip_burst_last_us; out[2] = c.q_burst_base_us; out[3] = c.q_burst_last_us;
}
std::mutex &enqueue_mutex(int device) { return g_dev[device].enqueue_mu; }
const uint32_t *abort_flag_dev(int device) { return static_cast<const uint32_t *>(g_dev[device].abort_dev); }

// A small block of host kernarg memory the device can read (a handle's flag-kernel argument).
void *host_kernarg_alloc(int device, size_t bytes) {
    DeviceCtx *c = device_ctx(device);
    void *p = nullptr;
    if (!c || g_api.hsa_amd_memory_pool_allocate(c->host_kernarg_pool, bytes, 0, &p) != HSA_STATUS_SUCCESS) return nullptr;
    if (g_api.hsa_amd_agents_allow_access(1, &c->gpu, nullptr, p) != HSA_STATUS_SUCCESS) { g_api.hsa_amd_memory_pool_free(p); return nullptr; }
    std::memset(p, 0, bytes);
    return p;
}
void host_kernarg_free(void *p) { if (p && g_api_ok) g_api.hsa_amd_memory_pool_free(p); }

bool queue_failed(const Queue *Q) { return Q->error.load() != 0; }

// Slot for the next packet.  Packets of a chain carry the barrier bit, so "packet j + 1 has been consumed" implies "packet j
// has completed"; waiting until fewer than size - 2 packets are outstanding keeps a slot's previous occupant finished.
static inline void *next_slot(Queue *Q) {
    const uint64_t idx = Q->widx;
    // (no doorbell ever covers packets on both sides of the ring buffer's end: under rocprofv3 --kernel-trace the runtime's intercept
    // queue hands the tool the new packets as one range and the tool read past the end of the buffer -- SIGSEGV at the buffer's
    // page-aligned end inside hsa_signal_store on the doorbell, round 4, whenever a batch of the every-fourth-step doorbells happened
    // to straddle the wrap; one doorbell more per trip round the ring)
    if ((idx & Q->mask) == 0 && Q->rung != idx) ring(Q);
    if (idx - g_api.hsa_queue_load_read_index_scacquire(Q->q) >= (uint64_t)Q->mask - 1) {
        ring(Q);                                        // (the consumer must see what we have written before we wait for it)
        while (idx - g_api.hsa_queue_load_read_index_scacquire(Q->q) >= (uint64_t)Q->mask - 1 && !Q->error.load())
            __builtin_ia32_pause();
    }
    return static_cast<uint8_t *>(Q->q->base_address) + (idx & Q->mask) * 64;
}

static inline void publish(void *pk, uint16_t header, uint16_t setup) {
    __atomic_store_n(static_cast<uint32_t *>(pk), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
}

constexpr uint16_t header_of(int type, bool barrier, int acquire, int release) {
    return (uint16_t)((type << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                      (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
}

void dispatch(Queue *Q, const Kernel &k, uint32_t grid_x, uint32_t block_x, uint32_t lds_dynamic, const void *kernarg_dev,
              bool barrier, int acquire_scope, int release_scope) {
    auto *pk = static_cast<hsa_kernel_dispatch_packet_t *>(next_slot(Q));
    pk->workgroup_size_x = (uint16_t)block_x; pk->workgroup_size_y = 1; pk->workgroup_size_z = 1; pk->reserved0 = 0;
    pk->grid_size_x = grid_x * block_x; pk->grid_size_y = 1; pk->grid_size_z = 1;
    pk->private_segment_size = k.private_size;
    pk->group_segment_size = k.group_static + lds_dynamic;
    pk->kernel_object = k.object;
    pk->kernarg_address = const_cast<void *>(kernarg_dev);
    pk->reserved2 = 0;
    pk->completion_signal.handle = Q->attach_signal ? Q->done_signal.handle : 0;
    publish(pk, header_of(HSA_PACKET_TYPE_KERNEL_DISPATCH, barrier, acquire_scope, release_scope), 1 /* dimensions */);
    Q->widx++;
}

void barrier_and(Queue *Q, uint64_t dep_signal_handle) {
    auto *pk = static_cast<hsa_barrier_and_packet_t *>(next_slot(Q));
    std::memset(reinterpret_cast<uint8_t *>(pk) + 4, 0, 60);
    pk->dep_signal[0].handle = dep_signal_handle;
    publish(pk, header_of(HSA_PACKET_TYPE_BARRIER_AND, true, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE), 0);
    Q->widx++;
}

void ring(Queue *Q) {
    if (Q->rung == Q->widx) return;
    g_api.hsa_queue_store_write_index_screlease(Q->q, Q->widx);
    g_api.hsa_signal_store_screlease(Q->q->doorbell_signal, (hsa_signal_value_t)(Q->widx - 1));
    Q->rung = Q->widx;
}

uint64_t write_index(const Queue *Q) { return Q->widx; }
uint64_t read_index(const Queue *Q) { return g_api.hsa_queue_load_read_index_scacquire(Q->q); }

// JOIN: after everything enqueued so far on Q, bump a join counter (`flag_kernarg`: host kernarg block holding the counter's
// device address); the packet's system-scope release makes the rollout's results visible to everybody.  (The caller's stream
// waits for the counter with ssd_wait_counter_kernel.)
void join(Queue *Q, const void *flag_kernarg) {
    // (measured: the packet without fences where the chain's stores were all write-through -- the waiting kernel's own end-of-kernel
    // release would do -- saves nothing: 6.63 against 6.56 us per step of the driver's 20-step call)
    dispatch(Q, Q->flag_kernel, 1, 64, 0, flag_kernarg, /*barrier=*/true, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_SYSTEM);
    ring(Q);
}

// Synchronous join (SSD_AQL_SYNC=1: profiling with serialised kernels, where a kernel that waits for another queue's kernel
// would wait forever): the join packet carries a completion signal and the HOST waits for it.
// (max_seconds > 0: give up after that long -- the drain behind a join that timed out must not hang on a queue that is stuck)
bool join_and_wait(Queue *Q, const void *flag_kernarg, double max_seconds) {
    if (!Q->done_signal.handle) return false;
    g_api.hsa_signal_store_screlease(Q->done_signal, 1);
    Q->attach_signal = true;
    join(Q, flag_kernarg);
    Q->attach_signal = false;
    const double t0 = now_us();
    while (g_api.hsa_signal_wait_scacquire(Q->done_signal, HSA_SIGNAL_CONDITION_LT, 1, 1000000, HSA_WAIT_STATE_BLOCKED) >= 1) {
        if (Q->error.load()) return false;
        if (max_seconds > 0 && now_us() - t0 > max_seconds * 1e6) return false;
    }
    return true;
}

// FORK signals
uint64_t signal_create(long long initial) {
    hsa_signal_t s{};
    if (!g_api_ok || g_api.hsa_signal_create(initial, 0, nullptr, &s) != HSA_STATUS_SUCCESS) return 0;
    return s.handle;
}
void signal_destroy(uint64_t h) {
    if (h && g_api_ok) { hsa_signal_t s; s.handle = h; g_api.hsa_signal_destroy(s); }
}
void signal_set(uint64_t h, long long v) {
    hsa_signal_t s; s.handle = h;
    g_api.hsa_signal_store_screlease(s, v);
}
long long *signal_value_ptr(uint64_t h) {
    // an hsa_signal_t handle is the address of its amd_signal_t; the value word is what packets and shaders poll / write
    return reinterpret_cast<long long *>(const_cast<int64_t *>(&reinterpret_cast<amd_signal_t *>(h)->value));
}

}  // namespace aql
}  // namespace ssd
