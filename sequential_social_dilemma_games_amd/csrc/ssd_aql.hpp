// csrc/ssd_aql.hpp -- the library's own AQL dispatch path (ssd_aql.hip): HSA queues it owns, packets it writes.
#pragma once
#include <stdint.h>

#include <mutex>

namespace ssd {
namespace aql {

struct Kernel { uint64_t object = 0; uint32_t kernarg_size = 0, group_static = 0, private_size = 0; };
struct Queue;

bool available(int device);                        // HSA runtime reachable, agent matched, code object loaded (SSD_AQL=0: never)
const char *why_not(int device);
int tool_attached_now();                           // a profiling / tracing tool is attached to the process (evaluated afresh)
bool sync_mode();                                  // host-side waits instead of waiting kernels: a tool is attached, or SSD_AQL_SYNC=1
bool lookup(int device, const void *host_fn, Kernel *out);   // kernel descriptor of the instantiation behind a HIP host stub

// The device's dispatch queues: a pool of at most pool_size(device) queues per device, shared by every handle on it (queues are
// scarce: ssd_aql.hip), created -- and probed for the hardware-queue cliff -- on first use and kept for the life of the process.
// One rollout call writes packets at a time (enqueue_mutex).
Queue *pool_queue(int device, int index);
int pool_size(int device);                         // what the rule of include/ssd.h allows, less what the probe turned down (0: none fits)
bool over_the_cliff(int device);                   // the probe turned the pool's FIRST queue down: the process holds too many active queues already
int pool_report(int device);                       // bits for ssd_rollout_path(): pool_size << 12 | 32 if the probe dropped a queue
void probe_figures(int device, double out[4]);     // us: HIP burst before the pool / after its last queue, queue burst first / last
std::mutex &enqueue_mutex(int device);
bool queue_failed(const Queue *q);                 // the runtime reported an error on the queue (the path is then abandoned)
uint64_t write_index(const Queue *q);              // index the next packet will get
uint64_t read_index(const Queue *q);               // packets below this index have been consumed by the command processor

// fence scopes: 0 none, 1 agent, 2 system (hsa_fence_scope_t)
void dispatch(Queue *q, const Kernel &k, uint32_t grid_x, uint32_t block_x, uint32_t lds_dynamic, const void *kernarg_dev,
              bool barrier, int acquire_scope, int release_scope);
void barrier_and(Queue *q, uint64_t dep_signal_handle);      // the queue waits until the signal's value is 0
void ring(Queue *q);                               // doorbell: hand everything written so far to the command processor
// after everything enqueued on q so far: the counter whose device address `flag_kernarg` (a host_kernarg_alloc block) holds += 1,
// with a system-scope release; join_and_wait: the host waits until that packet has completed
void join(Queue *q, const void *flag_kernarg);
bool join_and_wait(Queue *q, const void *flag_kernarg, double max_seconds = 0);   // max_seconds > 0: false when the wait exceeds it
void *host_kernarg_alloc(int device, size_t bytes);   // zeroed host kernarg memory the device can read
void host_kernarg_free(void *p);
const uint32_t *abort_flag_dev(int device);        // device address of the word that is set when a queue of the device fails

uint64_t signal_create(long long initial);         // 0 on failure
void signal_destroy(uint64_t handle);
void signal_set(uint64_t handle, long long value);
long long *signal_value_ptr(uint64_t handle);      // device-writable address of the signal's value

}  // namespace aql
}  // namespace ssd
