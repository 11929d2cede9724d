// csrc/ssd_aql.hpp -- the library's own AQL dispatch path (ssd_aql.hip): HSA queues it owns, packets it writes.
#pragma once
#include <stdint.h>

namespace ssd {
namespace aql {

struct Kernel { uint64_t object = 0; uint32_t kernarg_size = 0, group_static = 0, private_size = 0; };
struct Queue;

bool available(int device);                        // HSA runtime reachable, agent matched, code object loaded (SSD_AQL=0: never)
const char *why_not(int device);
bool lookup(int device, const void *host_fn, Kernel *out);   // kernel descriptor of the instantiation behind a HIP host stub

// `join_counter`: device memory, 8 bytes, zeroed; every join() on the queue adds 1 to it once the queue's earlier packets are done.
// `abort_flag`: host memory (device-mapped), set to 1 when the runtime reports an error on the queue.
Queue *queue_create(int device, unsigned long long *join_counter, uint32_t *abort_flag);
void queue_destroy(Queue *q);
bool queue_failed(const Queue *q);                 // the runtime reported an error on the queue (the path is then abandoned)
uint64_t write_index(const Queue *q);              // index the next packet will get
uint64_t read_index(const Queue *q);               // packets below this index have been consumed by the command processor

// fence scopes: 0 none, 1 agent, 2 system (hsa_fence_scope_t)
void dispatch(Queue *q, const Kernel &k, uint32_t grid_x, uint32_t block_x, uint32_t lds_dynamic, const void *kernarg_dev,
              bool barrier, int acquire_scope, int release_scope);
void barrier_and(Queue *q, uint64_t dep_signal_handle);      // the queue waits until the signal's value is 0
void ring(Queue *q);                               // doorbell: hand everything written so far to the command processor
bool join_and_wait(Queue *q);                      // synchronous form: the host waits until the join packet has completed
void join(Queue *q);                               // after everything enqueued on q so far: join_counter += 1 (and a system-scope release)

uint64_t signal_create(long long initial);         // 0 on failure
void signal_destroy(uint64_t handle);
void signal_set(uint64_t handle, long long value);
long long *signal_value_ptr(uint64_t handle);      // device-writable address of the signal's value

}  // namespace aql
}  // namespace ssd
