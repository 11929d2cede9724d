// tools/microbench/aql_header_spy.c -- LD_PRELOAD shim: which AQL packet headers does the HIP runtime write?
// Interposes hsa_queue_create, remembers every queue, and a sampler thread histograms the 16-bit headers (+ setup word and
// completion-signal presence) it sees in the rings while the program runs.  Diagnostic only; prints at exit.
//   gcc -O2 -shared -fPIC -I/opt/rocm/include -o aql_header_spy.so aql_header_spy.c -ldl -lpthread
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <hsa/hsa.h>

static hsa_queue_t *g_q[64];
static volatile int g_nq = 0;
static pthread_t g_thr;
static volatile int g_run = 0;
static uint64_t g_hist[65536];          // header value -> times seen (sampled, not exact)
static uint64_t g_sig[65536];           // ... with a completion signal

static void *sampler(void *arg) {
    (void)arg;
    while (g_run) {
        for (int i = 0; i < g_nq; ++i) {
            hsa_queue_t *q = g_q[i];
            const uint8_t *base = (const uint8_t *)q->base_address;
            for (uint32_t k = 0; k < q->size; ++k) {
                const uint16_t h = *(volatile const uint16_t *)(base + 64ull * k);
                const uint64_t sig = *(volatile const uint64_t *)(base + 64ull * k + 56);
                g_hist[h]++;
                if (sig) g_sig[h]++;
            }
        }
        usleep(50);
    }
    return NULL;
}

static void report(void) {
    g_run = 0;
    fprintf(stderr, "[aql spy] %d queues\n", g_nq);
    for (int h = 0; h < 65536; ++h)
        if (g_hist[h])
            fprintf(stderr, "[aql spy] header 0x%04x type=%d barrier=%d acquire=%d release=%d : %llu samples (%llu with completion signal)\n", h,
                    h & 0xff, (h >> 8) & 1, (h >> 9) & 3, (h >> 11) & 3, (unsigned long long)g_hist[h], (unsigned long long)g_sig[h]);
}

hsa_status_t hsa_queue_create(hsa_agent_t agent, uint32_t size, hsa_queue_type32_t type,
                              void (*callback)(hsa_status_t, hsa_queue_t *, void *), void *data, uint32_t private_segment_size,
                              uint32_t group_segment_size, hsa_queue_t **queue) {
    static hsa_status_t (*real)(hsa_agent_t, uint32_t, hsa_queue_type32_t, void (*)(hsa_status_t, hsa_queue_t *, void *), void *, uint32_t,
                                uint32_t, hsa_queue_t **) = NULL;
    if (!real) real = dlsym(RTLD_NEXT, "hsa_queue_create");
    hsa_status_t st = real(agent, size, type, callback, data, private_segment_size, group_segment_size, queue);
    if (st == HSA_STATUS_SUCCESS && g_nq < 64) {
        g_q[g_nq] = *queue;
        fprintf(stderr, "[aql spy] queue %d: size %u type %u\n", g_nq, size, (unsigned)type);
        g_nq++;
        if (!g_run) { g_run = 1; pthread_create(&g_thr, NULL, sampler, NULL); atexit(report); }
    }
    return st;
}
