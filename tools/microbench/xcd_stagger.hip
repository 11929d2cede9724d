// tools/microbench/xcd_stagger.hip -- do the 8 XCDs of an MI355X start a kernel at the same time?
// 8 workgroups (round-robin = one per XCD) each take a ticket and spin until all 8 have arrived (bounded spin).
// If XCD x really starts d_x later than the first one, the early XCDs WAIT ~ (max d - d_x) in their own clock;
// if the per-XCD s_memrealtime counters are merely skewed, all waits are about one atomic round trip.
//   hipcc --offload-arch=gfx950 -O2 -o xcd_stagger xcd_stagger.hip && ./xcd_stagger
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Rec { unsigned long long t0, t1; unsigned xcc, iters, ticket, pad; };

__global__ void probe(unsigned *arrived, Rec *out, int nblocks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned ticket = 0, iters = 0;
    if (threadIdx.x == 0) {
        ticket = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nblocks && iters < 200000u) ++iters;
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x] = Rec{t0, t1, xcc & 0xF, iters, ticket, 0};
    }
}

__global__ void empty_kernel() {}

struct Big { unsigned *arrived; Rec *out; int nblocks; int pad[77]; };   // ~336 B of kernel arguments, like ssd::Params
__global__ void probe_big(const Big b) {
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (b.pad[76] == 12345) lds[threadIdx.x] = 1;                       // keep the LDS allocation alive
    if (threadIdx.x == 0 && blockIdx.x < 8) {
        unsigned iters = 0;
        const unsigned ticket = __hip_atomic_fetch_add(b.arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(b.arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u && iters < 200000u) ++iters;
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        b.out[blockIdx.x] = Rec{t0, t1, xcc & 0xF, iters, ticket, 0};
    }
}

int main() {
    const int nb = 8, reps = 12;
    unsigned *arrived; Rec *out;
    hipMalloc(&arrived, 4 * reps); hipMalloc(&out, sizeof(Rec) * nb * reps);
    hipMemset(arrived, 0, 4 * reps);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(64), 0, 0);   // warm up
    hipDeviceSynchronize();
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe, dim3(nb), dim3(64), 0, 0, arrived + r, out + r * nb, nb);
    hipDeviceSynchronize();
    std::vector<Rec> h(nb * reps);
    hipMemcpy(h.data(), out, sizeof(Rec) * nb * reps, hipMemcpyDeviceToHost);
    for (int r = reps - 4; r < reps; ++r) {
        unsigned long long tmin = ~0ull;
        for (int b = 0; b < nb; ++b) if (h[r * nb + b].t0 < tmin) tmin = h[r * nb + b].t0;
        printf("launch %d\n", r);
        for (int b = 0; b < nb; ++b) {
            const Rec &x = h[r * nb + b];
            printf("  block %d xcc %u ticket %u: start +%.2f us (its clock), waited %.2f us, %u spins\n", b, x.xcc, x.ticket,
                   (x.t0 - tmin) * 0.01, (x.t1 - x.t0) * 0.01, x.iters);
        }
    }
    for (int variant = 0; variant < 3; ++variant) {
        const int grid = variant == 0 ? 8 : 2048, ldsb = variant == 2 ? 55296 : 0;
        hipMemset(arrived, 0, 4 * reps);
        for (int r = 0; r < reps; ++r) {
            Big bg{}; bg.arrived = arrived + r; bg.out = out + r * nb; bg.nblocks = nb;
            hipLaunchKernelGGL(probe_big, dim3(grid), dim3(128), ldsb, 0, bg);
        }
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, sizeof(Rec) * nb * reps, hipMemcpyDeviceToHost);
        printf("big kernarg, grid %d x 128, %d B LDS (last launch)\n", grid, ldsb);
        const int r = reps - 1;
        unsigned long long tmin = ~0ull;
        for (int b = 0; b < nb; ++b) if (h[r * nb + b].t0 < tmin) tmin = h[r * nb + b].t0;
        for (int b = 0; b < nb; ++b) {
            const Rec &x = h[r * nb + b];
            printf("  block %d xcc %u ticket %u: start +%.2f us, waited %.2f us\n", b, x.xcc, x.ticket, (x.t0 - tmin) * 0.01, (x.t1 - x.t0) * 0.01);
        }
    }
    // launch-to-launch time of an empty kernel, back to back on one stream
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {8, 256, 2048}) {
        hipEventRecord(a, 0);
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(128), 0, 0);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("empty kernel, grid %d x 128: %.2f us per launch\n", grid, ms * 1000.f / 2000);
    }
    return 0;
}
