// tools/microbench/launch_cost_mt.hip -- host cost of hipLaunchKernel from TWO threads on two streams (what the rollout's
// chains do), by kernel-argument size: aggregate microseconds per launch.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
struct A64 { void *p; int pad[14]; };
struct A120 { void *p; int pad[28]; };
struct A400 { void *p; int pad[98]; };
__global__ void k64(const A64 a) { if (a.p == (void *)1) *(int *)a.p = 0; }
__global__ void k120(const A120 a) { if (a.p == (void *)1) *(int *)a.p = a.pad[27]; }
__global__ void k400(const A400 a) { if (a.p == (void *)1) *(int *)a.p = a.pad[97]; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename F> void run(const char *name, int nthreads, F f) {
    hipStream_t s[4];
    for (int i = 0; i < 4; ++i) (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    const int n = 20000;
    auto body = [&](int t) { for (int i = 0; i < n; ++i) f(s[t]); };
    for (int i = 0; i < 500; ++i) f(s[0]);
    (void)hipDeviceSynchronize();
    double t0 = now();
    std::thread th[4];
    for (int t = 0; t < nthreads; ++t) th[t] = std::thread(body, t);
    for (int t = 0; t < nthreads; ++t) th[t].join();
    double t1 = now();
    (void)hipDeviceSynchronize();
    printf("%-22s %d thread(s): %.2f us per launch (aggregate)\n", name, nthreads, (t1 - t0) * 1e6 / (n * nthreads));
}
int main() {
    A64 a64{}; A120 a120{}; A400 a400{};
    for (int nt = 1; nt <= 3; ++nt) {
        run("64 B kernarg", nt, [&](hipStream_t s) { hipLaunchKernelGGL(k64, dim3(1024), dim3(128), 8192, s, a64); });
        run("120 B kernarg", nt, [&](hipStream_t s) { hipLaunchKernelGGL(k120, dim3(1024), dim3(128), 8192, s, a120); });
        run("400 B kernarg", nt, [&](hipStream_t s) { hipLaunchKernelGGL(k400, dim3(1024), dim3(128), 8192, s, a400); });
    }
    return 0;
}
