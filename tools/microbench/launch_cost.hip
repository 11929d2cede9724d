// tools/microbench/launch_cost.hip -- host cost of hipLaunchKernel by kernel-argument size (empty kernels, one stream).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
struct A8 { void *p; };
struct A64 { void *p; int pad[14]; };
struct A340 { void *p; int pad[83]; };
__global__ void k8(const A8 a) { if (a.p == (void *)1) *(int *)a.p = 0; }
__global__ void k64(const A64 a) { if (a.p == (void *)1) *(int *)a.p = 0; }
__global__ void k340(const A340 a) { if (a.p == (void *)1) *(int *)a.p = a.pad[82]; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename F> void run(const char *name, F f) {
    for (int i = 0; i < 500; ++i) f();
    hipDeviceSynchronize();
    const int n = 20000;
    double t0 = now();
    for (int i = 0; i < n; ++i) f();
    double t1 = now();
    hipDeviceSynchronize();
    double t2 = now();
    printf("%-28s host %.2f us/launch, total %.2f us/launch\n", name, (t1 - t0) * 1e6 / n, (t2 - t0) * 1e6 / n);
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    A8 a8{}; A64 a64{}; A340 a340{};
    run("8 B kernarg", [&] { hipLaunchKernelGGL(k8, dim3(256), dim3(128), 0, s, a8); });
    run("64 B kernarg", [&] { hipLaunchKernelGGL(k64, dim3(256), dim3(128), 0, s, a64); });
    run("340 B kernarg", [&] { hipLaunchKernelGGL(k340, dim3(256), dim3(128), 0, s, a340); });
    run("340 B kernarg + 55 KB LDS", [&] { hipLaunchKernelGGL(k340, dim3(256), dim3(128), 55296, s, a340); });
    void *args[] = {&a340};
    run("hipLaunchKernel directly", [&] { hipLaunchKernel((const void *)k340, dim3(256), dim3(128), args, 0, s); });
    // driver-style launches: function handle looked up once, arguments as an array or as one packed buffer
    hipFunction_t f = nullptr;
    if (hipGetFuncBySymbol(&f, (const void *)k340) == hipSuccess && f) {
        run("hipModuleLaunchKernel (args)", [&] { hipModuleLaunchKernel(f, 256, 1, 1, 128, 1, 1, 0, s, args, nullptr); });
        size_t sz = sizeof(a340);
        void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a340, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        run("hipModuleLaunchKernel (buf)", [&] { hipModuleLaunchKernel(f, 256, 1, 1, 128, 1, 1, 0, s, nullptr, extra); });
        run("hipExtModuleLaunchKernel (buf)", [&] { hipExtModuleLaunchKernel(f, 256 * 128, 1, 1, 128, 1, 1, 0, s, nullptr, extra, nullptr, nullptr, 0); });
    } else {
        printf("hipGetFuncBySymbol unavailable\n");
    }
    return 0;
}
