// tools/microbench/graph_chains.hip -- host cost of stepping two independent launch chains: two streams fed by one
// thread, vs one hipGraph holding B steps of both chains, replayed.  The kernel is small (a few us), so the time per
// step is what the host / command processor can sustain.
//   hipcc --offload-arch=gfx950 -O2 -o graph_chains graph_chains.hip && ./graph_chains
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

struct Big { float *p; int n; int pad[80]; };            // ~340 B of kernel arguments, like ssd::Params

__global__ void work(const Big b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float x = b.p[i];
    for (int k = 0; k < b.n; ++k) x = x * 1.0001f + 0.5f;
    b.p[i] = x;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int blocks = 256, threads = 512, steps = 4000, B = 50;
    float *a, *c;
    hipMalloc(&a, blocks * threads * 4); hipMalloc(&c, blocks * threads * 4);
    hipMemset(a, 0, blocks * threads * 4); hipMemset(c, 0, blocks * threads * 4);
    hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipEvent_t fork, join; hipEventCreateWithFlags(&fork, hipEventDisableTiming); hipEventCreateWithFlags(&join, hipEventDisableTiming);
    for (int iters : {200, 2000}) {                      // kernel length: short, and roughly the step kernel's
        Big ba{}; ba.p = a; ba.n = iters; Big bc{}; bc.p = c; bc.n = iters;
        // (a) two streams, one thread
        for (int k = 0; k < 200; ++k) { hipLaunchKernelGGL(work, dim3(blocks), dim3(threads), 0, s0, ba); hipLaunchKernelGGL(work, dim3(blocks), dim3(threads), 0, s1, bc); }
        hipDeviceSynchronize();
        double t0 = now();
        for (int k = 0; k < steps; ++k) { hipLaunchKernelGGL(work, dim3(blocks), dim3(threads), 0, s0, ba); hipLaunchKernelGGL(work, dim3(blocks), dim3(threads), 0, s1, bc); }
        double t1 = now();
        hipDeviceSynchronize();
        double t2 = now();
        printf("iters %d: two streams, one thread: host %.2f us/step, total %.2f us/step\n", iters, (t1 - t0) * 1e6 / steps, (t2 - t0) * 1e6 / steps);
        // one chain only, for the kernel's own duration
        t0 = now();
        for (int k = 0; k < steps; ++k) hipLaunchKernelGGL(work, dim3(blocks), dim3(threads), 0, s0, ba);
        hipDeviceSynchronize();
        t2 = now();
        printf("iters %d: one stream: total %.2f us/launch\n", iters, (t2 - t0) * 1e6 / steps);
        // (b) graph: B steps of both chains captured once
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal);
        hipEventRecord(fork, s0); hipStreamWaitEvent(s1, fork, 0);
        for (int k = 0; k < B; ++k) { hipLaunchKernelGGL(work, dim3(blocks), dim3(threads), 0, s0, ba); hipLaunchKernelGGL(work, dim3(blocks), dim3(threads), 0, s1, bc); }
        hipEventRecord(join, s1); hipStreamWaitEvent(s0, join, 0);
        hipError_t e1 = hipStreamEndCapture(s0, &g);
        hipError_t e2 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        printf("capture %s, instantiate %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
        if (e1 != hipSuccess || e2 != hipSuccess) return 1;
        for (int k = 0; k < 4; ++k) hipGraphLaunch(ge, s0);
        hipStreamSynchronize(s0);
        t0 = now();
        for (int k = 0; k < steps / B; ++k) hipGraphLaunch(ge, s0);
        t1 = now();
        hipStreamSynchronize(s0);
        t2 = now();
        printf("iters %d: graph of %d steps x 2 chains: host %.2f us/step, total %.2f us/step\n", iters, B, (t1 - t0) * 1e6 / steps, (t2 - t0) * 1e6 / steps);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
