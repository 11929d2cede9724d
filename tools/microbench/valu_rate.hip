// tools/microbench/valu_rate.hip -- cycles per wave64 instruction for integer / logic / select VALU ops with
// 1, 2, 4, 8 waves per SIMD (MI355X).  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ void k(unsigned *out, unsigned seed, int iters, unsigned long long *cyc) {
    unsigned a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = b ^ 9, f = c + 11, g = d ^ 3, h = e + 5;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) { a += b; c += d; e += f; g += h; b += a; d += c; f += e; h += g; }                       // v_add_u32
            if (KIND == 1) { a ^= b; c &= d | 1; e |= f; g ^= h; b ^= a; d ^= c; f ^= e; h ^= g; }                  // logic
            if (KIND == 2) { a = a > b ? c : d; c = c > d ? e : f; e = e > f ? g : h; g = g > h ? a : b;
                             b = b > a ? d : c; d = d > c ? f : e; f = f > e ? h : g; h = h > g ? b : a; }          // cmp + cndmask
            if (KIND == 3) { a *= b; c *= d; e *= f; g *= h; b *= a | 1; d *= c | 1; f *= e | 1; h *= g | 1; }      // v_mul_lo_u32
            if (KIND == 4) { float x = __uint_as_float(a), y = __uint_as_float(b), z = __uint_as_float(c), w = __uint_as_float(d);
                             x = x * y + z; y = y * z + w; z = z * w + x; w = w * x + y; x = x * y + z; y = y * z + w; z = z * w + x; w = w * x + y;
                             a = __float_as_uint(x); b = __float_as_uint(y); c = __float_as_uint(z); d = __float_as_uint(w); }   // v_fma_f32
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, int per_iter) {
    unsigned *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&cyc, 2048 * 8);
    const int iters = 2000;
    printf("%-14s", name);
    for (int wps : {1, 2, 4, 8}) {                 // waves per SIMD: block = wps*4 waves, one block per CU
        const int threads = wps * 4 * 64;
        const int blocks = 256;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads > 1024 ? 1024 : threads), 0, 0, out, 1u, iters, cyc);
        if (threads > 1024) { hipDeviceSynchronize(); }
        hipDeviceSynchronize();
        // for 8 waves/SIMD use 2 blocks of 1024 threads per CU
        int nb = blocks;
        if (threads > 1024) { nb = 512; hipLaunchKernelGGL(k<KIND>, dim3(nb), dim3(1024), 0, 0, out, 1u, iters, cyc); hipDeviceSynchronize(); }
        unsigned long long h[2048]; hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < nb; ++i) s += (double)h[i];
        s /= nb;
        // cycles per instruction as seen by ONE wave; SIMD-level cycles per instruction = that / waves per SIMD
        printf("  wps=%d: %.2f cyc/inst/wave (%.2f per SIMD)", wps, s / (iters * 8.0 * per_iter), s / (iters * 8.0 * per_iter) / wps);
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_add_u32", 8); run<1>("logic", 8); run<2>("cmp+cndmask", 16); run<3>("v_mul_lo_u32", 8); run<4>("v_fma_f32", 8);
    return 0;
}
