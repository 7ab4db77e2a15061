// Microbenchmark: issue rate of v_fma_f64 (and v_fma_f32) with 8 independent accumulators, one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/f64r tools/micro/fma_f64_rate.hip && /tmp/f64r
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
template <int F64>
__global__ void k(int iters, unsigned long long* out, double* sink) {
    double a[8]; float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; f[i] = threadIdx.x + i; }
    double b = 1.0000001, c = 0.5; float fb = 1.0001f, fc = 0.5f;
    unsigned long long t0 = now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fb), "v"(fc));
            }
    }
    unsigned long long t1 = now();
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + f[i];
    sink[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
    unsigned long long* out; double* sink; unsigned long long h;
    hipMalloc(&out, 64); hipMalloc(&sink, 8192);
    const int iters = 4000;
    for (int f64 = 1; f64 >= 0; --f64)
        for (int thr = 256; thr <= 1024; thr *= 2) {
            if (f64) hipLaunchKernelGGL(k<1>, dim3(256), dim3(thr), 0, 0, iters, out, sink); else hipLaunchKernelGGL(k<0>, dim3(256), dim3(thr), 0, 0, iters, out, sink);
            hipDeviceSynchronize(); hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
            printf("%s, %d wave(s) per SIMD: %.2f cycles per instruction and wave, %.2f per instruction and SIMD\n", f64 ? "v_fma_f64" : "v_fma_f32", thr / 256,
                   (double)h / (iters * 32.0), (double)h / (iters * 32.0) / (thr / 256));
        }
    return 0;
}
