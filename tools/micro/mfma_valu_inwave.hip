// Microbenchmark: independent VALU instructions BETWEEN the fp32 MFMAs of one wave (1 wave per SIMD): do they hide in the MFMA's
// shadow (time stays 32 cycles per MFMA) or add to it?   hipcc --offload-arch=gfx950 -O3 -o /tmp/inw mfma_valu_inwave.hip && /tmp/inw
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

template <int NV, int KIND, int MF>
__global__ void __launch_bounds__(256) k(int iters, unsigned long long* out, float* sink) {
    const int lane = threadIdx.x & 63;
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = (float)lane, y = 1.0f, a = 0.3f, b = 1.0001f, c = 0.5f, d2 = 0.1f;
    bf16x8 xb = {1, 2, 3, 4, 5, 6, 7, 8}, yb = {1, 1, 1, 1, 1, 1, 1, 1};
    __shared__ float sh[4096];
    sh[threadIdx.x] = x;
    __syncthreads();
    unsigned long long t0 = now();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MF == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(xb), "v"(yb));
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (KIND == 0) { if (v & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d2) : "v"(b), "v"(c)); }
                if (KIND == 1) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(lane * 4)); }
            }
            if (MF == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a1) : "v"(xb), "v"(yb));
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (KIND == 0) { if (v & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d2) : "v"(b), "v"(c)); }
                if (KIND == 1) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(lane * 4)); }
            }
        }
        if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1));
    unsigned long long t1 = now();
    sink[threadIdx.x] = a0[0] + a1[1] + a + d2;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int NV, int KIND, int MF>
void run(const char* name) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 4096);
    unsigned long long h = 0;
    const int iters = 2000;
    hipLaunchKernelGGL((k<NV, KIND, MF>), dim3(256), dim3(256), 0, 0, iters, out, sink);
    hipDeviceSynchronize();
    hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("%-44s %2d between MFMAs: %6.2f cycles per MFMA\n", name, NV, (double)h / (iters * 16.0));
}
int main() {
    run<0, 0, 0>("f32 16x16x4 + v_fma_f32");
    run<1, 0, 0>("f32 16x16x4 + v_fma_f32");
    run<2, 0, 0>("f32 16x16x4 + v_fma_f32");
    run<4, 0, 0>("f32 16x16x4 + v_fma_f32");
    run<6, 0, 0>("f32 16x16x4 + v_fma_f32");
    run<8, 0, 0>("f32 16x16x4 + v_fma_f32");
    run<1, 1, 0>("f32 16x16x4 + ds_read_b32");
    run<2, 1, 0>("f32 16x16x4 + ds_read_b32");
    run<4, 1, 0>("f32 16x16x4 + ds_read_b32");
    run<0, 0, 1>("bf16 16x16x32 + v_fma_f32");
    run<2, 0, 1>("bf16 16x16x32 + v_fma_f32");
    run<4, 0, 1>("bf16 16x16x32 + v_fma_f32");
    return 0;
}
