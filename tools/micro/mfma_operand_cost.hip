// Microbenchmark: cycles per v_mfma_f32_16x16x4_f32 in the forms the rollout kernels use: A operand from VGPRs / AccVGPRs, with the
// `s_nop 1` of the inline-asm strings, with one ds_read_b128 per 4 MFMAs, 2 or 4 accumulation chains.  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

// MODE bit0: A operand in AccVGPR; bit1: s_nop 1 behind each MFMA; bit2: a ds_read_b128 per 4 MFMAs (B operand from it); bit3: 4 chains
template <int MODE>
__global__ void __launch_bounds__(256) k(int iters, unsigned long long* out, float* sink) {
    const int lane = threadIdx.x & 63;
    __shared__ float4 sh[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sh[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = (float)(lane + i);
    float4 b = sh[lane];
    unsigned long long t0 = now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float4 bn = b;
            if (MODE & 4) bn = sh[((it * 4 + kb) & 15) * 64 + lane];
            const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4& a = acc[(MODE & 8) ? u : (u & 1)];
                if (MODE & 1) {
                    if (MODE & 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(a) : "a"(w[kb * 4 + u]), "v"(bb[u]));
                    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a) : "a"(w[kb * 4 + u]), "v"(bb[u]));
                } else {
                    if (MODE & 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(a) : "v"(w[kb * 4 + u]), "v"(bb[u]));
                    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a) : "v"(w[kb * 4 + u]), "v"(bb[u]));
                }
            }
            b = bn;
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    unsigned long long t1 = now();
    sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int MODE>
void run() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 4096);
    unsigned long long h = 0;
    const int iters = 4000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, iters, out, sink);
    hipDeviceSynchronize();
    hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("A operand %-8s %-10s %-22s %d chains: %6.2f cycles per MFMA\n", (MODE & 1) ? "AccVGPR" : "VGPR", (MODE & 2) ? "s_nop 1" : "-", (MODE & 4) ? "ds_read_b128 / 4 MFMAs" : "-",
           (MODE & 8) ? 4 : 2, (double)h / (iters * 16.0));
}
int main() {
    run<0>(); run<1>(); run<2>(); run<3>(); run<4>(); run<5>(); run<6>(); run<7>(); run<8>(); run<9>(); run<15>(); run<13>();
    return 0;
}
