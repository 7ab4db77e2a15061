// micro-benchmark: issue rate of the f32 MFMA shapes (cycles per instruction per SIMD), 1 and 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k4x4(float* out, long long* cyc, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ void k16(float* out, long long* cyc, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ void k32(float* out, long long* cyc, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K>
void run(const char* name, K kern, int nacc, int threads, int flops_per) {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 1024 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f, 2.0f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
    double n = (double)iters * 8 * nacc;
    double waves_per_simd = threads / 64 / 4.0; if (waves_per_simd < 1) waves_per_simd = 1;
    printf("%-28s acc=%d threads=%4d : %.1f cyc/MFMA/wave, %.1f cyc/MFMA/SIMD, %.1f TFLOP/s chip\n", name, nacc, threads,
           h[0] / n, h[0] / n / waves_per_simd, 256.0 * (threads / 64) * n * flops_per / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main() {
    run("mfma_f32_4x4x1_16b", k4x4<1>, 1, 256, 512);
    run("mfma_f32_4x4x1_16b", k4x4<2>, 2, 256, 512);
    run("mfma_f32_4x4x1_16b", k4x4<4>, 4, 256, 512);
    run("mfma_f32_4x4x1_16b", k4x4<4>, 4, 512, 512);
    run("mfma_f32_4x4x1_16b", k4x4<8>, 8, 256, 512);
    run("mfma_f32_16x16x4", k16<1>, 1, 256, 2048);
    run("mfma_f32_16x16x4", k16<4>, 4, 256, 2048);
    run("mfma_f32_16x16x4", k16<4>, 4, 512, 2048);
    run("mfma_f32_32x32x2", k32<1>, 1, 256, 4096);
    run("mfma_f32_32x32x2", k32<2>, 2, 256, 4096);
    return 0;
}
