// micro-benchmark of the rollout's GEMM phase in isolation: every workgroup (8 waves) streams the
// same 1 MiB packed weight image from L2 (wave w -> column block w, 128 k-quads of 1 KiB) and feeds
// 4x4x1 MFMAs with LDS activations.  Variants isolate loads / MFMAs / LDS reads and the issue pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define KQ 128
#define LD 516

__device__ __forceinline__ f32x4 mf(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// MODE bit0: do loads, bit1: do MFMA, bit2: read A from LDS (else constant)
// PAT 0: burst of 8 loads then 8 quads of compute (double buffer)   PAT 1: ring of 16 with per-quad refill
template <int MODE, int PAT, int S>
__global__ void __launch_bounds__(512) kern(const float4* __restrict__ img, float* out, int reps, int nblk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 8 * LD; i += blockDim.x) lds[i] = 0.001f * (i & 31);
    __syncthreads();
    f32x4 acc[S][4];
    for (int s = 0; s < S; ++s) for (int q = 0; q < 4; ++q) acc[s][q] = (f32x4){0, 0, 0, 0};
    const float* ap0 = lds + (lane & 3) * LD;
    float4 dummy = make_float4(0, 0, 0, 0);
    for (int r = 0; r < reps; ++r) {
        if (MODE & 4) {                  // the activations change every phase in the real kernel: forbid hoisting the LDS reads
            __syncthreads();
            lds[threadIdx.x] += 1e-6f;
            __syncthreads();
        }
        const float4* wp = img + (long)((wave + r) % nblk) * KQ * 64 + lane;
        const float* ap = ap0;
        if constexpr (PAT >= 16) {             // rolling window of PAT loads in flight, accumulate into dummy
            constexpr int D = PAT >= 16 ? PAT : 16;
            float4 w[D];
#pragma unroll
            for (int i = 0; i < D; ++i) w[i] = wp[i * 64];
            for (int q = 0; q < KQ; q += D) {
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    dummy.x += w[i].x; dummy.y += w[i].y; dummy.z += w[i].z; dummy.w += w[i].w;
                    if (q + D + i < KQ) w[i] = wp[(q + D + i) * 64];
                }
            }
        } else if constexpr (PAT == 0) {
            float4 wc[8], wn[8];
            if (MODE & 1) { for (int i = 0; i < 8; ++i) wc[i] = wp[i * 64]; } else { for (int i = 0; i < 8; ++i) wc[i] = make_float4(1, 2, 3, 4); }
            for (int ch = 0; ch < KQ / 8; ++ch) {
                wp += 8 * 64;
                if ((MODE & 1) && ch + 1 < KQ / 8) { for (int i = 0; i < 8; ++i) wn[i] = wp[i * 64]; }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        float4 a = (MODE & 4) ? *reinterpret_cast<const float4*>(ap + s * 4 * LD + i * 4) : make_float4(1, 1, 1, 1);
                        if (MODE & 2) {
                            acc[s][0] = mf(a.x, wc[i].x, acc[s][0]); acc[s][1] = mf(a.y, wc[i].y, acc[s][1]);
                            acc[s][2] = mf(a.z, wc[i].z, acc[s][2]); acc[s][3] = mf(a.w, wc[i].w, acc[s][3]);
                        } else { dummy.x += a.x * wc[i].x; dummy.y += wc[i].y + a.y; dummy.z += wc[i].z; dummy.w += wc[i].w; }
                    }
                }
                ap += 32;
                if (MODE & 1) for (int i = 0; i < 8; ++i) wc[i] = wn[i];
            }
        } else {
            float4 A[8], B[8];
            for (int i = 0; i < 8; ++i) { A[i] = wp[i * 64]; B[i] = wp[(8 + i) * 64]; }
            for (int pr = 0; pr < KQ / 16; ++pr) {
                const bool more = pr + 1 < KQ / 16;
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float4 w = hlf ? B[i] : A[i];
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            float4 a = (MODE & 4) ? *reinterpret_cast<const float4*>(ap + s * 4 * LD + (hlf * 8 + i) * 4) : make_float4(1, 1, 1, 1);
                            if (MODE & 2) {
                                acc[s][0] = mf(a.x, w.x, acc[s][0]); acc[s][1] = mf(a.y, w.y, acc[s][1]);
                                acc[s][2] = mf(a.z, w.z, acc[s][2]); acc[s][3] = mf(a.w, w.w, acc[s][3]);
                            } else { dummy.x += a.x * w.x; dummy.y += w.y + a.y; dummy.z += w.z; dummy.w += w.w; }
                        }
                        if (more) { if (hlf) B[i] = wp[(24 + i) * 64]; else A[i] = wp[(16 + i) * 64]; }
                    }
                }
                wp += 16 * 64; ap += 64;
            }
        }
    }
    float sum = dummy.x + dummy.y + dummy.z + dummy.w;
    for (int s = 0; s < S; ++s) for (int q = 0; q < 4; ++q) sum += acc[s][q][0] + acc[s][q][1] + acc[s][q][2] + acc[s][q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <class K>
void run(const char* name, K k, const float4* img, float* out, int grid, int threads, int S, int nblk) {
    const int reps = 64;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * LD * 4);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 8 * LD * 4, 0, img, out, 4, nblk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 8 * LD * 4, 0, img, out, reps, nblk);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us_phase = ms * 1e3 / reps;
    const double bytes = (double)grid * (threads / 64) * KQ * 1024.0;   // per phase
    const double flops = (double)grid * (threads / 64) * KQ * 4 * S * 512.0;
    printf("%-44s grid=%3d thr=%3d S=%d: %7.2f us/phase  %6.2f TB/s L2->CU  %6.1f TFLOP/s\n", name, grid, threads, S, us_phase,
           bytes / us_phase * 1e-6, flops / us_phase * 1e-6);
}

int main() {
    const int nblk = 8;
    float4* img; float* out;
    hipMalloc(&img, (size_t)nblk * KQ * 64 * 16 * 4);       // room for 4x images
    hipMemset(img, 0, (size_t)nblk * KQ * 64 * 16 * 4);
    hipMalloc(&out, 1 << 22);
    run("loads only, burst/double-buffer", kern<1, 0, 1>, img, out, 256, 512, 1, nblk);
    run("loads only, ring refill", kern<1, 1, 1>, img, out, 256, 512, 1, nblk);
    run("mfma only", kern<2, 0, 1>, img, out, 256, 512, 1, nblk);
    run("mfma + lds", kern<6, 0, 1>, img, out, 256, 512, 1, nblk);
    run("loads + mfma, burst", kern<3, 0, 1>, img, out, 256, 512, 1, nblk);
    run("loads + mfma + lds, burst", kern<7, 0, 1>, img, out, 256, 512, 1, nblk);
    run("loads + mfma + lds, ring", kern<7, 1, 1>, img, out, 256, 512, 1, nblk);
    run("loads + mfma + lds, burst S=2", kern<7, 0, 2>, img, out, 128, 512, 2, nblk);
    run("loads + mfma + lds, ring  S=2", kern<7, 1, 2>, img, out, 128, 512, 2, nblk);
    run("loads + mfma + lds, burst, 32 blocks(4MB)", kern<7, 0, 1>, img, out, 256, 512, 1, 32);
    run("loads only, window 16", kern<1, 16, 1>, img, out, 256, 512, 1, nblk);
    run("loads only, window 32", kern<1, 32, 1>, img, out, 256, 512, 1, nblk);
    run("loads only, window 64", kern<1, 64, 1>, img, out, 256, 512, 1, nblk);
    run("loads only, window 32, 4 waves", kern<1, 32, 1>, img, out, 256, 256, 1, nblk);
    run("loads only, window 64, 4 waves", kern<1, 64, 1>, img, out, 256, 256, 1, nblk);
    run("loads only, burst, 1 WG", kern<1, 0, 1>, img, out, 1, 512, 1, nblk);
    run("loads only, burst, 32 WG", kern<1, 0, 1>, img, out, 32, 512, 1, nblk);
    run("loads only, burst, 128 WG", kern<1, 0, 1>, img, out, 128, 512, 1, nblk);
    return 0;
}
