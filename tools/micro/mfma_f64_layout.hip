// Probe: operand / result layout and issue rate of v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4 per instruction) on gfx950.
// A and B are filled with recognisable values per lane; the host searches which (block, row, k) / (block, k, col) assignment of the
// 64 lanes reproduces D, and prints it.   hipcc --offload-arch=gfx950 -O3 -o /tmp/f64l tools/micro/mfma_f64_layout.hip && /tmp/f64l
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <stdio.h>
#include <math.h>

__global__ void probe(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    double acc = 0.0;
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], acc, 0, 0, 0);
    d[l] = acc;
}

__global__ void rate(double* out, int iters, int chains) {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c < chains) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = 0;
    for (int c = 0; c < 8; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[4096] = (double)(t1 - t0);
}

int main() {
    double ha[64], hb[64], hd[64], *a, *b, *d;
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 8 * 8192);
    // A[lane] = 1 + lane, B[lane] = 100 + lane: brute-force the mapping
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0 + l; hb[l] = 100.0 + 3.0 * l; }
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
    // candidate layouts: lane = 16 blk + 4 x + y with (x, y) in {(k, i), (i, k)} for A, {(k, j), (j, k)} for B, {(i, j), (j, i)} for D
    for (int la = 0; la < 2; ++la) for (int lb = 0; lb < 2; ++lb) for (int ld = 0; ld < 2; ++ld) {
        bool ok = true;
        for (int blk = 0; blk < 4 && ok; ++blk) for (int i = 0; i < 4 && ok; ++i) for (int j = 0; j < 4 && ok; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) {
                const int al = 16 * blk + (la ? 4 * i + k : 4 * k + i), bl = 16 * blk + (lb ? 4 * j + k : 4 * k + j);
                s += ha[al] * hb[bl];
            }
            const int dl = 16 * blk + (ld ? 4 * j + i : 4 * i + j);
            if (fabs(s - hd[dl]) > 1e-9 * fabs(s)) ok = false;
        }
        if (ok) printf("layout: A lane = 16 blk + %s, B lane = 16 blk + %s, D lane = 16 blk + %s\n", la ? "4 i + k" : "4 k + i", lb ? "4 j + k" : "4 k + j", ld ? "4 j + i" : "4 i + j");
    }
    printf("D[0..7] = %g %g %g %g %g %g %g %g\n", hd[0], hd[1], hd[2], hd[3], hd[4], hd[5], hd[6], hd[7]);
    for (int chains = 1; chains <= 8; chains *= 2) {
        const int iters = 20000;
        hipLaunchKernelGGL(rate, dim3(1024), dim3(256), 0, 0, d, iters, chains);
        hipDeviceSynchronize();
        double cyc; hipMemcpy(&cyc, d + 4096, 8, hipMemcpyDeviceToHost);
        printf("v_mfma_f64_4x4x4_4b_f64: %d independent chain(s) per wave, 1 wave per SIMD: %.1f cycles per MFMA (512 FLOP)\n", chains, cyc / ((double)iters * chains));
    }
    return 0;
}
