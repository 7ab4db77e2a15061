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

template <int chains>
__global__ void rate(double* out, int iters) {
    double acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < chains; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = 0;
    for (int c = 0; c < 16; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1024 * 256 + 16] = (double)(t1 - t0);
}

int main() {
    double ha[64], hb[64], hd[64], *a, *b, *d;
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 8 * (1024 * 256 + 8192));
    // A[lane] = 1 + lane, B[lane] = 100 + lane: brute-force the mapping
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0 + l; hb[l] = 100.0 + 3.0 * l; }
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
    // candidate layouts: the three 2-bit fields of the lane index (bits 0-1, 2-3, 4-5) in any role: (x, blk, k) for A and B, (j, blk, i) for D
    const char* nm[3] = {"bits 0-1", "bits 2-3", "bits 4-5"};
    for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) for (int pd = 0; pd < 6; ++pd) {
        const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
        auto lane_of = [&](const int* p, int x, int blk, int k) { int f[3]; f[p[0]] = x; f[p[1]] = blk; f[p[2]] = k; return f[0] + 4 * f[1] + 16 * f[2]; };
        bool ok = true;
        for (int blk = 0; blk < 4 && ok; ++blk) for (int i = 0; i < 4 && ok; ++i) for (int j = 0; j < 4 && ok; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += ha[lane_of(perm[pa], i, blk, k)] * hb[lane_of(perm[pb], j, blk, k)];
            if (fabs(s - hd[lane_of(perm[pd], j, blk, i)]) > 1e-9 * fabs(s)) ok = false;
        }
        if (ok) printf("layout: A (i, blk, k) = (%s, %s, %s);  B (j, blk, k) = (%s, %s, %s);  D (j, blk, i) = (%s, %s, %s)\n", nm[perm[pa][0]], nm[perm[pa][1]], nm[perm[pa][2]],
                       nm[perm[pb][0]], nm[perm[pb][1]], nm[perm[pb][2]], nm[perm[pd][0]], nm[perm[pd][1]], nm[perm[pd][2]]);
    }
    printf("D[0..7] = %g %g %g %g %g %g %g %g\n", hd[0], hd[1], hd[2], hd[3], hd[4], hd[5], hd[6], hd[7]);
    const int iters = 20000;
#define RATE(C) do { hipLaunchKernelGGL(rate<C>, dim3(1024), dim3(256), 0, 0, d, iters); hipDeviceSynchronize(); \
        double cyc; hipMemcpy(&cyc, d + 1024 * 256 + 16, 8, hipMemcpyDeviceToHost); \
        printf("v_mfma_f64_4x4x4_4b_f64: %2d independent chain(s) per wave, 1 wave per SIMD: %.1f cycles per MFMA (512 FLOP)\n", C, cyc / ((double)iters * C)); } while (0)
    RATE(1); RATE(2); RATE(4); RATE(8); RATE(12); RATE(16);
    return 0;
}
