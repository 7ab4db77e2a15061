// Microbenchmark: do fp32 MFMA (v_mfma_f32_16x16x4_f32) and ordinary VALU work of ANOTHER wave on the same SIMD overlap?
// 8 waves per workgroup = 2 per SIMD; waves 0..3 run an MFMA stream (2 chains), waves 4..7 a VALU stream of the chosen kind.
// Prints cycles of each stream alone and together.   hipcc --offload-arch=gfx950 -O3 -o /tmp/ovl mfma_valu_overlap.hip && /tmp/ovl
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

template <int KIND, int nops>
__global__ void __launch_bounds__(512) k(int do_mfma, int do_valu, int iters, unsigned long long* out, float* sink, int prio, int swap) {
    const int wave = (threadIdx.x >> 6) ^ (swap ? 4 : 0), lane = threadIdx.x & 63;
    __shared__ float sh[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) sh[i] = (float)i;
    __syncthreads();
    if (prio == 1 && wave >= 4) asm volatile("s_setprio 3");
    if (prio == 2 && wave < 4) asm volatile("s_setprio 3");
    unsigned long long t0 = now();
    if (wave < 4) {
        if (do_mfma) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            float x = (float)lane, y = 1.0f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (nops == 0) {
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
                    } else if (nops == 1) {
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 15" : "+v"(a0) : "v"(x), "v"(y));
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 15" : "+v"(a1) : "v"(x), "v"(y));
                    } else if (nops == 2) {
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7" : "+v"(a0) : "v"(x), "v"(y));
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7" : "+v"(a1) : "v"(x), "v"(y));
                    } else if (nops == 3) {
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 11" : "+v"(a0) : "v"(x), "v"(y));
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 11" : "+v"(a1) : "v"(x), "v"(y));
                    } else if (nops == 4) {
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 7" : "+v"(a0) : "v"(x), "v"(y));
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 7" : "+v"(a1) : "v"(x), "v"(y));
                    } else {
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(a0) : "v"(x), "v"(y));
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(a1) : "v"(x), "v"(y));
                    }
                }
            }
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1));
            sink[threadIdx.x] = a0[0] + a1[1];
        }
    } else if (do_valu) {
        float a = (float)lane, b = 1.0001f, c = 0.5f, d2 = 0.25f;
        int ia = lane, ib = 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (KIND == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d2) : "v"(b), "v"(c)); }
                if (KIND == 1) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia) : "v"(ib)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ib) : "v"(ia)); }
                if (KIND == 2) { asm volatile("v_exp_f32 %0, %0" : "+v"(a)); asm volatile("v_rcp_f32 %0, %0" : "+v"(d2)); }
                if (KIND == 3) { float v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((lane * 4 + u * 256) & 16383)); a += v; }
                if (KIND == 4) { asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b)); asm volatile("v_mov_b32 %0, %1" : "=v"(d2) : "v"(c)); }
                if (KIND == 5) { asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc"); }
                if (KIND == 6) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a) : "v"(*(double*)&b)); }
            }
        }
        sink[threadIdx.x] = a + d2 + (float)(ia + ib);
    }
    unsigned long long t1 = now();
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

template <int KIND, int nops = 0>
void run(const char* name, int prio = 0, int swap = 0) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 4096);
    unsigned long long h[8];
    const int iters = 2000;
    for (int mode = 1; mode <= 3; ++mode) {
        hipLaunchKernelGGL((k<KIND, nops>), dim3(256), dim3(512), 0, 0, mode & 1, (mode >> 1) & 1, iters, out, sink, prio, swap);
        hipDeviceSynchronize();
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("%-28s %-9s mfma wave0 %8llu ticks (%5.1f / mfma)   valu wave4 %8llu ticks (%5.2f / instr)\n", name,
               mode == 1 ? "mfma only" : mode == 2 ? "valu only" : "both", h[0], (double)h[0] / (iters * 16.0), h[4], (double)h[4] / (iters * 32.0));
    }
}
int main() {
    printf("(ticks of s_memtime)\n");
    run<0>("v_fma_f32 x2");
    run<1>("v_add_u32 + v_xor_b32");
    run<2>("v_exp_f32 + v_rcp_f32");
    run<3>("ds_read_b32 + wait (x1)");
    run<4>("v_mov_b32 x2");
    run<5>("v_cmp + v_cndmask (x1)");
    run<6>("v_pk_fma_f32 (x1)");
    printf("--- the VALU waves at s_setprio 3\n");
    run<0>("v_fma_f32 x2", 1);
    run<2>("v_exp_f32 + v_rcp_f32", 1);
    run<3>("ds_read_b32 + wait (x1)", 1);
    printf("--- the MFMA waves at s_setprio 3\n");
    run<0>("v_fma_f32 x2", 2);
    printf("--- s_nop 15 behind every MFMA (it costs 4 x 16 cycles)\n");
    run<0, 1>("v_fma_f32 x2");
    printf("--- s_nop 15 + s_nop 3 behind every MFMA (4 x 20 cycles)\n");
    run<0, 5>("v_fma_f32 x2");
    printf("--- s_nop 15 + s_nop 7 behind every MFMA (4 x 24 cycles)\n");
    run<0, 2>("v_fma_f32 x2");
    printf("--- s_nop 15 + s_nop 11 behind every MFMA (4 x 28 cycles)\n");
    run<0, 3>("v_fma_f32 x2");
    printf("--- s_nop 7 behind every MFMA (4 x 8 cycles: inside the 32 of the MFMA)\n");
    run<0, 4>("v_fma_f32 x2");
    printf("--- VALU waves are waves 0..3 of the workgroup, MFMA waves 4..7\n");
    run<0>("v_fma_f32 x2", 0, 1);
    run<3>("ds_read_b32 + wait (x1)", 0, 1);
    return 0;
}
