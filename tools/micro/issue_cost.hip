// Issue cost of single VALU instructions on gfx950, one wave on one SIMD (and 4 waves on one SIMD): cycles per
// instruction of a stream of INDEPENDENT instructions of one kind (s_memtime around an unrolled loop).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/issue_cost.hip -o /tmp/issue_cost && /tmp/issue_cost
// Answers "what do v_rcp_f64 / v_rsq_f64 cost next to v_fma_f64" for the arithmetic diet of the fused Chambolle kernel.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ void k(unsigned long long *out, double seed, int iters) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3;
    int i0 = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %2, %2, %1, %2\n v_fma_f64 %3, %3, %1, %3\n v_fma_f64 %4, %4, %1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) :); ) }
        if (OP == 1) { REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :); ) }
        if (OP == 2) { REP16(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :); ) }
        if (OP == 3) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) :); ) }
        if (OP == 4) { REP16(asm volatile("v_mul_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_mul_f64 %2, %2, %2\n v_mul_f64 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :); ) }
        if (OP == 5) { REP16(asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :); ) }
        if (OP == 6) { REP16(asm volatile("v_cvt_f32_f64 %0, %2\n v_cvt_f32_f64 %1, %3\n v_cvt_f64_f32 %2, %0\n v_cvt_f64_f32 %3, %1" : "+v"(f0), "+v"(f1), "+v"(a2), "+v"(a3) :); ) }
        if (OP == 7) { REP16(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(f0), "+v"(f1) :); ) }
        if (OP == 8) { REP16(asm volatile("v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :); ) }
        if (OP == 9) { REP16(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) :); ) }
        if (OP == 10) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %0" : "+v"(f0), "+v"(f1) :); ) }
        if (OP == 11) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %1, %1, %2, %1\n v_fma_f32 %2, %2, %3, %2\n v_fma_f32 %3, %3, %0, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) :); ) }
        if (OP == 12) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %1, %1, %2, %1\n v_pk_fma_f32 %2, %2, %3, %2\n v_pk_fma_f32 %3, %3, %0, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :); ) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + i0;
    if (s == 1.2345e-77) out[1000] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
static void run(const char *name, unsigned long long *d, int threads) {
    const int iters = 200;
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, 1.5, iters);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, 1.5, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(16);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on this part: report shader cycles through the ratio to the FMA stream instead
    printf("%-28s threads %4d: %8.2f ticks per 64 instructions (wave 0)\n", name, threads, (double)h[0] / iters);
}

int main() {
    unsigned long long *d;
    hipMalloc(&d, 8 * 2048);
    for (int threads : {64, 256, 1024}) {
        run<0>("v_fma_f64 (x4 per group)", d, threads);
        run<4>("v_mul_f64", d, threads);
        run<5>("v_add_f64", d, threads);
        run<1>("v_rcp_f64", d, threads);
        run<2>("v_rsq_f64", d, threads);
        run<8>("v_sqrt_f64", d, threads);
        run<3>("v_rcp_f32", d, threads);
        run<9>("v_rsq_f32", d, threads);
        run<6>("v_cvt f32<->f64", d, threads);
        run<7>("v_mov_b32_dpp wave_shr/shl", d, threads);
        run<10>("v_mov_b32", d, threads);
        run<11>("v_fma_f32", d, threads);
        run<12>("v_pk_fma_f32", d, threads);
    }
    return 0;
}
