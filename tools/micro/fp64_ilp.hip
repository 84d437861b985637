// How many independent fp64 chains does a SIMD of gfx950 need to stay busy?  A wave runs `CH` independent dependent chains
// of v_fma_f64 (CH = 1, 2, 4, 8); W waves per SIMD (W = 1, 2, 4, 8: blocks of 256 * W threads on one CU).  Reported: wall
// time per wave-instruction on one SIMD = elapsed / (instructions per wave * W), from hipEvents around a launch with
// one block per CU (all CUs loaded alike), in ns and in cycles of 2.4 GHz.  Saturation = the value stops falling.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/fp64_ilp.hip -o /tmp/fp64_ilp && /tmp/fp64_ilp
#include <hip/hip_runtime.h>

#include <cstdio>

template <int CH>
__global__ void k(double *out, double seed, int iters) {
    double a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = seed + threadIdx.x + c;
    const double m = 1.0000001;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / CH; ++r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[c]) : "v"(m));
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += a[c];
    if (s == 1.2345e-77) out[0] = s;
}

template <int CH>
static void run(int W) {
    const int iters = 2000, blocks = 256;
    double *d;
    hipMalloc(&d, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256 * W), 0, 0, d, 1.5, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256 * W), 0, 0, d, 1.5, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)iters * 64 * W;      // one block per CU: W waves on each of its 4 SIMDs
    const double ns = ms * 1e6 / inst_per_simd;
    printf("chains/wave %d, waves/SIMD %d: %6.2f ns per wave-instruction on a SIMD = %5.2f cycles at 2.4 GHz\n", CH, W, ns, ns * 2.4);
    hipFree(d);
}

int main() {
    for (int W : {1, 2, 4, 8}) {
        if (W <= 4) {
            run<1>(W);
            run<2>(W);
            run<4>(W);
            run<8>(W);
        } else {            // 2048 threads per block do not exist: two blocks of 1024 per CU instead
            ;
        }
    }
    return 0;
}
