// VALU micro-benchmark (measurement tool, not product): issue rate of v_fma_f32 / v_pk_fma_f32 with VGPR and SGPR
// coefficient operands on gfx950.  One wave per SIMD x N waves, long dependent-free chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE> __global__ __launch_bounds__(256) void k(float* out, float c0, float c1, int iters)
{
    typedef float __attribute__((ext_vector_type(2))) f2;
    f2 a0 = {1.f, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f}, a4 = {1.f, 2.f}, a5 = {3.f, 4.f}, a6 = {5.f, 6.f}, a7 = {7.f, 8.f};
    f2 x = {threadIdx.x * 1e-3f, 0.5f};
    f2 cv = {c0, c0};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // v_pk_fma_f32, coefficient in an SGPR pair (op_sel broadcast), 8 independent chains
            REP16(asm volatile("v_pk_fma_f32 %0, %8, %9, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %8, %9, %1 op_sel_hi:[1,0,1]\n"
                               "v_pk_fma_f32 %2, %8, %9, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %8, %9, %3 op_sel_hi:[1,0,1]\n"
                               "v_pk_fma_f32 %4, %8, %9, %4 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %8, %9, %5 op_sel_hi:[1,0,1]\n"
                               "v_pk_fma_f32 %6, %8, %9, %6 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %8, %9, %7 op_sel_hi:[1,0,1]\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "s"(cv));)
        } else if (MODE == 1) {   // v_pk_fma_f32, coefficient in a VGPR pair
            REP16(asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n"
                               "v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                               "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n"
                               "v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(cv));)
        } else if (MODE == 2) {   // v_fma_f32 with an SGPR coefficient, 16 scalar chains (same FLOPs as 8 packed)
            REP16(asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                               "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                               : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(x.x), "s"(c0));
                  asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                               "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                               : "+v"(a0.y), "+v"(a1.y), "+v"(a2.y), "+v"(a3.y), "+v"(a4.y), "+v"(a5.y), "+v"(a6.y), "+v"(a7.y) : "v"(x.y), "s"(c1));)
        } else {                  // v_fmac_f32 (VOP2) with VGPR coefficient
            REP16(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                               "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                               : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(x.x), "v"(cv.x));
                  asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                               "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                               : "+v"(a0.y), "+v"(a1.y), "+v"(a2.y), "+v"(a3.y), "+v"(a4.y), "+v"(a5.y), "+v"(a6.y), "+v"(a7.y) : "v"(x.y), "v"(cv.y));)
        }
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <int MODE> void run(const char* name, int waves_per_simd)
{
    float* d;
    const int blocks = 256 * waves_per_simd;          // 256 threads = 4 waves = 1 per SIMD per block
    hipMalloc(&d, sizeof(float) * blocks * 256);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.9999f, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.9999f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double fma = (double)blocks * 256 * iters * 16 * 16;     // 16 reps x (8 packed = 16 fma | 16 scalar)
    printf("%-34s waves/SIMD %d: %.3f ms  %.1f TFLOP/s  (%.2f cycles per wave-instruction @2.4GHz per SIMD)\n", name, waves_per_simd, ms,
           2 * fma / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * 16 * (MODE <= 1 ? 8 : 16)));
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_pk_fma_f32  sgpr coefficient", w);
        run<1>("v_pk_fma_f32  vgpr coefficient", w);
        run<2>("v_fma_f32     sgpr coefficient", w);
        run<3>("v_fmac_f32    vgpr coefficient", w);
    }
    return 0;
}
