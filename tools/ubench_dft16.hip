// ubench_dft16.hip -- the 16-point complex DFT of the channeliser's second stage (K6, docs/SPEC.md 3.11), two ways:
//   valu : factored radix-4 x 4 in registers, one output instant per lane (what k_channelise does)
//   mfma : the same transform as a dense GEMM on the matrix cores -- [32 x 32] real block form of the complex 16 x 16
//          DFT matrix times [32 x 32 instants], v_mfma_f32_32x32x2_f32 (exact fp32, runs at the fp32 vector rate)
// Both variants keep their inputs in registers and loop R times, so the number is the arithmetic pipes' throughput
// (DFT16 per second), not memory.  Prints both rates, their ratio and the largest difference of the results.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_dft16 tools/ubench_dft16.hip && ./ubench_dft16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int R = 512;                    // transforms per lane (valu) / per 32-instant tile (mfma) and launch
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void bf4(float2& x0, float2& x1, float2& x2, float2& x3)     // 4-point DFT, kernel e^{+j 2 pi ab / 4}
{
    const float2 t0 = make_float2(x0.x + x2.x, x0.y + x2.y), t1 = make_float2(x0.x - x2.x, x0.y - x2.y);
    const float2 t2 = make_float2(x1.x + x3.x, x1.y + x3.y), t3 = make_float2(x1.x - x3.x, x1.y - x3.y);
    x0 = make_float2(t0.x + t2.x, t0.y + t2.y);
    x2 = make_float2(t0.x - t2.x, t0.y - t2.y);
    x1 = make_float2(t1.x - t3.y, t1.y + t3.x);
    x3 = make_float2(t1.x + t3.y, t1.y - t3.x);
}

// first input of instant n (every repeat then adds +-1/16 to it): a cheap deterministic pattern
__host__ __device__ inline float gen(int n, int p, int c, int r) { return (float)(((n * 7 + p * 13 + c * 5 + r * 3) % 31) - 15) * (1.0f / 16.0f); }

__global__ __launch_bounds__(64) void k_valu(float* out, const float* tw /*[16][2] V16^k*/)
{
    const int n = blockIdx.x * 64 + threadIdx.x;
    float2 acc[16];
    for (int k = 0; k < 16; ++k) acc[k] = make_float2(0.f, 0.f);
    float twr[16], twi[16];
    for (int k = 0; k < 16; ++k) { twr[k] = tw[2 * k]; twi[k] = tw[2 * k + 1]; }
    float2 in[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) in[p] = make_float2(gen(n, p, 0, 0), gen(n, p, 1, 0));
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        float2 b[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) { in[p].x += 0.0625f; in[p].y -= 0.0625f; b[p] = in[p]; }     // next input: one add per element
        // p2 = 4 a + bb, c2 = e + 4 f :  Z[bb][e] = sum_a b[4a + bb] V4^{ae};  Y[e + 4f] = sum_bb Z[bb][e] V16^{e bb} V4^{f bb}
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) bf4(b[bb], b[4 + bb], b[8 + bb], b[12 + bb]);
#pragma unroll
        for (int e = 1; e < 4; ++e)
#pragma unroll
            for (int bb = 1; bb < 4; ++bb) {
                const float2 v = b[4 * e + bb];
                const float wr = twr[(e * bb) & 15], wi = twi[(e * bb) & 15];
                b[4 * e + bb] = make_float2(fmaf(-v.y, wi, v.x * wr), fmaf(v.y, wr, v.x * wi));
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) bf4(b[4 * e], b[4 * e + 1], b[4 * e + 2], b[4 * e + 3]);
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc[k].x += b[k].x; acc[k].y += b[k].y; }       // slot 4e + f holds Y[e + 4f]
    }
    for (int e = 0; e < 4; ++e)
        for (int f = 0; f < 4; ++f) {
            out[(size_t)n * 32 + 2 * (e + 4 * f)] = acc[4 * e + f].x;
            out[(size_t)n * 32 + 2 * (e + 4 * f) + 1] = acc[4 * e + f].y;
        }
}

// D[32 x 32] = M[32 x 32] X[32 x 32 instants]: rows 0..15 = Re Y[c2], 16..31 = Im Y[c2]; X rows 0..15 = Re b[p2], 16..31 = Im b[p2].
// v_mfma_f32_32x32x2_f32: A 32 x 2 (lane l: row l % 32, k = l / 32), B 2 x 32 (lane l: column l % 32, k = l / 32),
// D 32 x 32 as 16 floats per lane (lane l: column l % 32, rows 8 (i / 4) + 4... -> written out by index below).
__global__ __launch_bounds__(64) void k_mfma(float* out, const float* M /*[32][32] row major*/)
{
    const int lane = threadIdx.x, col = lane & 31, kh = lane >> 5;
    const int n = blockIdx.x * 32 + col;                         // 32 instants per wave
    float a[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) a[t] = M[(lane & 31) * 32 + 2 * t + kh];     // A operand of K step t
    v16f acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float x[16], dx[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int row = 2 * t + kh;                              // X row of this lane's B element: Re rows 0..15, Im rows 16..31
        x[t] = gen(n, row & 15, row >> 4, 0);
        dx[t] = row < 16 ? 0.0625f : -0.0625f;
    }
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        v16f d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            x[t] += dx[t];                                        // next input: one add per element
            d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], x[t], d, 0, 0, 0);
        }
        acc += d;
    }
    // D layout (32x32, 16 regs): register j of lane l holds row 8 * (j / 4) + 4 * (l / 32) + j % 4, column l % 32
    for (int j = 0; j < 16; ++j) {
        const int row = 8 * (j / 4) + 4 * kh + (j % 4);
        out[(size_t)n * 32 + 2 * (row & 15) + (row >> 4)] = acc[j];
    }
}

int main()
{
    const int n_inst = 1 << 20;
    std::vector<float> tw(32), M(32 * 32);
    for (int k = 0; k < 16; ++k) { tw[2 * k] = (float)cos(2 * M_PI * k / 16.0); tw[2 * k + 1] = (float)sin(2 * M_PI * k / 16.0); }
    for (int c = 0; c < 16; ++c)
        for (int p = 0; p < 16; ++p) {
            const double ang = 2 * M_PI * ((c * p) % 16) / 16.0;
            M[c * 32 + p] = (float)cos(ang);        M[c * 32 + 16 + p] = (float)-sin(ang);       // Re = C br - S bi
            M[(16 + c) * 32 + p] = (float)sin(ang); M[(16 + c) * 32 + 16 + p] = (float)cos(ang);   // Im = S br + C bi
        }
    float *d_tw, *d_M, *o1, *o2;
    CHECK(hipMalloc(&d_tw, 32 * 4)); CHECK(hipMalloc(&d_M, 32 * 32 * 4));
    CHECK(hipMalloc(&o1, (size_t)n_inst * 32 * 4)); CHECK(hipMalloc(&o2, (size_t)n_inst * 32 * 4));
    CHECK(hipMemcpy(d_tw, tw.data(), 32 * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_M, M.data(), 32 * 32 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms_v = 0, ms_m = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_valu, dim3(n_inst / 64), dim3(64), 0, 0, o1, d_tw);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_v, e0, e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mfma, dim3(n_inst / 32), dim3(64), 0, 0, o2, d_M);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_m, e0, e1));
    }
    std::vector<float> h1((size_t)4096 * 32), h2((size_t)4096 * 32);
    CHECK(hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0;
    for (size_t i = 0; i < h1.size(); ++i) { maxd = fmax(maxd, fabs((double)h1[i] - h2[i])); maxv = fmax(maxv, fabs((double)h1[i])); }
    const double nd = (double)n_inst * R;
    printf("DFT16 x %d instants x %d repeats, inputs generated in registers (arithmetic pipes only)\n", n_inst, R);
    printf("  valu  radix-4x4, lane = instant          : %8.3f ms  %8.2f G DFT16/s\n", ms_v, nd / ms_v / 1e6);
    printf("  mfma  32x32 real GEMM, v_mfma_f32_32x32x2 : %8.3f ms  %8.2f G DFT16/s\n", ms_m, nd / ms_m / 1e6);
    printf("  mfma / valu time = %.2f   max |difference| = %.3g (max |value| %.3g)\n", ms_m / ms_v, maxd, maxv);
    printf("  K6 needs 12 DFT16 per output instant; at its measured store floor (4.08 ms per 1.44e7 instants) that is %.1f G DFT16/s\n",
           12.0 * 1.44e7 / 4.08e-3 / 1e9);
    return 0;
}
