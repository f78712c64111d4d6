// Micro-benchmarks behind round-3 kernel decisions (run on the GPU box):
//  1. does a wave64 VALU instruction with only lanes 0-31 enabled issue faster?
//  2. LDS cost of wave-uniform reads: ds_read2_b64 vs 2 x ds_read_b64 vs ds_read_b128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int ACTIVE>
__global__ __launch_bounds__(256) void valu_kernel(float* out, int iters)
{
    const int lane = threadIdx.x & 63;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    const float m = 1.0000001f, c = 1e-9f;
    if (lane < ACTIVE) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// MODE 0: two ds_read_b64 (forced apart), 1: ds_read2_b64 (what the compiler merges to), 2: one ds_read_b128
template <int MODE>
__global__ __launch_bounds__(256) void lds_kernel(float* out, int iters)
{
    __shared__ __align__(16) float tile[4][3][132];
    const int wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 3 * 132; i += 256) (&tile[0][0][0])[i] = float(i & 7);
    __syncthreads();
    float acc = 0.f;
    unsigned base = unsigned(reinterpret_cast<size_t>(&tile[wv][0][0]));   // LDS byte address (uniform per wave)
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            unsigned addr = base + ((i + k) & 31) * 8;
            if (MODE == 0) {
                float2 p, q;
                asm volatile("ds_read_b64 %0, %1" : "=v"(p) : "v"(addr));
                asm volatile("ds_read_b64 %0, %1 offset:1056" : "=v"(q) : "v"(addr));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                acc += p.x + p.y + q.x + q.y;
            } else if (MODE == 1) {
                float4 p;
                asm volatile("ds_read2_b64 %0, %1 offset1:132" : "=v"(p) : "v"(addr));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                acc += p.x + p.y + p.z + p.w;
            } else {
                float4 p;
                unsigned a16 = base + ((i + k) & 31) * 16;
                asm volatile("ds_read_b128 %0, %1" : "=v"(p) : "v"(a16));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                acc += p.x + p.y + p.z + p.w;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename F>
static float time_ms(F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float* out;
    const int blocks = 256 * 8;     // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    CK(hipMalloc(&out, blocks * 256 * sizeof(float)));
    const int iters = 2000;
    const double valu_insts = double(blocks) * 4 * iters * 64;     // wave-instructions
    float t64 = time_ms([&] { valu_kernel<64><<<blocks, 256>>>(out, iters); });
    float t32 = time_ms([&] { valu_kernel<32><<<blocks, 256>>>(out, iters); });
    float t16 = time_ms([&] { valu_kernel<16><<<blocks, 256>>>(out, iters); });
    printf("VALU v_fma wave-instr/s (G): exec 64 lanes %.1f | 32 lanes %.1f | 16 lanes %.1f   (ms %.3f %.3f %.3f)\n",
           valu_insts / t64 / 1e6, valu_insts / t32 / 1e6, valu_insts / t16 / 1e6, t64, t32, t16);
    const int li = 4000;
    const double reads = double(blocks) * 4 * li * 16;             // survivor fetches (wave-level)
    float l0 = time_ms([&] { lds_kernel<0><<<blocks, 256>>>(out, li); });
    float l1 = time_ms([&] { lds_kernel<1><<<blocks, 256>>>(out, li); });
    float l2 = time_ms([&] { lds_kernel<2><<<blocks, 256>>>(out, li); });
    // cycles per fetch per CU at 2.4 GHz: time * 2.4e9 / (reads / 256 CUs)
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / (reads / 256.0); };
    printf("uniform 16-byte fetch, LDS cycles per wave-fetch per CU: 2 x ds_read_b64 %.2f | ds_read2_b64 %.2f | ds_read_b128 %.2f\n",
           cyc(l0), cyc(l1), cyc(l2));
    CK(hipGetLastError());
    return 0;
}
