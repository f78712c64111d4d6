// Probe: which XCC / SE / CU does workgroup b land on?  (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 8) void probe(unsigned* out, int spin)
{
    __shared__ float pad[3800];   // ~15 KB like the trace kernel
    pad[threadIdx.x] = 0.f;
    unsigned hw = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup resident for a while so that the chip fills up as in the real launch
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    if (pad[threadIdx.x] != 0.f) out[0] = 0;
}
int main()
{
    const int nb = 2048;
    unsigned* d; hipMalloc(&d, nb * 8);
    probe<<<nb, 256>>>(d, 200000);
    std::vector<unsigned> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < 96; ++b) {
        unsigned hw = h[2 * b], x = h[2 * b + 1];
        // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
        printf("wg %4d xcc %u se %u sh %u cu %2u simd %u wave %u\n", b, x & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
    }
    // workgroups per (xcc, se, cu)
    return 0;
}
