// Comparator (NOT part of the product, never linked into libgrace_hip.so): the vendor library's
// radix sort on the build's workload -- 10^7 30-bit keys with a 16-byte payload -- timed with HIP
// events next to grace_sort_pairs_u32.  hipcc --offload-arch=gfx950 -O2 -I include
//   -L grace-devel_amd/lib -lgrace_hip profiles/bench_rocprim_sort.hip
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "grace_hip.h"

struct alignas(16) Payload { float x, y, z, w; };

int main()
{
    const size_t n = 10000000;
    std::vector<unsigned> hk(n);
    std::vector<Payload> hp(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hk[i] = s >> 2; hp[i] = { float(i), 0, 0, 0 }; }
    unsigned *k0, *k1, *kw; Payload *p0, *p1, *pw;
    hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&kw, n * 4);
    hipMalloc(&p0, n * 16); hipMalloc(&p1, n * 16); hipMalloc(&pw, n * 16);
    hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(p0, hp.data(), n * 16, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    size_t tmp_bytes = 0; void* tmp = nullptr;
    rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, p0, p1, n, 0, 30);
    hipMalloc(&tmp, tmp_bytes);
    float best_rp = 1e9f, best_g = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, p0, p1, n, 0, 30);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best_rp) best_rp = ms;
    }
    for (int rep = 0; rep < 5; ++rep) {
        hipMemcpy(kw, k0, n * 4, hipMemcpyDeviceToDevice); hipMemcpy(pw, p0, n * 16, hipMemcpyDeviceToDevice);
        hipDeviceSynchronize();
        hipEventRecord(a);
        if (grace_sort_pairs_u32(kw, pw, n, 16, 0, 30, nullptr, nullptr) != GRACE_OK) { std::printf("grace sort failed\n"); return 1; }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best_g) best_g = ms;
    }
    // same result?
    std::vector<unsigned> r1(n), r2(n);
    hipMemcpy(r1.data(), k1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), kw, n * 4, hipMemcpyDeviceToHost);
    std::vector<Payload> q1(n), q2(n);
    hipMemcpy(q1.data(), p1, n * 16, hipMemcpyDeviceToHost); hipMemcpy(q2.data(), pw, n * 16, hipMemcpyDeviceToHost);
    size_t diff = 0;
    for (size_t i = 0; i < n; ++i) diff += (r1[i] != r2[i]) || (q1[i].x != q2[i].x);
    std::printf("10^7 keys (30 bits) + 16-byte payload: rocprim::radix_sort_pairs %.3f ms (out of place, %zu MB scratch), grace_sort_pairs_u32 %.3f ms (in place); outputs differ in %zu places\n",
                best_rp, tmp_bytes >> 20, best_g, diff);
    return diff != 0;
}
