// helper/random.cuh -- the reference tests' sphere generator (tests/helper/random.cuh:20-29,
// 56-111), restated for the host: element n seeds a minstd engine with a Wang/Jenkins hash of n
// and draws x, y, z, w uniformly in [low, high).  Generated on the host and copied to the
// device vector (the reference runs thrust::transform; no Thrust algorithm is used here).
#pragma once

#include <hip/hip_runtime.h>
#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

#include <cstdint>

inline uint32_t random_hash(uint32_t a)
{
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

template <typename Real4>
inline void random_real4(const Real4 low, const Real4 high, const size_t N,
                         thrust::device_vector<Real4>& d_out)
{
    typedef decltype(low.x) Real;
    thrust::host_vector<Real4> h(N);
    const uint64_t m = 2147483647ull;                       // minstd_rand: x <- 48271 x mod 2^31 - 1
    const float denom = float(2147483645u) + 1.0f;          // uniform_real_distribution<float>
    for (size_t n = 0; n < N; ++n) {
        uint64_t x = random_hash(uint32_t(n)) % m;
        if (x == 0) x = 1;
        Real v[4];
        const Real lo[4] = { low.x, low.y, low.z, low.w }, hi[4] = { high.x, high.y, high.z, high.w };
        for (int k = 0; k < 4; ++k) {
            x = (48271ull * x) % m;
            const float u = float(x - 1) / denom;
            v[k] = Real(u * float(hi[k] - lo[k]) + float(lo[k]));
        }
        h[n].x = v[0]; h[n].y = v[1]; h[n].z = v[2]; h[n].w = v[3];
    }
    d_out = h;
}
