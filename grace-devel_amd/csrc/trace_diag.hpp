// Diagnostic build of the traversal (-DGRACE_STAMPS; never the product): s_memtime stamps around
// the phases of a wave's life, accumulated per wave and summarised on the host after every launch.
// In the product build the macros expand to nothing.
#pragma once

#include "common.hpp"

#ifdef GRACE_STAMPS
#include <algorithm>
#include <vector>
#define STAMP_NOW() __builtin_amdgcn_s_memtime()
#define STAMP_ADD(acc, t0) do { acc += __builtin_amdgcn_s_memtime() - (t0); } while (0)
__device__ unsigned long long g_stamp_acc[8];
__device__ unsigned long long g_stamp_log[1 << 16][4];
__device__ unsigned int g_stamp_n;
#else
#define STAMP_NOW() 0ull
#define STAMP_ADD(acc, t0) do { } while (0)
#endif

#ifdef GRACE_STAMPS
// Per-launch summary on stderr (synchronises the device).
inline grace_status stamps_report(const int MODE)
{
    {
        unsigned long long h[8];
        GRACE_TRY_HIP(hipDeviceSynchronize());
        GRACE_TRY_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp_acc), sizeof(h)));
        const double w = double(h[7] ? h[7] : 1);
        std::fprintf(stderr, "[stamps] mode %d waves %llu: per wave (s_memtime ticks) total %.0f walk %.0f cluster %.0f "
                             "cull %.0f survivors %.0f | rounds %.1f survivors %.1f\n", MODE, h[7], h[0] / w,
                     h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6] / w);
        unsigned long long z[8] = {};
        GRACE_TRY_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_acc), z, sizeof(z)));
        {
            unsigned nlog = 0;
            GRACE_TRY_HIP(hipMemcpyFromSymbol(&nlog, HIP_SYMBOL(g_stamp_n), sizeof(nlog)));
            if (nlog > (1u << 16)) nlog = 1u << 16;
            std::vector<unsigned long long> lg(size_t(nlog) * 4);
            if (nlog) GRACE_TRY_HIP(hipMemcpyFromSymbol(lg.data(), HIP_SYMBOL(g_stamp_log), lg.size() * 8));
            unsigned long long t0 = ~0ull, t1 = 0;
            std::vector<double> life(nlog), start(nlog), surv(nlog);
            for (unsigned i = 0; i < nlog; ++i) { t0 = std::min(t0, lg[4 * i]); t1 = std::max(t1, lg[4 * i + 1]); }
            for (unsigned i = 0; i < nlog; ++i) {
                life[i] = double(lg[4 * i + 1] - lg[4 * i]); start[i] = double(lg[4 * i] - t0); surv[i] = double(lg[4 * i + 2]);
            }
            auto pct = [](std::vector<double> v, double q) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[size_t(q * (v.size() - 1))]; };
            std::fprintf(stderr, "[stamps] span %.0f | life p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f | start p50 %.0f p90 %.0f max %.0f | nsurv p10 %.0f p50 %.0f p90 %.0f max %.0f\n",
                         double(t1 - t0), pct(life, .1), pct(life, .5), pct(life, .9), pct(life, .99), pct(life, 1.), pct(start, .5),
                         pct(start, .9), pct(start, 1.), pct(surv, .1), pct(surv, .5), pct(surv, .9), pct(surv, 1.));
            unsigned zero = 0;
            GRACE_TRY_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_n), &zero, sizeof(zero)));
        }
    }
    return GRACE_OK;
}
#else
inline grace_status stamps_report(const int) { return GRACE_OK; }
#endif
