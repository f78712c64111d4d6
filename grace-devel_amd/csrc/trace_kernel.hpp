// The traversal kernel of libgrace_hip.so (see trace.hip for the design notes): device helpers,
// beam / pencil / cluster culls, and trace_kernel<MODE, SPLIT, ALT, LAT>.  Included by trace.hip only.
#pragma once

#include "trace_diag.hpp"
#include "trace_state.hpp"

#include <type_traits>

namespace {

using namespace grace_hip;

// include/grace/cuda/trace_sph.cuh:32-48
__constant__ double c_kernel_table[N_TABLE] = {
    1.90986019771937, 1.90563449910964, 1.89304415940934, 1.87230928086763,
    1.84374947679902, 1.80776276033034, 1.76481079856299, 1.71540816859939,
    1.66011373131439, 1.59952322363667, 1.53426266082279, 1.46498233888091,
    1.39235130929287, 1.31705223652377, 1.23977618317103, 1.16121278415369,
    1.08201943664419, 1.00288866679720, 0.924475767210246, 0.847415371038733,
    0.772316688105931, 0.699736940377312, 0.630211918937167, 0.564194562399538,
    0.502076205853037, 0.444144023534733, 0.390518196140658, 0.341148855945766,
    0.295941946237307, 0.254782896476983, 0.217538645099225, 0.184059547649710,
    0.154181189781890, 0.127726122453554, 0.104505535066266,
    8.432088120445191E-002, 6.696547102921641E-002, 5.222604427168923E-002,
    3.988433820097490E-002, 2.971866601747601E-002, 2.150552303075515E-002,
    1.502124104014533E-002, 1.004371608622562E-002, 6.354242122978656E-003,
    3.739494884706115E-003, 1.993729589156428E-003, 9.212900163813992E-004,
    3.395908945333921E-004, 8.287326418242995E-005, 7.387919939044624E-006,
    0.000000000000000E+000
};


__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// Integer min/max on float bit patterns, as the reference's vmin/vmax PTX
// (include/grace/cuda/device/intrinsics.cuh:8-51).
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// include/grace/cuda/device/intersect.cuh:10-40.  Box corners are wave-uniform (SGPRs).
__device__ __forceinline__ void aabbs_hit(const float ix, const float iy, const float iz,
                                          const float ox, const float oy, const float oz,
                                          const float len, const float4 L, const float4 R,
                                          const float4 Z, bool& hit_l, bool& hit_r)
{
    const float bx_L = (L.x - ox) * ix, tx_L = (L.y - ox) * ix;
    const float by_L = (L.z - oy) * iy, ty_L = (L.w - oy) * iy;
    const float bz_L = (Z.x - oz) * iz, tz_L = (Z.y - oz) * iz;
    const float bx_R = (R.x - ox) * ix, tx_R = (R.y - ox) * ix;
    const float by_R = (R.z - oy) * iy, ty_R = (R.w - oy) * iy;
    const float bz_R = (Z.z - oz) * iz, tz_R = (Z.w - oz) * iz;

    const int zero = __float_as_int(0.0f), ilen = __float_as_int(len);
    const int tmin_L = imax(imax(__float_as_int(fminf(bx_L, tx_L)), __float_as_int(fminf(by_L, ty_L))),
                            imax(imin(__float_as_int(bz_L), __float_as_int(tz_L)), zero));
    const int tmax_L = imin(imin(__float_as_int(fmaxf(bx_L, tx_L)), __float_as_int(fmaxf(by_L, ty_L))),
                            imin(imax(__float_as_int(bz_L), __float_as_int(tz_L)), ilen));
    const int tmin_R = imax(imax(__float_as_int(fminf(bx_R, tx_R)), __float_as_int(fminf(by_R, ty_R))),
                            imax(imin(__float_as_int(bz_R), __float_as_int(tz_R)), zero));
    const int tmax_R = imin(imin(__float_as_int(fmaxf(bx_R, tx_R)), __float_as_int(fmaxf(by_R, ty_R))),
                            imin(imax(__float_as_int(bz_R), __float_as_int(tz_R)), ilen));
    // two bare comparisons: their ballots fold onto the v_cmp results
    hit_r = __int_as_float(tmax_R) >= __int_as_float(tmin_R);
    hit_l = __int_as_float(tmax_L) >= __int_as_float(tmin_L);
}

// OnHit_sphere_cumulate / _individual arithmetic (functors/trace.cuh:181-186) with lerp
// (include/grace/generic/interpolate.h:11-39, device branch).  lut[i] = (y_i, y_{i+1} - y_i);
// ir = 1/h and ir2 = ir*ir come from the pre-pass (same fp32 operations).  The lerp weight
// t = double(b) - int(b) is formed as float(b - float(int(b))), which is exact (b < 64,
// Sterbenz), then widened: one fp64 conversion instead of two and an fp64 subtract.
// Correctly rounded sqrt for x = 0 or x >= 2^-96 (finite): v_sqrt_f32 is within 1 ulp, the
// two FMA residuals pick the neighbour if it is closer -- the core of hipcc's own expansion
// without its input scaling and class test (seven instructions the hit path executes for
// every candidate).  x = 0 falls through unchanged (the residuals are NaN / -0).
__device__ __forceinline__ float sqrt_rn_normal(const float x)
{
    const float y = __builtin_amdgcn_sqrtf(x);
    const float ym = __int_as_float(__float_as_int(y) - 1);
    const float yp = __int_as_float(__float_as_int(y) + 1);
    const float rm = __builtin_fmaf(-ym, y, x);
    const float rp = __builtin_fmaf(-yp, y, x);
    float r = (0.0f >= rm) ? ym : y;
    r = (0.0f < rp) ? yp : r;
    return r;
}

__device__ __forceinline__ float hit_integral(const float b2, const float ir, const float ir2,
                                              const double2* lut)
{
    // Tiny non-zero b2 (a ray within ~1e-15 of a centre) takes the general sqrtf; the branch
    // is wave-uniform and practically never taken.
    const bool tiny = b2 < 1.2621774e-29f && b2 > 0.0f; // 2^-96
    const float root = __builtin_amdgcn_ballot_w64(tiny) ? __builtin_sqrtf(b2) : sqrt_rn_normal(b2);
    const float b = (N_TABLE - 1) * (root * ir);
    int x_idx = static_cast<int>(b);
    // t = double(b) - x_idx is exact in fp32 (b < 64, Sterbenz) -> one widening conversion.
    float t32 = b - static_cast<float>(x_idx);
    // Table end (b == N_table - 1 exactly, i.e. sqrt(b2)/h rounded to 1): x = 50, x_idx = 49,
    // t = 1.  Practically never taken; the vote keeps it off the common path.
    if (__builtin_amdgcn_ballot_w64(x_idx >= N_TABLE - 1)) {
        t32 = x_idx >= N_TABLE - 1 ? 1.0f : t32;
        x_idx = x_idx >= N_TABLE - 1 ? N_TABLE - 2 : x_idx;
    }
    const double2 y = lut[x_idx];
    float integral = static_cast<float>(__builtin_fma(static_cast<double>(t32), y.y, y.x));
    integral *= ir2;
    return integral;
}

// The column-density trace's default evaluation of the same line integral (tolerance, not
// bit, parity -- DESIGN.md section 4): v_sqrt_f32 as is (1 ulp), table position
// b = sqrt(b2) * (50/h) with 50/h from the pre-pass, ONE fp32 FMA on an fp32 table of
// (y_i - i dy_i, dy_i) rounded from the fp64 one: fma(dy_i, b, y_i - i dy_i), i = int(b).  Entries
// from N_TABLE - 1 on are (0, 0), so b == 50 (sqrt(b2)/h rounded up to 1) needs no clamp.
// Seven VALU instructions instead of twenty-five; each term within a few ulp of the exact one.
// Returns the table value; the caller applies 1/h^2 inside its accumulating FMA.
__device__ __forceinline__ float hit_integral_fast(const float b2, const float ir50,
                                                   const float2* lutf)
{
    const float b = __builtin_amdgcn_sqrtf(b2) * ir50;
    const float2 y = lutf[static_cast<int>(b)];
    return __builtin_fmaf(y.y, b, y.x);
}

// Moeller-Trumbore with back-face culling, tests/profile_trace_triangle/triangle.cuh:54-88,
// with the fp64 dot/cross of tests/helper/vector_math.cu narrowed to float exactly where the
// reference assigns to float / float3.  tri = {v, e1, e2} in fp64 (wave-uniform); dd = the
// ray direction in fp64.  `float inv_det = 1. / det` is a double division narrowed to float;
// for a float det that equals the correctly rounded fp32 quotient (1/det cannot lie within
// 2^-49 of a float midpoint), so the fp32 divide is used.
__device__ __forceinline__ bool tri_intersect(const double ddx, const double ddy, const double ddz,
                                              const float ox, const float oy, const float oz,
                                              const double* __restrict__ tri, float* t_out)
{
    const double vx = tri[0], vy = tri[1], vz = tri[2];
    const double e1x = tri[3], e1y = tri[4], e1z = tri[5];
    const double e2x = tri[6], e2y = tri[7], e2z = tri[8];
    const float Px = float(ddy * e2z - ddz * e2y);
    const float Py = float(ddz * e2x - ddx * e2z);
    const float Pz = float(ddx * e2y - ddy * e2x);
    const float det = float((e1x * double(Px) + e1y * double(Py)) + e1z * double(Pz));
    bool reject = det < 1E-14f;
    const float inv_det = 1.0f / det;
    const float OVx = ox - float(vx), OVy = oy - float(vy), OVz = oz - float(vz);
    const double dOVx = OVx, dOVy = OVy, dOVz = OVz;
    const float u = float(((dOVx * double(Px) + dOVy * double(Py)) + dOVz * double(Pz)) * double(inv_det));
    reject = reject || (u < 0.f || u > 1.f);
    const float Qx = float(dOVy * e1z - dOVz * e1y);
    const float Qy = float(dOVz * e1x - dOVx * e1z);
    const float Qz = float(dOVx * e1y - dOVy * e1x);
    const float v = float(((ddx * double(Qx) + ddy * double(Qy)) + ddz * double(Qz)) * double(inv_det));
    reject = reject || (v < 0.f || u + v > 1.f);
    *t_out = float(((e2x * double(Qx) + e2y * double(Qy)) + e2z * double(Qz)) * double(inv_det));
    return !reject;
}

// The per-hit arithmetic on plain arrays (tests pin it against the oracle on inputs no
// traversal would produce: zeros, denormals, b2 -> h^2, huge/small h).
__global__ __launch_bounds__(256) void hit_integrals_kernel(const float* __restrict__ b2,
                                                            const float* __restrict__ h, size_t n,
                                                            float* __restrict__ out)
{
    __shared__ double2 s_lut[N_TABLE];
    if (threadIdx.x < N_TABLE) {
        const double y0 = c_kernel_table[threadIdx.x];
        const double y1 = threadIdx.x + 1 < N_TABLE ? c_kernel_table[threadIdx.x + 1] : y0;
        s_lut[threadIdx.x] = make_double2(y0, y1 - y0);
    }
    __syncthreads();
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x) {
        const float ir = 1.f / h[i];
        out[i] = hit_integral(b2[i], ir, ir * ir, s_lut);
    }
}

// Bounding boxes of the packet's origins and directions (wave-uniform, SGPRs).
struct Beam {
    float olo[3], ohi[3], dlo[3], dhi[3];
};

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// Conservative: returns false only if sphere_hit (generic/intersect.h:10-55) is false for
// every ray with origin in [olo, ohi] and direction in [dlo, dhi].
//   true b^2 = |p|^2 - (p.d)^2, p = c - o.  Lower bound over the beam:
//   |p|^2_lo - max((p.d)_lo^2, (p.d)_hi^2), component-wise interval arithmetic.
// sphere_hit's computed b2 differs from the true value by < ~16 u |p|^2 (u = 2^-24; |b| <= |p|)
// and the interval end points carry similar rounding; the margin 2^-18 |p|^2_hi covers both
// with a factor > 16 to spare.  Any NaN keeps the sphere.
__device__ __forceinline__ bool beam_may_hit(const float4 s, const Beam& bm,
                                             const float margin = 3.814697265625e-06f /* 2^-18 */)
{
    float p2_lo = 0.f, p2_hi = 0.f, t_lo = 0.f, t_hi = 0.f;
    const float c[3] = { s.x, s.y, s.z };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float plo = c[k] - bm.ohi[k], phi = c[k] - bm.olo[k];
        const float a2 = plo * plo, b2 = phi * phi;
        p2_hi += fmaxf(a2, b2);
        p2_lo += (plo <= 0.f && phi >= 0.f) ? 0.f : fminf(a2, b2);
        const float q0 = plo * bm.dlo[k], q1 = plo * bm.dhi[k];
        const float q2 = phi * bm.dlo[k], q3 = phi * bm.dhi[k];
        t_lo += fminf(fminf(q0, q1), fminf(q2, q3));
        t_hi += fmaxf(fmaxf(q0, q1), fmaxf(q2, q3));
    }
    const float t2_hi = fmaxf(t_lo * t_lo, t_hi * t_hi);
    const float b2_lo = p2_lo - t2_hi - margin * p2_hi;
    return !(b2_lo >= s.w);
}

// Axis-aligned packet (every direction = +-e_AX): a ray's b2 is fl(fl(q1^2) + fl(q2^2)) with
// q = fl(s - o) in the two perpendicular components.  Rounding is monotone, so replacing each
// o by the point of the packet's origin interval nearest to s bounds every lane's b2 from
// below EXACTLY -- no margin, eight instructions.
// Pencil packet (every ray starts at the same point: point sources, HEALPix / isotropic
// bundles, pinhole cameras): the rays lie in the cone of half-angle theta around the
// normalised mean direction a.  A ray at angle <= theta from a passes within h of centre c only
// if angle(c - o, a) < theta + asin(h / |c - o|) (or the origin is within h of c).  Interval
// arithmetic on separate origin/direction boxes loses that correlation: on 10^5 isotropic
// rays through 10^6 spheres it kept 40 k candidates per packet of which 12 k were hit by some
// ray.  Conservative by an absolute 1e-5 on the cosine and a relative 1e-5 on h^2; explicit
// FMAs are fine here (a cull, not a result).
struct Pencil {
    float ox, oy, oz;     // common origin
    float ax, ay, az;     // unit axis
    float sin_t, cos_t;   // half-angle
    // Four planes through the origin bounding the bundle in the tangent frame (u, v) of the
    // axis: outward unit normals.  A sphere wholly outside any of them (n . (c - o) > h) cannot
    // be hit.  Tightens the cone where the bundle's footprint is not round.
    float nx[4], ny[4], nz[4];
};

__device__ __forceinline__ bool pencil_may_hit(const float4 s, const Pencil& pc)
{
    const float vx = s.x - pc.ox, vy = s.y - pc.oy, vz = s.z - pc.oz;
    const float d2 = __builtin_fmaf(vx, vx, __builtin_fmaf(vy, vy, vz * vz));
    const float va = __builtin_fmaf(vx, pc.ax, __builtin_fmaf(vy, pc.ay, vz * pc.az));
    const float inv = __builtin_amdgcn_rsqf(d2);
    const float sin_a = fminf(1.0f, __builtin_amdgcn_sqrtf(s.w) * inv * 1.00001f);
    const float cos_a = __builtin_amdgcn_sqrtf(fmaxf(0.0f, __builtin_fmaf(-sin_a, sin_a, 1.0f)));
    const float cos_limit = __builtin_fmaf(pc.cos_t, cos_a, -pc.sin_t * sin_a) - 1e-5f;
    const float h = __builtin_amdgcn_sqrtf(s.w) * 1.00001f + 1e-6f * __builtin_amdgcn_sqrtf(d2);
    float out = -1.0f;   // largest signed distance beyond a side plane, in units of length
#pragma unroll
    for (int k = 0; k < 4; ++k)
        out = fmaxf(out, __builtin_fmaf(vx, pc.nx[k], __builtin_fmaf(vy, pc.ny[k], vz * pc.nz[k])) - h);
    // !(a < b) forms keep the sphere on any NaN (d2 = 0: the origin is the centre).
    return !(d2 > s.w * 1.00001f) || (!(va * inv < cos_limit) && !(out > 0.0f));
}

// FUSED: the fast column-density trace forms b2 as fma(q1, q1, q2 q2) (one instruction fewer per
// survivor; tolerance parity); its cull must bound THAT expression -- equally monotone.
template <int AX, bool FUSED = false>
__device__ __forceinline__ bool axis_beam_may_hit(const float4 s, const Beam& bm)
{
    constexpr int D1 = AX == 0 ? 1 : 0, D2 = AX == 2 ? 1 : 2;
    const float s1 = AX == 0 ? s.y : s.x;
    const float s2 = AX == 2 ? s.y : s.z;
    const float q1 = s1 - __builtin_amdgcn_fmed3f(s1, bm.olo[D1], bm.ohi[D1]);
    const float q2 = s2 - __builtin_amdgcn_fmed3f(s2, bm.olo[D2], bm.ohi[D2]);
    const float b2_lo = FUSED ? __builtin_fmaf(q1, q1, q2 * q2) : q1 * q1 + q2 * q2;
    return !(b2_lo >= s.w);
}

// Cluster tests (see cluster_boxes_kernel): may any ray of the packet hit any member of the
// cluster with box [blo, bhi]?  Axis-aligned packets: the box against the packet's origin
// rectangle in the two perpendicular components (a member is hit only by a ray whose origin
// lies inside the member's own inflated box, which the cluster box contains).  Other packets: the
// box's circumscribed sphere through the same conservative tests as a single candidate, with a
// wider margin (a hit member at distance < h (1 + e) of a ray puts the centre of the cluster
// within |c - C| + h (1 + e) <= R (1 + e) of it).  Any NaN keeps the cluster.
__device__ __forceinline__ float4 cluster_sphere(const float4 blo, const float4 bhi)
{
    const float cx = 0.5f * (blo.x + bhi.x), cy = 0.5f * (blo.y + bhi.y), cz = 0.5f * (blo.z + bhi.z);
    const float ex = bhi.x - cx, ey = bhi.y - cy, ez = bhi.z - cz;
    const float fx = cx - blo.x, fy = cy - blo.y, fz = cz - blo.z;
    const float rx = fmaxf(ex, fx), ry = fmaxf(ey, fy), rz = fmaxf(ez, fz);
    return make_float4(cx, cy, cz, (rx * rx + ry * ry + rz * rz) * 1.001f);
}

template <int AX>
__device__ __forceinline__ bool cluster_may_hit(const float4 blo, const float4 bhi, const Beam& bm,
                                                const Pencil* pc)
{
    if constexpr (AX >= 0) {
        constexpr int D1 = AX == 0 ? 1 : 0, D2 = AX == 2 ? 1 : 2;
        const float lo1 = AX == 0 ? blo.y : blo.x, hi1 = AX == 0 ? bhi.y : bhi.x;
        const float lo2 = AX == 2 ? blo.y : blo.z, hi2 = AX == 2 ? bhi.y : bhi.z;
        return !(lo1 > bm.ohi[D1]) && !(hi1 < bm.olo[D1]) && !(lo2 > bm.ohi[D2]) && !(hi2 < bm.olo[D2]);
    } else if constexpr (AX == -2) {
        // Pencil packet: the bundle lies inside the wedge of its four side planes (outward unit
        // normals n_k through the common origin) and in front of the origin along the axis.  The
        // box (already inflated by the members' radii) is wholly outside a plane if even its
        // innermost corner is: min over the box of n . (p - o) = sum_i min(n_i (lo_i - o_i),
        // n_i (hi_i - o_i)) > 0; wholly behind if max over the box of a . (p - o) < 0.  Sharper
        // than the circumscribed sphere for the elongated boxes Morton clusters often have; the
        // sphere test stays as a second opinion (either may drop the cluster).  Slack: 1e-5 of
        // the box's distance scale, far above the rounding of these few products.
        const float lx = blo.x - pc->ox, ly = blo.y - pc->oy, lz = blo.z - pc->oz;
        const float hx = bhi.x - pc->ox, hy = bhi.y - pc->oy, hz = bhi.z - pc->oz;
        const float scale = fmaxf(fmaxf(fmaxf(fabsf(lx), fabsf(hx)), fmaxf(fabsf(ly), fabsf(hy))),
                                  fmaxf(fabsf(lz), fabsf(hz)));
        const float slack = 1e-5f * scale;
        float worst = -1.0f;   // largest "innermost corner beyond plane k"
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float m = fminf(pc->nx[k] * lx, pc->nx[k] * hx) + fminf(pc->ny[k] * ly, pc->ny[k] * hy)
                + fminf(pc->nz[k] * lz, pc->nz[k] * hz);
            worst = fmaxf(worst, m);
        }
        const float front = fmaxf(pc->ax * lx, pc->ax * hx) + fmaxf(pc->ay * ly, pc->ay * hy)
            + fmaxf(pc->az * lz, pc->az * hz);
        // (!(a > b) forms: any NaN keeps the cluster)
        return !(worst > slack) && !(front < -slack) && pencil_may_hit(cluster_sphere(blo, bhi), *pc);
    } else {
        return beam_may_hit(cluster_sphere(blo, bhi), bm, 1.52587890625e-05f /* 2^-16 */);
    }
}

// ALT selects the mode's alternative code path: the fast kernel integral of the column-density
// trace, the LDS-staged outputs of the per-hit trace.
// The class-split instantiations are held to 8 waves per SIMD (<= 64 VGPRs, <= 80 SGPRs: the
// compiler parks ~28 scalars in VGPR lanes): they exist for small batches, where resident waves
// are what is scarce (1/8-image shard: K = 4 fits the chip at once, 0.82 -> 0.71 ms).
// LAT: the instantiation with the origin-lattice cull (see the packet set-up).  Both variants of
// a trace are launched; a device flag set from the batch's ray spacing and the scene's smallest
// sphere (choose_lattice, in ray_keys_kernel) lets exactly one of them run -- the test costs the
// class-split kernels registers they do not have, and the frame kernel 2 %, so scenes without
// sub-spacing spheres must not carry it.
template <int MODE, bool SPLIT, bool ALT = false, bool LAT = false>
// The one-wave-per-packet hit-count / column-density kernels are held to 6 waves per SIMD (<= 80
// VGPRs): left to itself the register allocator drifts between 77 and 102 VGPRs from one edit of
// this file to the next, and at 5 waves per SIMD the frame kernel loses 20 % (2.9 -> 3.55 ms,
// measured when an unrelated change tipped it over).
__global__ __launch_bounds__(TRACE_BLOCK, (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE) ? (SPLIT ? 8 : 6) : 1)
void trace_kernel(const TraceArgs a)
{
    static_assert(!ALT || MODE == MODE_CUMULATIVE || MODE == MODE_HITS, "no alternative path for this mode");
    static_assert(!LAT || MODE == MODE_COUNT || MODE == MODE_CUMULATIVE || MODE == MODE_HITS, "no lattice cull for this mode");
    if (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE || MODE == MODE_HITS) {
        if (a.lat_dev ? (*a.lat_dev != 0) != LAT : LAT) return;   // (workgroup-uniform)
        if (a.stage_dev && *a.stage_dev != a.stage_want) return;
    }
    constexpr bool FAST = ALT && MODE == MODE_CUMULATIVE;
    __shared__ double2 s_lut[FAST ? 1 : N_TABLE];
    // Fast integral: entry i = (y_i - i dy_i, dy_i), so that the lerp at table position b is ONE fma,
    // fma(dy_i, b, y_i - i dy_i) with i = int(b) -- no fractional part to extract.  Entries from
    // N_TABLE - 1 on are (0, 0): a position clamped to the table's end, or beyond it (misses of
    // the test-free rounds), contributes exactly +0.
    constexpr int LUTF_N = 256;
    __shared__ float2 s_lutf[FAST ? LUTF_N : 1];
    // Per-wave tile of the candidates of the current culling round (MODE_TRI keeps its
    // fp64 triangles on the scalar path).
    constexpr bool D4 = (MODE == MODE_COUNT_D4 || MODE == MODE_CUM_D4 || MODE == MODE_HITS_D4);
    constexpr bool LDS_TILE = (MODE != MODE_TRI && !D4);
    // Three 8-byte planes per wave -- (x, y), (z, h^2), (1/h terms) -- so that one address
    // (plane base + 8 j) serves all of a survivor's reads through immediate offsets.
    // (66 slots: the survivor loop reads up to two slots past the round's last survivor)
    __shared__ __align__(16) float2 s_tile[LDS_TILE ? TRACE_BLOCK / 64 : 1][LDS_TILE ? 3 : 1][LDS_TILE ? 66 : 1];
    // *_D4 modes: the round's candidates as doubles, lane-indexed: {x, y, z, w w, 1/w, (1/w)^2}
    // (the division is done once per candidate by its lane, not once per survivor by the wave).
    __shared__ double s_tile_d[D4 ? TRACE_BLOCK / 64 : 1][D4 ? 64 : 1][D4 ? 6 : 1];
    const int lane = threadIdx.x & 63;
    constexpr bool SPLITTABLE = (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE || MODE == MODE_HITS
                                 || MODE == MODE_COUNT_D4 || MODE == MODE_CUM_D4);
    static_assert(!SPLIT || SPLITTABLE, "triangle and stats walks do not split");
    // Hit counts and column densities split a packet by summation class (interleaved granules);
    // the per-hit trace, whose output is ordered, by contiguous chunk ranges chosen per packet.
    constexpr bool RANGE_SPLIT = SPLIT && MODE == MODE_HITS;
    // Waves per packet: as launched, or fewer when the device-side choice (choose_split)
    // says so.  The working waves are packed into the first workgroups -- surplus workgroups exit
    // whole, before touching LDS, so that they do not hold resources of the working ones.
    const int split = !SPLIT ? 1 : (!RANGE_SPLIT && a.split_dev) ? *a.split_dev : a.split;
    const int n_packets = (a.n_rays + a.width - 1) / a.width;
    const int nb = (n_packets * split + TRACE_BLOCK / 64 - 1) / (TRACE_BLOCK / 64);   // working workgroups
    // Workgroups b and b + 8 share an XCD (round-robin dispatch; speed only, never
    // correctness): give each XCD a contiguous run of packets.
    const int q = nb >> 3, r8 = nb & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (slot >= q + (xcd < r8 ? 1 : 0)) return;
    const int vblock = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + slot;
    const int wave_id = __builtin_amdgcn_readfirstlane(vblock * (TRACE_BLOCK / 64)
                                                       + (threadIdx.x >> 6));
    if (MODE == MODE_CUMULATIVE || MODE == MODE_HITS || MODE == MODE_CUM_D4 || MODE == MODE_HITS_D4) {
        if (FAST) {
            static_assert(TRACE_BLOCK >= 256, "one table entry per thread");
            float2 e = make_float2(0.f, 0.f);
            if (threadIdx.x < N_TABLE - 1) {
                const double y0 = c_kernel_table[threadIdx.x], dy = c_kernel_table[threadIdx.x + 1] - y0;
                e = make_float2(float(y0 - double(threadIdx.x) * dy), float(dy));
            }
            if (threadIdx.x < LUTF_N) s_lutf[threadIdx.x] = e;
        } else if (threadIdx.x < N_TABLE) {
            const int i0 = threadIdx.x;
            const double y0 = c_kernel_table[i0];
            const double y1 = i0 + 1 < N_TABLE ? c_kernel_table[i0 + 1] : y0;
            s_lut[threadIdx.x] = make_double2(y0, y1 - y0);
        }
        __syncthreads();
    }
    int packet = wave_id / split, part = wave_id - packet * split;
    // Primitive range owned by this wave (RANGE_SPLIT).
    int prim_lo = 0, prim_hi = 0x7fffffff;
    if (RANGE_SPLIT) {
        if (wave_id >= *a.n_wave_map) return;
        const int4 wm = a.wave_map[wave_id];
        packet = wm.x; part = 0;
        prim_lo = wm.y << a.chunk_shift;
        prim_hi = wm.z << a.chunk_shift;
        if (prim_lo >= prim_hi) return;
    }
    const int first_ray = packet * a.width;
    if (first_ray >= a.n_rays) return;
    // Summation classes owned by this wave: [own_lo, own_hi).
    const int classes_per_part = SUM_CLASSES / split;
    const int own_lo = part * classes_per_part, own_hi = own_lo + classes_per_part;
    auto owns_granule = [&](const int g) {
        if (RANGE_SPLIT) { const int p = g << GRANULE_SHIFT; return p >= prim_lo && p < prim_hi; }
        const int c = g & (SUM_CLASSES - 1);
        return c >= own_lo && c < own_hi;
    };
    // True if no primitive of [first, first + count) belongs to this wave (class split: ranges
    // of up to two granules are decided exactly; longer ones are descended / swept).
    auto foreign_range = [&](const int first, const int count) {
        if (RANGE_SPLIT) return first + count <= prim_lo || first >= prim_hi;
        const int g0 = first >> GRANULE_SHIFT, g1 = (first + count - 1) >> GRANULE_SHIFT;
        return g1 - g0 <= 1 && !owns_granule(g0) && !owns_granule(g1);
    };
    const int slot_index = first_ray + lane;
    const bool valid = lane < a.width && slot_index < a.n_rays;
    // Idle and tail lanes re-trace the packet's last ray so that they do not widen the packet.
    const int slot_clamped = valid ? slot_index : min(first_ray + a.width, a.n_rays) - 1;
    const int ray_index = a.perm ? int(a.perm[slot_clamped]) : slot_clamped;
    const float* rp = a.rays + 7 * size_t(ray_index);
    const float dx = rp[0], dy = rp[1], dz = rp[2];
    const float ox = rp[3], oy = rp[4], oz = rp[5];
    const float len = rp[6];
    const float ix = 1.f / dx, iy = 1.f / dy, iz = 1.f / dz; // bintree_trace.cuh:111-114

    Beam beam;
    beam.olo[0] = wave_min(ox); beam.ohi[0] = wave_max(ox);
    beam.olo[1] = wave_min(oy); beam.ohi[1] = wave_max(oy);
    beam.olo[2] = wave_min(oz); beam.ohi[2] = wave_max(oz);
    beam.dlo[0] = wave_min(dx); beam.dhi[0] = wave_max(dx);
    beam.dlo[1] = wave_min(dy); beam.dhi[1] = wave_max(dy);
    beam.dlo[2] = wave_min(dz); beam.dhi[2] = wave_max(dz);

    // Axis-aligned packet?  (wave-uniform; tail lanes replicate a valid ray)
    int axis = -1;
    if (MODE != MODE_HITS && MODE != MODE_TRI && MODE != MODE_HITS_D4) {
        const unsigned long long all = ~0ull;
        const bool zx = dx == 0.f, zy = dy == 0.f, zz = dz == 0.f;
        if (__builtin_amdgcn_ballot_w64(zy && zz && fabsf(dx) == 1.f) == all) axis = 0;
        else if (__builtin_amdgcn_ballot_w64(zx && zz && fabsf(dy) == 1.f) == all) axis = 1;
        else if (__builtin_amdgcn_ballot_w64(zx && zy && fabsf(dz) == 1.f) == all) axis = 2;
    }
    // Permuted per-lane constants for the axis path: along-axis origin/direction, then the
    // two perpendicular origins in component order.
    const float oa = axis == 0 ? ox : axis == 1 ? oy : oz;
    const float da = axis == 0 ? dx : axis == 1 ? dy : dz;
    // (s_a - o_a) * d_a with d_a = +-1 is the correctly rounded +-(s_a - o_a): one FMA
    // s_a * d_a + (-o_a * d_a) gives the same bits (both products are exact).
    const float noda = -(oa * da);
    // Pencil packets: one origin, directions inside a cone narrower than 60 degrees.
    // The 20 constants live in LDS (one record per wave) and are re-read by every culling round
    // of a pencil sweep: held in registers they would be live across the whole walk and cost
    // every instantiation 16 VGPRs -- two waves of occupancy for the orthographic kernels that
    // never use them.
    __shared__ Pencil s_pencil[TRACE_BLOCK / 64];
    Pencil pencil;
    bool is_pencil = false;
    if (axis < 0 && MODE != MODE_STATS && beam.olo[0] == beam.ohi[0] && beam.olo[1] == beam.ohi[1]
        && beam.olo[2] == beam.ohi[2]) {
        float sx = dx, sy = dy, sz = dz;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); sz += __shfl_xor(sz, off);
        }
        const float inv = 1.0f / sqrtf(sx * sx + sy * sy + sz * sz);
        const float ax = sx * inv, ay = sy * inv, az = sz * inv;
        // |a x d| = sin(angle): accurate for the small angles that matter; + margin for the
        // rounding of a and of the (unit) directions.
        const float cx = ay * dz - az * dy, cy = az * dx - ax * dz, cz = ax * dy - ay * dx;
        const float sin_t = wave_max(sqrtf(cx * cx + cy * cy + cz * cz)) + 2e-6f;
        const float cos_min = wave_min(ax * dx + ay * dy + az * dz);
        if (cos_min > 0.5f && sin_t < 0.8660254f) {   // also false for NaN (zero mean direction)
            is_pencil = true;
            pencil.ox = beam.olo[0]; pencil.oy = beam.olo[1]; pencil.oz = beam.olo[2];
            pencil.ax = ax; pencil.ay = ay; pencil.az = az;
            pencil.sin_t = sin_t;
            pencil.cos_t = sqrtf(1.0f - sin_t * sin_t);
            // tangent frame: u perpendicular to a (built from the axis' smallest component), v = a x u
            float ux, uy, uz;
            if (fabsf(ax) <= fabsf(ay) && fabsf(ax) <= fabsf(az)) { ux = 0.f; uy = -az; uz = ay; }
            else if (fabsf(ay) <= fabsf(az)) { ux = az; uy = 0.f; uz = -ax; }
            else { ux = -ay; uy = ax; uz = 0.f; }
            const float un = 1.0f / sqrtf(ux * ux + uy * uy + uz * uz);
            ux *= un; uy *= un; uz *= un;
            const float vx = ay * uz - az * uy, vy = az * ux - ax * uz, vz = ax * uy - ay * ux;
            // gnomonic co-ordinates of every direction (d . a > 0.5 here), their extremes
            const float da_ = ax * dx + ay * dy + az * dz;
            const float tu = (ux * dx + uy * dy + uz * dz) / da_, tv = (vx * dx + vy * dy + vz * dz) / da_;
            const float m = 4e-6f;   // rounding of the frame and of the directions
            const float tu_lo = wave_min(tu) - m, tu_hi = wave_max(tu) + m;
            const float tv_lo = wave_min(tv) - m, tv_hi = wave_max(tv) + m;
            // side plane "t_u <= tu_hi": points p with p.u - tu_hi p.a <= 0; outward normal u - tu_hi a
            auto plane = [&](int k, float cu, float cv, float ca) {
                float px = cu * ux + cv * vx + ca * ax, py = cu * uy + cv * vy + ca * ay,
                      pz = cu * uz + cv * vz + ca * az;
                const float pn = 1.0f / sqrtf(px * px + py * py + pz * pz);
                pencil.nx[k] = px * pn; pencil.ny[k] = py * pn; pencil.nz[k] = pz * pn;
            };
            plane(0, 1.f, 0.f, -tu_hi);
            plane(1, -1.f, 0.f, tu_lo);
            plane(2, 0.f, 1.f, -tv_hi);
            plane(3, 0.f, -1.f, tv_lo);
            if (lane == 0) s_pencil[threadIdx.x >> 6] = pencil;
        }
    }
    // For the range-check-free sweep (below): the packet's extremes of -o_a d_a and of the ray
    // length, and whether all rays point the same way along the axis.
    float noda_lo = 0.f, noda_hi = 0.f, len_lo = 0.f, da0 = 0.f;
    bool same_sense = false;
    if (axis >= 0) {
        noda_lo = wave_min(noda); noda_hi = wave_max(noda); len_lo = wave_min(len);
        const unsigned long long fwd = __builtin_amdgcn_ballot_w64(da > 0.f);
        same_sense = fwd == 0ull || fwd == ~0ull;
        da0 = fwd ? 1.f : -1.f;
    }
    const float o1 = axis == 0 ? oy : ox;
    const float o2 = axis == 2 ? oy : oz;
    // Axis-aligned packets, fast integral: a sphere hit by SOME ray of the packet lies within
    // h + D of every other ray (D = diagonal of the origins' box), i.e. at a table position below
    // 50 (1 + D / h).  For h >= D / 3 that stays below 200 < LUTF_N: such survivors need no clamp.
    float fat_r2 = INFINITY;
    if (FAST && axis >= 0) {
        const float ex = beam.ohi[0] - beam.olo[0], ey = beam.ohi[1] - beam.olo[1], ez = beam.ohi[2] - beam.olo[2];
        const float e1 = axis == 0 ? ey : ex, e2 = axis == 2 ? ey : ez;
        fat_r2 = (e1 * e1 + e2 * e2) * (1.0f / 9.0f) * 1.0001f;
    }
    const int treelet = axis >= 0 ? a.treelet_axis : a.treelet;
    // Origin lattice of an axis-aligned packet.  The beam cull bounds b^2 at the point of the
    // origin RECTANGLE nearest to the sphere; a sphere smaller than the ray spacing can lie
    // inside the rectangle and still between the rays -- in the dense cores of clustered SPH
    // data most do (h << pixel), and every one of them used to cost all 64 lanes a test (10^7
    // particles, 90 % of them in 50 clumps: 180 863 surviving candidates in the heaviest packet
    // against 2273 in the median one, whose wave outlived the launch 30x).  If the packet's
    // origins take at most 8 distinct values in each perpendicular co-ordinate (pixel grids do:
    // 8 x 8 tiles), the tables of those values give the exact minimum of the rays' own b^2
    // expression over the lattice {x_i} x {y_j} -- a superset of the rays --: |s - x| rounds
    // monotonically in the true difference, so the nearest table value minimises the rounded |q|
    // in each co-ordinate, and b^2 is monotone in both.  No margin, same bits as the ray's test.
    constexpr bool LATTICE = LAT;
    __shared__ float s_lat[LATTICE ? TRACE_BLOCK / 64 : 1][2][8];
    // Spheres with r^2 below this can fall between the rays; 0 = no lattice.  Kept in LDS and
    // re-read where it is used (once per group of cluster tests): the split kernels have no
    // scalar register to spare.
    __shared__ float s_lat_r2[LATTICE ? TRACE_BLOCK / 64 : 1];
    if (LATTICE && lane == 0) s_lat_r2[threadIdx.x >> 6] = 0.f;
    if (LATTICE && axis >= 0) {
        // The distinct values of each co-ordinate, in any order (the nearest one is found by a
        // plain minimum): take the first lane not yet accounted for, strike every lane that
        // holds its value, eight times at most.  NaN origins strike nobody: no lattice.
        bool ok = true;
        float cell2 = 0.f;
#pragma unroll
        for (int dim = 0; dim < 2; ++dim) {
            const float o = dim ? o2 : o1;
            unsigned long long todo = ~0ull;
            int n_val = 0;
            float v = 0.f, v_lo = INFINITY, v_hi = -INFINITY;
#pragma unroll 1
            for (int k = 0; k < 8 && todo != 0ull; ++k) {
                v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o),
                                                                        __builtin_ctzll(todo)));
                v_lo = fminf(v_lo, v); v_hi = fmaxf(v_hi, v);
                todo &= ~__builtin_amdgcn_ballot_w64(o == v);
                if (lane == 0) s_lat[threadIdx.x >> 6][dim][k] = v;
                ++n_val;
            }
            ok = ok && todo == 0ull;
            if (lane == 0)
                for (int k = n_val; k < 8; ++k) s_lat[threadIdx.x >> 6][dim][k] = v;   // padding repeats
            // mean spacing (a gate only: it decides which spheres are worth the lattice test)
            const float gap = n_val > 1 ? (v_hi - v_lo) / float(n_val - 1) : 0.f;
            cell2 += gap * gap;
        }
        // spheres wider than the cell diagonal meet a ray wherever they lie inside the lattice
        if (lane == 0) s_lat_r2[threadIdx.x >> 6] = ok ? cell2 : 0.f;
    }

    int count = 0;
    // Chunk bookkeeping of the split per-hit trace (see TraceArgs): the counting pass adds each
    // lane's hits of a chunk to chunk_counts when the walk leaves the chunk; the per-hit pass
    // repositions each lane's output cursor when it enters one.
    constexpr bool CHUNKED = (SPLIT && (MODE == MODE_COUNT || MODE == MODE_HITS));
    int cur_chunk = -1;        // wave-uniform
    int count_at_chunk = 0;
    float sum = 0.f;        // accumulator of the current granule's class (MODE_CUMULATIVE)
    // Class accumulators of this wave's lanes (one wave = one row of the workgroup's array).
    // (MODE_CUM_D4: the same classes, accumulated and combined in double)
    constexpr bool CLASSES_F = (MODE == MODE_CUMULATIVE), CLASSES_D = (MODE == MODE_CUM_D4);
    constexpr bool CLASSES = CLASSES_F || CLASSES_D;
    __shared__ float s_class[CLASSES_F ? TRACE_BLOCK / 64 : 1][CLASSES_F ? SUM_CLASSES : 1][CLASSES_F ? 64 : 1];
    __shared__ double s_class_d[CLASSES_D ? TRACE_BLOCK / 64 : 1][CLASSES_D ? SUM_CLASSES : 1][CLASSES_D ? 64 : 1];
    const int wv_acc = threadIdx.x >> 6;
    if (CLASSES_F) {
#pragma unroll
        for (int c = 0; c < SUM_CLASSES; ++c) s_class[wv_acc][c][lane] = 0.f;
    }
    if (CLASSES_D) {
#pragma unroll
        for (int c = 0; c < SUM_CLASSES; ++c) s_class_d[wv_acc][c][lane] = 0.0;
    }
    double sum_d = 0.0;     // MODE_CUM_D4: accumulator of the current granule's class
    int cur_granule = -1;            // wave-uniform
    int cur_granule_end = 0;         // first primitive past the current granule
    bool cur_owned = true;
    auto enter_granule = [&](const int prim) {
        if (cur_granule >= 0) {
            if (CLASSES_D) s_class_d[wv_acc][cur_granule & (SUM_CLASSES - 1)][lane] = sum_d;
            else s_class[wv_acc][cur_granule & (SUM_CLASSES - 1)][lane] = sum;
        }
        cur_granule = prim >> GRANULE_SHIFT;
        cur_granule_end = (cur_granule + 1) << GRANULE_SHIFT;
        cur_owned = !SPLIT || owns_granule(cur_granule);
        if (CLASSES_D) sum_d = s_class_d[wv_acc][cur_granule & (SUM_CLASSES - 1)][lane];
        else sum = s_class[wv_acc][cur_granule & (SUM_CLASSES - 1)][lane];
    };
    int write_at = 0;
    auto leave_chunk = [&]() {
        if (MODE == MODE_COUNT && cur_chunk >= 0 && valid && count != count_at_chunk)
            atomicAdd(&a.chunk_counts[size_t(ray_index) * a.n_chunks + cur_chunk],
                      count - count_at_chunk);
        count_at_chunk = count;
    };
    // MODE_TRI: RayEntry_tri (tris_trace.cuh:63-73): closest index -1, t_min = length (1 + eps)
    int tri_data = -1;
    float tri_tmin = len * (1.f + 0.000001f);
    const double ddx = dx, ddy = dy, ddz = dz;
    if (MODE == MODE_HITS || MODE == MODE_HITS_D4) write_at = a.offsets[ray_index];
    const double rdx = dx, rdy = dy, rdz = dz;
    // MODE_HITS: every ray owns a contiguous output segment, so lanes writing hit by hit
    // touch 64 different cache lines per store and the partial lines thrash L2 (measured:
    // 48 GB/s of useful output).  Hits are staged per lane in LDS (HIT_CAP entries, entry-major,
    // padded to 65 columns so that neither the per-lane appends nor the per-ray drains conflict)
    // and drained by the whole wave: HIT_CAP lanes per ray write HIT_CAP consecutive elements
    // (2.1 -> 5.0 ms ... 56 -> 20 ms at 0.4 ... 2.1 G hits).  With few packets the walk is
    // latency-bound and the extra instructions cost more than the stores: the host picks the
    // staged instantiation from the packet count.
    constexpr int HIT_CAP = 8;
    constexpr bool STAGE_HITS = ALT && MODE == MODE_HITS;
    __shared__ float s_hits[STAGE_HITS ? TRACE_BLOCK / 64 : 1][STAGE_HITS ? 3 : 1]
                           [STAGE_HITS ? HIT_CAP : 1][STAGE_HITS ? 65 : 1];
    int staged = 0;            // hits of this lane waiting in LDS; they belong at write_at - staged
    auto drain_hits = [&]() {
        const int wvh = threadIdx.x >> 6;
        constexpr int RAYS_PER_PASS = 64 / HIT_CAP;
        const int g = lane / HIT_CAP, e = lane % HIT_CAP;
#pragma unroll 1
        for (int pass = 0; pass < HIT_CAP; ++pass) {
            const int src = pass * RAYS_PER_PASS + g;               // the lane whose hits these are
            const int n_src = __shfl(staged, src);
            const int first = __shfl(write_at - staged, src);
            if (e < n_src) {
                a.hit_idx[first + e] = __float_as_int(s_hits[wvh][0][e][src]);
                a.hit_integral[first + e] = s_hits[wvh][STAGE_HITS ? 1 : 0][e][src];
                a.hit_dist[first + e] = s_hits[wvh][STAGE_HITS ? 2 : 0][e][src];
            }
        }
        staged = 0;
    };
    auto enter_chunk = [&](const int chunk) {
        leave_chunk();
        cur_chunk = chunk;
        if (MODE == MODE_HITS) {
            // the staged hits belong to the chunk being left: out before the cursor moves
            if (STAGE_HITS && __builtin_amdgcn_ballot_w64(staged != 0) != 0ull) drain_hits();
            write_at = a.chunk_off[size_t(ray_index) * a.n_chunks + chunk];
        }
    };
    uint32_t st_nodes = 0, st_leaves = 0, st_tested = 0;

    // Packet stack: entry e lives in lane (e & 63) of stk0 (e < 64) or stk1.
    int stk0 = 0, stk1 = 0;
    // MODE_STATS: per entry, the lanes that reach it on their own.
    int ml0 = 0, mh0 = 0, ml1 = 0, mh1 = 0;
    int sp = -1;
    int junk = 0;
    bool overflow = false;

    // v_writelane is not exposed as a builtin by this hipcc; a push is a lane-select
    // (v_cmp_eq + v_cndmask with the scalar stack pointer), a pop is v_readlane.
    auto push = [&](const int value, const unsigned long long alive) {
        if (sp >= 127) { overflow = true; return; }
        ++sp;
        if (sp < 64) {
            const bool me = lane == sp;
            stk0 = me ? value : stk0;
            if (MODE == MODE_STATS) {
                ml0 = me ? int(uint32_t(alive)) : ml0;
                mh0 = me ? int(uint32_t(alive >> 32)) : mh0;
            }
        } else {
            const bool me = lane == sp - 64;
            stk1 = me ? value : stk1;
            if (MODE == MODE_STATS) {
                ml1 = me ? int(uint32_t(alive)) : ml1;
                mh1 = me ? int(uint32_t(alive >> 32)) : mh1;
            }
        }
    };

    // Axis-aligned packets of the hit-count / column-density traces do not walk the tree: the
    // primitives are Morton-sorted, so GROUPS of 2^group_shift consecutive ones (4096 up to 16.7 M
    // primitives; <= 4096 groups) are compact cells, and their boxes -- the unions of their
    // clusters' boxes, written by the pre-pass behind the cluster records -- are tested 64 at a
    // time, lane j <-> group g + j, against the packet's origin rectangle and its extent along
    // the axis.  Every surviving group is swept like a subtree the walk would have stopped at
    // (cluster tests, then culling rounds), in ascending order.  What the walk cost -- ~100 node
    // visits per packet, each a dependent load, repeated by every wave of a split packet -- is
    // replaced by n / 2^18 independent, coalesced passes (39 at 10^7 primitives).  Conservative
    // like the walk and the cluster tests: a group is dropped only if no ray of the packet can
    // hit any member, so the per-ray hit sets, and with them every sum, are unchanged.
    constexpr bool FLAT_OK = (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE || MODE == MODE_COUNT_D4
                              || MODE == MODE_CUM_D4);
    // Only packets whose group test is sharp: axis-aligned ones (origin rectangle) and pencils
    // (one origin: the bundle's side planes).  A GENERAL packet's beam -- boxes around its origins
    // and directions -- keeps nearly every group, where the walk's per-ray slab tests prune
    // exactly: 262144 random rays through 10^7 spheres took 990 ms with the group passes against
    // 145 ms with the walk.  Those packets keep the walk.
    bool flat = FLAT_OK && (axis >= 0 || is_pencil);
    int groups_kept = 0;                    // surviving groups of this packet (wave-uniform)
    const float4* const group_boxes = a.C + 2 * ((size_t(a.n_prims) + 63) >> 6) + 1;
    const int n_groups = FLAT_OK ? int((size_t(a.n_prims) + (size_t(1) << a.group_shift) - 1) >> a.group_shift) : 0;
    unsigned long long group_mask = 0ull;   // surviving groups of the current pass, not yet swept
    int group_next = 0;                     // first group of the next pass
    // All passes are made up front, their loads in flight together (made one by one between the
    // sweeps, each pass exposed its load latency: 39 of them per wave at 10^7 primitives); the
    // survivor masks wait in LDS.
    __shared__ unsigned long long s_group_mask[FLAT_OK ? TRACE_BLOCK / 64 : 1][FLAT_OK ? 64 : 1];
    if (flat && axis < 0) {
        // pencil packets: their cluster test on the group boxes
        const int wvg = threadIdx.x >> 6;
        for (int g0 = 0; g0 < n_groups; g0 += 128) {
            float4 lo[2], hi[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int gj = min(g0 + 64 * k + lane, n_groups - 1);
                lo[k] = group_boxes[2 * size_t(gj)];
                hi[k] = group_boxes[2 * size_t(gj) + 1];
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int gk = g0 + 64 * k;
                if (gk < n_groups) {
                    const bool may = cluster_may_hit<-2>(lo[k], hi[k], beam, &s_pencil[wvg]);
                    const int n_g = n_groups - gk;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(may)
                        & (n_g >= 64 ? ~0ull : ((1ull << n_g) - 1ull));
                    if (lane == 0) s_group_mask[wvg][gk >> 6] = m;
                    groups_kept += __builtin_popcountll(m);
                }
            }
        }
    } else if (flat) {
        const int d1 = axis == 0 ? 1 : 0, d2 = axis == 2 ? 1 : 2;
        const float flat_lo1 = d1 == 0 ? beam.olo[0] : beam.olo[1], flat_hi1 = d1 == 0 ? beam.ohi[0] : beam.ohi[1];
        const float flat_lo2 = d2 == 1 ? beam.olo[1] : beam.olo[2], flat_hi2 = d2 == 1 ? beam.ohi[1] : beam.ohi[2];
        // the packet's extent along the axis: every ray's segment [o, o + d len], d = +-1
        const float oa = axis == 0 ? ox : axis == 1 ? oy : oz;
        const float da = axis == 0 ? dx : axis == 1 ? dy : dz;
        const float ea = oa + da * len;
        const float flat_alo = wave_min(fminf(oa, ea));
        const float flat_ahi = wave_max(fmaxf(oa, ea));
        const int wvg = threadIdx.x >> 6;
        auto group_test = [&](const float4 glo, const float4 ghi, const int g0) {
            const float l1 = axis == 0 ? glo.y : glo.x, h1 = axis == 0 ? ghi.y : ghi.x;
            const float l2 = axis == 2 ? glo.y : glo.z, h2 = axis == 2 ? ghi.y : ghi.z;
            const float la = axis == 0 ? glo.x : axis == 1 ? glo.y : glo.z;
            const float ha = axis == 0 ? ghi.x : axis == 1 ? ghi.y : ghi.z;
            // (negated forms: a NaN bound keeps the group)
            const bool may = !(l1 > flat_hi1) && !(h1 < flat_lo1) && !(l2 > flat_hi2) && !(h2 < flat_lo2)
                && !(la > flat_ahi) && !(ha < flat_alo);
            const int n_g = n_groups - g0;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(may)
                & (n_g >= 64 ? ~0ull : n_g <= 0 ? 0ull : ((1ull << n_g) - 1ull));
            if (lane == 0) s_group_mask[wvg][g0 >> 6] = m;
            groups_kept += __builtin_popcountll(m);
        };
        for (int g0 = 0; g0 < n_groups; g0 += 256) {
            // four passes' boxes fetched before the first test (the last group's box pads the tail)
            float4 lo[4], hi[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int gj = min(g0 + 64 * k + lane, n_groups - 1);
                lo[k] = group_boxes[2 * size_t(gj)];
                hi[k] = group_boxes[2 * size_t(gj) + 1];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (g0 + 64 * k < n_groups) group_test(lo[k], hi[k], g0 + 64 * k);
        }
    }
    // A packet that keeps many groups (a wide pencil: few rays per origin; spheres far larger than
    // their cells) would sweep each of them with a pass of cluster tests, where the walk's per-ray
    // slab tests prune whole subtrees: beyond FLAT_MAX_GROUPS it walks after all (the passes made
    // are lost: <= 64, ~2 k instructions).  32768 rays from each of two origins through 10^7 small
    // spheres: 16.5 ms walking, 31 ms with every packet on the group passes, 16.6 ms with this rule.
    constexpr int FLAT_MAX_GROUPS = 256;
    if (flat && groups_kept > FLAT_MAX_GROUPS) flat = false;
    if (!flat) push(*a.root, ~0ull);
    unsigned long long st_walk = 0, st_cluster = 0, st_cull = 0, st_surv = 0, st_rounds = 0, st_nsurv = 0;
    const unsigned long long st_begin = STAMP_NOW();
    (void)st_walk; (void)st_cluster; (void)st_cull; (void)st_surv; (void)st_rounds; (void)st_nsurv; (void)st_begin;

    for (;;) {
        const unsigned long long st_t0 = STAMP_NOW(); (void)st_t0;
        int sweep_first = 0, sweep_count = 0;
        bool sweep = false;
        if (flat) {
            while (group_mask == 0ull && group_next < n_groups) {
                group_mask = s_group_mask[threadIdx.x >> 6][group_next >> 6];
                group_mask = (unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(group_mask))))
                    | ((unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(group_mask >> 32)))) << 32);
                group_next += 64;
            }
            if (group_mask == 0ull) break;
            const int g = group_next - 64 + __builtin_ctzll(group_mask);
            group_mask &= group_mask - 1ull;
            sweep_first = g << a.group_shift;
            sweep_count = min(1 << a.group_shift, a.n_prims - sweep_first);
            // A wave of a split packet skips groups none of whose granules it owns.
            if (SPLIT) {
                bool mine = false;
                for (int gr = sweep_first >> GRANULE_SHIFT; gr <= (sweep_first + sweep_count - 1) >> GRANULE_SHIFT
                         && gr < (sweep_first >> GRANULE_SHIFT) + SUM_CLASSES; ++gr)
                    mine = mine || owns_granule(gr);
                if (!mine) continue;
            }
            sweep = true;
        } else {
        if (sp < 0) break;
        int idx;
        unsigned long long alive_mask = ~0ull;
        if (sp < 64) {
            idx = __builtin_amdgcn_readlane(stk0, sp);
            if (MODE == MODE_STATS)
                alive_mask = (unsigned long long)uint32_t(__builtin_amdgcn_readlane(ml0, sp))
                    | ((unsigned long long)uint32_t(__builtin_amdgcn_readlane(mh0, sp)) << 32);
        } else {
            idx = __builtin_amdgcn_readlane(stk1, sp - 64);
            if (MODE == MODE_STATS)
                alive_mask = (unsigned long long)uint32_t(__builtin_amdgcn_readlane(ml1, sp - 64))
                    | ((unsigned long long)uint32_t(__builtin_amdgcn_readlane(mh1, sp - 64)) << 32);
        }
        --sp;
        const bool alive = (alive_mask >> lane) & 1ull;

        if (idx < a.n_nodes) {
            const float4* np = a.nodes + 4 * size_t(idx);
            // Node and span are fetched together (one scalar-load round trip).
            const float4 n0 = np[0];
            const float4 L = np[1];
            const float4 R = np[2];
            const float4 Z = np[3];
            int2 span = make_int2(0, 0x7fffffff);
#ifdef GRACE_PACKET_STATS
            if (treelet > 0) span = a.node_prims[idx];
#else
            if (MODE != MODE_STATS && (treelet > 0 || SPLIT)) span = a.node_prims[idx];
#endif
            // A wave of a split packet skips subtrees outside its primitive range.
            if (SPLIT && foreign_range(span.x, span.y)) continue;
            if (span.y <= treelet) {
                sweep = true; sweep_first = span.x; sweep_count = span.y;
            } else {
            // (A wave-uniform box-overlap test of the packet's bounding box instead of this per-ray
            // slab test was tried three times for axis-aligned packets -- twelve compares with scalar
            // operands in rounds 1 and 2; in round 3 ONE vector compare, lane j holding dword j of
            // the node record and its own bound, 2 vector instructions per node instead of ~45: same
            // node count, no gain each time (2.91 vs 2.92 ms).  The ~100 node tests per packet are a
            // chain of dependent loads; their vector work is not what the walk waits for.)
            bool hit_l, hit_r;
            aabbs_hit(ix, iy, iz, ox, oy, oz, len, L, R, Z, hit_l, hit_r);
            const unsigned long long vote_r = __builtin_amdgcn_ballot_w64(hit_r);
            const unsigned long long vote_l = __builtin_amdgcn_ballot_w64(hit_l);
#ifdef GRACE_PACKET_STATS
#else
            if (MODE == MODE_STATS && alive) ++st_nodes;
#endif
            if (vote_r) push(__float_as_int(n0.y),
                             MODE == MODE_STATS ? __builtin_amdgcn_ballot_w64(hit_r && alive) : 0ull);
            if (vote_l) push(__float_as_int(n0.x),
                             MODE == MODE_STATS ? __builtin_amdgcn_ballot_w64(hit_l && alive) : 0ull);
            }
        } else {
            const int4 lf = a.leaves[idx - a.n_nodes];
            if (SPLIT && foreign_range(lf.x, lf.y)) continue;
            sweep = true; sweep_first = lf.x; sweep_count = lf.y;
#ifndef GRACE_PACKET_STATS
            if (MODE == MODE_STATS && alive) { ++st_leaves; st_tested += uint32_t(lf.y); }
#endif
        }
        } // tree walk
        STAMP_ADD(st_walk, st_t0);
        if (sweep) {
            const int2 leaf = make_int2(sweep_first, sweep_count);
            // Touch the next stack entry's cache line now; its pop follows this leaf.
            int warm = 0;
            if (sp >= 0) {
                const int nxt = sp < 64 ? __builtin_amdgcn_readlane(stk0, sp)
                                        : __builtin_amdgcn_readlane(stk1, sp - 64);
                warm = nxt < a.n_nodes
                    ? reinterpret_cast<const int*>(a.nodes)[16 * size_t(nxt)]
                    : reinterpret_cast<const int*>(a.leaves)[4 * size_t(nxt - a.n_nodes)];
            }
            // The sweep is instantiated per packet kind (general / axis x, y, z) so that the
            // component selection of the axis path is resolved at compile time.
            auto sweep_range = [&](auto ax_tag) {
                constexpr int AX = decltype(ax_tag)::value;
            constexpr bool NEED_B = (MODE == MODE_CUMULATIVE || MODE == MODE_HITS);
            constexpr bool LEAN4 = FAST && AX >= 0;
            const int wv = threadIdx.x >> 6;
            float4* const tile4 = reinterpret_cast<float4*>(&s_tile[wv][0][0]);   // (the same bytes, as 16-byte records)
            const int r_lo = leaf.x, r_hi = leaf.x + leaf.y;   // the swept primitives (wave-uniform)
            const int c_first = r_lo >> 6, c_last = (r_hi - 1) >> 6;
            // Lane j's candidate of cluster c: primitive 64 c + j, clamped into the range (idle
            // lanes then hold a valid candidate and the tests need no control flow).
            double4 mined_next = make_double4(0., 0., 0., 0.);   // *_D4: the candidate's double4 record
            // (Round 3, measured and rejected: buffer loads for the candidates -- a resource based at
            // the range's first cluster, the cluster as scalar offset, the lane as constant vector
            // offset: no per-round address arithmetic (six vector instructions) and no clamps, the
            // hardware's range check returning zeros for the masked lanes -- same images, frame
            // kernel +-0, hit counts -1.7 %, 1/8 shard +2.3 %: not kept.)
            auto load_cluster = [&](const int c, float4& m4, float2& m2) {
                const int pj = min(max((c << 6) + lane, r_lo), r_hi - 1);
                m4 = a.A[pj];
                if (LDS_TILE && NEED_B) m2 = a.B[pj];
                if (D4) mined_next = reinterpret_cast<const double4*>(a.spheres_d)[pj];
            };
            // The range's clusters, 64 at a time: lane j decides for cluster cg + j whether ANY ray
            // of the packet can hit ANY of its members (cluster_may_hit); culling rounds then run
            // over the surviving clusters only, in ascending order.
            for (int cg = c_first; cg <= c_last; cg += 64) {
                const unsigned long long st_t1 = STAMP_NOW(); (void)st_t1;
                unsigned long long cmask = 1ull;
                unsigned long long small_mask = ~0ull;   // clusters with members smaller than the ray spacing
                if (c_last != c_first) {   // (one cluster -- a small leaf -- goes straight to its round)
                    const int cj = min(cg + lane, c_last);
                    const float4 blo = a.C[2 * size_t(cj)], bhi = a.C[2 * size_t(cj) + 1];
                    const bool c_may = cluster_may_hit<AX>(blo, bhi, beam, &s_pencil[wv]);
                    if (LATTICE && AX >= 0) small_mask = __builtin_amdgcn_ballot_w64(blo.w < s_lat_r2[wv]);
                    const int n_c = min(64, c_last - cg + 1);
                    cmask = __builtin_amdgcn_ballot_w64(c_may)
                        & (n_c >= 64 ? ~0ull : ((1ull << n_c) - 1ull));
                    // A wave of a split packet sweeps its own clusters only (a cluster lies inside
                    // one granule: 1024 = 16 x 64).
                    if (SPLIT) cmask &= __builtin_amdgcn_ballot_w64(owns_granule(cj >> (GRANULE_SHIFT - 6)));
#ifdef GRACE_PACKET_STATS
                    if (MODE == MODE_STATS) st_leaves += 1;
#endif
                }
                STAMP_ADD(st_cluster, st_t1);
                if (cmask == 0ull) continue;
                int cnext = cg + __builtin_ctzll(cmask);
                cmask &= cmask - 1ull;
                // A round's 64 candidates are fetched one round ahead (vector loads, 16 B/lane,
                // coalesced) so that their latency hides behind the previous round's survivors.
                // (Round 3, measured and rejected: two register sets taking turns with the round
                // body instantiated twice -- no copies, but +11 % on the frame kernel; fetching only
                // after the round has staged its survivors, into the same registers -- no copies
                // either, +2.5 %: the loads need the whole round's lead; staging the survivors of
                // several test-free rounds behind one another and running the survivor loop once
                // per ~64 of them -- a third of the loop prologues, bit-identical images, +6 %:
                // the rounds without a loop leave the prefetch no time.)
                float4 mine_next;
                float2 mineb_next = make_float2(0.f, 0.f);
                load_cluster(cnext, mine_next, mineb_next);
                for (;;) {
                    const unsigned long long st_t2 = STAMP_NOW(); (void)st_t2;
                    const int pbase = cnext << 6;          // first primitive of this round's cluster
                    const float4 mine = mine_next;
                    const float2 mineb = mineb_next;
                    const double4 mined = mined_next;
                    const bool more = cmask != 0ull;
                    if (more) {
                        cnext = cg + __builtin_ctzll(cmask);
                        cmask &= cmask - 1ull;
                        load_cluster(cnext, mine_next, mineb_next);
                    }
                    const int lo_bit = max(r_lo - pbase, 0), hi_bit = min(r_hi - pbase, 64);
                    const unsigned long long m_mask =
                        (hi_bit >= 64 ? ~0ull : ((1ull << hi_bit) - 1ull)) & (~0ull << lo_bit);
                // Lane j: can ANY ray of the beam come within h of sphere j?
                // (The tests run on every lane -- idle lanes hold a clamped, valid candidate -- so
                // there is no control flow; lane masks are formed from ballots of the bare
                // comparisons and combined on the scalar unit: a ballot of a combined boolean
                // costs two extra vector instructions each.)
                bool may_hit;
                if constexpr (AX >= 0) may_hit = axis_beam_may_hit<AX, FAST>(mine, beam);
                else if constexpr (AX == -2) may_hit = pencil_may_hit(mine, s_pencil[wv]);
                else may_hit = beam_may_hit(mine, beam);
                unsigned long long rest = __builtin_amdgcn_ballot_w64(may_hit) & m_mask;
                bool keep = __builtin_amdgcn_inverse_ballot_w64(rest);   // (the scalar mask itself: no lane compares)
                // Axis packets: if every kept candidate lies inside every ray's [0, length)
                // along the axis -- decided per candidate with the same FMA the rays use, which
                // is monotone in its addend -- the round's survivors skip the two range tests.
                bool lean_round = false;
                if constexpr (AX >= 0 && (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE)) {
                    const float sa = AX == 0 ? mine.x : AX == 1 ? mine.y : mine.z;
                    const unsigned long long inside =
                        __builtin_amdgcn_ballot_w64(__builtin_fmaf(sa, da0, noda_lo) >= 0.0f)
                        & __builtin_amdgcn_ballot_w64(__builtin_fmaf(sa, da0, noda_hi) < len_lo);
                    lean_round = same_sense & ((rest & ~inside) == 0ull);
                }
                bool fat_round = false;
                if constexpr (LEAN4)
                    fat_round = (rest & ~__builtin_amdgcn_ballot_w64(mine.w >= fat_r2)) == 0ull;
                if constexpr (LATTICE && AX >= 0) {
                    // Origin-lattice cull (see the packet set-up): only in rounds over clusters that
                    // hold small spheres, and only if one of them survived the rectangle test.  Kept
                    // behind the round's main cull so that rounds that never take it (every round of
                    // a scene without sub-spacing spheres) run the same instruction stream as before
                    // plus one scalar test.
                    if (((small_mask >> ((pbase >> 6) - cg)) & 1ull) && rest != 0ull
                        && __builtin_amdgcn_ballot_w64(keep && mine.w < s_lat_r2[wv]) != 0ull) {
                        const float s1 = AX == 0 ? mine.y : mine.x;
                        const float s2 = AX == 2 ? mine.y : mine.z;
                        float q1 = INFINITY, q2 = INFINITY;
#pragma unroll 1   // (a rare path: keep its sixteen table values out of the rounds' register budget)
                        for (int k = 0; k < 8; ++k) {
                            q1 = fminf(q1, fabsf(s1 - s_lat[wv][0][k]));
                            q2 = fminf(q2, fabsf(s2 - s_lat[wv][1][k]));
                        }
                        // (fminf drops a NaN; a NaN centre must stay -- sphere_hit's negated
                        // comparisons let it "hit", generic/intersect.h:37-52)
                        const float nan_if_nan = (s1 + s2) * 0.0f;
                        const float b2_lo = (FAST ? __builtin_fmaf(q1, q1, q2 * q2) : q1 * q1 + q2 * q2) + nan_if_nan;
                        keep = keep && !(b2_lo >= mine.w);
                        rest = __builtin_amdgcn_ballot_w64(keep);
                        // (lean_round was decided on a superset of the survivors: still valid)
                    }
                }
#ifdef GRACE_PACKET_STATS
                if (MODE == MODE_STATS) { st_tested += __builtin_popcountll(rest); }
#endif
                // Hit counts and column densities need no candidate index: their tile holds the
                // survivors only, in ascending order (slot = number of kept lanes below), so the
                // k-th survivor sits at slot k -- no bit scanning, and slot addresses that differ
                // by immediates.  The per-hit and triangle modes keep lane-indexed tiles.
                constexpr bool COMPACT = (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE);
                // The round's survivors belong to ONE granule (a cluster never straddles two), so
                // the class accumulator switch and the ownership test of a split packet stay out
                // of the per-survivor loop.
                bool skip_round = rest == 0ull;
                if (!skip_round && (CLASSES || SPLIT)) {
                    const int pf = pbase + __builtin_ctzll(rest);
                    if (CLASSES && pf >= cur_granule_end) enter_granule(pf); // ascending index
                    if (SPLIT && !(CLASSES ? cur_owned : owns_granule(pf >> GRANULE_SHIFT)))
                        skip_round = true;
                    else if (CHUNKED && (MODE == MODE_HITS || a.chunk_counts) && (pf >> a.chunk_shift) != cur_chunk)
                        enter_chunk(pf >> a.chunk_shift);
                }
                if (!skip_round) {
                if (LDS_TILE) {
                    // Stage the round's candidates; survivors are then broadcast-read from LDS
                    // into VGPRs (in-order LDS returns, no scalar-load round trips, VGPR operands).
                    const int slot = COMPACT
                        ? int(__builtin_amdgcn_mbcnt_hi(uint32_t(rest >> 32),
                                                        __builtin_amdgcn_mbcnt_lo(uint32_t(rest), 0u)))
                        : lane;
                    if (LEAN4 && lean_round) {
                        // test-free round of an axis-aligned packet: all a survivor needs is the two
                        // perpendicular co-ordinates and the two 1/h terms -- ONE 16-byte record
                        const float s1 = AX == 0 ? mine.y : mine.x;
                        const float s2 = AX == 2 ? mine.y : mine.z;
                        if (keep) tile4[slot] = make_float4(s1, s2, mineb.x, mineb.y);
                    } else if (!COMPACT || keep) {
                        s_tile[wv][0][slot] = make_float2(mine.x, mine.y);
                        s_tile[wv][LDS_TILE ? 1 : 0][slot] = make_float2(mine.z, mine.w);
                        if (NEED_B) s_tile[wv][LDS_TILE ? 2 : 0][slot] = mineb;
                    }
                }
                if (D4) {
                    double* t = s_tile_d[wv][lane];
                    const double ir = 1.f / mined.w;                  // functors/trace.cuh:181
                    t[0] = mined.x; t[1] = mined.y; t[2] = mined.z;
                    t[3] = mined.w * mined.w;                         // generic/intersect.h:37
                    t[4] = ir; t[5] = ir * ir;
                }
                const unsigned long long todo = rest;
                STAMP_ADD(st_cull, st_t2);
                const unsigned long long st_t3 = STAMP_NOW(); (void)st_t3;
#ifdef GRACE_STAMPS
                st_rounds += 1; st_nsurv += __builtin_popcountll(todo);
#endif
                // One survivor: the packet's 64 rays against candidate jj (wave-uniform primitive
                // index; 0 for the compacted tiles, which do not need it).
                auto process = [&](auto lean_tag, const float4 s, const float2 sb, const int jj) {
                    constexpr bool LEAN = decltype(lean_tag)::value;
                    if constexpr (D4) {
                        // sphere_hit<double4, double> (generic/intersect.h:16-54: ray members are
                        // float, everything else double) and OnHit_sphere_cumulate / _individual with
                        // Real = double (functors/trace.cuh:164-186, 196-235: ir = 1.f / w,
                        // b = (N - 1) (sqrt(b2) ir), lerp<double> with the device branch's fma,
                        // integral *= ir ir), on the caller's double4 record (wave-uniform load).
                        // (the record staged by the candidate's lane: wave-uniform LDS reads)
                        const double* sp = s_tile_d[wv][jj - pbase];
                        const double sx = sp[0], sy = sp[1], sz = sp[2], sw2 = sp[3];
                        double dot_p, b2;
                        if (AX >= 0) {
                            // Axis-aligned packet (d = +-e_AX exactly): sphere_hit collapses under
                            // IEEE rules exactly as in float -- the two perpendicular products
                            // with 0 vanish, dot = (s_a - o_a) d_a, b = p - dot d leaves the two
                            // perpendicular components untouched and cancels the third to 0.
                            const double sa = AX == 0 ? sx : AX == 1 ? sy : sz;
                            const double s1 = AX == 0 ? sy : sx;
                            const double s2 = AX == 2 ? sy : sz;
                            const double q1 = s1 - double(o1), q2 = s2 - double(o2);
                            dot_p = (sa - double(oa)) * double(da);
                            b2 = q1 * q1 + q2 * q2;
                        } else {
                            const double px = sx - ox, py = sy - oy, pz = sz - oz;
                            dot_p = px * rdx + py * rdy + pz * rdz;
                            const double bx = px - dot_p * rdx, by = py - dot_p * rdy, bz = pz - dot_p * rdz;
                            b2 = bx * bx + by * by + bz * bz;
                        }
                        const bool hit = !(b2 >= sw2) && !(dot_p < 0.0f) && !(dot_p >= len);
                        if (MODE == MODE_COUNT_D4) {
                            count += hit ? 1 : 0;
                        } else if (hit) {
                            const double ir = sp[4];
                            double x = (N_TABLE - 1) * (sqrt(b2) * ir);
                            int x_idx = static_cast<int>(x);
                            if (x_idx >= N_TABLE - 1) { x = double(N_TABLE - 1); x_idx = N_TABLE - 2; }
                            const double2 y = s_lut[x_idx];
                            double integral = __builtin_fma(x - x_idx, y.y, y.x);
                            integral *= sp[5];
                            if (MODE == MODE_CUM_D4) {
                                sum_d += integral;
                            } else if (valid) {
                                a.hit_idx[write_at] = jj;
                                a.hit_integral_d[write_at] = integral;
                                a.hit_dist_d[write_at] = dot_p;
                                ++write_at;
                            }
                        }
                    } else if (MODE == MODE_TRI) {
                        // RayIntersect_tri + OnHit_tri (tris_trace.cuh:24-61)
                        float t;
                        if (tri_intersect(ddx, ddy, ddz, ox, oy, oz, a.T64 + 9 * size_t(jj), &t)) {
                            if (t <= tri_tmin && t >= 1E-14f) {
                                tri_tmin = t;
                                tri_data = jj;
                            }
                        }
                    } else {
                        float b2, dot_p;
                        if (AX >= 0) {
                            // sphere_hit collapsed for d = +-e_AX (see the file header)
                            const float sa = AX == 0 ? s.x : AX == 1 ? s.y : s.z;
                            const float s1 = AX == 0 ? s.y : s.x;
                            const float s2 = AX == 2 ? s.y : s.z;
                            const float q1 = s1 - o1, q2 = s2 - o2;
                            dot_p = LEAN ? 0.0f : __builtin_fmaf(sa, da, noda);
                            // (fast integral: fused -- the same value a general packet computes
                            // for an axis-aligned ray below, so a ray's term does not depend on
                            // the kind of packet it travels in)
                            b2 = FAST ? __builtin_fmaf(q1, q1, q2 * q2) : q1 * q1 + q2 * q2;
                        } else {
                            // sphere_hit, include/grace/generic/intersect.h:16-54; s.w = h*h
                            const float px = s.x - ox, py = s.y - oy, pz = s.z - oz;
                            dot_p = px * dx + py * dy + pz * dz;
                            const float bx = px - dot_p * dx;
                            const float by = py - dot_p * dy;
                            const float bz = pz - dot_p * dz;
                            b2 = FAST ? __builtin_fmaf(bx, bx, __builtin_fmaf(by, by, bz * bz))
                                      : bx * bx + by * by + bz * bz;
                        }
                        if constexpr (FAST && LEAN) {
                            // No hit test at all: a candidate the ray misses has b2 >= h^2, hence a
                            // table position >= 50 (the product of the two rounded factors is monotone),
                            // which clamps to the table's last entry (y = 0, dy = 0): it adds exactly
                            // +0.  Same bits as the tested path, without the compare, the EXEC
                            // round trip and the branch.
                            const float b = fminf(__builtin_amdgcn_sqrtf(b2) * sb.x, float(N_TABLE - 1));
                            const float2 y = s_lutf[static_cast<int>(b)];
                            sum = __builtin_fmaf(__builtin_fmaf(y.y, b, y.x), sb.y, sum);
                            return;
                        }
                        const bool hit = LEAN ? !(b2 >= s.w)
                                              : !(b2 >= s.w) && !(dot_p < 0.0f) && !(dot_p >= len);
#ifdef GRACE_PACKET_STATS
                        if (MODE == MODE_STATS && __builtin_amdgcn_ballot_w64(hit) != 0ull) ++st_nodes;
#endif
                        if (MODE == MODE_COUNT || MODE == MODE_STATS) {
                            count += hit ? 1 : 0;
                        } else if (hit) {
                            const float w = FAST ? hit_integral_fast(b2, sb.x, s_lutf)
                                                 : hit_integral(b2, sb.x, sb.y, s_lut);
                            if (FAST) {
                                sum = __builtin_fmaf(w, sb.y, sum);
                            } else if (MODE == MODE_CUMULATIVE) {
                                sum += w;
                            } else if (valid && !STAGE_HITS) {
                                a.hit_idx[write_at] = jj;
                                a.hit_integral[write_at] = w;
                                a.hit_dist[write_at] = dot_p;
                                ++write_at;
                            } else if (valid) {
                                const int wvh = threadIdx.x >> 6;
                                s_hits[wvh][0][staged][lane] = __int_as_float(jj);
                                s_hits[wvh][STAGE_HITS ? 1 : 0][staged][lane] = w;
                                s_hits[wvh][STAGE_HITS ? 2 : 0][staged][lane] = dot_p;
                                ++staged;
                                ++write_at;
                            }
                        }
                        if (STAGE_HITS && __builtin_amdgcn_ballot_w64(staged == HIT_CAP) != 0ull)
                            drain_hits();
                    }
                };
                // Fetch a survivor from the wave's LDS tile: slot k of the compacted tile, or the
                // lowest set bit of `td` (which always carries bit 63 as a sentinel) otherwise.
                // Issued UNCONDITIONALLY, up to two past the last survivor: lgkmcnt counts in
                // order, so a fetch on only one of two merging paths makes the compiler wait for
                // everything outstanding -- the just-issued reads included -- before each test.
                auto fetch = [&](unsigned long long& td, int& k, float4& c, float2& cb, int& jj) {
                    int at;
                    if (COMPACT) {
                        at = k++;
                        jj = 0;
                    } else {
                        at = __builtin_ctzll(td);
                        td = (td & ~(1ull << at)) | 0x8000000000000000ull;
                        jj = min(pbase + at, r_hi - 1);
                    }
                    if (LDS_TILE) {
                        const float2 xy = s_tile[wv][0][at];
                        const float2 zw = s_tile[wv][LDS_TILE ? 1 : 0][at];
                        c = make_float4(xy.x, xy.y, zw.x, zw.y);
                        if (NEED_B) cb = s_tile[wv][LDS_TILE ? 2 : 0][at];
                        // Keep the reads here -- ahead of the survivors in between -- instead of
                        // letting the scheduler sink them next to their use.
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (!D4) {
                        c = a.A[jj];
                    }
                };
                auto run = [&](auto lean_tag) {
                    // Two survivors ahead, rotating through three register sets: each is loaded
                    // while the other two are being processed; no copies between survivors.
                    float4 c0, c1, c2;
                    float2 b0 = make_float2(0.f, 0.f), b1 = b0, b2 = b0;
                    int j0, j1 = 0, j2 = 0;
                    int left = __builtin_popcountll(todo);
                    unsigned long long td = todo | 0x8000000000000000ull;
                    int k = 0;
                    fetch(td, k, c0, b0, j0);
                    fetch(td, k, c1, b1, j1);
                    for (;;) {
                        fetch(td, k, c2, b2, j2);
                        process(lean_tag, c0, b0, j0);
                        if (--left == 0) break;
                        fetch(td, k, c0, b0, j0);
                        process(lean_tag, c1, b1, j1);
                        if (--left == 0) break;
                        fetch(td, k, c1, b1, j1);
                        process(lean_tag, c2, b2, j2);
                        if (--left == 0) break;
                    }
                };
                // Test-free rounds of axis-aligned packets with the fast integral (every round of an
                // orthographic projection through the box): survivors as single 16-byte records.
                // A wave-uniform ds_read_b128 occupies the LDS for 5.5 cycles where the ds_read2_b64
                // the compiler forms from two 8-byte plane reads takes 9 (measured, scratch
                // micro-benchmark; MI355X_MICROARCH.md, LDS) -- and the LDS pipe, shared by the CU's
                // four SIMDs, is what these rounds saturate first.  10 VALU instructions per
                // survivor: sub sub mul fma | sqrt mul min cvt shl | LDS | fma fma.
                auto run_lean4 = [&](auto fat_tag) {
                    // FAT: every survivor's h is at least a third of the packet's origin diagonal, so
                    // no ray's table position can pass the end of the zero-padded table (some ray
                    // hits the survivor: any other is within h + diagonal of its centre) -- the
                    // clamp goes: 10 VALU instructions per survivor.
                    // (Tried and measured no faster: finishing a survivor one step later, behind the
                    // next one's table read -- the loop is bound by VALU issue, ~2.7 cycles per
                    // instruction with the chip's clocks under this load, not by latency.)
                    constexpr bool FAT = decltype(fat_tag)::value;
                    float4 c0, c1, c2;
                    int left = __builtin_popcountll(todo);
                    const float4* t4 = tile4;
                    auto proc = [&](const float4 c) {
                        const float q1 = c.x - o1, q2 = c.y - o2;
                        float b = __builtin_amdgcn_sqrtf(__builtin_fmaf(q1, q1, q2 * q2)) * c.z;
                        if (!FAT) b = fminf(b, float(N_TABLE - 1));
                        const float2 y = s_lutf[static_cast<int>(b)];
                        sum = __builtin_fmaf(__builtin_fmaf(y.y, b, y.x), c.w, sum);
                    };
                    c0 = t4[0]; __builtin_amdgcn_sched_barrier(0);
                    c1 = t4[1]; __builtin_amdgcn_sched_barrier(0);
                    for (;;) {
                        c2 = t4[2]; __builtin_amdgcn_sched_barrier(0);
                        proc(c0);
                        if (--left == 0) break;
                        c0 = t4[3]; __builtin_amdgcn_sched_barrier(0);
                        proc(c1);
                        if (--left == 0) break;
                        c1 = t4[4]; __builtin_amdgcn_sched_barrier(0);
                        proc(c2);
                        if (--left == 0) break;
                        t4 += 3;
                    }
                };
                if constexpr (LEAN4) {
                    if (lean_round) {
                        if (fat_round) run_lean4(std::true_type());
                        else run_lean4(std::false_type());
                    } else run(std::false_type());
                } else {
                    if (lean_round) run(std::true_type());
                    else run(std::false_type());
                }
                STAMP_ADD(st_surv, st_t3);
                } // !skip_round
                    if (!more) break;
                } // rounds over the surviving clusters
            }
            };
            switch (axis) {
            case 0: sweep_range(std::integral_constant<int, 0>()); break;
            case 1: sweep_range(std::integral_constant<int, 1>()); break;
            case 2: sweep_range(std::integral_constant<int, 2>()); break;
            default:
                if (is_pencil) sweep_range(std::integral_constant<int, -2>());
                else sweep_range(std::integral_constant<int, -1>());
                break;
            }
            // Keep the warming load alive (child / primitive indices are never negative).
            junk |= warm;
        }
    }

#ifdef GRACE_STAMPS
    if (lane == 0) {
        atomicAdd(&g_stamp_acc[0], __builtin_amdgcn_s_memtime() - st_begin);
        atomicAdd(&g_stamp_acc[1], st_walk); atomicAdd(&g_stamp_acc[2], st_cluster);
        atomicAdd(&g_stamp_acc[3], st_cull); atomicAdd(&g_stamp_acc[4], st_surv);
        atomicAdd(&g_stamp_acc[5], st_rounds); atomicAdd(&g_stamp_acc[6], st_nsurv);
        atomicAdd(&g_stamp_acc[7], 1ull);
        {
            const unsigned slot = atomicAdd(&g_stamp_n, 1u) & 0xffffu;
            g_stamp_log[slot][0] = st_begin; g_stamp_log[slot][1] = __builtin_amdgcn_s_memtime();
            g_stamp_log[slot][2] = st_nsurv; g_stamp_log[slot][3] = st_walk;
        }
    }
#endif
    if (STAGE_HITS) drain_hits();
    if (CHUNKED && MODE == MODE_COUNT && a.chunk_counts) leave_chunk();
    if ((overflow || junk < 0) && lane == 0) *a.status = GRACE_STACK_OVERFLOW;
    if (!valid) return;
    if (MODE == MODE_COUNT) {
        if (!SPLIT) a.out_counts[ray_index] = count;
        else if (count) atomicAdd(&a.out_counts[ray_index], count); // output zeroed by the host
    }
    if (MODE == MODE_TRI) a.out_counts[ray_index] = tri_data;
    if (MODE == MODE_COUNT_D4) {
        if (!SPLIT) a.out_counts[ray_index] = count;
        else if (count) atomicAdd(&a.out_counts[ray_index], count); // output zeroed by the host
    }
    if (MODE == MODE_CUM_D4) {
        // the float path's class-ordered sum, in double: pairwise over the 8 classes (this wave's
        // subtree of it when the packet is split)
        if (cur_granule >= 0) s_class_d[wv_acc][cur_granule & (SUM_CLASSES - 1)][lane] = sum_d;
        const int c_lo = SPLIT ? own_lo : 0, c_n = SPLIT ? classes_per_part : SUM_CLASSES;
        for (int w = 1; w < c_n; w *= 2)
            for (int c = c_lo; c < c_lo + c_n; c += 2 * w) {
                const double x = s_class_d[wv_acc][c][lane], y = s_class_d[wv_acc][c + w][lane];
                s_class_d[wv_acc][c][lane] = x + y;
            }
        if (!SPLIT) a.out_sums_d[ray_index] = s_class_d[wv_acc][0][lane];
        else a.partial_d[size_t(ray_index) * split + part] = s_class_d[wv_acc][c_lo][lane];
    }
    if (MODE == MODE_CUMULATIVE) {
        if (cur_granule >= 0) s_class[wv_acc][cur_granule & (SUM_CLASSES - 1)][lane] = sum;
        // Pairwise sum of this wave's classes (a subtree of the summation tree).
        float t[SUM_CLASSES];
#pragma unroll
        for (int c = 0; c < SUM_CLASSES; ++c) t[c] = s_class[wv_acc][c][lane];
        float result = 0.f;
        if (!SPLIT) {
#pragma unroll
            for (int w = 1; w < SUM_CLASSES; w *= 2)
#pragma unroll
                for (int c = 0; c < SUM_CLASSES; c += 2 * w) t[c] = t[c] + t[c + w];
            result = t[0];
            a.out_sums[ray_index] = result;
        } else {
            // classes own_lo .. own_hi-1: reduce with the same pairing, then publish
            for (int w = 1; w < classes_per_part; w *= 2)
                for (int c = own_lo; c < own_hi; c += 2 * w) {
                    // t[] is indexed with wave-uniform runtime indices only here (rare path)
                    const float x = s_class[wv_acc][c][lane], y = s_class[wv_acc][c + w][lane];
                    s_class[wv_acc][c][lane] = x + y;
                }
            a.partial[size_t(ray_index) * split + part] = s_class[wv_acc][own_lo][lane];
        }
    }
    if (MODE == MODE_STATS) {
        reinterpret_cast<uint4*>(a.stats)[ray_index] =
            make_uint4(st_nodes, st_leaves, st_tested, uint32_t(count));
    }
}

} // namespace
