// BVH traversal with ray-sphere tests and SPH-kernel line integrals, for gfx950.
//
// Replaces trace_kernel (reference include/grace/cuda/kernels/bintree_trace.cuh:52-197) as
// instantiated by trace_hitcounts_sph / trace_cumulative_sph / trace_sph
// (include/grace/cuda/trace_sph.cuh:58-168) with the functors of
// include/grace/cuda/functors/trace.cuh, AABBs_hit (include/grace/cuda/device/intersect.cuh:
// 10-40) and sphere_hit (include/grace/generic/intersect.h:10-55).
//
// Same traversal semantics: a packet of rays shares one stack; a child is pushed if ANY
// ray of the packet hits its box (right first, so the left subtree is walked first and
// every ray meets its hits in ascending primitive index); every ray of the packet is
// tested against every sphere of every leaf the packet enters.  Results per ray therefore
// equal the brute-force loop over all spheres (the reference's own criterion,
// tests/tree_traversal/tree_traversal.cu:65-100).
//
// CDNA4 design (not the reference's; measurements in DESIGN.md section 4/6):
//   * packet = one 64-lane wavefront (the reference: a 32-thread warp); for the per-hit and
//     triangle walks of small batches, 32 or 16 rays per wave;
//   * the packet's stack lives in TWO VGPRs indexed by lane (pop = v_readlane, push =
//     lane-select, with a scalar stack pointer): 128 entries, no LDS traffic;
//   * node records are wave-uniform: child indices / primitive spans are consumed as scalars,
//     the per-ray slab test keeps the reference's arithmetic (~100 node tests per packet);
//   * no FMA contraction (-ffp-contract=off), IEEE 1/x and sqrt in everything that decides a
//     hit or feeds a bit-exact output: the reference's CPU/GPU equality test is built with
//     -fmad=false (tests/tree_traversal/Makefile:5-8).  Explicit FMAs appear only where they
//     provably return the unfused bits (axis dot product), in culls, and in the fast integral;
//   * a streaming pre-pass hoists per-sphere work out of the (ray x sphere) loop:
//     A[i] = {x, y, z, h*h}, B[i] = {1/h or 50/h, (1/h)^2};
//   * ray coherence order: 64 consecutive rays of a space-filling order of the ray
//     co-ordinates that vary form a packet (15-bit round-to-nearest keys; a Hilbert curve for
//     two varying co-ordinates and, over the octahedral map of the direction, for batches from
//     one origin -- its runs never join distant patches; Z-order for power-of-two pixel grids,
//     whose tiles are the same and whose tile order suits the dispatcher better, and for
//     general batches); results do not depend on it; a batch traced repeatedly can have its
//     order prepared once (grace_trace_prepare_rays);
//   * treelet sweep with a cluster level: a node whose subtree holds <= T primitives (contiguous
//     indices; T = 16384 for axis-aligned packets, 8192 otherwise, 512 for triangles) is not
//     descended: one box per 64 consecutive primitives (a pre-pass) is tested first, lane j
//     for cluster j, and only the surviving clusters go through culling rounds of 64;
//   * scenes with spheres smaller than the ray spacing (dense cores of clustered SPH data) run
//     a separate instantiation (LAT), selected by a device flag: an exact cull against the
//     packet's origin lattice, and four waves per packet for big batches (their packets are
//     very unequal);
//   * beam culling per round, lane j deciding for candidate j whether ANY ray of the packet
//     can hit it: exact lower bound of b^2 for axis-aligned packets (rounding is monotone),
//     cone + four side planes for packets from one origin, interval arithmetic otherwise;
//   * survivors of a round are written compacted (slot = v_mbcnt) to the wave's LDS tile as
//     three 8-byte planes and read back two survivors ahead into three rotating register sets
//     (unconditional reads pinned by sched_barrier; immediates address the slots);
//   * axis-aligned packets (orthographic projections): sphere_hit collapses exactly under IEEE
//     rules to dot = fma(s_k, d_k, -o_k d_k), b2 = q1^2 + q2^2; rounds whose candidates are
//     provably inside every ray's [0, length) skip the range tests (6 instructions per test);
//   * kernel integral of the column-density trace: hardware sqrt + fp32 lerp of an fp32
//     (y, dy) table in LDS (7 instructions, tolerance parity) by default; the reference's
//     arithmetic bit for bit (correctly rounded sqrt, fp64 lerp) on request and always for
//     the per-hit outputs;
//   * class-ordered sums and packet splitting: primitives are dealt to 8 CLASSES in granules
//     of 1024 consecutive indices (class = (index >> 10) & 7).  A ray's column density is
//     defined as the balanced pairwise (binary-tree) fp32 sum of its 8 class sums, each class
//     summed in ascending primitive order.  The value is a function of the ray and the scene
//     only -- not of how rays are batched, ordered or sharded -- and differs from the
//     reference's single running sum only in the last bits (both within 1e-6 of the exact
//     sum; tolerance 1e-5).  What it buys: a packet can be walked by K = 1, 2, 4 or 8
//     waves, wave k owning the 8/K classes [k 8/K, (k+1) 8/K) -- a subtree of the summation
//     tree, and, because classes interleave along the Morton order, an even share of the
//     work for any beam -- with NO change of the result: each wave reduces its classes, a
//     tiny kernel finishes the tree.  K is the smallest power of two that puts >= 16384 waves
//     in flight; for batches of one direction (light packets) a device-side choice lowers it
//     (choose_split).  A wave keeps its class accumulators in LDS (2 KiB) and switches at
//     granule boundaries, once per culling round at most.  Hit counts split the same way
//     (integers);
//   * per-hit outputs (ordered per ray): large batches stage hits per lane in LDS and drain
//     them eight entries per ray; small batches split a packet over K waves by contiguous
//     chunk ranges with per-(ray, chunk) output offsets from a counting walk (hits_plan_kernel);
//   * packets are dealt to workgroups so that the workgroups sharing an XCD (blockIdx % 8)
//     walk a contiguous range of packets: neighbouring packets touch the same subtree and
//     each XCD's 4 MiB L2 keeps it.
//
// Roofline: divergent tree walk, integer/fp32 scalar-operand work -- no MFMA.  Algorithmic
// bytes per ray: 28 + 64 * nodes + 16 * leaves + 16 * spheres tested + 4,
// counted per ray by the `stats` instantiation below.
#include "common.hpp"

#ifdef GRACE_STAMPS
#include <algorithm>
#include <vector>
#endif
#include <cstdlib>
#include <type_traits>

using namespace grace_hip;

// Diagnostic build (-DGRACE_STAMPS, never the product): s_memtime stamps around the phases of a
// wave's life, accumulated per wave and summarised on the host after every launch.
#ifdef GRACE_STAMPS
#define STAMP_NOW() __builtin_amdgcn_s_memtime()
#define STAMP_ADD(acc, t0) do { acc += __builtin_amdgcn_s_memtime() - (t0); } while (0)
__device__ unsigned long long g_stamp_acc[8];
__device__ unsigned long long g_stamp_log[1 << 16][4];
__device__ unsigned int g_stamp_n;
#else
#define STAMP_NOW() 0ull
#define STAMP_ADD(acc, t0) do { } while (0)
#endif

namespace {

constexpr int TRACE_BLOCK = 256;
constexpr int N_TABLE = 51;
constexpr int MAX_HIT_CHUNKS = 256; // chunk ranges of the split per-hit trace
constexpr int SUM_CLASSES = 8;   // summation classes (leaves of the pairwise sum tree)
constexpr int GRANULE_SHIFT = 10; // 1024 consecutive primitives share a class

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

enum { MODE_COUNT = 0, MODE_CUMULATIVE = 1, MODE_HITS = 2, MODE_STATS = 3, MODE_TRI = 4,
       // Real4 = double4, Real = double (trace_sph.cuh:57-241 instantiated in double): the walk and
       // every cull run on float records that CONTAIN the double spheres; each survivor is then
       // tested and integrated in double against the caller's double4 record.
       MODE_COUNT_D4 = 5, MODE_CUM_D4 = 6, MODE_HITS_D4 = 7 };

struct TraceArgs {
    const float* rays;      // 7 floats per ray
    const uint32_t* perm;   // packet slot -> ray index (coherence order), or null
    int n_rays;
    const float4* spheres;
    const int2* node_prims; // pre-pass: per node {first primitive, primitive count}
    int treelet;            // nodes with <= treelet primitives are swept as one leaf (0: off)
    int treelet_axis;       // the same for axis-aligned packets (whose cluster test is much sharper)
    const float4* A;        // pre-pass: {x, y, z, h*h}, padded by 4 entries
    const float2* B;        // pre-pass: {1/h, (1/h)^2}, padded by 4 entries
    const float4* C;        // pre-pass: per CLUSTER (64 consecutive primitives) {lo.xyz, -}, {hi.xyz, -}:
                            // the box of the member spheres, slightly inflated (cluster_boxes_kernel)
    const double* T64;      // MODE_TRI pre-pass: {v, e1, e2} widened to fp64, 9 per triangle
    const double* spheres_d; // *_D4 modes: the caller's double4 spheres
    double* out_sums_d;      // MODE_CUM_D4
    double* hit_integral_d;  // MODE_HITS_D4
    double* hit_dist_d;
    int split;              // waves per packet (1, 2, 4, 8); each owns SUM_CLASSES / split classes
    int n_prims;
    float* partial;         // split > 1, cumulative: [n_rays][split] subtree sums
    // Class split only: the number of waves per packet that actually work (a power of two <=
    // split, chosen on the device from the batch's coherence); waves beyond it exit at once.
    const int* split_dev;
    const int* lat_dev;     // which of the LAT = false / true instantiations runs (null: false)
    // Split per-hit trace (small batches): primitives are cut into n_chunks ranges of
    // 2^chunk_shift consecutive indices.  The counting pass fills chunk_counts[ray][chunk];
    // the per-hit pass lets wave w own chunks [wave_map[w].y, wave_map[w].z) of packet
    // wave_map[w].x -- heavier packets get more waves (hits_assign_kernel) -- and writes a ray's
    // hits of a chunk from chunk_off[ray][chunk] on.
    int* chunk_counts;
    const int* chunk_off;
    const int4* wave_map;
    const int* n_wave_map;
    bool keep_chunks;       // host only: a hit-count trace whose chunk counts the per-hit trace will reuse
    int chunk_shift, n_chunks;
    int width;              // rays per packet: 64, or 32 / 16 for small batches of the modes that
                            // cannot split a packet (lanes >= width re-trace the packet's last ray)
    const float4* nodes;    // 4 x float4 per node
    int n_nodes;
    const int4* leaves;
    const int* root;
    int* out_counts;        // MODE_COUNT
    float* out_sums;        // MODE_CUMULATIVE
    const int* offsets;     // MODE_HITS
    int* hit_idx;
    float* hit_integral;
    float* hit_dist;
    uint32_t* stats;        // MODE_STATS, 4 per ray
    int* status;            // set to GRACE_STACK_OVERFLOW on stack exhaustion
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
// b = sqrt(b2) * (50/h) with 50/h from the pre-pass, weight v_fract_f32(b), fp32 FMA on an
// fp32 (y_i, y_{i+1} - y_i) table rounded from the fp64 one.  lutf has N_TABLE + 1 entries,
// the last two being (y_50, 0), so b == 50 (sqrt(b2)/h rounded up to 1) needs no clamp.
// Eight VALU instructions instead of twenty-five; each term within ~3 ulp of the exact one.
// Returns the table value; the caller applies 1/h^2 inside its accumulating FMA.
__device__ __forceinline__ float hit_integral_fast(const float b2, const float ir50,
                                                   const float2* lutf)
{
    const float b = __builtin_amdgcn_sqrtf(b2) * ir50;
    const int x_idx = static_cast<int>(b);
    const float t = __builtin_amdgcn_fractf(b);
    const float2 y = lutf[x_idx];
    return __builtin_fmaf(t, y.y, y.x);
}

// One wave, one cluster: the box of the 64 records a(i), i = 64 c + lane (lanes with i >= n hold
// nothing); lane 0 writes the cluster record and returns the cluster's smallest r^2 in every lane.
__device__ __forceinline__ float cluster_box_of_wave(const float4 s, const bool have, const size_t c,
                                                     float4* __restrict__ C)
{
    const int lane = threadIdx.x & 63;
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    float r2_min = INFINITY;
    if (have) {
        r2_min = s.w;
        const float r = sqrtf(s.w) * 1.00001f;   // sqrt(fl(h h)) can round below h
        const float ctr[3] = { s.x, s.y, s.z };
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float slack = (fabsf(ctr[k]) + r) * 4.76837158203125e-07f; // 2^-21
            lo[k] = (ctr[k] - r) - slack;
            hi[k] = (ctr[k] + r) + slack;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) r2_min = fminf(r2_min, __shfl_xor(r2_min, off));
    if (lane == 0) {
        // .w of the low corner: the smallest r^2 of the members (the origin-lattice cull of
        // axis-aligned packets looks only at clusters that hold spheres smaller than the
        // packet's ray spacing)
        C[2 * c] = make_float4(lo[0], lo[1], lo[2], r2_min);
        C[2 * c + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
    return r2_min;
}

// The scene's smallest r^2 (positive floats order like their bit patterns): one float4 past the
// last cluster record, pre-set to a huge value by scene_fill; one atomic per workgroup (one per
// cluster serialised 156 k atomics on one address at 10^7 primitives: +1.8 ms per unprepared call).
// Every thread of the workgroup must call it.
__device__ __forceinline__ void publish_r2_min(const float wave_r2_min, float4* __restrict__ C_tail)
{
    __shared__ float s_r2_min[4];
    if ((threadIdx.x & 63) == 0) s_r2_min[(threadIdx.x >> 6) & 3] = wave_r2_min;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = s_r2_min[0];
        for (unsigned w = 1; w < blockDim.x / 64 && w < 4; ++w) m = fminf(m, s_r2_min[w]);
        if (m < INFINITY) atomicMin(reinterpret_cast<unsigned int*>(C_tail), __float_as_uint(m));
    }
}

// C != null: the cluster boxes in the same pass (a wave's 64 consecutive records ARE a cluster:
// the loop stride is a multiple of the workgroup size) -- one launch and one read of A fewer.
__global__ __launch_bounds__(256) void trace_prepass_kernel(const float4* __restrict__ spheres,
                                                            size_t n, float4* __restrict__ A,
                                                            float2* __restrict__ B,
                                                            const float b_scale,
                                                            float4* __restrict__ C = nullptr)
{
    float wave_r2_min = INFINITY;
    // (whole waves enter every iteration: the bound is rounded up to the wave's first record)
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; (i & ~size_t(63)) < n + 4;
         i += size_t(gridDim.x) * blockDim.x) {

        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 b = make_float2(0.f, 0.f);
        if (i < n) {
            const float4 s = spheres[i];
            a = make_float4(s.x, s.y, s.z, s.w * s.w); // sphere.w * sphere.w, intersect.h:37
            if (B) {
                const float ir = 1.f / s.w;            // functors/trace.cuh:181
                b = make_float2(ir * b_scale, ir * ir); // functors/trace.cuh:184 (b_scale 1, or 50: fast)
                // The fast integral's branch-free rounds add (table value 0) * (1/h^2) for a
                // candidate the ray misses: keep that product 0 when 1/h^2 overflows (h < 5e-20).
                if (b_scale != 1.0f) b.y = fminf(b.y, 3.4028234664e38f);
            }
        }
        if (i < n + 4) {
            A[i] = a;
            if (B) B[i] = b;
        }
        // (wave-uniform condition: every lane of the wave takes part in the shuffles)
        if (C && (i & ~size_t(63)) < n) wave_r2_min = fminf(wave_r2_min, cluster_box_of_wave(a, i < n, i >> 6, C));
    }
    if (C) publish_r2_min(wave_r2_min, C + 2 * ((n + 63) / 64));
}

// double4 spheres: the float record {x, y, z, r^2} that drives the walk's culls must CONTAIN the
// double sphere -- the centre is narrowed (error <= half a float ulp per co-ordinate) and the
// double hit test is close to exact, so the radius is inflated by 2^-18 relative plus 2^-21 of
// the co-ordinate magnitudes before squaring (the float culls' own margins then cover their own
// rounding as for float spheres).
__global__ __launch_bounds__(256) void trace_prepass_d4_kernel(const double* __restrict__ spheres,
                                                               size_t n, float4* __restrict__ A)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n + 4;
         i += size_t(gridDim.x) * blockDim.x) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) {
            const double* s = spheres + 4 * i;
            const double r = fabs(s[3]) * (1.0 + 3.814697265625e-06)
                + (fabs(s[0]) + fabs(s[1]) + fabs(s[2]) + fabs(s[3])) * 4.76837158203125e-07;
            a = make_float4(float(s[0]), float(s[1]), float(s[2]), float(r * r * (1.0 + 1e-6)));
        }
        A[i] = a;
    }
}

// ---- cluster boxes ---------------------------------------------------------------------------
// Primitives are Morton-sorted, so 64 consecutive ones (a CLUSTER: indices [64 c, 64 c + 64)) are
// a compact clump about as wide as a smoothing length.  One box per cluster -- the union of
// the member spheres' boxes [c - h, c + h] -- lets a sweep drop 64 candidates with one lane's
// test instead of 64 lanes' tests: a swept subtree first tests its clusters (lane j <-> cluster
// j), then runs culling rounds only over the clusters that survive.  Never a result: a cluster
// is dropped only if no ray of the packet can hit any member, so per-ray hit sets are unchanged.
// The half-width is inflated by 4 ulp of the co-ordinate magnitude: sphere_hit's own rounding
// (q = fl(s - o), b2 = fl(fl(q1^2) + fl(q2^2)) < h^2 admits |s - o| up to h (1 + 3 u) + u |s|).
// A = {x, y, z, r^2} as written by the pre-passes (spheres: r = h; triangles: bounding radius).
__global__ __launch_bounds__(256) void cluster_boxes_kernel(const float4* __restrict__ A, size_t n,
                                                            float4* __restrict__ C)
{
    const size_t n_clusters = (n + 63) / 64;
    const int lane = threadIdx.x & 63;
    float wave_r2_min = INFINITY;   // over the clusters this wave handles
    for (size_t c = blockIdx.x * size_t(blockDim.x / 64) + (threadIdx.x >> 6); c < n_clusters;
         c += size_t(gridDim.x) * (blockDim.x / 64)) {
        const size_t i = c * 64 + lane;
        const bool have = i < n;
        const float4 s = have ? A[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        wave_r2_min = fminf(wave_r2_min, cluster_box_of_wave(s, have, c, C));
    }
    publish_r2_min(wave_r2_min, C + 2 * n_clusters);
}

// ---- triangle primitives (tests/profile_trace_triangle) -----------------------------------
// Pre-pass: a bounding sphere per triangle (for the beam culling: a ray that meets the
// triangle passes within r of the centre; r^2 is inflated by 2^-10 against fp32 rounding) and
// the triangle widened to fp64 (the reference's dot/cross products are fp64 products of
// float operands, tests/helper/vector_math.cu:27-52; widening once is exact).
__global__ __launch_bounds__(256) void tri_prepass_kernel(const float* __restrict__ tris, size_t n,
                                                          float4* __restrict__ A,
                                                          double* __restrict__ T64)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n + 4;
         i += size_t(gridDim.x) * blockDim.x) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) {
            const float* t = tris + 9 * i;
            float c[3], r2 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) c[k] = t[k] + (t[3 + k] + t[6 + k]) * (1.0f / 3.0f);
#pragma unroll
            for (int vtx = 0; vtx < 3; ++vtx) {
                float d2 = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float p = (vtx == 0 ? t[k] : vtx == 1 ? t[k] + t[3 + k] : t[k] + t[6 + k]) - c[k];
                    d2 += p * p;
                }
                r2 = fmaxf(r2, d2);
            }
            a = make_float4(c[0], c[1], c[2], r2 * 1.0009765625f + 1e-37f);
#pragma unroll
            for (int k = 0; k < 9; ++k) T64[9 * i + k] = double(t[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) T64[9 * i + k] = 0.0;
        }
        A[i] = a;
    }
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

// Primitive range of every node: a node's leaves are consecutive (nodes.h:27-28) and so are
// their primitives, [leaves[first].x, leaves[last].x + leaves[last].y).
__global__ __launch_bounds__(256) void node_prims_kernel(const int4* __restrict__ nodes4,
                                                         const int4* __restrict__ leaves, int n_nodes,
                                                         int2* __restrict__ out,
                                                         uint32_t* __restrict__ r2_min_slot)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // (the slot the cluster-box pass reduces the scene's smallest r^2 into starts out huge: this
    // launch precedes that pass on the stream)
    if (i == 0 && r2_min_slot) *r2_min_slot = 0x7f7f7f7fu;
    if (i >= n_nodes) return;
    const int4 n0 = nodes4[4 * size_t(i)];
    const int4 lf = leaves[n0.z], ll = leaves[n0.w];
    out[i] = make_int2(lf.x, ll.x + ll.y - lf.x);
}

// ---- ray coherence order ---------------------------------------------------------------
// Packets are 64 consecutive rays of an ORDER chosen here, not of the caller's array: the
// per-ray results do not depend on which rays share a packet (each equals the brute-force
// loop), but the number of boxes and spheres a packet touches does.  Rays are keyed by a
// Morton code over those of their six coordinates (origin, direction) that actually vary,
// quantised over their extents, and sorted (stable radix sort, sort.hip).  The reference
// leaves this to the caller (its generators sort by direction or end point,
// include/grace/cuda/kernels/gen_rays.cuh:483,520,577,615).
__device__ __forceinline__ uint32_t f2ord_u(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float ord2f_u(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void ray_ext_init_kernel(uint32_t* __restrict__ ext16)
{
    if (threadIdx.x < 16) ext16[threadIdx.x] = threadIdx.x < 6 ? 0xFFFFFFFFu : 0u;
}

__global__ __launch_bounds__(256) void ray_extents_kernel(const float* __restrict__ rays, int n,
                                                          uint32_t* __restrict__ ext12)
{
    float lo[6], hi[6];
    float len_hi = -INFINITY;   // the longest ray (slot 15: choose_lattice's scale for one-origin batches)
#pragma unroll
    for (int k = 0; k < 6; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* r = rays + 7 * size_t(i);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float v = r[k];
            lo[k] = fminf(lo[k], v);
            hi[k] = fmaxf(hi[k], v);
        }
        len_hi = fmaxf(len_hi, r[6]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) len_hi = fmaxf(len_hi, __shfl_xor(len_hi, off));
    if ((threadIdx.x & 63) == 0 && len_hi > -INFINITY) atomicMax(&ext12[15], f2ord_u(len_hi));
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    __shared__ float s_lo[4][6], s_hi[4][6];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { s_lo[wave][k] = lo[k]; s_hi[wave][k] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float l = s_lo[0][k], h = s_hi[0][k];
        for (int w = 1; w < 4; ++w) { l = fminf(l, s_lo[w][k]); h = fmaxf(h, s_hi[w][k]); }
        atomicMin(&ext12[k], f2ord_u(l));
        atomicMax(&ext12[6 + k], f2ord_u(h));
    }
}

// How many of the launched waves per packet should work.  The host sizes the launch for an
// incoherent batch (whose packets are heavy: >= 16384 waves in flight pay off); a batch whose
// rays all share one direction (orthographic shards) has light packets, for which every extra
// wave repeats the upper-tree walk and the cluster tests.  ext12 = the ray
// extents of the coherence pass (order-preserving uints: minima then maxima of d, o).
__device__ int choose_split(const uint32_t* __restrict__ ext12, int n_packets, int launched, bool lattice)
{
    const bool one_direction = ext12[0] == ext12[6] && ext12[1] == ext12[7] && ext12[2] == ext12[8];
    int k = launched;
    // (a scene with spheres smaller than the ray spacing has packets of very unequal weight: it
    // keeps every launched wave -- see lat_split in launch_trace)
    if (one_direction && !lattice) {
        // Measured on 1/8 ... 1/1 shards of the 1024^2 frame (2048 ... 16384 packets): best K =
        // 4, 2, 2, 1.  The split kernels run 8 waves per SIMD: 8192 waves fill the chip once;
        // from 6144 packets on a second wave per packet still pays (16384 waves).
        k = 1;
        while (k < launched && n_packets * k < 8192) k *= 2;
        if (k < launched && n_packets >= 6144 && n_packets * k < 16384) k *= 2;
    }
    return k;
}

// Position of cell (x, y) of a 2^15 x 2^15 grid along the Hilbert curve (30 bits).  Unlike the
// Z-order curve it has no jumps: ANY 64 consecutive rays of the sorted order form one connected
// patch, where a Z-order run that straddles a high-level cell boundary joins two distant patches
// into one very wide packet (whose wave then outlives the rest of the launch).
__device__ __forceinline__ uint32_t hilbert2d_15(uint32_t x, uint32_t y)
{
    uint32_t d = 0;
    for (uint32_t s = 1u << 14; s > 0; s >>= 1) {
        const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d = (d << 2) | ((3u * rx) ^ ry);
        if (ry == 0) {
            if (rx) { x = 32767u - x; y = 32767u - y; }
            const uint32_t t = x; x = y; y = t;
        }
    }
    return d;
}

// Is the batch a power-of-two pixel grid?  (Two varying co-ordinates, N1 x N2 = n points with N1,
// N2 powers of two >= 8, every ray on a lattice point.)  Such a batch gets Z-order keys: the
// tiles are the Hilbert curve's, but in the order that spreads a CU's workgroups evenly over its
// XCD's block (5 % on a 1/8-image shard); every other batch gets the Hilbert
// curve, whose runs never join distant patches.  flag: 0 on entry; any thread that finds the
// batch unfit sets it.
__global__ __launch_bounds__(256) void ray_lattice_kernel(const float* __restrict__ rays, int n,
                                                          const uint32_t* __restrict__ ext12,
                                                          uint32_t* __restrict__ flag)
{
    int dims[2] = { 0, 0 }, nvar = 0;
    float lo[2] = { 0.f, 0.f }, span[2] = { 0.f, 0.f };
    for (int k = 5; k >= 0; --k) {          // (the order ray_keys_kernel takes them in)
        const float l = ord2f_u(ext12[k]), sp = ord2f_u(ext12[6 + k]) - l;
        if (sp > 0.f && sp < INFINITY) {
            if (nvar < 2) { dims[nvar] = k; lo[nvar] = l; span[nvar] = sp; }
            ++nvar;
        }
    }
    bool fit = nvar == 2 && n >= 64 && (n & (n - 1)) == 0;
    float m1 = 0.f, m2 = 0.f;                // N1 - 1, N2 - 1
    if (fit) {
        int log_n = 0;
        while ((1 << log_n) < n) ++log_n;
        fit = false;
        const float ratio = span[0] / span[1];
        for (int a = 3; a <= log_n - 3; ++a) {
            const float c1 = float((1 << a) - 1), c2 = float((1 << (log_n - a)) - 1);
            if (fabsf(c1 / c2 - ratio) <= 1e-3f * ratio) { fit = true; m1 = c1; m2 = c2; break; }
        }
    }
    if (!fit) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(flag, 1u);
        return;
    }
    const float s1 = m1 / span[0], s2 = m2 / span[1];
    bool off = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* r = rays + 7 * size_t(i);
        const float u1 = (r[dims[0]] - lo[0]) * s1, u2 = (r[dims[1]] - lo[1]) * s2;
        off = off || !(fabsf(u1 - rintf(u1)) <= 0.01f) || !(fabsf(u2 - rintf(u2)) <= 0.01f);
    }
    if (__builtin_amdgcn_ballot_w64(off) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// Device-side choices of a trace launch, made by one thread from the batch's ray extents:
// choose_lattice -- the LAT instantiation runs if all rays share one axis-aligned direction and the
// scene holds spheres smaller than the diagonal of the batch's mean ray cell -- and choose_split
// (class-split hit-count / cumulative launches): how many of the launched waves per packet work.
__device__ void choose_variants(const uint32_t* __restrict__ ext12, int n, const float4* __restrict__ scene_min,
                                uint32_t* __restrict__ lat_flag, int split_packets, int split_launched,
                                int* __restrict__ split_dev)
{
    if (lat_flag) {
        int n_dir = 0;
        bool one_dir = true;
        float e1 = 0.f, e2 = 0.f;   // the two largest origin extents
        for (int k = 0; k < 3; ++k) {
            const float dl = ord2f_u(ext12[k]), dh = ord2f_u(ext12[6 + k]);
            one_dir = one_dir && dl == dh;
            n_dir += dl != 0.f ? 1 : 0;
            const float e = ord2f_u(ext12[9 + k]) - ord2f_u(ext12[3 + k]);
            if (e > e1) { e2 = e1; e1 = e; } else if (e > e2) e2 = e;
        }
        const float spacing2 = e1 * e2 / float(n);
        const float r2_min = scene_min->x;
        bool lat = one_dir && n_dir == 1 && spacing2 > 0.f && r2_min < 2.0f * spacing2;
        // One origin (point sources, cameras): the rays' spacing at the far end of the longest ray,
        // 4 pi L^2 / n for a full sphere (an upper bound for partial ones).  There is no lattice to
        // cull against, but such scenes have the same very unequal packets: the flag sends the batch
        // to four waves per packet (launch_trace) all the same.
        if (!lat && e1 == 0.f && ext12[15] != 0u) {
            const float len = ord2f_u(ext12[15]);
            if (len > 0.f && len < INFINITY) lat = r2_min < 2.0f * (12.566371f * len * len / float(n));
        }
        *lat_flag = lat ? 1u : 0u;
    }
    if (split_dev)
        *split_dev = choose_split(ext12, split_packets, split_launched, lat_flag ? *lat_flag != 0u : false);
}

// The same choices for a call whose ray order is cached (grace_trace_prepare_rays).
__global__ void choose_variants_kernel(const uint32_t* __restrict__ ext12, int n, const float4* __restrict__ scene_min,
                                       uint32_t* __restrict__ lat_flag, int split_packets, int split_launched,
                                       int* __restrict__ split_dev)
{
    choose_variants(ext12, n, scene_min, lat_flag, split_packets, split_launched, split_dev);
}

__global__ __launch_bounds__(256) void ray_keys_kernel(const float* __restrict__ rays, int n,
                                                       const uint32_t* __restrict__ ext12,
                                                       uint32_t* __restrict__ keys,
                                                       const float4* __restrict__ scene_min,
                                                       uint32_t* __restrict__ lat_flag,
                                                       int split_packets, int split_launched,
                                                       int* __restrict__ split_dev,
                                                       const uint32_t* __restrict__ not_grid)
{
    const bool z_order_2d = not_grid && *not_grid == 0u;   // a power-of-two pixel grid (ray_lattice_kernel)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        choose_variants(ext12, n, scene_min, lat_flag, split_packets, split_launched, split_dev);
    float lo[6], scale[6], span[6];
    int nvar = 0;
    // One scale for the three direction components and one for the three origin components (the
    // largest extent of each group): cells of the curve are then cubes in ray space whatever the
    // batch's aspect ratio.  (Scaling every component by its own extent made the packets of a
    // 1024 x 128-pixel shard 23 x 3-pixel strips instead of 8 x 8 tiles: 9966 surviving
    // candidates per packet instead of 6687, measured with the stamped diagnostic build.)
    float span_d = 0.f, span_o = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        lo[k] = ord2f_u(ext12[k]);
        span[k] = ord2f_u(ext12[6 + k]) - lo[k];
        const bool varies = span[k] > 0.f && span[k] < INFINITY;
        if (varies) { if (k < 3) span_d = fmaxf(span_d, span[k]); else span_o = fmaxf(span_o, span[k]); }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const bool varies = span[k] > 0.f && span[k] < INFINITY;
        scale[k] = varies ? 1.0f / (k < 3 ? span_d : span_o) : 0.f;
        nvar += varies ? 1 : 0;
    }
    const int bits = nvar ? min(15, 30 / nvar) : 0;
    const float qmax = float((1 << bits) - 1);
    // One origin (cameras, cones, point sources): a 3-D curve over the direction components is a
    // poor order for points of a 2-D surface (and a camera looking along z has a tiny, non-linear
    // z extent).  Map the direction to the unit square with the octahedral parametrisation and
    // order THAT along a 15-bit Hilbert curve: a pinhole camera's pixel grid becomes compact
    // 64-ray patches (closest-hit trace of 10^6 triangles: 13.8 -> 7.5 ms with the 2-D order),
    // and because the Hilbert curve never jumps no packet joins two distant patches -- with
    // Z-order keys 10^5 isotropic rays had packets of up to 6.7x the mean candidate count whose
    // waves outlived the launch's mean wave 3x (stamped build: surviving candidates per wave
    // 1977 -> 1576 mean, 13258 -> 3902 max; HEALPix source 1764 -> 1300, 8117 -> 3983).
    const bool pencil = scale[3] == 0.f && scale[4] == 0.f && scale[5] == 0.f && nvar > 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* r = rays + 7 * size_t(i);
        if (pencil) {
            const float l1 = fabsf(r[0]) + fabsf(r[1]) + fabsf(r[2]);
            float u = r[0] / l1, v = r[1] / l1;
            if (r[2] < 0.f) {
                const float fu = (1.f - fabsf(v)) * (u >= 0.f ? 1.f : -1.f);
                const float fv = (1.f - fabsf(u)) * (v >= 0.f ? 1.f : -1.f);
                u = fu; v = fv;
            }
            // NaN (zero direction) quantises to 0
            const uint32_t qu = uint32_t(fminf(32767.f, fmaxf(0.f, (u * 0.5f + 0.5f) * 32767.f + 0.5f)));
            const uint32_t qv = uint32_t(fminf(32767.f, fmaxf(0.f, (v * 0.5f + 0.5f) * 32767.f + 0.5f)));
            keys[i] = hilbert2d_15(qu, qv);
            continue;
        }
        uint32_t q[6];
#pragma unroll
        // Round to nearest: a regular ray grid then maps to distinct, evenly spaced cells whatever
        // the rounding of the scaling (a truncated 1023.9999 would merge two pixel columns and
        // skew every 8x8 tile after it).
        for (int k = 0; k < 6; ++k) q[k] = uint32_t(fminf(qmax, (r[k] - lo[k]) * scale[k] * qmax + 0.5f));
#ifndef GRACE_MORTON2D
        if (nvar == 2 && !z_order_2d) {
            // Two varying co-ordinates (orthographic and plane-parallel batches): the Hilbert curve
            // again.  A power-of-two pixel grid gives the same 8x8 tiles as the Z-order curve; any
            // other grid, or jittered origins, gives connected patches where Z-order runs straddle.
            uint32_t xy[2] = {0, 0};
            int m = 0;
#pragma unroll
            for (int k = 5; k >= 0; --k)
                if (scale[k] > 0.f) { if (m < 2) xy[m] = q[k]; ++m; }
            keys[i] = hilbert2d_15(xy[1], xy[0]);
            continue;
        }
#endif
        uint32_t key = 0;
        for (int b = bits - 1; b >= 0; --b) {
#pragma unroll
            for (int k = 5; k >= 0; --k) // origin x is the least significant dimension
                if (scale[k] > 0.f) key = (key << 1) | ((q[k] >> b) & 1u);
        }
        keys[i] = key << (30 - bits * nvar);   // left-aligned in 30 bits (the host sorts the top bits)
    }
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

bool g_ray_reorder = true;
int g_treelet = -1; // -1: chosen per call from the packet count (512 on a full GPU, else 256)

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
__global__ __launch_bounds__(TRACE_BLOCK, (SPLIT && MODE != MODE_HITS) ? 8 : 1) void trace_kernel(const TraceArgs a)
{
    static_assert(!ALT || MODE == MODE_CUMULATIVE || MODE == MODE_HITS, "no alternative path for this mode");
    static_assert(!LAT || MODE == MODE_COUNT || MODE == MODE_CUMULATIVE || MODE == MODE_HITS, "no lattice cull for this mode");
    if (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE || MODE == MODE_HITS) {
        if (a.lat_dev ? (*a.lat_dev != 0) != LAT : LAT) return;   // (workgroup-uniform)
    }
    constexpr bool FAST = ALT && MODE == MODE_CUMULATIVE;
    __shared__ double2 s_lut[FAST ? 1 : N_TABLE];
    __shared__ float2 s_lutf[FAST ? N_TABLE + 1 : 1];
    // Per-wave tile of the candidates of the current culling round (MODE_TRI keeps its
    // fp64 triangles on the scalar path).
    constexpr bool D4 = (MODE == MODE_COUNT_D4 || MODE == MODE_CUM_D4 || MODE == MODE_HITS_D4);
    constexpr bool LDS_TILE = (MODE != MODE_TRI && !D4);
    // Three 8-byte planes per wave -- (x, y), (z, h^2), (1/h terms) -- so that one address
    // (plane base + 8 j) serves all of a survivor's reads through immediate offsets.
    // (66 slots: the survivor loop reads up to two slots past the round's last survivor)
    __shared__ float2 s_tile[LDS_TILE ? TRACE_BLOCK / 64 : 1][LDS_TILE ? 3 : 1][LDS_TILE ? 66 : 1];
    // *_D4 modes: the round's candidates as doubles, lane-indexed: {x, y, z, w w, 1/w, (1/w)^2}
    // (the division is done once per candidate by its lane, not once per survivor by the wave).
    __shared__ double s_tile_d[D4 ? TRACE_BLOCK / 64 : 1][D4 ? 64 : 1][D4 ? 6 : 1];
    const int lane = threadIdx.x & 63;
    constexpr bool SPLITTABLE = (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE || MODE == MODE_HITS);
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
        if (threadIdx.x < N_TABLE + (FAST ? 1 : 0)) {
            const int i0 = threadIdx.x < N_TABLE ? threadIdx.x : N_TABLE - 1;
            const double y0 = c_kernel_table[i0];
            const double y1 = i0 + 1 < N_TABLE ? c_kernel_table[i0 + 1] : y0;
            if (FAST) s_lutf[threadIdx.x] = make_float2(float(y0), float(y1 - y0));
            else s_lut[threadIdx.x] = make_double2(y0, y1 - y0);
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
    constexpr bool CLASSES = (MODE == MODE_CUMULATIVE);
    __shared__ float s_class[CLASSES ? TRACE_BLOCK / 64 : 1][CLASSES ? SUM_CLASSES : 1][CLASSES ? 64 : 1];
    const int wv_acc = threadIdx.x >> 6;
    if (CLASSES) {
#pragma unroll
        for (int c = 0; c < SUM_CLASSES; ++c) s_class[wv_acc][c][lane] = 0.f;
    }
    int cur_granule = -1;            // wave-uniform
    int cur_granule_end = 0;         // first primitive past the current granule
    bool cur_owned = true;
    auto enter_granule = [&](const int prim) {
        if (cur_granule >= 0) s_class[wv_acc][cur_granule & (SUM_CLASSES - 1)][lane] = sum;
        cur_granule = prim >> GRANULE_SHIFT;
        cur_granule_end = (cur_granule + 1) << GRANULE_SHIFT;
        cur_owned = !SPLIT || owns_granule(cur_granule);
        sum = s_class[wv_acc][cur_granule & (SUM_CLASSES - 1)][lane];
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
    double sum_d = 0.0;     // MODE_CUM_D4: one running double sum per ray, ascending primitive index
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

    push(*a.root, ~0ull);
    unsigned long long st_walk = 0, st_cluster = 0, st_cull = 0, st_surv = 0, st_rounds = 0, st_nsurv = 0;
    const unsigned long long st_begin = STAMP_NOW();
    (void)st_walk; (void)st_cluster; (void)st_cull; (void)st_surv; (void)st_rounds; (void)st_nsurv; (void)st_begin;

    while (sp >= 0) {
        const unsigned long long st_t0 = STAMP_NOW(); (void)st_t0;
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

        int sweep_first = 0, sweep_count = 0;
        bool sweep = false;
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
            // (A wave-uniform box-overlap test of the packet's bounding box -- twelve compares
            // instead of this per-ray slab test -- was tried twice for axis-aligned packets: same
            // node count, no gain (node tests are ~320 per packet, ~12 % of the vector work).)
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
            const int wv = threadIdx.x >> 6;
            const int r_lo = leaf.x, r_hi = leaf.x + leaf.y;   // the swept primitives (wave-uniform)
            const int c_first = r_lo >> 6, c_last = (r_hi - 1) >> 6;
            // Lane j's candidate of cluster c: primitive 64 c + j, clamped into the range (idle
            // lanes then hold a valid candidate and the tests need no control flow).
            double4 mined_next = make_double4(0., 0., 0., 0.);   // *_D4: the candidate's double4 record
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
                bool keep = may_hit & (lane >= lo_bit) & (lane < hi_bit);
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
                    if (!COMPACT || keep) {
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
                            const int x_idx = static_cast<int>(b);
                            const float t = __builtin_amdgcn_fractf(b);
                            const float2 y = s_lutf[x_idx];
                            sum = __builtin_fmaf(__builtin_fmaf(t, y.y, y.x), sb.y, sum);
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
                if (lean_round) run(std::true_type());
                else run(std::false_type());
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
    if (MODE == MODE_COUNT_D4) a.out_counts[ray_index] = count;
    if (MODE == MODE_CUM_D4) a.out_sums_d[ray_index] = sum_d;
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

int* g_status = nullptr; // one device int, allocated on first use
// Measurement hook (grace_trace_last_lattice): the device flag of the last trace launch (lives in
// the call's workspace frame: valid until the next library call on the device).
const int* g_last_lat_dev = nullptr;
hipStream_t g_last_lat_stream = nullptr;
bool g_timing = false;   // record HIP events around the traversal kernel itself
hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
bool g_ev_valid = false;

grace_status ensure_status(hipStream_t stream)
{
    if (!g_status) {
        GRACE_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&g_status), sizeof(int)));
        GRACE_TRY_HIP(hipMemsetAsync(g_status, 0, sizeof(int), stream));
    }
    return GRACE_OK;
}

// Upper levels of the pairwise summation tree for split packets: K subtree sums per ray.
__global__ __launch_bounds__(256) void combine_classes_kernel(const float* __restrict__ partial,
                                                              int n_rays, int split,
                                                              const int* __restrict__ split_dev,
                                                              float* __restrict__ out,
                                                              const int* __restrict__ run_if = nullptr)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    if (run_if && *run_if == 0) return;   // the one-wave-per-packet kernel ran: `out` is final
    if (split_dev) split = *split_dev;
    float t[SUM_CLASSES];
    for (int k = 0; k < split; ++k) t[k] = partial[size_t(r) * split + k];
    for (int w = 1; w < split; w *= 2)
        for (int k = 0; k < split; k += 2 * w) t[k] = t[k] + t[k + w];
    out[r] = t[0];
}

// Plan of the split per-hit trace.
// (1) hits_offsets_kernel / hits_plan_kernel: each ray's chunk counts become output offsets
//     (exclusive scan along the chunks, starting at the ray's own offset); the packet's running
//     chunk totals are kept (pk_prefix) with its grand total (pk_total).
// (2) hits_assign_kernel, one workgroup: the W launched waves are dealt to the packets in
//     proportion to their hit totals -- K_p = 1 + floor((W - P) H_p / H) -- so that waves, not
//     packets, carry equal work (HEALPix / isotropic bundles: rays along a box diagonal collect
//     1.7x the hits of rays along an axis; with a fixed K the slowest packet set the kernel time
//     at 2.5x the mean wave's).
// (3) hits_bounds_kernel, one wave per packet: its chunks are cut into K_p contiguous ranges of
//     about equal hit totals; wave first_p + k gets {packet, first chunk, end chunk}.
// (1a) one wavefront per RAY: the ray's row of chunk counts (contiguous: coalesced) becomes its
//      row of output offsets.
__global__ __launch_bounds__(256) void hits_offsets_kernel(const int* __restrict__ chunk_counts,
                                                           const int* __restrict__ ray_offsets,
                                                           int n_rays, int n_chunks,
                                                           int* __restrict__ chunk_off)
{
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int* row = chunk_counts + size_t(ray) * n_chunks;
    int* out = chunk_off + size_t(ray) * n_chunks;
    int carry = ray_offsets[ray];
    for (int c0 = 0; c0 < n_chunks; c0 += 64) {
        const int c = c0 + lane;
        const int v = c < n_chunks ? row[c] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
        if (c < n_chunks) out[c] = carry + incl - v;
        carry += __shfl(incl, 63);
    }
}

// (1b) one workgroup per PACKET, one thread per chunk: the packet's hits per chunk (sum over its
//      64 rays, coalesced along the chunks), their running totals and the grand total.
__global__ __launch_bounds__(MAX_HIT_CHUNKS) void hits_plan_kernel(const int* __restrict__ chunk_counts,
                                                                   const uint32_t* __restrict__ perm,
                                                                   int n_rays, int n_chunks,
                                                                   uint32_t* __restrict__ pk_prefix,
                                                                   uint32_t* __restrict__ pk_total)
{
    __shared__ uint32_t s_wave[MAX_HIT_CHUNKS / 64];
    const int packet = blockIdx.x, c = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t t = 0;
    if (c < n_chunks)
        for (int r = 0; r < 64; ++r) {
            const int slot = packet * 64 + r;
            if (slot >= n_rays) break;
            const int ray = perm ? int(perm[slot]) : slot;
            t += uint32_t(chunk_counts[size_t(ray) * n_chunks + c]);
        }
    uint32_t incl = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if (lane >= o) incl += up; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    if (c < n_chunks) pk_prefix[size_t(packet) * n_chunks + c] = before + incl;   // inclusive
    if (c == n_chunks - 1) pk_total[packet] = before + incl;
}

__global__ __launch_bounds__(1024) void hits_assign_kernel(const uint32_t* __restrict__ pk_total,
                                                           int n_packets, int n_waves, int n_chunks,
                                                           int* __restrict__ pk_first,
                                                           int* __restrict__ pk_parts,
                                                           int* __restrict__ n_used)
{
    __shared__ unsigned long long s_red[16];
    __shared__ int s_scan[16];
    __shared__ int s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long h = 0;
    for (int p = threadIdx.x; p < n_packets; p += blockDim.x) h += pk_total[p];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
    if (lane == 0) s_red[wave] = h;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    unsigned long long H = 0;
    for (int w = 0; w < 16; ++w) H += s_red[w];
    const unsigned long long pool = (unsigned long long)(n_waves > n_packets ? n_waves - n_packets : 0);
    for (int base = 0; base < n_packets; base += blockDim.x) {
        const int p = base + threadIdx.x;
        int k = 0;
        if (p < n_packets) {
            k = 1 + (H ? int(pool * pk_total[p] / H) : 0);
            if (k > n_chunks) k = n_chunks;
        }
        int incl = k;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        int before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_scan[w];
        if (p < n_packets) { pk_first[p] = before + incl - k; pk_parts[p] = k; }
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *n_used = s_carry;
        *reinterpret_cast<unsigned long long*>(n_used + 2) = H;   // the batch's hit total, for the host
    }
}

__global__ __launch_bounds__(64) void hits_bounds_kernel(const uint32_t* __restrict__ pk_prefix,
                                                         const uint32_t* __restrict__ pk_total,
                                                         const int* __restrict__ pk_first,
                                                         const int* __restrict__ pk_parts, int n_chunks,
                                                         int4* __restrict__ wave_map)
{
    const int packet = blockIdx.x, lane = threadIdx.x;
    const uint32_t* pre = pk_prefix + size_t(packet) * n_chunks;
    const unsigned long long run = pk_total[packet];
    const int parts = pk_parts[packet], first = pk_first[packet];
    // Boundary j = first chunk whose inclusive prefix reaches j / parts of the total.
    auto boundary = [&](const int j) {
        if (j <= 0) return 0;
        if (j >= parts) return n_chunks;
        const unsigned long long want = (run * (unsigned long long)j + parts - 1) / parts;
        int lo = 0, hi = n_chunks - 1;           // smallest c with pre[c] >= want
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] >= want) hi = mid; else lo = mid + 1;
        }
        return lo + 1 > n_chunks ? n_chunks : lo + 1;   // chunks [.., lo] belong to the parts before
    };
    for (int k = lane; k < parts; k += 64)
        wave_map[first + k] = make_int4(packet, boundary(k), boundary(k + 1), 0);
}

int g_split = -1; // waves per packet; -1: automatic
// waves per packet for big batches of scenes with sub-spacing spheres (measurement switch:
// GRACE_LAT_SPLIT=0 keeps one wave per packet; 2, 4 (default), 8)
const int g_lat_split = [] {
    const char* e = std::getenv("GRACE_LAT_SPLIT");
    const int k = e ? std::atoi(e) : 4;
    return (k == 0 || k == 2 || k == 4 || k == 8) ? k : 4;
}();
int g_width = -1; // rays per packet of the per-hit / triangle traces; -1: automatic
bool g_exact_integrals = false; // column-density trace: bit-reproducible per-hit arithmetic

// ---- scene-constant pre-pass data --------------------------------------------------------
// A, B, the nodes' primitive spans and the cluster boxes depend on the primitives and the tree
// only.  By default every trace call recomputes them into the workspace (the reference's
// trace calls are stateless too); grace_trace_prepare_f4 / _tri computes them ONCE into buffers
// of their own, and later trace calls whose primitive / node / leaf pointers and sizes match
// reuse them (0.25 ms per call at 10^7 particles -- a fifth of a 1/8-image shard's trace).  The
// caller promises not to change those arrays until grace_trace_release(); this library's own
// sort and build entry points drop the cache when they write to one of them.
struct Scene {
    bool valid = false, tri = false;
    const void* prims = nullptr; const void* nodes = nullptr; const void* leaves = nullptr;
    size_t n_prims = 0, n_nodes = 0;
    float4* A = nullptr; float2* B1 = nullptr; float2* B50 = nullptr; double* T64 = nullptr;
    int2* node_prims = nullptr; float4* C = nullptr;
};
Scene g_scene;

grace_status scene_release()
{
    if (g_scene.A || g_scene.B1 || g_scene.B50 || g_scene.T64 || g_scene.node_prims || g_scene.C)
        GRACE_TRY_HIP(hipDeviceSynchronize());
    void* bufs[] = { g_scene.A, g_scene.B1, g_scene.B50, g_scene.T64, g_scene.node_prims, g_scene.C };
    for (void* b : bufs)
        if (b) GRACE_TRY_HIP(hipFree(b));
    g_scene = Scene();
    return GRACE_OK;
}

// Fills the scene-constant arrays (any of B1 / B50 / T64 may be null).  kind: 0 float4 spheres,
// 1 triangles, 2 double4 spheres.
grace_status scene_fill(int kind, const void* prims, size_t n_prims, const float4* nodes,
                        size_t n_nodes, const int4* leaves, float4* A, float2* B1, float2* B50,
                        double* T64, int2* node_prims, float4* C, hipStream_t stream)
{
    node_prims_kernel<<<ceil_div(n_nodes, 256), 256, 0, stream>>>(
        reinterpret_cast<const int4*>(nodes), leaves, int(n_nodes), node_prims,
        reinterpret_cast<uint32_t*>(C + 2 * ((n_prims + 63) / 64)));
    GRACE_CHECK_LAUNCH();
    if (kind == 1) {
        tri_prepass_kernel<<<stream_grid(n_prims + 4, 256), 256, 0, stream>>>(
            static_cast<const float*>(prims), n_prims, A, T64);
        GRACE_CHECK_LAUNCH();
    } else if (kind == 2) {
        trace_prepass_d4_kernel<<<stream_grid(n_prims + 4, 256), 256, 0, stream>>>(
            static_cast<const double*>(prims), n_prims, A);
        GRACE_CHECK_LAUNCH();
    } else {
        trace_prepass_kernel<<<stream_grid(n_prims + 4, 256), 256, 0, stream>>>(
            static_cast<const float4*>(prims), n_prims, A, B1 ? B1 : B50,
            B1 ? 1.0f : float(N_TABLE - 1), C);      // (+ the cluster boxes: fused)
        GRACE_CHECK_LAUNCH();
        if (B1 && B50) {
            trace_prepass_kernel<<<stream_grid(n_prims + 4, 256), 256, 0, stream>>>(
                static_cast<const float4*>(prims), n_prims, A, B50, float(N_TABLE - 1));
            GRACE_CHECK_LAUNCH();
        }
    }
    if (kind != 0) {
        cluster_boxes_kernel<<<stream_grid((n_prims + 63) / 64, 4), 256, 0, stream>>>(A, n_prims, C);
        GRACE_CHECK_LAUNCH();
    }
    return GRACE_OK;
}

grace_status scene_prepare(bool tri, const void* prims, size_t n_prims, const int* d_nodes,
                           size_t n_nodes, const int* d_leaves, hipStream_t stream)
{
    GRACE_REQUIRE(prims && d_nodes && d_leaves, "trace_prepare: null pointer");
    GRACE_REQUIRE(n_prims > 0 && n_nodes >= 1, "trace_prepare: empty scene");
    GRACE_TRY(scene_release());
    Scene sc;
    auto alloc = [&](void** ptr, size_t bytes) -> grace_status {
        hipError_t e = hipMalloc(ptr, bytes);
        if (e != hipSuccess)
            return set_error(GRACE_OUT_OF_MEMORY, __FILE__, __LINE__, hipGetErrorString(e));
        return GRACE_OK;
    };
    grace_status st = alloc(reinterpret_cast<void**>(&sc.A), (n_prims + 4) * sizeof(float4));
    if (st == GRACE_OK && !tri) st = alloc(reinterpret_cast<void**>(&sc.B1), (n_prims + 4) * sizeof(float2));
    if (st == GRACE_OK && !tri) st = alloc(reinterpret_cast<void**>(&sc.B50), (n_prims + 4) * sizeof(float2));
    if (st == GRACE_OK && tri) st = alloc(reinterpret_cast<void**>(&sc.T64), 72 * (n_prims + 4));
    if (st == GRACE_OK) st = alloc(reinterpret_cast<void**>(&sc.node_prims), n_nodes * sizeof(int2));
    if (st == GRACE_OK) st = alloc(reinterpret_cast<void**>(&sc.C), (2 * ((n_prims + 63) / 64) + 1) * sizeof(float4));
    g_scene = sc;   // so that a failure below releases what was allocated
    if (st != GRACE_OK) { scene_release(); return st; }
    st = scene_fill(tri ? 1 : 0, prims, n_prims, reinterpret_cast<const float4*>(d_nodes), n_nodes,
                    reinterpret_cast<const int4*>(d_leaves), sc.A, sc.B1, sc.B50, sc.T64,
                    sc.node_prims, sc.C, stream);
    if (st != GRACE_OK) { scene_release(); return st; }
    // (a one-time call: wait for the records, so that traces on ANY stream may use them)
    GRACE_TRY_HIP(hipStreamSynchronize(stream));
    g_scene.valid = true; g_scene.tri = tri;
    g_scene.prims = prims; g_scene.nodes = d_nodes; g_scene.leaves = d_leaves;
    g_scene.n_prims = n_prims; g_scene.n_nodes = n_nodes;
    return GRACE_OK;
}

// Prepared ray batch (grace_trace_prepare_rays): the ray coherence order -- extents, keys, the
// partial sort: ten small launches, ~0.08 ms, a seventh of a 1/8-image shard's call -- depends on
// the rays alone.  The reference leaves ray ordering to the caller (its generators sort at
// generation time, gen_rays.cuh:483,520,577,615); a caller that traces the SAME batch repeatedly
// (a fixed camera over an evolving scene, a benchmark loop) computes it once here.  Keyed on
// (pointer, count): the caller promises not to change the rays until grace_trace_release_rays();
// this library's own ray generators drop the cache when they write to the array.
struct RayOrder {
    bool valid = false;
    const float* rays = nullptr;
    size_t n = 0;
    uint32_t* perm = nullptr;   // n
    uint32_t* ext = nullptr;    // 12 extents (order-preserving uints: minima then maxima of d, o)
};
RayOrder g_rays;

grace_status rays_release()
{
    if (g_rays.perm || g_rays.ext) GRACE_TRY_HIP(hipDeviceSynchronize());
    if (g_rays.perm) GRACE_TRY_HIP(hipFree(g_rays.perm));
    if (g_rays.ext) GRACE_TRY_HIP(hipFree(g_rays.ext));
    g_rays = RayOrder();
    return GRACE_OK;
}

// extents -> keys -> partial sort (see launch_trace); keys: n words of scratch
grace_status ray_order(const float* d_rays, size_t n_rays, uint32_t* ext, uint32_t* keys, uint32_t* perm,
                       const float4* scene_min, uint32_t* lat_flag, int n_packets, int split, int* split_dev,
                       hipStream_t stream)
{
    // minima at the top of the order, maxima / per-call choices / grid flag at zero: one tiny launch
    // (two hipMemsetAsync of 24 and 40 bytes became four fill kernels)
    ray_ext_init_kernel<<<1, 64, 0, stream>>>(ext);
    GRACE_CHECK_LAUNCH();
    ray_extents_kernel<<<(stream_grid(n_rays, 256, 8) < 256 ? stream_grid(n_rays, 256, 8) : 256), 256, 0, stream>>>(
        d_rays, int(n_rays), ext);
    GRACE_CHECK_LAUNCH();
    ray_lattice_kernel<<<(stream_grid(n_rays, 256, 8) < 256 ? stream_grid(n_rays, 256, 8) : 256), 256, 0, stream>>>(
        d_rays, int(n_rays), ext, ext + 14);
    GRACE_CHECK_LAUNCH();
    ray_keys_kernel<<<stream_grid(n_rays, 256), 256, 0, stream>>>(d_rays, int(n_rays), ext, keys, scene_min,
                                                                 lat_flag, n_packets, split, split_dev, ext + 14);
    GRACE_CHECK_LAUNCH();
    // Only the key bits that decide which PACKET a ray joins need sorting: the order of
    // the rays inside a packet is irrelevant (log2(packets) + 2 bits, in whole 8-bit
    // passes; keys are left-aligned in 30 bits).  The sort is stable, so ties keep the
    // caller's order.
    const size_t packets64 = ceil_div(n_rays, size_t(64));
    int want_bits = 2;
    while ((size_t(1) << (want_bits - 2)) < packets64 && want_bits < 30) ++want_bits;
    want_bits = ((want_bits + 7) / 8) * 8;
    const int begin_bit = want_bits >= 30 ? 0 : 30 - want_bits;
    return sort_pairs_u32_nested(keys, nullptr, n_rays, 0, begin_bit, 30, perm, stream);
}

grace_status rays_prepare(const float* d_rays, size_t n_rays, hipStream_t stream)
{
    GRACE_REQUIRE(d_rays || n_rays == 0, "trace_prepare_rays: null pointer");
    GRACE_REQUIRE(n_rays < (size_t(1) << 31), "trace_prepare_rays: bad ray count");
    GRACE_TRY(rays_release());
    if (n_rays <= 64) return GRACE_OK;          // one packet: nothing to order
    RayOrder ro;
    if (hipMalloc(reinterpret_cast<void**>(&ro.perm), n_rays * 4) != hipSuccess
        || hipMalloc(reinterpret_cast<void**>(&ro.ext), 64) != hipSuccess) {
        if (ro.perm) (void)hipFree(ro.perm);
        return set_error(GRACE_OUT_OF_MEMORY, __FILE__, __LINE__, "trace_prepare_rays: out of device memory");
    }
    g_rays = ro;
    grace_status st = Workspace::begin(Workspace::aligned(n_rays * 4) + sort_ws_bytes(n_rays, 4, 0) + 1024, stream);
    if (st == GRACE_OK) {
        uint32_t* keys = Workspace::take<uint32_t>(n_rays);
        st = ray_order(d_rays, n_rays, ro.ext, keys, ro.perm, nullptr, nullptr, 0, 0, nullptr, stream);
    }
    if (st != GRACE_OK) { rays_release(); return st; }
    // (a one-time call: wait for the order, so that traces on ANY stream may use it)
    if (hipStreamSynchronize(stream) != hipSuccess) {
        rays_release();
        return set_error(GRACE_HIP_ERROR, __FILE__, __LINE__, "trace_prepare_rays: stream synchronisation failed");
    }
    g_rays.valid = true; g_rays.rays = d_rays; g_rays.n = n_rays;
    return GRACE_OK;
}

// trace_sph walks twice -- hit counts (for the offsets), then the per-hit pass -- and the split
// per-hit pass of small batches needs hits per (ray, chunk), a third walk.  The hit-count call
// made on behalf of trace_sph (grace_trace_hitcounts_keep_f4) records them into this buffer of
// its own (the workspace is reset by the scan in between); the per-hit call that follows on the
// same rays and spheres consumes them.
struct HitsCache {
    int* chunk_counts = nullptr;
    size_t capacity = 0;      // ints
    bool valid = false;
    const void* rays = nullptr; const void* prims = nullptr;
    size_t n_rays = 0, n_prims = 0;
    int n_chunks = 0;
};
HitsCache g_hits;
// split per-hit walk: stage hits in LDS (measurement switch: GRACE_HITS_STAGE=0 -> direct stores)
const bool g_hits_stage_split = [] { const char* e = std::getenv("GRACE_HITS_STAGE"); return !e || e[0] != '0'; }();

template <int MODE>
grace_status launch_trace(TraceArgs a, size_t n_rays, size_t n_spheres, size_t n_nodes,
                          hipStream_t stream)
{
    GRACE_REQUIRE(a.rays && a.spheres && a.nodes && a.leaves && a.root, "trace: null pointer");
    GRACE_REQUIRE(n_rays < (size_t(1) << 31), "trace: bad ray count");
    GRACE_REQUIRE(n_nodes >= 1 && n_nodes < (size_t(1) << 30), "trace: bad node count");
    GRACE_REQUIRE(n_spheres > 0 && n_spheres < (size_t(1) << 31), "trace: bad primitive count");
    GRACE_TRY(ensure_status(stream));
    // Split per-hit trace for small batches (see TraceArgs / hits_plan_kernel): chunk size =
    // a power of two >= one granule giving at most MAX_HIT_CHUNKS chunks.
    int hit_chunk_shift = GRANULE_SHIFT;
    while ((((n_spheres - 1) >> hit_chunk_shift) + 1) > size_t(MAX_HIT_CHUNKS)) ++hit_chunk_shift;
    const int hit_chunks = int(((n_spheres - 1) >> hit_chunk_shift) + 1);
    const size_t hit_packets = ceil_div(n_rays, size_t(64));
    int hit_split = 1;
    if (MODE == MODE_HITS && g_width <= 0 && hit_packets < 4096 && hit_chunks >= 8) {
        if (g_split > 0) hit_split = g_split;
        else while (hit_split < 8 && hit_packets * hit_split < 16384) hit_split *= 2;
    }
    const bool hits_split = hit_split > 1;
    int* chunk_counts = nullptr; int* chunk_off = nullptr;
    int* scratch_counts = nullptr;
    int4* wave_map = nullptr; int* n_wave_map = nullptr;
    uint32_t* pk_prefix = nullptr; uint32_t* pk_total = nullptr; int* pk_first = nullptr; int* pk_parts = nullptr;
    // Would the per-hit trace of this batch use the split path?  (the same rule, for the hit-count
    // call that is asked to keep its chunk counts)
    const bool keep_chunks = MODE == MODE_COUNT && a.keep_chunks && g_width <= 0 && hit_packets < 4096
        && hit_chunks >= 8 && g_split != 1;
    const bool reuse_chunks = MODE == MODE_HITS && hits_split && g_hits.valid && g_hits.rays == a.rays
        && g_hits.n_rays == n_rays && g_hits.prims == static_cast<const void*>(a.spheres)
        && g_hits.n_prims == n_spheres && g_hits.n_chunks == hit_chunks;
    if (MODE == MODE_HITS || MODE == MODE_COUNT) g_hits.valid = false;   // consumed, or stale from here on
    if (keep_chunks) {
        const size_t need = n_rays * size_t(hit_chunks);
        if (g_hits.capacity < need) {
            if (g_hits.chunk_counts) { GRACE_TRY_HIP(hipDeviceSynchronize()); GRACE_TRY_HIP(hipFree(g_hits.chunk_counts)); }
            g_hits.chunk_counts = nullptr; g_hits.capacity = 0;
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&g_hits.chunk_counts), need * sizeof(int));
            if (e != hipSuccess) return set_error(GRACE_OUT_OF_MEMORY, __FILE__, __LINE__, hipGetErrorString(e));
            g_hits.capacity = need;
        }
    }
    // Per-hit and triangle traces cannot split a packet among waves (their outputs are ordered
    // / reduced per ray inside one wave); with few rays they use narrower packets instead:
    // 2-4x the waves, each with a tighter beam, on a chip that would otherwise sit idle.
    int width = 64;
    if (g_width > 0) width = g_width;
    else if ((MODE == MODE_HITS && !hits_split) || MODE == MODE_TRI || MODE == MODE_COUNT_D4
             || MODE == MODE_CUM_D4 || MODE == MODE_HITS_D4)
        while (width > 16 && ceil_div(n_rays, size_t(width)) < 4096) width /= 2;
    // Hit counts and column densities split packets eight ways at most; a batch too small to fill
    // the chip even then (< 512 packets) also gets narrower packets (10^7 particles, 12288 HEALPix
    // rays: 3.5 -> 2.0 ms at 16 rays per packet; from 49152 rays on it loses: config 3 0.87 -> 0.95 ms).
    else if ((MODE == MODE_COUNT || MODE == MODE_CUMULATIVE) && g_split <= 0)
        while (width > 16 && ceil_div(n_rays, size_t(width)) * SUM_CLASSES < 4096) width /= 2;
    a.width = width;
    const int n_packets = ceil_div(n_rays, size_t(width));
    // Waves per packet: two resident sets of waves (2 x 8192) for small ray batches.
    int split = 1;
    if (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE) {
        if (g_split > 0) split = g_split;
        else while (split < SUM_CLASSES && size_t(n_packets) * split < 16384) split *= 2;
    }
    if (hits_split) split = hit_split;
    a.split = split;
    a.split_dev = nullptr;
    // (the working waves per packet are chosen on the device, by ray_keys_kernel)
    const bool dev_split = (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE) && split > 1 && g_split <= 0;
    {
        constexpr bool need_b = (MODE == MODE_CUMULATIVE || MODE == MODE_HITS);
        const bool fast_b = MODE == MODE_CUMULATIVE && !g_exact_integrals;
        const bool reorder = g_ray_reorder && n_rays > 64;
        constexpr bool D4 = (MODE == MODE_COUNT_D4 || MODE == MODE_CUM_D4 || MODE == MODE_HITS_D4);
        const bool cached = !D4 && g_scene.valid && g_scene.tri == (MODE == MODE_TRI)
            && g_scene.prims == static_cast<const void*>(a.spheres) && g_scene.n_prims == n_spheres
            && g_scene.nodes == static_cast<const void*>(a.nodes) && g_scene.n_nodes == n_nodes
            && g_scene.leaves == static_cast<const void*>(a.leaves);
        const size_t n_clusters = (n_spheres + 63) / 64;
        GRACE_TRY(Workspace::begin((cached ? 0 : Workspace::aligned((n_spheres + 4) * sizeof(float4))
                                               + Workspace::aligned((n_spheres + 4) * sizeof(float2))
                                               + Workspace::aligned(n_nodes * sizeof(int2))
                                               + Workspace::aligned((2 * n_clusters + 1) * sizeof(float4))
                                               + (MODE == MODE_TRI ? Workspace::aligned(72 * (n_spheres + 4)) : 0))
                                   + (hits_split ? 2 * Workspace::aligned(n_rays * size_t(hit_chunks) * 4)
                                                   + Workspace::aligned(hit_packets * hit_split * sizeof(int4))
                                                   + Workspace::aligned(hit_packets * size_t(hit_chunks) * 4)
                                                   + 4 * Workspace::aligned(hit_packets * 4 + 64)
                                                   + Workspace::aligned(n_rays * 4) : 0)
                                   + (MODE == MODE_CUMULATIVE ? Workspace::aligned(n_rays * SUM_CLASSES * 4) : 0)
                                   + (reorder ? 2 * Workspace::aligned(n_rays * 4)
                                                + sort_ws_bytes(n_rays, 4, 0) : 0) + 1024, stream));
        if (cached) {
            a.A = g_scene.A;
            a.B = need_b ? (fast_b ? g_scene.B50 : g_scene.B1) : nullptr;
            a.T64 = g_scene.T64;
            a.node_prims = g_scene.node_prims;
            a.C = g_scene.C;
        } else {
            float4* A = Workspace::take<float4>(n_spheres + 4);
            float2* B = need_b ? Workspace::take<float2>(n_spheres + 4) : nullptr;
            double* T64 = (MODE == MODE_TRI) ? Workspace::take<double>(9 * (n_spheres + 4)) : nullptr;
            int2* node_prims = Workspace::take<int2>(n_nodes);
            float4* C = Workspace::take<float4>(2 * n_clusters + 1);
            GRACE_TRY(scene_fill(MODE == MODE_TRI ? 1 : D4 ? 2 : 0,
                                 D4 ? static_cast<const void*>(a.spheres_d) : a.spheres, n_spheres, a.nodes, n_nodes, a.leaves, A,
                                 fast_b ? nullptr : B, fast_b ? B : nullptr, T64, node_prims, C, stream));
            a.A = A; a.B = B; a.T64 = T64; a.node_prims = node_prims; a.C = C;
        }
        a.partial = (MODE == MODE_CUMULATIVE) ? Workspace::take<float>(n_rays * SUM_CLASSES) : nullptr;
        if (hits_split) {
            chunk_counts = Workspace::take<int>(n_rays * size_t(hit_chunks));
            chunk_off = Workspace::take<int>(n_rays * size_t(hit_chunks));
            wave_map = Workspace::take<int4>(hit_packets * hit_split);
            pk_prefix = Workspace::take<uint32_t>(hit_packets * size_t(hit_chunks));
            pk_total = Workspace::take<uint32_t>(hit_packets + 16);
            pk_first = Workspace::take<int>(hit_packets + 16);
            pk_parts = Workspace::take<int>(hit_packets + 16);
            n_wave_map = Workspace::take<int>(16);
            scratch_counts = Workspace::take<int>(n_rays);
        }
        // Subtrees of up to this many primitives are swept -- cluster tests, then culling rounds
        // over the surviving clusters -- rather than descended.
        // Axis-aligned packets test a cluster's box against their origin rectangle (sharp: large
        // subtrees pay, 16384 measured best on full frames and shards alike); pencil packets test it
        // against the bundle's side planes, general packets its circumscribed sphere.
        // (re-measured after the pencil cluster test became a box-against-side-planes test: sphere
        // scenes now prefer 8192 there too -- config 2 column densities 1.74 -> 1.46 ms, hit counts
        // 1.54 -> 1.19, config 3 0.94 -> 0.87 --; triangles, culled through bounding spheres, keep 512:
        // 5.3 / 4.6 / 2.8 ms for the three cameras against 6.9 / 5.8 / 3.0 at 8192)
        const int auto_treelet = (MODE == MODE_TRI) ? 512 : 8192, auto_treelet_axis = 16384;
#ifdef GRACE_PACKET_STATS
        a.treelet = g_treelet < 0 ? auto_treelet : g_treelet;
        a.treelet_axis = g_treelet < 0 ? auto_treelet_axis : g_treelet;
#else
        a.treelet = (MODE == MODE_STATS) ? 0 : (g_treelet < 0 ? auto_treelet : g_treelet);
        a.treelet_axis = (MODE == MODE_STATS) ? 0 : (g_treelet < 0 ? auto_treelet_axis : g_treelet);
#endif
        if (reorder) {
            // ext: 12 extents + [12] the device-side split + [13] the lattice flag (per call)
            uint32_t* ext = Workspace::take<uint32_t>(16);
            uint32_t* keys = Workspace::take<uint32_t>(n_rays);
            uint32_t* perm = Workspace::take<uint32_t>(n_rays);
            constexpr bool lat_mode = (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE || MODE == MODE_HITS);
            uint32_t* lat_flag = lat_mode ? ext + 13 : nullptr;
            int* split_dev = dev_split ? reinterpret_cast<int*>(ext + 12) : nullptr;
            if (g_rays.valid && g_rays.rays == a.rays && g_rays.n == n_rays) {
                // prepared ray batch: only this call's device-side choices remain
                if (lat_flag || split_dev) {
                    choose_variants_kernel<<<1, 1, 0, stream>>>(g_rays.ext, int(n_rays), a.C + 2 * n_clusters, lat_flag,
                                                                n_packets, split, split_dev);
                    GRACE_CHECK_LAUNCH();
                }
                a.perm = g_rays.perm;
            } else {
                GRACE_TRY(ray_order(a.rays, n_rays, ext, keys, perm, a.C + 2 * n_clusters, lat_flag, n_packets,
                                    split, split_dev, stream));
                a.perm = perm;
            }
            if (lat_mode) a.lat_dev = reinterpret_cast<const int*>(ext + 13);
            if (dev_split) a.split_dev = reinterpret_cast<const int*>(ext + 12);
        }
    }
    a.n_rays = int(n_rays);
    a.n_nodes = int(n_nodes);
    a.status = g_status;
    a.n_prims = int(n_spheres);
    a.chunk_shift = hit_chunk_shift;
    a.n_chunks = hit_chunks;
    a.chunk_counts = nullptr;
    a.chunk_off = chunk_off;
    a.wave_map = wave_map;
    a.n_wave_map = n_wave_map;
    if (split > 1 && MODE == MODE_COUNT)
        GRACE_TRY_HIP(hipMemsetAsync(a.out_counts, 0, n_rays * sizeof(int), stream));
    if (keep_chunks && split > 1) {
        a.chunk_counts = g_hits.chunk_counts;
        GRACE_TRY_HIP(hipMemsetAsync(g_hits.chunk_counts, 0, n_rays * size_t(hit_chunks) * 4, stream));
        g_hits.valid = true;
        g_hits.rays = a.rays; g_hits.n_rays = n_rays;
        g_hits.prims = a.spheres; g_hits.n_prims = n_spheres; g_hits.n_chunks = hit_chunks;
    }
    if (g_timing) {
        if (!g_ev0) {
            GRACE_TRY_HIP(hipEventCreate(&g_ev0));
            GRACE_TRY_HIP(hipEventCreate(&g_ev1));
        }
        GRACE_TRY_HIP(hipEventRecord(g_ev0, stream));
    }
    const int grid = ceil_div(size_t(n_packets) * split, TRACE_BLOCK / 64);
    g_last_lat_dev = a.lat_dev; g_last_lat_stream = stream;
    // Both variants of a kernel with a lattice instantiation (the device flag lets one run).
    auto both = [&](auto mode_tag, auto split_tag, auto alt_tag, const TraceArgs& args) {
        constexpr int M = decltype(mode_tag)::value;
        constexpr bool S = decltype(split_tag)::value, A = decltype(alt_tag)::value;
        trace_kernel<M, S, A, false><<<grid, TRACE_BLOCK, 0, stream>>>(args);
        if (args.lat_dev) trace_kernel<M, S, A, true><<<grid, TRACE_BLOCK, 0, stream>>>(args);
    };
    using T = std::true_type; using F = std::false_type;
    using M_ = std::integral_constant<int, MODE>;
    // A batch of >= 16384 packets runs one wave per packet -- unless the device flag says the scene
    // holds spheres smaller than the ray spacing (clustered SPH data: dense cores).  Such scenes
    // have packets dozens of times heavier than the median (10^7 particles, 90 % of them in 50
    // clumps: with the lattice cull the heaviest of 16384 waves still lived 14x the mean and set
    // the kernel time), so the lattice instantiation of these batches is the class-split kernel
    // with four waves per packet: the heaviest packets' work is spread over four SIMDs (measured
    // on two clustered scenes: K = 2 / 4 / 8 -> 3.62 / 3.47 / 4.24 ms and 4.24 / 3.42 / 3.71 ms;
    // one wave: 4.66 and 6.78 ms).  Same class sums, same bits.
    const bool lat_split = (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE) && split == 1 && a.lat_dev && width == 64
        && g_lat_split > 0 && g_split <= 0;   // (an explicit grace_trace_set_packet_split is obeyed)
    auto one_or_split = [&](auto alt_tag) -> grace_status {
        constexpr bool A = decltype(alt_tag)::value;
        constexpr int M = (MODE == MODE_COUNT) ? MODE_COUNT : MODE_CUMULATIVE;
        if (MODE == MODE_COUNT) GRACE_TRY_HIP(hipMemsetAsync(a.out_counts, 0, n_rays * sizeof(int), stream));
        trace_kernel<M, false, A, false><<<grid, TRACE_BLOCK, 0, stream>>>(a);
        GRACE_CHECK_LAUNCH();
        TraceArgs a8 = a;
        a8.split = g_lat_split; a8.split_dev = nullptr;
        trace_kernel<M, true, A, true><<<ceil_div(size_t(n_packets) * g_lat_split, TRACE_BLOCK / 64), TRACE_BLOCK, 0, stream>>>(a8);
        GRACE_CHECK_LAUNCH();
        if (MODE == MODE_CUMULATIVE) {
            combine_classes_kernel<<<ceil_div(n_rays, 256), 256, 0, stream>>>(a.partial, int(n_rays), g_lat_split,
                                                                              nullptr, a.out_sums, a.lat_dev);
            GRACE_CHECK_LAUNCH();
        }
        return GRACE_OK;
    };
    if constexpr (MODE == MODE_CUMULATIVE) {
        if (g_exact_integrals) {
            if (split > 1) both(M_(), T(), F(), a);
            else if (lat_split) GRACE_TRY(one_or_split(F()));
            else both(M_(), F(), F(), a);
        } else {
            if (split > 1) both(M_(), T(), T(), a);
            else if (lat_split) GRACE_TRY(one_or_split(T()));
            else both(M_(), F(), T(), a);
        }
    } else if constexpr (MODE == MODE_HITS) {
        if (hits_split) {
            // 1. hits per (ray, chunk): the counting walk, split by summation class -- unless the
            //    hit-count call made for this trace_sph has kept them (grace_trace_hitcounts_keep_f4)
            const int* counts = g_hits.chunk_counts;
            if (!reuse_chunks) {
                TraceArgs c = a;
                c.chunk_counts = chunk_counts;
                c.out_counts = scratch_counts;
                GRACE_TRY_HIP(hipMemsetAsync(chunk_counts, 0, n_rays * size_t(hit_chunks) * 4, stream));
                GRACE_TRY_HIP(hipMemsetAsync(scratch_counts, 0, n_rays * 4, stream));
                both(std::integral_constant<int, MODE_COUNT>(), T(), F(), c);
                GRACE_CHECK_LAUNCH();
                counts = chunk_counts;
            }
            // 2. output offsets per (ray, chunk); the launched waves dealt to the packets by hit
            //    totals; each packet's chunks cut into its waves' ranges
            hits_offsets_kernel<<<ceil_div(n_rays, 4), 256, 0, stream>>>(counts, a.offsets, int(n_rays),
                                                                         hit_chunks, chunk_off);
            GRACE_CHECK_LAUNCH();
            hits_plan_kernel<<<n_packets, MAX_HIT_CHUNKS, 0, stream>>>(counts, a.perm, int(n_rays),
                                                                       hit_chunks, pk_prefix, pk_total);
            GRACE_CHECK_LAUNCH();
            hits_assign_kernel<<<1, 1024, 0, stream>>>(pk_total, n_packets, n_packets * split, hit_chunks,
                                                       pk_first, pk_parts, n_wave_map);
            GRACE_CHECK_LAUNCH();
            hits_bounds_kernel<<<n_packets, 64, 0, stream>>>(pk_prefix, pk_total, pk_first, pk_parts,
                                                             hit_chunks, wave_map);
            GRACE_CHECK_LAUNCH();
            // 3. the per-hit walk, wave w owning wave_map[w]'s range of chunks
            //    Heavy packets (output-bandwidth-bound: 10^5 isotropic rays through 10^6 large spheres,
            //    410 k hits per packet: 18.0 -> 11.0 ms) stage their hits in LDS and store them eight
            //    per ray at a time; light ones (61 M hits over 768 packets: 3.6 ms direct, 4.5 staged)
            //    store directly.  The hit total is known on the device only: one 16-byte read-back.
            unsigned long long h_plan[2] = { 0, 0 };
            GRACE_TRY_HIP(hipMemcpyAsync(h_plan, n_wave_map, 16, hipMemcpyDeviceToHost, stream));
            GRACE_TRY_HIP(hipStreamSynchronize(stream));
            const bool stage = g_hits_stage_split && h_plan[1] / (unsigned long long)n_packets >= 200000ull;
            if (stage) both(M_(), T(), T(), a);
            else both(M_(), T(), F(), a);
        } else if (n_packets >= 4096) {
            both(M_(), F(), T(), a);
        } else {
            both(M_(), F(), F(), a);
        }
    } else if constexpr (MODE == MODE_COUNT) {
        if (split > 1) both(M_(), T(), F(), a);
        else if (lat_split) GRACE_TRY(one_or_split(F()));
        else both(M_(), F(), F(), a);
    } else {
        trace_kernel<MODE, false><<<grid, TRACE_BLOCK, 0, stream>>>(a);
    }
    GRACE_CHECK_LAUNCH();
#ifdef GRACE_STAMPS
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
#endif
    if (MODE == MODE_CUMULATIVE && split > 1) {
        combine_classes_kernel<<<ceil_div(n_rays, 256), 256, 0, stream>>>(a.partial, int(n_rays),
                                                                          split, a.split_dev, a.out_sums);
        GRACE_CHECK_LAUNCH();
    }
    if (g_timing) {
        GRACE_TRY_HIP(hipEventRecord(g_ev1, stream));
        g_ev_valid = true;
    }
    return GRACE_OK;
}

} // namespace

namespace grace_hip {
// Called by this library's entry points that WRITE caller arrays (sort payloads, tree builds):
// a prepared scene over that array is stale from here on.
grace_status rays_invalidate_if_written(const void* d_written)
{
    if (g_rays.valid && d_written && d_written == static_cast<const void*>(g_rays.rays)) return rays_release();
    return GRACE_OK;
}

grace_status scene_invalidate_if_written(const void* d_written)
{
    if (g_scene.valid && d_written
        && (d_written == g_scene.prims || d_written == g_scene.nodes || d_written == g_scene.leaves))
        return scene_release();
    return GRACE_OK;
}
} // namespace grace_hip

extern "C" {

grace_status grace_trace_hitcounts_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                      size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                      const int* d_leaves, const int* d_root,
                                      int* d_hit_counts, grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_hit_counts, "trace_hitcounts: null output");
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_spheres);
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    a.out_counts = d_hit_counts;
    return launch_trace<MODE_COUNT>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_hitcounts_keep_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                           size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                           const int* d_leaves, const int* d_root,
                                           int* d_hit_counts, grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;
    GRACE_REQUIRE(d_hit_counts, "trace_hitcounts: null output");
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_spheres);
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    a.out_counts = d_hit_counts;
    a.keep_chunks = true;
    return launch_trace<MODE_COUNT>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_cumulative_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                       size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                       const int* d_leaves, const int* d_root,
                                       float* d_cumulated, grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_cumulated, "trace_cumulative: null output");
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_spheres);
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    a.out_sums = d_cumulated;
    return launch_trace<MODE_CUMULATIVE>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_hits_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                 size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                 const int* d_leaves, const int* d_root,
                                 const int* d_ray_offsets, int* d_hit_indices,
                                 float* d_hit_integrals, float* d_hit_distances,
                                 grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_ray_offsets && d_hit_indices && d_hit_integrals && d_hit_distances,
                  "trace_hits: null output");
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_spheres);
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    a.offsets = d_ray_offsets;
    a.hit_idx = d_hit_indices;
    a.hit_integral = d_hit_integrals;
    a.hit_dist = d_hit_distances;
    return launch_trace<MODE_HITS>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_closest_tri(const void* d_rays, size_t n_rays, const float* d_tris,
                                     size_t n_tris, const int* d_nodes, size_t n_nodes,
                                     const int* d_leaves, const int* d_root, int* d_closest,
                                     grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_closest, "trace_closest_tri: null output");
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_tris); // 9 floats per triangle
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    a.out_counts = d_closest;
    return launch_trace<MODE_TRI>(a, n_rays, n_tris, n_nodes, as_stream(stream));
}

// ---- double4 spheres (Real4 = double4, Real = double) ----------------------------------------
static TraceArgs d4_args(const void* d_rays, const double* d_spheres, const int* d_nodes,
                         const int* d_leaves, const int* d_root)
{
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_spheres);   // (non-null check only)
    a.spheres_d = d_spheres;
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    return a;
}

grace_status grace_trace_hitcounts_d4(const void* d_rays, size_t n_rays, const double* d_spheres,
                                      size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                      const int* d_leaves, const int* d_root, int* d_hit_counts,
                                      grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_hit_counts, "trace_hitcounts (double4): null output");
    TraceArgs a = d4_args(d_rays, d_spheres, d_nodes, d_leaves, d_root);
    a.out_counts = d_hit_counts;
    return launch_trace<MODE_COUNT_D4>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_cumulative_d4(const void* d_rays, size_t n_rays, const double* d_spheres,
                                       size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                       const int* d_leaves, const int* d_root, double* d_sums,
                                       grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_sums, "trace_cumulative (double4): null output");
    TraceArgs a = d4_args(d_rays, d_spheres, d_nodes, d_leaves, d_root);
    a.out_sums_d = d_sums;
    return launch_trace<MODE_CUM_D4>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_hits_d4(const void* d_rays, size_t n_rays, const double* d_spheres,
                                 size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                 const int* d_leaves, const int* d_root, const int* d_ray_offsets,
                                 int* d_hit_indices, double* d_hit_integrals,
                                 double* d_hit_distances, grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_ray_offsets && d_hit_indices && d_hit_integrals && d_hit_distances,
                  "trace_hits (double4): null output");
    TraceArgs a = d4_args(d_rays, d_spheres, d_nodes, d_leaves, d_root);
    a.offsets = d_ray_offsets;
    a.hit_idx = d_hit_indices;
    a.hit_integral_d = d_hit_integrals;
    a.hit_dist_d = d_hit_distances;
    return launch_trace<MODE_HITS_D4>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_status_d4(grace_stream stream) { return grace_trace_status(stream); }

grace_status grace_trace_stats_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                  size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                  const int* d_leaves, const int* d_root,
                                  uint32_t* d_stats4, grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_stats4, "trace_stats: null output");
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_spheres);
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    a.stats = d_stats4;
    return launch_trace<MODE_STATS>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_hit_integrals_f32(const float* d_b2, const float* d_h, size_t n, float* d_out,
                                     grace_stream stream)
{
    GRACE_REQUIRE(n == 0 || (d_b2 && d_h && d_out), "hit_integrals: null pointer");
    if (n == 0) return GRACE_OK;
    hit_integrals_kernel<<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(d_b2, d_h, n, d_out);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

grace_status grace_trace_prepare_f4(const float* d_spheres, size_t n_spheres, const int* d_nodes,
                                    size_t n_nodes, const int* d_leaves, grace_stream stream)
{
    return scene_prepare(false, d_spheres, n_spheres, d_nodes, n_nodes, d_leaves, as_stream(stream));
}

grace_status grace_trace_prepare_tri(const float* d_tris, size_t n_tris, const int* d_nodes,
                                     size_t n_nodes, const int* d_leaves, grace_stream stream)
{
    return scene_prepare(true, d_tris, n_tris, d_nodes, n_nodes, d_leaves, as_stream(stream));
}

grace_status grace_trace_prepare_rays(const void* d_rays, size_t n_rays, grace_stream stream)
{
    return rays_prepare(static_cast<const float*>(d_rays), n_rays, as_stream(stream));
}

grace_status grace_trace_release_rays(void)
{
    return rays_release();
}

grace_status grace_trace_release(void)
{
    if (g_hits.chunk_counts) {
        GRACE_TRY_HIP(hipDeviceSynchronize());
        GRACE_TRY_HIP(hipFree(g_hits.chunk_counts));
    }
    g_hits = HitsCache();
    return scene_release();
}

grace_status grace_trace_enable_timing(int enabled)
{
    g_timing = enabled != 0;
    g_ev_valid = false;
    return GRACE_OK;
}

grace_status grace_trace_last_kernel_ms(float* h_ms)
{
    GRACE_REQUIRE(h_ms, "null output");
    GRACE_REQUIRE(g_timing && g_ev_valid, "no timed traversal launch recorded");
    GRACE_TRY_HIP(hipEventSynchronize(g_ev1));
    GRACE_TRY_HIP(hipEventElapsedTime(h_ms, g_ev0, g_ev1));
    return GRACE_OK;
}

grace_status grace_trace_last_lattice(int* h_lattice)
{
    GRACE_REQUIRE(h_lattice, "null output");
    *h_lattice = 0;
    if (g_last_lat_dev) {
        GRACE_TRY_HIP(hipStreamSynchronize(g_last_lat_stream));
        GRACE_TRY_HIP(hipMemcpy(h_lattice, g_last_lat_dev, sizeof(int), hipMemcpyDeviceToHost));
    }
    return GRACE_OK;
}

grace_status grace_trace_set_packet_split(int waves_per_packet)
{
    GRACE_REQUIRE(waves_per_packet == -1 || waves_per_packet == 1 || waves_per_packet == 2
                      || waves_per_packet == 4 || waves_per_packet == 8,
                  "packet split must be 1, 2, 4, 8 or -1 (automatic)");
    g_split = waves_per_packet;
    return GRACE_OK;
}

grace_status grace_trace_set_packet_width(int rays_per_packet)
{
    GRACE_REQUIRE(rays_per_packet == -1 || rays_per_packet == 16 || rays_per_packet == 32
                      || rays_per_packet == 64,
                  "packet width must be 16, 32, 64 or -1 (automatic)");
    g_width = rays_per_packet;
    return GRACE_OK;
}

grace_status grace_trace_set_exact_integrals(int enabled)
{
    g_exact_integrals = enabled != 0;
    return GRACE_OK;
}

grace_status grace_trace_set_treelet_size(int max_primitives)
{
    GRACE_REQUIRE(max_primitives >= -1, "treelet size must be >= 0 (or -1 for automatic)");
    g_treelet = max_primitives;
    return GRACE_OK;
}

grace_status grace_trace_set_ray_reorder(int enabled)
{
    g_ray_reorder = enabled != 0;
    return GRACE_OK;
}

grace_status grace_trace_status(grace_stream stream)
{
    if (!g_status) return GRACE_OK;
    int h = 0;
    GRACE_TRY_HIP(hipMemcpyAsync(&h, g_status, sizeof(int), hipMemcpyDeviceToHost,
                                 as_stream(stream)));
    GRACE_TRY_HIP(hipStreamSynchronize(as_stream(stream)));
    if (h != 0) {
        GRACE_TRY_HIP(hipMemsetAsync(g_status, 0, sizeof(int), as_stream(stream)));
        return set_error(GRACE_STACK_OVERFLOW, __FILE__, __LINE__,
                         "trace: packet stack (128 entries) exhausted");
    }
    return GRACE_OK;
}

} // extern "C"
