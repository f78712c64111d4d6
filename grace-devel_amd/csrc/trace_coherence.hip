// Ray coherence order of the traversal, for gfx950: packets are 64 consecutive rays of a
// space-filling-curve order over the ray co-ordinates that vary (extents -> 15-bit keys -> partial
// stable radix sort), plus the device-side choices a trace launch makes from the batch's extents
// (origin-lattice instantiation, waves per packet).  Small streaming passes over the rays (28 B
// each); results of a trace never depend on the order.
#include "trace_state.hpp"

using namespace grace_hip;

namespace {

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

__global__ __launch_bounds__(1024) void ray_extents_kernel(const float* __restrict__ rays, int n,
                                                          uint32_t* __restrict__ ext12,
                                                          const uint32_t* __restrict__ run_if)
{
    if (run_if && *run_if == 0u) return;
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
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    __shared__ float s_lo[16][6], s_hi[16][6], s_len[16];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { s_lo[wave][k] = lo[k]; s_hi[wave][k] = hi[k]; }
        s_len[wave] = len_hi;
    }
    __syncthreads();
    // (13 atomics per workgroup, all on one cache line: they serialise at ~7 ns each, so the
    // launch is kept to 128 large workgroups -- 1024 workgroups of 256 measured 76 us, 256: 28 us)
    const int n_waves = int(blockDim.x >> 6);
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float l = s_lo[0][k], h = s_hi[0][k];
        for (int w = 1; w < n_waves; ++w) { l = fminf(l, s_lo[w][k]); h = fmaxf(h, s_hi[w][k]); }
        atomicMin(&ext12[k], f2ord_u(l));
        atomicMax(&ext12[6 + k], f2ord_u(h));
    } else if (threadIdx.x == 6) {
        float l = s_len[0];
        for (int w = 1; w < n_waves; ++w) l = fmaxf(l, s_len[w]);
        if (l > -INFINITY) atomicMax(&ext12[15], f2ord_u(l));
    }
}

// How many of the launched waves per packet should work.  The host sizes the launch for an
// incoherent batch (whose packets are heavy: >= 16384 waves in flight pay off); a batch whose
// rays all share one direction (orthographic shards) has light packets, for which every extra
// wave repeats the upper-tree walk and the cluster tests.  ext12 = the ray
// extents of the coherence pass (order-preserving uints: minima then maxima of d, o).
__device__ int choose_split(const uint32_t* __restrict__ ext12, int n_packets, int launched_arg, bool lattice)
{
    const bool one_direction = ext12[0] == ext12[6] && ext12[1] == ext12[7] && ext12[2] == ext12[8];
    const int launched = launched_arg & 0xFF;
    const int budget = (launched_arg & SPLIT_WIDE_BUDGET) ? 32768 : 16384;
    int k = launched;
    // (a scene with spheres smaller than the ray spacing has packets of very unequal weight: it
    // keeps every launched wave -- see lat_split in launch_trace)
    if (one_direction && !lattice) {
        // Measured on 1/8 ... 1/1 shards of the 1024^2 frame (2048 ... 16384 packets): best K =
        // 4, 2, 2, 1.  The split kernels run 8 waves per SIMD: 8192 waves fill the chip once;
        // from 6144 packets on a second wave per packet still pays (16384 waves).
        // Round 3, after the walk was replaced by the flat group passes (the duplicated part of
        // a split packet got cheaper): column densities are faster with two waves per packet up
        // to the full frame -- 16384 packets: 2.65 ms at K = 1, 2.56 at K = 2, 2.75 at K = 4;
        // hit counts are not (1.83 / 1.87 / 2.17): their budget stays at 16384 waves.
        k = 1;
        while (k < launched && n_packets * k < 8192) k *= 2;
        if (k < launched && n_packets >= 6144 && n_packets * k * 2 <= budget) k *= 2;
    } else if (!lattice) {
        // incoherent batches: the smallest K that puts 16384 waves in flight (the launch may be
        // sized for more: column densities, see launch_trace)
        k = 1;
        while (k < launched && n_packets * k < 16384) k *= 2;
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
                                                          uint32_t* __restrict__ flag,
                                                          const uint32_t* __restrict__ run_if)
{
    if (run_if && *run_if == 0u) return;
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
                                                       const uint32_t* __restrict__ not_grid,
                                                       const uint32_t* __restrict__ run_if)
{
    if (run_if && *run_if == 0u) return;
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

} // namespace

namespace grace_hip {

grace_status launch_choose_variants(const uint32_t* ext12, int n, const float4* scene_min, uint32_t* lat_flag,
                                    int split_packets, int split_launched, int* split_dev, hipStream_t stream)
{
    choose_variants_kernel<<<1, 1, 0, stream>>>(ext12, n, scene_min, lat_flag, split_packets, split_launched,
                                                split_dev);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

static grace_status rays_free_buffers(RayOrder& ro)
{
    if (ro.perm || ro.ext || ro.ctl) GRACE_TRY_HIP(hipDeviceSynchronize());
    if (ro.perm) GRACE_TRY_HIP(hipFree(ro.perm));
    if (ro.ext) GRACE_TRY_HIP(hipFree(ro.ext));
    if (ro.ctl) GRACE_TRY_HIP(hipFree(ro.ctl));
    ro.perm = nullptr; ro.ext = nullptr; ro.ctl = nullptr;
    ro.valid = false;
    ro.pinned = false;
    return GRACE_OK;
}

grace_status rays_release(TraceState& ts)
{
    GRACE_TRY(rays_free_buffers(ts.rays));
    ts.rays = RayOrder();
    return GRACE_OK;
}

grace_status rays_cache_alloc(TraceState& ts, const RayKey& key)
{
    RayOrder& ro = ts.rays;
    GRACE_TRY(rays_free_buffers(ro));
    if (hipMalloc(reinterpret_cast<void**>(&ro.perm), key.n * 4) != hipSuccess
        || hipMalloc(reinterpret_cast<void**>(&ro.ext), 64) != hipSuccess
        || hipMalloc(reinterpret_cast<void**>(&ro.ctl), sizeof(CacheCtl)) != hipSuccess) {
        (void)hipGetLastError();
        (void)rays_free_buffers(ro);
        return set_error(GRACE_OUT_OF_MEMORY, __FILE__, __LINE__, "ray order cache: out of device memory");
    }
    ro.key = key;
    return GRACE_OK;
}

grace_status ray_order(const float* d_rays, size_t n_rays, uint32_t* ext, uint32_t* keys, uint32_t* perm,
                       const float4* scene_min, uint32_t* lat_flag, int n_packets, int split, int* split_dev,
                       hipStream_t stream, const uint32_t* run_if)
{
    // minima at the top of the order, maxima / per-call choices / grid flag at zero: one tiny launch
    // (a gated pass finds its extents initialised by the signature check)
    if (!run_if) {
        ray_ext_init_kernel<<<1, 64, 0, stream>>>(ext);
        GRACE_CHECK_LAUNCH();
    }
    const int grid_small = stream_grid(n_rays, 256, 4) < 1024 ? stream_grid(n_rays, 256, 4) : 1024;
    const int grid_ext = stream_grid(n_rays, 1024, 4) < 128 ? stream_grid(n_rays, 1024, 4) : 128;
    ray_extents_kernel<<<grid_ext, 1024, 0, stream>>>(d_rays, int(n_rays), ext, run_if);
    GRACE_CHECK_LAUNCH();
    ray_lattice_kernel<<<grid_small, 256, 0, stream>>>(d_rays, int(n_rays), ext, ext + 14, run_if);
    GRACE_CHECK_LAUNCH();
    ray_keys_kernel<<<stream_grid(n_rays, 256), 256, 0, stream>>>(d_rays, int(n_rays), ext, keys, scene_min,
                                                                 lat_flag, n_packets, split, split_dev, ext + 14,
                                                                 run_if);
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
    return sort_pairs_u32_nested(keys, nullptr, n_rays, 0, begin_bit, 30, perm, stream, run_if);
}

// grace_trace_prepare_rays: the cache filled NOW (and kept until released or replaced by another
// prepare), instead of at the second call on the same batch.  The reference leaves ray ordering to
// the caller (its generators sort at generation time, gen_rays.cuh:483,520,577,615).
grace_status rays_prepare(TraceState& ts, const float* d_rays, size_t n_rays, hipStream_t stream)
{
    GRACE_REQUIRE(d_rays || n_rays == 0, "trace_prepare_rays: null pointer");
    GRACE_REQUIRE(n_rays < (size_t(1) << 31), "trace_prepare_rays: bad ray count");
    GRACE_TRY(rays_release(ts));
    if (n_rays <= 64) return GRACE_OK;          // one packet: nothing to order
    RayKey key;
    key.rays = d_rays; key.n = n_rays;
    GRACE_TRY(rays_cache_alloc(ts, key));
    RayOrder& ro = ts.rays;
    FrameGuard frame;
    grace_status st = frame.begin(Workspace::aligned(n_rays * 4) + sort_ws_bytes(n_rays, 4, 0)
                                      + Workspace::aligned(sig_partial_words() * 8) + 1024, stream);
    if (st == GRACE_OK) {
        unsigned long long* partial = Workspace::take<unsigned long long>(sig_partial_words());
        SigRequest rq;
        rq.rays = d_rays; rq.rays_bytes = n_rays * 28;
        rq.rays_ctl = ro.ctl; rq.rays_force = true; rq.rays_ext = ro.ext;
        st = launch_signatures(rq, partial, stream);
    }
    if (st == GRACE_OK) {
        uint32_t* keys = Workspace::take<uint32_t>(n_rays);
        st = ray_order(d_rays, n_rays, ro.ext, keys, ro.perm, nullptr, nullptr, 0, 0, nullptr, stream, &ro.ctl->stale);
    }
    if (st != GRACE_OK) { (void)rays_release(ts); return st; }
    ro.valid = true;
    ro.pinned = true;
    ts.rays.seen = key;
    return GRACE_OK;
}

} // namespace grace_hip
