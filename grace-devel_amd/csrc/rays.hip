// Deterministic ray generators for gfx950 (inputs of the traversal).
//
// orthogonal_z : tests/helper/rays.cuh:55-79 (orthogonal_rays_z) ->
//                orthographic_projection_rays (include/grace/cuda/kernels/gen_rays.cuh:
//                319-360, 667-725) for the -z view; bit-identical to the reference's
//                arithmetic for this axis-aligned case (every cross term is exactly zero).
// healpix      : one source, HEALPix nested pixel centres, the role of
//                RayVectorGeneration/src/generateRays.c:57-59 (pix2vec_nest).
// isotropic    : uniform_random_rays (gen_rays.cuh:104-170, 401-490): Gaussian-normalised
//                directions sorted by ray_dir_morton_key (gen_rays.cuh:38-43).  The
//                reference's cuRAND XORWOW streams are device-specific by its own account
//                (gen_rays.cuh:21-24); here a counter-based generator (splitmix64 of
//                (seed, ray index)) makes the rays reproducible on any device.
// Streaming kernels: 28 B written per ray.
#include "common.hpp"

#include <cmath>

using namespace grace_hip;

namespace {

struct Ray7 { float dx, dy, dz, ox, oy, oz, length; };

__global__ __launch_bounds__(256) void ortho_z_kernel(int n_side, float cam_x, float cam_y,
                                                      float cam_z, float vx, float uy,
                                                      float length, float* __restrict__ rays)
{
    const int n = n_side * n_side;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int i = t % n_side, j = t / n_side;
        // image_plane_coord, gen_rays.cuh:76-97 (+0.5 = pixel centres), aspect ratio 1.
        const float x = (2 * ((i + 0.5f) / n_side) - 1) * 1.0f;
        const float y = 1 - 2 * ((j + 0.5f) / n_side);
        float* r = rays + 7 * size_t(t);
        r[0] = 0.f; r[1] = 0.f; r[2] = -1.f;
        r[3] = cam_x + x * vx;
        r[4] = cam_y + y * uy;
        r[5] = cam_z;
        r[6] = length;
    }
}

// HEALPix nested -> unit vector (Gorski et al. 2005, eqs. for the nested scheme).
__device__ __forceinline__ void pix2vec_nest(int nside, int ipix, double* v)
{
    const int jrll[12] = { 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4 };
    const int jpll[12] = { 1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7 };
    const double halfpi = 1.570796326794896619231321691639751442099;
    const int npface = nside * nside;
    const long long npix = 12ll * npface;
    const int face = ipix / npface;
    const int ipf = ipix % npface;
    int ix = 0, iy = 0;
    for (int b = 0; b < 15; ++b) {
        ix |= ((ipf >> (2 * b)) & 1) << b;
        iy |= ((ipf >> (2 * b + 1)) & 1) << b;
    }
    const int nl4 = 4 * nside;
    const int jr = jrll[face] * nside - ix - iy - 1;
    const double fact2 = 4.0 / npix, fact1 = (nside << 1) * fact2;
    int nr, kshift;
    double z;
    if (jr < nside) { nr = jr; z = 1.0 - double(nr) * nr * fact2; kshift = 0; }
    else if (jr > 3 * nside) { nr = nl4 - jr; z = double(nr) * nr * fact2 - 1.0; kshift = 0; }
    else { nr = nside; z = (2 * nside - jr) * fact1; kshift = (jr - nside) & 1; }
    int jp = (jpll[face] * nr + ix - iy + 1 + kshift) / 2;
    if (jp > nl4) jp -= nl4;
    if (jp < 1) jp += nl4;
    const double phi = (jp - (kshift + 1) * 0.5) * (halfpi / nr);
    const double st = sqrt((1.0 - z) * (1.0 + z));
    v[0] = st * cos(phi); v[1] = st * sin(phi); v[2] = z;
}

__global__ __launch_bounds__(256) void healpix_kernel(int nside, float ox, float oy, float oz,
                                                      float length, float* __restrict__ rays)
{
    const int n = 12 * nside * nside;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        double v[3];
        pix2vec_nest(nside, t, v);
        float* r = rays + 7 * size_t(t);
        r[0] = float(v[0]); r[1] = float(v[1]); r[2] = float(v[2]);
        r[3] = ox; r[4] = oy; r[5] = oz; r[6] = length;
    }
}

// perspective_projection_rays_kernel (include/grace/cuda/kernels/gen_rays.cuh:362-395)
__global__ __launch_bounds__(256) void pinhole_kernel(int res_x, int res_y, float aspect,
                                                      float cx, float cy, float cz, float vx,
                                                      float vy, float vz, float ux, float uy,
                                                      float uz, float nx, float ny, float nz,
                                                      float length, float* __restrict__ rays)
{
    const int n = res_x * res_y;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int i = t % res_x, j = t / res_x;
        const float x = (2 * ((i + 0.5f) / res_x) - 1) * aspect;
        const float y = 1 - 2 * ((j + 0.5f) / res_y);
        const float dx = x * vx + y * ux + 1.f * nx;
        const float dy = x * vy + y * uy + 1.f * ny;
        const float dz = x * vz + y * uz + 1.f * nz;
        const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        float* r = rays + 7 * size_t(t);
        r[0] = dx * inv; r[1] = dy * inv; r[2] = dz * inv;
        r[3] = cx; r[4] = cy; r[5] = cz; r[6] = length;
    }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ float u01(uint32_t bits) // (0, 1]
{
    return (float(bits >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ uint32_t space10(uint32_t x)
{
    x &= 1023u;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

// octant < 0: the whole sphere (gen_uniform_rays_kernel, kernels/gen_rays.cuh:125-159);
// octant 0..7: sign * |normal| per component, bit 2 = x, bit 1 = y, bit 0 = z, set = positive
// (gen_uniform_rays_single_octant_kernel, :161-204; enum Octants, grace/types.h:36-45).
__global__ __launch_bounds__(256) void isotropic_kernel(size_t n, float ox, float oy, float oz,
                                                        float length, uint64_t seed, int octant,
                                                        float* __restrict__ rays,
                                                        uint32_t* __restrict__ keys)
{
    for (size_t t = blockIdx.x * size_t(blockDim.x) + threadIdx.x; t < n;
         t += size_t(gridDim.x) * blockDim.x) {
        const uint64_t a = splitmix64(seed ^ splitmix64(2 * t));
        const uint64_t b = splitmix64(seed ^ splitmix64(2 * t + 1));
        // Three standard normals by Box-Muller, then normalise (gen_rays.cuh:127-147).
        const float r1 = sqrtf(-2.0f * logf(u01(uint32_t(a))));
        const float r2 = sqrtf(-2.0f * logf(u01(uint32_t(b))));
        const float t1 = 6.283185307179586f * u01(uint32_t(a >> 32));
        const float t2 = 6.283185307179586f * u01(uint32_t(b >> 32));
        float gx = r1 * cosf(t1), gy = r1 * sinf(t1), gz = r2 * cosf(t2);
        if (octant >= 0) {
            gx = ((octant & 4) ? 1 : -1) * fabsf(gx);
            gy = ((octant & 2) ? 1 : -1) * fabsf(gy);
            gz = ((octant & 1) ? 1 : -1) * fabsf(gz);
        }
        float norm2 = gx * gx + gy * gy + gz * gz;
        if (!(norm2 > 0.f)) { gx = 1.f; gy = 0.f; gz = 0.f; norm2 = 1.f; }
        const float inv = 1.0f / sqrtf(norm2);
        const float dx = gx * inv, dy = gy * inv, dz = gz * inv;
        float* r = rays + 7 * t;
        r[0] = dx; r[1] = dy; r[2] = dz; r[3] = ox; r[4] = oy; r[5] = oz; r[6] = length;
        // ray_dir_morton_key: morton_key((d + 1) / 2) with span 1023 (generic/morton.h:32-42)
        const uint32_t kx = uint32_t(1023u * ((dx + 1) / 2.f));
        const uint32_t ky = uint32_t(1023u * ((dy + 1) / 2.f));
        const uint32_t kz = uint32_t(1023u * ((dz + 1) / 2.f));
        keys[t] = space10(kz) << 2 | space10(ky) << 1 | space10(kx);
    }
}

__device__ __forceinline__ uint32_t ray_dir_key(float dx, float dy, float dz)
{
    // ray_dir_morton_key: morton_key((d + 1) / 2) with span 1023 (gen_rays.cuh:38-43)
    const uint32_t kx = uint32_t(1023u * ((dx + 1) / 2.f));
    const uint32_t ky = uint32_t(1023u * ((dy + 1) / 2.f));
    const uint32_t kz = uint32_t(1023u * ((dz + 1) / 2.f));
    return space10(kz) << 2 | space10(ky) << 1 | space10(kx);
}

// one_to_many_rays_kernel (kernels/gen_rays.cuh:206-243): direction = (point - origin)
// normalised, length = |point - origin| as float(1.0 / double(1/|.|)).
template <typename Elem>
__global__ __launch_bounds__(256) void one_to_many_kernel(size_t n, float ox, float oy, float oz,
                                                          const Elem* __restrict__ pts, int stride,
                                                          float* __restrict__ rays,
                                                          uint32_t* __restrict__ dir_keys)
{
    for (size_t t = blockIdx.x * size_t(blockDim.x) + threadIdx.x; t < n;
         t += size_t(gridDim.x) * blockDim.x) {
        const Elem* q = pts + t * size_t(stride);
        const float dx = float(q[0] - ox), dy = float(q[1] - oy), dz = float(q[2] - oz);
        const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
        float* r = rays + 7 * t;
        r[0] = ux; r[1] = uy; r[2] = uz; r[3] = ox; r[4] = oy; r[5] = oz;
        r[6] = float(1.0 / double(inv));
        if (dir_keys) dir_keys[t] = ray_dir_key(ux, uy, uz);
    }
}

// plane_parallel_random_rays_kernel (kernels/gen_rays.cuh:245-317): one ray per grid cell, its
// origin at a random point of the cell, O = base + W dw + H dh; own counter-based generator
// (uniforms in (0, 1] like curand_uniform).
__global__ __launch_bounds__(256) void plane_parallel_kernel(int width, size_t n, float bx, float by,
                                                             float bz, float dwx, float dwy,
                                                             float dwz, float dhx, float dhy,
                                                             float dhz, float length, float nx,
                                                             float ny, float nz, uint64_t seed,
                                                             float* __restrict__ rays)
{
    for (size_t t = blockIdx.x * size_t(blockDim.x) + threadIdx.x; t < n;
         t += size_t(gridDim.x) * blockDim.x) {
        const int i = int(t % size_t(width)), j = int(t / size_t(width));
        const uint64_t a = splitmix64(seed ^ splitmix64(t));
        const float rw = u01(uint32_t(a)), rh = u01(uint32_t(a >> 32));
        // zero_one_to_a_b(f, a, b) = f * (b - a) + a  (gen_rays.cuh:66-74)
        const float awx = i * dwx, awy = i * dwy, awz = i * dwz;
        const float bwx = (i + 1) * dwx, bwy = (i + 1) * dwy, bwz = (i + 1) * dwz;
        const float ahx = j * dhx, ahy = j * dhy, ahz = j * dhz;
        const float bhx = (j + 1) * dhx, bhy = (j + 1) * dhy, bhz = (j + 1) * dhz;
        const float wx = rw * (bwx - awx) + awx, wy = rw * (bwy - awy) + awy,
                    wz = rw * (bwz - awz) + awz;
        const float hx = rh * (bhx - ahx) + ahx, hy = rh * (bhy - ahy) + ahy,
                    hz = rh * (bhz - ahz) + ahz;
        float* r = rays + 7 * t;
        r[0] = nx; r[1] = ny; r[2] = nz;
        r[3] = bx + wx + hx; r[4] = by + wy + hy; r[5] = bz + wz + hz;
        r[6] = length;
    }
}

// orthographic_projection_rays_kernel (kernels/gen_rays.cuh:319-360) with image_plane_coord
// (:76-95), aspect 1 and n = 0 rolled in.
__global__ __launch_bounds__(256) void ortho_projection_kernel(int res_x, int res_y, float cx,
                                                               float cy, float cz, float dx,
                                                               float dy, float dz, float vx,
                                                               float vy, float vz, float ux,
                                                               float uy, float uz, float length,
                                                               float* __restrict__ rays)
{
    const size_t n = size_t(res_x) * res_y;
    for (size_t t = blockIdx.x * size_t(blockDim.x) + threadIdx.x; t < n;
         t += size_t(gridDim.x) * blockDim.x) {
        const int i = int(t % size_t(res_x)), j = int(t / size_t(res_x));
        const float x = (2 * ((i + 0.5f) / res_x) - 1) * 1.f;
        const float y = 1 - 2 * ((j + 0.5f) / res_y);
        const float z = 1.f;
        const float px = x * vx + y * ux + z * 0.f;
        const float py = x * vy + y * uy + z * 0.f;
        const float pz = x * vz + y * uz + z * 0.f;
        float* r = rays + 7 * t;
        r[0] = dx; r[1] = dy; r[2] = dz;
        r[3] = cx + px; r[4] = cy + py; r[5] = cz + pz;
        r[6] = length;
    }
}

// normalize3 / cross on the host (generic/vecmath.h:9-52): products in float, the norm in
// double, each component narrowed back to float.
inline void host_cross(const float* u, const float* v, float* out)
{
    out[0] = u[1] * v[2] - u[2] * v[1];
    out[1] = u[2] * v[0] - u[0] * v[2];
    out[2] = u[0] * v[1] - u[1] * v[0];
}
inline void host_normalize(float* v)
{
    const double N = 1. / std::sqrt(double(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
    for (int k = 0; k < 3; ++k) v[k] = float(v[k] * N);
}

} // namespace

extern "C" {

grace_status grace_rays_orthogonal_z(int n_side, const float* h_mins4, const float* h_maxs4,
                                     void* d_rays, float* h_area, grace_stream stream)
{
    GRACE_REQUIRE(n_side > 0 && h_mins4 && h_maxs4 && d_rays, "orthogonal_rays_z: bad argument");
    GRACE_TRY(rays_invalidate_if_written(d_rays));   // a prepared ray batch over this array is stale
    GRACE_REQUIRE(size_t(n_side) * n_side < (size_t(1) << 31), "orthogonal_rays_z: too many rays");
    // box_center / box_span, tests/helper/rays.cuh:11-29 (float sums, double halving).
    const float cx = float((h_mins4[0] + h_maxs4[0]) / 2.);
    const float cy = float((h_mins4[1] + h_maxs4[1]) / 2.);
    float sx = h_maxs4[0] - h_mins4[0] + 2 * h_maxs4[3];
    float sy = h_maxs4[1] - h_mins4[1] + 2 * h_maxs4[3];
    const float sz = h_maxs4[2] - h_mins4[2] + 2 * h_maxs4[3];
    if (sx > sy) sy = sx; else if (sy > sx) sx = sy;
    if (h_area) *h_area = (sx / n_side) * (sy / n_side);
    // v = (1,0,0) * horizontal_extent / 2, u = (0,1,0) * vertical_extent / 2
    // (gen_rays.cuh:696-706); camera at (cx, cy, span.z), length 2 * span.z.
    const float vx = float(1.f * (sy / 2.));
    const float uy = float(1.f * (sy / 2.));
    const size_t n = size_t(n_side) * n_side;
    ortho_z_kernel<<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(
        n_side, cx, cy, sz, vx, uy, 2 * sz, static_cast<float*>(d_rays));
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

grace_status grace_rays_healpix(int nside, float ox, float oy, float oz, float length,
                                void* d_rays, grace_stream stream)
{
    GRACE_REQUIRE(nside >= 1 && nside <= 8192 && (nside & (nside - 1)) == 0 && d_rays,
                  "healpix: nside must be a power of two in [1, 8192]");
    GRACE_TRY(rays_invalidate_if_written(d_rays));   // a prepared ray batch over this array is stale
    const size_t n = 12 * size_t(nside) * nside;
    healpix_kernel<<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(
        nside, ox, oy, oz, length, static_cast<float*>(d_rays));
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

grace_status grace_rays_pinhole(int res_x, int res_y, const float* h_camera, const float* h_look_at,
                                const float* h_view_up, float fovy, float length, void* d_rays,
                                grace_stream stream)
{
    GRACE_REQUIRE(res_x > 0 && res_y > 0 && h_camera && h_look_at && h_view_up && d_rays,
                  "pinhole rays: bad argument");
    GRACE_TRY(rays_invalidate_if_written(d_rays));   // a prepared ray batch over this array is stale
    // pinhole_camera_rays, gen_rays.cuh:727-789 (Real = float; normalize3 in fp64 on the host)
    const float* c = h_camera;
    float vd[3] = { h_look_at[0] - c[0], h_look_at[1] - c[1], h_look_at[2] - c[2] };
    const float* up = h_view_up;
    float c1[3] = { vd[1] * up[2] - vd[2] * up[1], vd[2] * up[0] - vd[0] * up[2],
                    vd[0] * up[1] - vd[1] * up[0] };
    double N = 1. / std::sqrt(double(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]));
    float v[3] = { float(c1[0] * N), float(c1[1] * N), float(c1[2] * N) };
    float c2[3] = { v[1] * vd[2] - v[2] * vd[1], v[2] * vd[0] - v[0] * vd[2],
                    v[0] * vd[1] - v[1] * vd[0] };
    N = 1. / std::sqrt(double(c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2]));
    float u[3] = { float(c2[0] * N), float(c2[1] * N), float(c2[2] * N) };
    N = 1. / std::sqrt(double(vd[0] * vd[0] + vd[1] * vd[1] + vd[2] * vd[2]));
    float nn[3] = { float(vd[0] * N), float(vd[1] * N), float(vd[2] * N) };
    const float pre = float(1. / std::tan(fovy / 2.));
    for (int k = 0; k < 3; ++k) nn[k] *= pre;
    const float aspect = float(res_x) / res_y;
    const size_t n = size_t(res_x) * res_y;
    pinhole_kernel<<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(
        res_x, res_y, aspect, c[0], c[1], c[2], v[0], v[1], v[2], u[0], u[1], u[2], nn[0], nn[1],
        nn[2], length, static_cast<float*>(d_rays));
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

static grace_status isotropic_rays(size_t n_rays, float ox, float oy, float oz, float length,
                                   uint64_t seed, int octant, void* d_rays, grace_stream stream)
{
    GRACE_REQUIRE(n_rays > 0 && d_rays, "isotropic rays: bad argument");
    GRACE_TRY(rays_invalidate_if_written(d_rays));   // a prepared ray batch over this array is stale
    hipStream_t st = as_stream(stream);
    FrameGuard frame;
    GRACE_TRY(frame.begin(Workspace::aligned(n_rays * 4) + sort_ws_bytes(n_rays, 4, 28), st));
    uint32_t* keys = Workspace::take<uint32_t>(n_rays);
    isotropic_kernel<<<stream_grid(n_rays, 256), 256, 0, st>>>(
        n_rays, ox, oy, oz, length, seed, octant, static_cast<float*>(d_rays), keys);
    GRACE_CHECK_LAUNCH();
    return sort_pairs_u32_nested(keys, d_rays, n_rays, 28, 0, 30, nullptr, st);
}

grace_status grace_rays_isotropic(size_t n_rays, float ox, float oy, float oz, float length,
                                  uint64_t seed, void* d_rays, grace_stream stream)
{
    return isotropic_rays(n_rays, ox, oy, oz, length, seed, -1, d_rays, stream);
}

grace_status grace_rays_isotropic_octant(size_t n_rays, float ox, float oy, float oz, float length,
                                         int octant, uint64_t seed, void* d_rays,
                                         grace_stream stream)
{
    GRACE_REQUIRE(octant >= 0 && octant <= 7, "single-octant rays: octant must be 0 (MMM) .. 7 (PPP)");
    return isotropic_rays(n_rays, ox, oy, oz, length, seed, octant, d_rays, stream);
}

grace_status grace_rays_one_to_many(size_t n_rays, float ox, float oy, float oz,
                                    const void* d_points, int is_double, int elems_per_point,
                                    int sort_type, const float* h_bot, const float* h_top,
                                    void* d_rays, grace_stream stream)
{
    GRACE_REQUIRE(n_rays > 0 && d_points && d_rays, "one_to_many_rays: bad argument");
    GRACE_TRY(rays_invalidate_if_written(d_rays));   // a prepared ray batch over this array is stale
    GRACE_REQUIRE(elems_per_point >= 3 && elems_per_point <= 16,
                  "one_to_many_rays: elements per point must be 3..16");
    // gen_rays.cuh:126-131: an unknown sort type throws std::invalid_argument
    GRACE_REQUIRE(sort_type >= 0 && sort_type <= 2, "Ray sort type not recognized");
    GRACE_REQUIRE(sort_type != 2 || (h_bot && h_top), "one_to_many_rays: end-point sort needs the points' bounds");
    hipStream_t st = as_stream(stream);
    uint32_t* keys = nullptr;
    FrameGuard frame;
    if (sort_type != 0) {
        GRACE_TRY(frame.begin(Workspace::aligned(n_rays * 4) + sort_ws_bytes(n_rays, 4, 28), st));
        keys = Workspace::take<uint32_t>(n_rays);
    }
    uint32_t* dir_keys = sort_type == 1 ? keys : nullptr;
    if (is_double)
        one_to_many_kernel<double><<<stream_grid(n_rays, 256), 256, 0, st>>>(
            n_rays, ox, oy, oz, static_cast<const double*>(d_points), elems_per_point,
            static_cast<float*>(d_rays), dir_keys);
    else
        one_to_many_kernel<float><<<stream_grid(n_rays, 256), 256, 0, st>>>(
            n_rays, ox, oy, oz, static_cast<const float*>(d_points), elems_per_point,
            static_cast<float*>(d_rays), dir_keys);
    GRACE_CHECK_LAUNCH();
    if (sort_type == 0) return GRACE_OK;
    if (sort_type == 2)   // morton_keys(points, bot, top, CentroidSphere), gen_rays.cuh:603-604
        GRACE_TRY(grace_morton_keys30_points(d_points, n_rays, is_double, elems_per_point, h_bot,
                                             h_top, keys, stream));
    return sort_pairs_u32_nested(keys, d_rays, n_rays, 28, 0, 30, nullptr, st);
}

grace_status grace_rays_plane_parallel_random(int width, int height, const float* h_base,
                                              const float* h_w, const float* h_h, float length,
                                              uint64_t seed, void* d_rays, grace_stream stream)
{
    GRACE_REQUIRE(width > 0 && height > 0 && h_base && h_w && h_h && d_rays,
                  "plane_parallel_random_rays: bad argument");
    GRACE_TRY(rays_invalidate_if_written(d_rays));   // a prepared ray batch over this array is stale
    const size_t n = size_t(width) * height;
    GRACE_REQUIRE(n < (size_t(1) << 31), "plane_parallel_random_rays: too many rays");
    // gen_rays.cuh:628-642 (Real3 = float3)
    const float dw[3] = { h_w[0] / width, h_w[1] / width, h_w[2] / width };
    const float dh[3] = { h_h[0] / height, h_h[1] / height, h_h[2] / height };
    float dir[3];
    host_cross(h_w, h_h, dir);
    host_normalize(dir);
    plane_parallel_kernel<<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(
        width, n, h_base[0], h_base[1], h_base[2], dw[0], dw[1], dw[2], dh[0], dh[1], dh[2],
        length, dir[0], dir[1], dir[2], seed, static_cast<float*>(d_rays));
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

grace_status grace_rays_orthographic_projection(int res_x, int res_y, const float* h_camera,
                                                const float* h_look_at, const float* h_view_up,
                                                float vertical_extent, float length,
                                                void* d_rays, grace_stream stream)
{
    GRACE_REQUIRE(res_x > 0 && res_y > 0 && h_camera && h_look_at && h_view_up && d_rays,
                  "orthographic_projection_rays: bad argument");
    GRACE_TRY(rays_invalidate_if_written(d_rays));   // a prepared ray batch over this array is stale
    // gen_rays.cuh:667-725 (Real = float)
    const float aspect = float(res_x) / res_y;
    const float horizontal_extent = vertical_extent * aspect;
    float vd[3] = { h_look_at[0] - h_camera[0], h_look_at[1] - h_camera[1],
                    h_look_at[2] - h_camera[2] };
    host_normalize(vd);
    float v[3], u[3];
    host_cross(vd, h_view_up, v);
    host_normalize(v);
    host_cross(v, vd, u);
    host_normalize(u);
    for (int k = 0; k < 3; ++k) {
        v[k] = float(v[k] * (horizontal_extent / 2.));
        u[k] = float(u[k] * (vertical_extent / 2.));
    }
    const size_t n = size_t(res_x) * res_y;
    ortho_projection_kernel<<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(
        res_x, res_y, h_camera[0], h_camera[1], h_camera[2], vd[0], vd[1], vd[2], v[0], v[1], v[2],
        u[0], u[1], u[2], length, static_cast<float*>(d_rays));
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

} // extern "C"
