// A caller-defined primitive through the GENERIC forms of the drop-in headers, shaped like the
// reference's alternate-primitive programs (tests/profile_trace_triangle/tris_tree.cuh:17-30:
// morton_keys with a centroid functor -> sort_by_key -> compute_deltas(DeltaXOR) -> build_ALBVH
// with an AABB functor; tris_trace.cu:43-62: trace_texref with the caller's intersect / on-hit /
// ray-entry functors and RayExit_to_array).  Authored here: the primitive, its three functors and
// the ray payload are this file's own; nothing below names a libgrace_hip.so triangle entry point.
// tests/test_gpu_dropin.py compares the sorted primitives, the tree and the closest hits with the
// library's built-in triangle path (grace_*_tri) bit for bit.
//
//   dropin_triangles <tris.f32> <rays.f32> <max_per_leaf> <out prefix> [greater]
//     tris.f32: N x 9 floats {v, e1, e2};  rays.f32: M x 7 floats (grace::Ray), M % 32 == 0
//     with "greater": deltas are bit-flipped and the tree built with thrust::greater -- the same
//     tree, through the other comparator.
#include "grace/cuda/functors/trace.cuh"
#include "grace/cuda/kernels/albvh.cuh"
#include "grace/cuda/kernels/bintree_trace.cuh"
#include "grace/cuda/kernels/morton.cuh"
#include "grace/cuda/nodes.h"
#include "grace/cuda/util/bound_iter.cuh"
#include "grace/generic/functors/albvh.h"
#include "grace/ray.h"
#include "grace/types.h"

#include <thrust/device_vector.h>
#include <thrust/host_vector.h>
#ifdef USE_THRUST_SORT
#include <thrust/sort.h>
#endif

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

// One vertex and the two edges leaving it.
struct Tri
{
    float3 v, e1, e2;
};

// Box of the three corners; a face that is flat along an axis gets a sliver of thickness there.
struct TriBox
{
    __host__ __device__ void operator()(const Tri& t, float3* bot, float3* top) const
    {
        const float c0[3] = { t.v.x, t.v.y, t.v.z };
        const float c1[3] = { t.v.x + t.e1.x, t.v.y + t.e1.y, t.v.z + t.e1.z };
        const float c2[3] = { t.v.x + t.e2.x, t.v.y + t.e2.y, t.v.z + t.e2.z };
        float lo[3], hi[3];
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(c0[k], fminf(c1[k], c2[k]));
            hi[k] = fmaxf(c0[k], fmaxf(c1[k], c2[k]));
            if (lo[k] == hi[k]) {
                const float pad = 0.000001f * fabsf(lo[k]);
                lo[k] -= pad;
                hi[k] += pad;
            }
        }
        *bot = make_float3(lo[0], lo[1], lo[2]);
        *top = make_float3(hi[0], hi[1], hi[2]);
    }
};

// Mean of the corners: v + (e1 + e2) / 3, the third taken in double and the product in float.
struct TriCentre
{
    __host__ __device__ float3 operator()(const Tri& t) const
    {
        const float third = float(1. / 3.);
        return make_float3(t.v.x + third * (t.e1.x + t.e2.x), t.v.y + third * (t.e1.y + t.e2.y),
                           t.v.z + third * (t.e1.z + t.e2.z));
    }
};

struct Nearest
{
    int data;        // index of the nearest face so far (RayExit_to_array copies .data out)
    float t_min;
};

// Products of float components accumulated in double, results narrowed where they are stored in
// float -- the arithmetic of the reference's vector helpers (tests/helper/vector_math.cu:27-52).
__device__ inline double dot3(const float3 a, const float3 b)
{
    return ((double)a.x * b.x + (double)a.y * b.y) + (double)a.z * b.z;
}
__device__ inline float3 cross3(const float3 a, const float3 b)
{
    return make_float3(float((double)a.y * b.z - (double)a.z * b.y),
                       float((double)a.z * b.x - (double)a.x * b.z),
                       float((double)a.x * b.y - (double)a.y * b.x));
}

// Moeller-Trumbore, front faces only; accepts a hit no farther than the nearest one so far.
struct HitTri
{
    __device__ bool operator()(const grace::Ray& ray, const Tri& tri, Nearest& near, const int,
                               const grace::gpu::BoundIter<char>) const
    {
        const float3 d = make_float3(ray.dx, ray.dy, ray.dz);
        const float3 p = cross3(d, tri.e2);
        const float det = float(dot3(tri.e1, p));
        if (det < 1E-14f) return false;
        const float inv_det = float(1. / det);
        const float3 ov = make_float3(ray.ox - tri.v.x, ray.oy - tri.v.y, ray.oz - tri.v.z);
        const float u = float(dot3(ov, p) * inv_det);
        if (u < 0.f || u > 1.f) return false;
        const float3 q = cross3(ov, tri.e1);
        const float v = float(dot3(d, q) * inv_det);
        if (v < 0.f || u + v > 1.f) return false;
        const float t = float(dot3(tri.e2, q) * inv_det);
        if (t <= near.t_min && t >= 1E-14f) {
            near.t_min = t;
            return true;
        }
        return false;
    }
};

struct KeepTri
{
    __device__ void operator()(const int, const grace::Ray&, Nearest& near, const int tri_idx,
                               const Tri&, const int, const grace::gpu::BoundIter<char>) const
    {
        near.data = tri_idx;
    }
};

struct StartRay
{
    __device__ void operator()(const int, const grace::Ray& ray, Nearest& near,
                               const grace::gpu::BoundIter<char>) const
    {
        near.data = -1;
        near.t_min = ray.length * (1.f + 0.000001f);
    }
};

__global__ void flip_bits(grace::uinteger32* d, size_t n)
{
    const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (i < n) d[i] = ~d[i];
}

template <typename T>
static std::vector<T> read_file(const char* name)
{
    std::FILE* f = std::fopen(name, "rb");
    if (!f) { std::perror(name); std::exit(2); }
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<T> v(size_t(bytes) / sizeof(T));
    if (std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) std::exit(2);
    std::fclose(f);
    return v;
}

template <typename T>
static void write_file(const std::string& name, const T* data, size_t n)
{
    std::FILE* f = std::fopen(name.c_str(), "wb");
    if (!f || std::fwrite(data, sizeof(T), n, f) != n) { std::perror(name.c_str()); std::exit(2); }
    std::fclose(f);
}

int main(int argc, char* argv[])
{
    if (argc < 5) return 2;
    static_assert(sizeof(Tri) == 36, "three float3");
    const std::vector<Tri> h_tris_in = read_file<Tri>(argv[1]);
    const std::vector<grace::Ray> h_rays_in = read_file<grace::Ray>(argv[2]);
    const int max_per_leaf = std::atoi(argv[3]);
    const std::string out = argv[4];
    const bool greater = argc > 5 && std::string(argv[5]) == "greater";

    thrust::device_vector<Tri> d_tris(h_tris_in.begin(), h_tris_in.end());
    thrust::device_vector<grace::Ray> d_rays(h_rays_in.begin(), h_rays_in.end());
    grace::Tree d_tree(d_tris.size(), max_per_leaf);

    thrust::device_vector<grace::uinteger32> d_keys(d_tris.size());
    thrust::device_vector<grace::uinteger32> d_deltas(d_tris.size() + 1);
    float3 bots, tops;
    grace::morton_keys(d_tris, d_keys, TriCentre(), &bots, &tops);
#ifdef USE_THRUST_SORT
    // the reference's line (tris_tree.cuh:28): rocThrust's stable radix sort gives the same order
    thrust::sort_by_key(d_keys.begin(), d_keys.end(), d_tris.begin());
#else
    GRACE_STATUS_CHECK(grace_sort_pairs_u32(thrust::raw_pointer_cast(d_keys.data()),
                                            thrust::raw_pointer_cast(d_tris.data()), d_tris.size(),
                                            int(sizeof(Tri)), 0, 30, NULL, NULL));
#endif
    grace::compute_deltas(d_keys, d_deltas, grace::DeltaXOR());
    if (greater) {
        flip_bits<<<unsigned((d_deltas.size() + 255) / 256), 256>>>(thrust::raw_pointer_cast(d_deltas.data()),
                                                                    d_deltas.size());
        grace::build_ALBVH(d_tree, d_tris, d_deltas, thrust::greater<grace::uinteger32>(), TriBox());
    } else {
        grace::build_ALBVH(d_tree, d_tris, d_deltas, TriBox());
    }

    thrust::device_vector<int> d_closest(d_rays.size());
    grace::trace_texref<Nearest>(d_rays, d_tris, d_tree, 0, grace::Init_null(), HitTri(), KeepTri(),
                                 StartRay(),
                                 grace::RayExit_to_array<int>(thrust::raw_pointer_cast(d_closest.data())));

    thrust::host_vector<Tri> h_tris = d_tris;
    thrust::host_vector<int4> h_nodes = d_tree.nodes, h_leaves = d_tree.leaves;
    thrust::host_vector<int> h_closest = d_closest;
    int root = -1;
    GRACE_HIP_CHECK(hipMemcpy(&root, d_tree.root_index_ptr, sizeof(int), hipMemcpyDeviceToHost));
    write_file(out + ".tris", h_tris.data(), h_tris.size());
    write_file(out + ".nodes", h_nodes.data(), h_nodes.size());
    write_file(out + ".leaves", h_leaves.data(), h_leaves.size());
    write_file(out + ".closest", h_closest.data(), h_closest.size());
    const float bounds[6] = { bots.x, bots.y, bots.z, tops.x, tops.y, tops.z };
    write_file(out + ".bounds", bounds, 6);
    std::printf("%zu triangles, %zu leaves, root %d, %zu rays\n", h_tris.size(), h_leaves.size(), root,
                h_closest.size());
    write_file(out + ".root", &root, 1);
    return 0;
}
