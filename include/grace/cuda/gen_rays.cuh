// grace/cuda/gen_rays.cuh -- the ray generators of the reference
// (include/grace/cuda/gen_rays.cuh:25-399) with their signatures, dispatching to the
// deterministic generators of libgrace_hip.so (csrc/rays.hip).  The reference draws from
// cuRAND, whose streams are device-specific by its own account (kernels/gen_rays.cuh:21-24):
// only the distributions and the ordering contracts are kept, never the random stream.
// Positions, directions and lengths are computed in float (Real = float arithmetic); a double
// Real or Real3 argument is narrowed.
#pragma once

#include "grace/detail/raw.h"
#include "grace/ray.h"

namespace grace {

namespace detail {

// Point records the generators accept: x y z first, `elems` components of float or double.
template <typename PointType> struct point_traits;
template <> struct point_traits<float3>  { static const int is_double = 0, elems = 3; };
template <> struct point_traits<float4>  { static const int is_double = 0, elems = 4; };
template <> struct point_traits<double3> { static const int is_double = 1, elems = 3; };
template <> struct point_traits<double4> { static const int is_double = 1, elems = 4; };

} // namespace detail

// Isotropic rays from one origin, sorted by the 30-bit Morton key of their direction
// (gen_rays.cuh:25-60).
template <typename Real>
GRACE_HOST void uniform_random_rays(
    Ray* const d_rays_ptr,
    const size_t N_rays,
    const Real ox,
    const Real oy,
    const Real oz,
    const Real length,
    const unsigned long long seed = 1234)
{
    GRACE_STATUS_CHECK(grace_rays_isotropic(N_rays, float(ox), float(oy), float(oz), float(length),
                                            seed, d_rays_ptr, NULL));
}

template <typename Real>
GRACE_HOST void uniform_random_rays(
    thrust::device_vector<Ray>& d_rays,
    const Real ox,
    const Real oy,
    const Real oz,
    const Real length,
    const unsigned long long seed = 1234)
{
    uniform_random_rays(detail::raw(d_rays), d_rays.size(), ox, oy, oz, length, seed);
}

// gen_rays.cuh:62-97
template <typename Real>
GRACE_HOST void uniform_random_rays_single_octant(
    Ray* const d_rays_ptr,
    const size_t N_rays,
    const Real ox,
    const Real oy,
    const Real oz,
    const Real length,
    const enum Octants octant = PPP,
    const unsigned long long seed = 1234)
{
    GRACE_STATUS_CHECK(grace_rays_isotropic_octant(N_rays, float(ox), float(oy), float(oz),
                                                   float(length), int(octant), seed, d_rays_ptr, NULL));
}

template <typename Real>
GRACE_HOST void uniform_random_rays_single_octant(
    thrust::device_vector<Ray>& d_rays,
    const Real ox,
    const Real oy,
    const Real oz,
    const Real length,
    const enum Octants octant = PPP,
    const unsigned long long seed = 1234)
{
    uniform_random_rays_single_octant(detail::raw(d_rays), d_rays.size(), ox, oy, oz, length,
                                      octant, seed);
}

// One ray from (ox, oy, oz) to each point (gen_rays.cuh:99-158).  With EndPointSort the
// points' bounds are computed first, as the reference does (min_vec3 / max_vec3), and BOTH are
// used (the reference passes AABB_bot twice, gen_rays.cuh:121-122: not reproduced).  An unknown
// sort type throws std::invalid_argument.
template <typename Real, typename PointType>
GRACE_HOST void one_to_many_rays(
    Ray* const d_rays_ptr,
    const size_t N_rays,
    const Real ox,
    const Real oy,
    const Real oz,
    const PointType* const d_points_ptr,
    const enum RaySortType sort_type = DirectionSort)
{
    float bot[3], top[3];
    const bool endpoint = sort_type == EndPointSort;
    if (endpoint)
        GRACE_STATUS_CHECK(grace_centroid_bounds_points(d_points_ptr, N_rays,
                                                        detail::point_traits<PointType>::is_double,
                                                        detail::point_traits<PointType>::elems,
                                                        bot, top, NULL));
    GRACE_STATUS_CHECK(grace_rays_one_to_many(N_rays, float(ox), float(oy), float(oz), d_points_ptr,
                                              detail::point_traits<PointType>::is_double,
                                              detail::point_traits<PointType>::elems,
                                              int(sort_type), endpoint ? bot : NULL,
                                              endpoint ? top : NULL, d_rays_ptr, NULL));
}

template <typename Real, typename PointType>
GRACE_HOST void one_to_many_rays(
    thrust::device_vector<Ray>& d_rays,
    const Real ox,
    const Real oy,
    const Real oz,
    const thrust::device_vector<PointType>& d_points,
    const enum RaySortType sort_type = DirectionSort)
{
    // If d_rays.size() < d_points.size(), d_rays will be resized.
    if (d_rays.size() < d_points.size()) d_rays.resize(d_points.size());
    one_to_many_rays(detail::raw(d_rays), d_points.size(), ox, oy, oz, detail::raw(d_points),
                     sort_type);
}

// EndPointSort within the given bounds (gen_rays.cuh:160-208).
template <typename Real, typename Real3, typename PointType>
GRACE_HOST void one_to_many_rays(
    Ray* const d_rays_ptr,
    const size_t N_rays,
    const Real ox,
    const Real oy,
    const Real oz,
    const PointType* const d_points_ptr,
    const Real3 AABB_bot,
    const Real3 AABB_top)
{
    float bot[3], top[3];
    detail::xyz(AABB_bot, bot);
    detail::xyz(AABB_top, top);
    GRACE_STATUS_CHECK(grace_rays_one_to_many(N_rays, float(ox), float(oy), float(oz), d_points_ptr,
                                              detail::point_traits<PointType>::is_double,
                                              detail::point_traits<PointType>::elems,
                                              int(EndPointSort), bot, top, d_rays_ptr, NULL));
}

template <typename Real, typename Real3, typename PointType>
GRACE_HOST void one_to_many_rays(
    thrust::device_vector<Ray>& d_rays,
    const Real ox,
    const Real oy,
    const Real oz,
    const thrust::device_vector<PointType>& d_points,
    const Real3 AABB_bot,
    const Real3 AABB_top)
{
    if (d_rays.size() < d_points.size()) d_rays.resize(d_points.size());
    one_to_many_rays(detail::raw(d_rays), d_points.size(), ox, oy, oz, detail::raw(d_points),
                     AABB_bot, AABB_top);
}

// A width x height grid of cells spanned by w and h from base; one ray per cell from a random
// point of the cell, direction normalize(cross(w, h)) (gen_rays.cuh:210-262).
template <typename Real, typename Real3>
GRACE_HOST void plane_parallel_random_rays(
    Ray* const d_rays_ptr,
    const int width,
    const int height,
    const Real3 base,
    const Real3 w,
    const Real3 h,
    const Real length,
    const unsigned long long seed = 1234)
{
    float b[3], wv[3], hv[3];
    detail::xyz(base, b);
    detail::xyz(w, wv);
    detail::xyz(h, hv);
    GRACE_STATUS_CHECK(grace_rays_plane_parallel_random(width, height, b, wv, hv, float(length), seed,
                                                        d_rays_ptr, NULL));
}

template <typename Real, typename Real3>
GRACE_HOST void plane_parallel_random_rays(
    thrust::device_vector<Ray>& d_rays,
    const int width,
    const int height,
    const Real3 base,
    const Real3 w,
    const Real3 h,
    const Real length,
    const unsigned long long seed = 1234)
{
    if (d_rays.size() < (size_t)width * height) d_rays.resize((size_t)width * height);
    plane_parallel_random_rays(detail::raw(d_rays), width, height, base, w, h, length, seed);
}

// Orthographic projection: ray 0 is the top-left pixel, x fastest (gen_rays.cuh:264-329).
template <typename Real, typename Real3>
GRACE_HOST void orthographic_projection_rays(
    Ray* const d_rays_ptr,
    const int resolution_x,
    const int resolution_y,
    const Real3 camera_position,
    const Real3 look_at,
    const Real3 view_up,
    const Real vertical_extent,
    const Real length)
{
    float c[3], l[3], u[3];
    detail::xyz(camera_position, c);
    detail::xyz(look_at, l);
    detail::xyz(view_up, u);
    GRACE_STATUS_CHECK(grace_rays_orthographic_projection(resolution_x, resolution_y, c, l, u,
                                                          float(vertical_extent), float(length),
                                                          d_rays_ptr, NULL));
}

template <typename Real, typename Real3>
GRACE_HOST void orthographic_projection_rays(
    thrust::device_vector<Ray>& d_rays,
    const int resolution_x,
    const int resolution_y,
    const Real3 camera_position,
    const Real3 look_at,
    const Real3 view_up,
    const Real vertical_extent,
    const Real length)
{
    if (d_rays.size() < (size_t)resolution_x * resolution_y) d_rays.resize((size_t)resolution_x * resolution_y);
    orthographic_projection_rays(detail::raw(d_rays), resolution_x, resolution_y, camera_position,
                                 look_at, view_up, vertical_extent, length);
}

// Pinhole camera; FOVy in radians (gen_rays.cuh:331-399).
template <typename Real, typename Real3>
GRACE_HOST void pinhole_camera_rays(
    Ray* const d_rays_ptr,
    const int resolution_x,
    const int resolution_y,
    const Real3 camera_position,
    const Real3 look_at,
    const Real3 view_up,
    const Real FOVy,
    const Real length)
{
    float c[3], l[3], u[3];
    detail::xyz(camera_position, c);
    detail::xyz(look_at, l);
    detail::xyz(view_up, u);
    GRACE_STATUS_CHECK(grace_rays_pinhole(resolution_x, resolution_y, c, l, u, float(FOVy),
                                          float(length), d_rays_ptr, NULL));
}

template <typename Real, typename Real3>
GRACE_HOST void pinhole_camera_rays(
    thrust::device_vector<Ray>& d_rays,
    const int resolution_x,
    const int resolution_y,
    const Real3 camera_position,
    const Real3 look_at,
    const Real3 view_up,
    const Real FOVy,
    const Real length)
{
    if (d_rays.size() < (size_t)resolution_x * resolution_y) d_rays.resize((size_t)resolution_x * resolution_y);
    pinhole_camera_rays(detail::raw(d_rays), resolution_x, resolution_y, camera_position, look_at,
                        view_up, FOVy, length);
}

} // namespace grace
