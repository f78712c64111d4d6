/*
 * grace_oracle.c -- CPU oracle for the GRACE BVH-build + SPH ray-traversal hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (grace-devel_amd/, include/)
 * may call, link or import this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker.
 *
 * It is a plain-C restatement of the reference's algorithm (spthm/grace-devel).  Every
 * function cites the reference file:line it follows (paths relative to the reference
 * root).  Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp): contraction
 * is OFF because the reference's own CPU/GPU equality test is built with -fmad=false
 * (tests/tree_traversal/Makefile:5-8).
 *
 * Pinning: the Morton restatement is checked against the reference's known-answer
 * vectors (tests/morton_key/30bit_key.cu:20-26, 63bit_key.cu:20-26); the input RNG
 * against the three vectors recorded in SURVEY.md section 8c; HEALPix against
 * oracle/_ref (the reference's own chealpix.c compiled here); traversal against the
 * brute-force criterion of tests/tree_traversal/tree_traversal.cu:65-100; the line
 * integral against the volume-integral KAT of tests/integrate/integrate.cu:90-101.
 * See tests/test_oracle_*.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ------------------------------------------------------------------------- */
/* Types                                                                      */
/* ------------------------------------------------------------------------- */

/* include/grace/ray.h:5-10 : 7 floats, direction first. */
typedef struct { float dx, dy, dz, ox, oy, oz, length; } go_ray;
typedef struct { float x, y, z, w; } go_f4;
typedef struct { int x, y, z, w; } go_i4;

static inline int32_t f2i(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static inline float i2f(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

/* ------------------------------------------------------------------------- */
/* Morton keys                                                                */
/* ------------------------------------------------------------------------- */

/* include/grace/generic/bits.h:24-33 */
uint32_t go_space_by_two_10bit(uint32_t x)
{
    x &= (1u << 10) - 1;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x <<  8)) & 0x0300F00Fu;
    x = (x | (x <<  4)) & 0x030C30C3u;
    x = (x | (x <<  2)) & 0x09249249u;
    return x;
}

/* include/grace/generic/bits.h:35-46 */
uint64_t go_space_by_two_21bit(uint64_t x)
{
    x &= (1u << 21) - 1;
    x = (x | x << 32) & 0x001f00000000ffffull;
    x = (x | x << 16) & 0x001f0000ff0000ffull;
    x = (x | x <<  8) & 0x100f00f00f00f00full;
    x = (x | x <<  4) & 0x10c30c30c30c30c3ull;
    x = (x | x <<  2) & 0x1249249249249249ull;
    return x;
}

/* include/grace/generic/morton.h:14-20 */
uint32_t go_morton_key30(uint32_t x, uint32_t y, uint32_t z)
{
    return go_space_by_two_10bit(z) << 2 | go_space_by_two_10bit(y) << 1
           | go_space_by_two_10bit(x);
}

/* include/grace/generic/morton.h:23-29 */
uint64_t go_morton_key63(uint64_t x, uint64_t y, uint64_t z)
{
    return go_space_by_two_21bit(z) << 2 | go_space_by_two_21bit(y) << 1
           | go_space_by_two_21bit(x);
}

/* include/grace/generic/morton.h:32-42 : floats in (0,1), span * x in fp32. */
uint32_t go_morton_key30_unit(float x, float y, float z)
{
    unsigned int span = (1u << 10) - 1;
    return go_morton_key30((uint32_t)(span * x), (uint32_t)(span * y),
                           (uint32_t)(span * z));
}

/* include/grace/generic/morton.h:45-55 */
uint64_t go_morton_key63_unit(double x, double y, double z)
{
    unsigned int span = (1u << 21) - 1;
    return go_morton_key63((uint64_t)(span * x), (uint64_t)(span * y),
                           (uint64_t)(span * z));
}

/* Device formula: scale = span / (top - bot) on the host in Real3 precision
 * (include/grace/cuda/kernels/morton.cuh:107-113), key coordinate =
 * KeyType(scale * (centre - min)) per axis (morton.cuh:43-50); the centroid of a
 * sphere is its float3 centre (generic/functors/centroid.h:40-48). */
void go_morton_keys30_f4(const go_f4* prims, size_t n, const float* bot,
                         const float* top, uint32_t* keys)
{
    const int span = (1u << 10) - 1;
    float sx = span / (top[0] - bot[0]);
    float sy = span / (top[1] - bot[1]);
    float sz = span / (top[2] - bot[2]);
    for (size_t i = 0; i < n; ++i) {
        uint32_t x = (uint32_t)(sx * (prims[i].x - bot[0]));
        uint32_t y = (uint32_t)(sy * (prims[i].y - bot[1]));
        uint32_t z = (uint32_t)(sz * (prims[i].z - bot[2]));
        keys[i] = go_morton_key30(x, y, z);
    }
}

/* The same loop over all host threads (bench.py's cpu_baseline reports it next to the serial
 * one; the reference's tests/morton_key loop is serial). */
void go_morton_keys30_f4_omp(const go_f4* prims, size_t n, const float* bot,
                             const float* top, uint32_t* keys)
{
    const int span = (1u << 10) - 1;
    float sx = span / (top[0] - bot[0]);
    float sy = span / (top[1] - bot[1]);
    float sz = span / (top[2] - bot[2]);
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        uint32_t x = (uint32_t)(sx * (prims[i].x - bot[0]));
        uint32_t y = (uint32_t)(sy * (prims[i].y - bot[1]));
        uint32_t z = (uint32_t)(sz * (prims[i].z - bot[2]));
        keys[i] = go_morton_key30(x, y, z);
    }
}

/* Same, 63-bit keys, Real3 = float3 bounds (fp32 arithmetic). */
void go_morton_keys63_f4(const go_f4* prims, size_t n, const float* bot,
                         const float* top, uint64_t* keys)
{
    const int span = (1u << 21) - 1;
    float sx = span / (top[0] - bot[0]);
    float sy = span / (top[1] - bot[1]);
    float sz = span / (top[2] - bot[2]);
    for (size_t i = 0; i < n; ++i) {
        uint64_t x = (uint64_t)(sx * (prims[i].x - bot[0]));
        uint64_t y = (uint64_t)(sy * (prims[i].y - bot[1]));
        uint64_t z = (uint64_t)(sz * (prims[i].z - bot[2]));
        keys[i] = go_morton_key63(x, y, z);
    }
}

/* Same, 63-bit keys, Real3 = double3 bounds: float centre promoted to double. */
void go_morton_keys63_f4_d3(const go_f4* prims, size_t n, const double* bot,
                            const double* top, uint64_t* keys)
{
    const int span = (1u << 21) - 1;
    double sx = span / (top[0] - bot[0]);
    double sy = span / (top[1] - bot[1]);
    double sz = span / (top[2] - bot[2]);
    for (size_t i = 0; i < n; ++i) {
        uint64_t x = (uint64_t)(sx * (prims[i].x - bot[0]));
        uint64_t y = (uint64_t)(sy * (prims[i].y - bot[1]));
        uint64_t z = (uint64_t)(sz * (prims[i].z - bot[2]));
        keys[i] = go_morton_key63(x, y, z);
    }
}

/* Component-wise min/max of centroids: include/grace/cuda/kernels/morton.cuh:153-164
 * (compute_centroids + min_vec3/max_vec3).  Exact (min/max only). */
void go_centroid_bounds_f4(const go_f4* prims, size_t n, float* bot, float* top)
{
    float lo[3] = { INFINITY, INFINITY, INFINITY };
    float hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = 0; i < n; ++i) {
        const float c[3] = { prims[i].x, prims[i].y, prims[i].z };
        for (int k = 0; k < 3; ++k) {
            if (c[k] < lo[k]) lo[k] = c[k];
            if (c[k] > hi[k]) hi[k] = c[k];
        }
    }
    for (int k = 0; k < 3; ++k) { bot[k] = lo[k]; top[k] = hi[k]; }
}

/* ------------------------------------------------------------------------- */
/* Input generator of the reference's tests                                   */
/* ------------------------------------------------------------------------- */

/* tests/helper/random.cuh:20-29 (Wang/Jenkins integer hash). */
uint32_t go_hash(uint32_t a)
{
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

/* tests/helper/random.cuh:56-111 random_real4_functor: engine = minstd_rand
 * (x <- 48271 x mod 2^31-1) seeded with hash(n); uniform_real_distribution<float>
 * = float(x - 1) / (float(2147483645) + 1) * (hi - lo) + lo; draws in x,y,z,w order. */
void go_random_real4(uint32_t first, size_t n, const float* lo, const float* hi,
                     go_f4* out)
{
    const uint64_t m = 2147483647ull;
    const float denom = (float)2147483645u + 1.0f;
    for (size_t i = 0; i < n; ++i) {
        uint64_t x = go_hash(first + (uint32_t)i) % m;
        if (x == 0) x = 1;
        float v[4];
        for (int k = 0; k < 4; ++k) {
            x = (48271ull * x) % m;
            float u = (float)(uint32_t)(x - 1);
            u /= denom;
            v[k] = u * (hi[k] - lo[k]) + lo[k];
        }
        out[i].x = v[0]; out[i].y = v[1]; out[i].z = v[2]; out[i].w = v[3];
    }
}

/* ------------------------------------------------------------------------- */
/* Deltas                                                                     */
/* ------------------------------------------------------------------------- */

/* include/grace/generic/functors/albvh.h:59-81 with compute_deltas_kernel
 * (include/grace/cuda/kernels/albvh.cuh:33-47): deltas[i] = delta(i-1), i in [0,N]. */
void go_deltas_euclid_f4(const go_f4* p, size_t n, float* deltas)
{
    for (size_t t = 0; t <= n; ++t) {
        long i = (long)t - 1;
        if (i < 0 || (size_t)(i + 1) >= n) { deltas[t] = INFINITY; continue; }
        float dx = p[i].x - p[i + 1].x;
        float dy = p[i].y - p[i + 1].y;
        float dz = p[i].z - p[i + 1].z;
        deltas[t] = dx * dx + dy * dy + dz * dz;
    }
}

/* include/grace/generic/functors/albvh.h:84-123 with AABBSphere
 * (generic/functors/aabb.h:9-26). */
void go_deltas_area_f4(const go_f4* p, size_t n, float* deltas)
{
    for (size_t t = 0; t <= n; ++t) {
        long i = (long)t - 1;
        if (i < 0 || (size_t)(i + 1) >= n) { deltas[t] = INFINITY; continue; }
        go_f4 a = p[i], b = p[i + 1];
        float Lx = fmaxf(a.x + a.w, b.x + b.w) - fminf(a.x - a.w, b.x - b.w);
        float Ly = fmaxf(a.y + a.w, b.y + b.w) - fminf(a.y - a.w, b.y - b.w);
        float Lz = fmaxf(a.z + a.w, b.z + b.w) - fminf(a.z - a.w, b.z - b.w);
        deltas[t] = (Lx * Ly) + (Lx * Lz) + (Ly * Lz);
    }
}

/* include/grace/generic/functors/albvh.h:17-33 */
void go_deltas_xor_u32(const uint32_t* k, size_t n, uint32_t* deltas)
{
    for (size_t t = 0; t <= n; ++t) {
        long i = (long)t - 1;
        if (i < 0 || (size_t)(i + 1) >= n) { deltas[t] = 0xFFFFFFFFu; continue; }
        deltas[t] = k[i] ^ k[i + 1];
    }
}

/* include/grace/generic/functors/albvh.h:35-48 */
void go_deltas_xor_u64(const uint64_t* k, size_t n, uint64_t* deltas)
{
    for (size_t t = 0; t <= n; ++t) {
        long i = (long)t - 1;
        if (i < 0 || (size_t)(i + 1) >= n) { deltas[t] = ~0ull; continue; }
        deltas[t] = k[i] ^ k[i + 1];
    }
}

/* ------------------------------------------------------------------------- */
/* ALBVH build (sequential restatement)                                       */
/* ------------------------------------------------------------------------- */

/* Delta arrays are passed with the reference's +1 shift: d[0] is delta(-1).  The
 * comparator is thrust::less (albvh.cuh:1045-1057): left parent iff dL < dR. */
#define GO_DEFINE_LEAVES(NAME, T)                                                  \
/* build_leaves_kernel (albvh.cuh:77-234) without the per-block windows (a leaf's   \
 * climb is followed by some block whose window holds it, albvh.cuh:147-150), then  \
 * write_leaves_kernel (albvh.cuh:236-295) and the stable compaction                \
 * remove_if(is_empty_node) (albvh.cuh:826-846).  Returns n_leaves; leaves must hold\
 * n entries.  Returns -1 if n <= max_per_leaf (std::invalid_argument,              \
 * albvh.cuh:795-799). */                                                           \
long NAME(const T* deltas_shifted, size_t n, int max_per_leaf, go_i4* leaves)       \
{                                                                                   \
    if (n <= (size_t)max_per_leaf) return -1;                                       \
    const T* d = deltas_shifted + 1; /* d[-1] .. d[n-1] */                          \
    size_t n_nodes = n - 1;                                                         \
    int* nx = (int*)calloc(n_nodes, sizeof(int));                                   \
    int* ny = (int*)calloc(n_nodes, sizeof(int));                                   \
    unsigned char* flag = (unsigned char*)calloc(n_nodes, 1);                       \
    for (size_t idx = 0; idx < n; ++idx) {                                          \
        long left = (long)idx, right = (long)idx;                                   \
        for (;;) {                                                                  \
            long parent;                                                            \
            if (d[left - 1] < d[right]) { parent = left - 1; ny[parent] = (int)right; } \
            else { parent = right; nx[parent] = (int)left; }                        \
            if (flag[parent]++ == 0) break; /* first arrival stops */               \
            left = nx[parent]; right = ny[parent];                                  \
            if (right - left + 1 > max_per_leaf) break;                             \
            /* the root has no parent; it is only reached when n <= mpl */          \
        }                                                                           \
    }                                                                               \
    go_i4* big = (go_i4*)calloc(n, sizeof(go_i4));                                  \
    for (size_t tid = 0; tid < n_nodes; ++tid) {                                    \
        int left = nx[tid], right = ny[tid];                                        \
        int size = right - left + 1;                                                \
        int left_size = (int)tid - left + 1;                                        \
        int right_size = right > 0 ? right - (int)tid : max_per_leaf + 1;           \
        int left_leaf = left_size <= max_per_leaf;                                  \
        int right_leaf = right_size <= max_per_leaf;                                \
        int write_check = (left_leaf != right_leaf) ? 1 : (size > max_per_leaf);    \
        if (left_leaf && write_check) { big[left].x = left; big[left].y = left_size; } \
        if (right_leaf && write_check) { big[right].x = (int)tid + 1; big[right].y = right_size; } \
    }                                                                               \
    long n_leaves = 0;                                                              \
    for (size_t i = 0; i < n; ++i)                                                  \
        if (big[i].y != 0) { leaves[n_leaves].x = big[i].x; leaves[n_leaves].y = big[i].y; \
                             leaves[n_leaves].z = 0; leaves[n_leaves].w = 0; ++n_leaves; } \
    free(nx); free(ny); free(flag); free(big);                                      \
    return n_leaves;                                                                \
}

GO_DEFINE_LEAVES(go_albvh_leaves_f32, float)
GO_DEFINE_LEAVES(go_albvh_leaves_u32, uint32_t)
GO_DEFINE_LEAVES(go_albvh_leaves_u64, uint64_t)

/* copy_leaf_deltas_kernel (albvh.cuh:51-74); output has n_leaves + 1 entries. */
#define GO_DEFINE_LEAF_DELTAS(NAME, T)                                              \
void NAME(const go_i4* leaves, size_t n_leaves, const T* deltas_shifted, T* out)    \
{                                                                                   \
    out[0] = deltas_shifted[0];                                                     \
    for (size_t i = 0; i < n_leaves; ++i)                                           \
        out[i + 1] = deltas_shifted[1 + leaves[i].x + leaves[i].y - 1];             \
}
GO_DEFINE_LEAF_DELTAS(go_leaf_deltas_f32, float)
GO_DEFINE_LEAF_DELTAS(go_leaf_deltas_u32, uint32_t)
GO_DEFINE_LEAF_DELTAS(go_leaf_deltas_u64, uint64_t)

/* AABB of a primitive.  kind 0: sphere float4 (generic/functors/aabb.h:9-26);
 * kind 1: triangle 9 floats {v, e1, e2} (tests/profile_trace_triangle/triangle.cuh,
 * TriangleAABB: min/max over v, v+e1, v+e2, degenerate extents inflated by
 * AABB_EPSILON). */
#define GO_TRI_AABB_EPS 0.000001f
static void prim_aabb(const float* prims, int kind, size_t i, float* bot, float* top)
{
    if (kind == 0) {
        const float* s = prims + 4 * i;
        for (int k = 0; k < 3; ++k) { bot[k] = s[k] - s[3]; top[k] = s[k] + s[3]; }
    } else if (kind == 2) {
        /* double4 spheres: AABBSphere with Real4 = double4 (generic/functors/aabb.h:9-26):
         * the sum is formed in double and narrowed to the float3 corner */
        const double* s = (const double*)prims + 4 * i;
        for (int k = 0; k < 3; ++k) { bot[k] = (float)(s[k] - s[3]); top[k] = (float)(s[k] + s[3]); }
    } else {
        const float* t = prims + 9 * i;
        for (int k = 0; k < 3; ++k) {
            float v0 = t[k], v1 = t[k] + t[3 + k], v2 = t[k] + t[6 + k];
            bot[k] = fminf(v0, fminf(v1, v2));
            top[k] = fmaxf(v0, fmaxf(v1, v2));
            if (bot[k] == top[k]) { /* triangle.cu:21-35: inflate by eps * |coordinate| */
                float scale = fabsf(bot[k]);
                bot[k] -= GO_TRI_AABB_EPS * scale; top[k] += GO_TRI_AABB_EPS * scale;
            }
        }
    }
}

/* build_nodes (albvh.cuh:848-940): the final contents of the node array after all
 * slices.  A node's index is the leaf index of its split (g_left - 1 or g_right,
 * albvh.cuh:470,491), children are node indices or n_nodes + leaf index
 * (albvh.cuh:510), .z/.w the first/last leaf covered (fix_node_ranges,
 * albvh.cuh:717-761), child AABBs laid out as in include/grace/cuda/nodes.h:22-37,
 * root = the node whose range is every leaf (albvh.cuh:572-573).  nodes must hold
 * 16 * (n_leaves - 1) ints/floats. */
#define GO_DEFINE_NODES(NAME, T)                                                    \
int NAME(const go_i4* leaves, size_t n_leaves, const float* prims, int prim_kind,   \
         const T* leaf_deltas_shifted, int* nodes_i, int* root_index)               \
{                                                                                   \
    const T* d = leaf_deltas_shifted + 1;                                           \
    size_t n_nodes = n_leaves - 1;                                                  \
    float* nodes_f = (float*)nodes_i;                                               \
    unsigned char* flag = (unsigned char*)calloc(n_nodes ? n_nodes : 1, 1);         \
    *root_index = -1;                                                               \
    for (size_t leaf = 0; leaf < n_leaves; ++leaf) {                                \
        float bot[3] = { INFINITY, INFINITY, INFINITY };                            \
        float top[3] = { -INFINITY, -INFINITY, -INFINITY };                         \
        for (int i = 0; i < leaves[leaf].y; ++i) {                                  \
            float b[3], t[3];                                                       \
            prim_aabb(prims, prim_kind, (size_t)leaves[leaf].x + i, b, t);          \
            for (int k = 0; k < 3; ++k) { bot[k] = fminf(bot[k], b[k]); top[k] = fmaxf(top[k], t[k]); } \
        }                                                                           \
        long g_left = (long)leaf, g_right = (long)leaf;                             \
        int g_cur = (int)(n_nodes + leaf);                                          \
        for (;;) {                                                                  \
            long parent; int is_right_child;                                        \
            if (d[g_left - 1] < d[g_right]) { parent = g_left - 1; is_right_child = 1; } \
            else { parent = g_right; is_right_child = 0; }                          \
            if (parent < 0 || parent >= (long)n_nodes) break; /* cur is the root */ \
            int* n0 = nodes_i + 16 * parent;                                        \
            float* nf = nodes_f + 16 * parent;                                      \
            if (is_right_child) {                                                   \
                n0[1] = g_cur; n0[3] = (int)g_right;                                \
                nf[8] = bot[0]; nf[9] = top[0]; nf[10] = bot[1]; nf[11] = top[1];   \
                nf[14] = bot[2]; nf[15] = top[2];                                   \
            } else {                                                                \
                n0[0] = g_cur; n0[2] = (int)g_left;                                 \
                nf[4] = bot[0]; nf[5] = top[0]; nf[6] = bot[1]; nf[7] = top[1];     \
                nf[12] = bot[2]; nf[13] = top[2];                                   \
            }                                                                       \
            if (flag[parent]++ == 0) break;                                         \
            g_cur = (int)parent;                                                    \
            g_left = n0[2]; g_right = n0[3];                                        \
            if (g_right - g_left == (long)n_leaves - 1) *root_index = g_cur;        \
            bot[0] = fminf(nf[4], nf[8]);   top[0] = fmaxf(nf[5], nf[9]);           \
            bot[1] = fminf(nf[6], nf[10]);  top[1] = fmaxf(nf[7], nf[11]);          \
            bot[2] = fminf(nf[12], nf[14]); top[2] = fmaxf(nf[13], nf[15]);         \
        }                                                                           \
    }                                                                               \
    free(flag);                                                                     \
    return *root_index >= 0 ? 0 : -1;                                               \
}
GO_DEFINE_NODES(go_albvh_nodes_f32, float)
GO_DEFINE_NODES(go_albvh_nodes_u32, uint32_t)
GO_DEFINE_NODES(go_albvh_nodes_u64, uint64_t)

/* ------------------------------------------------------------------------- */
/* Intersection tests                                                         */
/* ------------------------------------------------------------------------- */

/* include/grace/generic/intersect.h:10-55, Real = float, no FMA. */
static inline int sphere_hit(const go_ray* ray, const go_f4* s, float* b2, float* dot_p)
{
    float px = s->x - ray->ox;
    float py = s->y - ray->oy;
    float pz = s->z - ray->oz;
    float rx = ray->dx, ry = ray->dy, rz = ray->dz;
    *dot_p = px * rx + py * ry + pz * rz;
    float bx = px - *dot_p * rx;
    float by = py - *dot_p * ry;
    float bz = pz - *dot_p * rz;
    *b2 = bx * bx + by * by + bz * bz;
    if (*b2 >= s->w * s->w) return 0;
    if (*dot_p < 0.0f) return 0;
    if (*dot_p >= ray->length) return 0;
    return 1;
}

int go_sphere_hit(const go_ray* ray, const go_f4* s, float* b2, float* dist)
{
    return sphere_hit(ray, s, b2, dist);
}

/* CUDA fminf/fmaxf: the non-NaN operand when one is NaN (C99 fminf does the same). */
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* include/grace/cuda/device/intersect.cuh:10-40 with the integer min/max of
 * include/grace/cuda/device/intrinsics.cuh:8-51 (signed-int compares on float
 * bits).  node = 16 floats laid out as include/grace/cuda/nodes.h:22-37.
 * Returns bit0 = right hit, bit1 = left hit. */
static inline int aabbs_hit(const float* invd, const float* o, float len, const float* nf)
{
    const float* L = nf + 4;  /* Lbx Ltx Lby Lty */
    const float* R = nf + 8;  /* Rbx Rtx Rby Rty */
    const float* Z = nf + 12; /* Lbz Ltz Rbz Rtz */
    float bx_L = (L[0] - o[0]) * invd[0], tx_L = (L[1] - o[0]) * invd[0];
    float by_L = (L[2] - o[1]) * invd[1], ty_L = (L[3] - o[1]) * invd[1];
    float bz_L = (Z[0] - o[2]) * invd[2], tz_L = (Z[1] - o[2]) * invd[2];
    float bx_R = (R[0] - o[0]) * invd[0], tx_R = (R[1] - o[0]) * invd[0];
    float by_R = (R[2] - o[1]) * invd[1], ty_R = (R[3] - o[1]) * invd[1];
    float bz_R = (Z[2] - o[2]) * invd[2], tz_R = (Z[3] - o[2]) * invd[2];

    /* maxf_vmaxf(a, b, c) = max(max(a,b),c); maxf_vminf(a,b,c) = max(min(a,b),c);
     * minf_vminf = min(min(a,b),c); minf_vmaxf(a,b,c) = min(max(a,b),c). */
    int tmin_L = imax(imax(f2i(fminf(bx_L, tx_L)), f2i(fminf(by_L, ty_L))),
                      imax(imin(f2i(bz_L), f2i(tz_L)), f2i(0.0f)));
    int tmax_L = imin(imin(f2i(fmaxf(bx_L, tx_L)), f2i(fmaxf(by_L, ty_L))),
                      imin(imax(f2i(bz_L), f2i(tz_L)), f2i(len)));
    int tmin_R = imax(imax(f2i(fminf(bx_R, tx_R)), f2i(fminf(by_R, ty_R))),
                      imax(imin(f2i(bz_R), f2i(tz_R)), f2i(0.0f)));
    int tmax_R = imin(imin(f2i(fmaxf(bx_R, tx_R)), f2i(fmaxf(by_R, ty_R))),
                      imin(imax(f2i(bz_R), f2i(tz_R)), f2i(len)));
    return (int)(i2f(tmax_R) >= i2f(tmin_R)) + 2 * (int)(i2f(tmax_L) >= i2f(tmin_L));
}

int go_aabbs_hit(const go_ray* ray, const float* node16)
{
    float invd[3] = { 1.f / ray->dx, 1.f / ray->dy, 1.f / ray->dz };
    float o[3] = { ray->ox, ray->oy, ray->oz };
    return aabbs_hit(invd, o, ray->length, node16);
}

/* ------------------------------------------------------------------------- */
/* SPH kernel line integral                                                   */
/* ------------------------------------------------------------------------- */

/* include/grace/cuda/trace_sph.cuh:22-50 */
#define GO_N_TABLE 51
static const double go_table[GO_N_TABLE] = {
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

const double* go_kernel_table(int* n) { if (n) *n = GO_N_TABLE; return go_table; }

/* include/grace/generic/interpolate.h:11-39, device branch (fma in the table's
 * precision, :33-34), Real = float, TableReal = double. */
static inline float lerp_table(float x)
{
    int x_idx = (int)x;
    if (x_idx >= GO_N_TABLE - 1) {
        x = (float)(double)(GO_N_TABLE - 1);
        x_idx = GO_N_TABLE - 2;
    }
    double y0 = go_table[x_idx];
    double y1 = go_table[x_idx + 1];
    double t = (double)x - x_idx;
    return (float)fma(t, y1 - y0, y0);
}

/* OnHit_sphere_cumulate / OnHit_sphere_individual arithmetic
 * (include/grace/cuda/functors/trace.cuh:181-186, 221-224). */
static inline float hit_integral(float b2, float h)
{
    float ir = 1.f / h;
    float b = (GO_N_TABLE - 1) * (sqrtf(b2) * ir);
    float integral = lerp_table(b);
    integral *= (ir * ir);
    return integral;
}

float go_hit_integral(float b2, float h) { return hit_integral(b2, h); }

void go_hit_integral_array(const float* b2, const float* h, size_t n, float* out)
{
    #pragma omp parallel for
    for (size_t i = 0; i < n; ++i) out[i] = hit_integral(b2[i], h[i]);
}

/* ------------------------------------------------------------------------- */
/* Brute force (the reference's own correctness criterion)                    */
/* ------------------------------------------------------------------------- */

/* tests/tree_traversal/tree_traversal.cu:65-79 */
void go_brute_hitcounts(const go_ray* rays, size_t n_rays, const go_f4* s, size_t n,
                        int* counts)
{
    #pragma omp parallel for schedule(dynamic, 16)
    for (size_t ri = 0; ri < n_rays; ++ri) {
        go_ray ray = rays[ri];
        int hits = 0; float b2, d;
        for (size_t si = 0; si < n; ++si)
            if (sphere_hit(&ray, &s[si], &b2, &d)) ++hits;
        counts[ri] = hits;
    }
}

/* Column density of one ray.  Hits are visited in ascending primitive index: the order in
 * which trace_kernel meets them (bintree_trace.cuh:128-192: left child pushed last, so popped
 * first; leaves scanned first..first+count).  The reference keeps ONE fp32 running sum
 * (RayData_sphere<float,float>.data).  This implementation's stated result is the
 * CLASS-ORDERED sum: primitive p belongs to class (p >> 10) & 7 (granules of 1024 consecutive
 * indices dealt round-robin to 8 classes); each class is summed in ascending primitive
 * order and the 8 class sums are added pairwise, ((c0+c1)+(c2+c3))+((c4+c5)+(c6+c7)) -- a fixed
 * association that lets up to 8 waves share a packet with an even share of the work each
 * (DESIGN.md section 4).  blocks = 1 gives the reference's single running sum; both lie
 * within ~1e-6 of the fp64 sum (tolerance 1e-5). */
#define GO_SUM_CLASSES 8
#define GO_GRANULE_SHIFT 10
typedef struct { float cls[GO_SUM_CLASSES]; int classes; } go_acc;
static inline void acc_init(go_acc* a, size_t n, int blocks)
{
    (void)n;
    for (int c = 0; c < GO_SUM_CLASSES; ++c) a->cls[c] = 0.f;
    a->classes = blocks > 1 ? GO_SUM_CLASSES : 1;
}
static inline void acc_add(go_acc* a, long prim, float w)
{
    const int c = a->classes > 1 ? (int)((prim >> GO_GRANULE_SHIFT) & (GO_SUM_CLASSES - 1)) : 0;
    a->cls[c] += w;
}
static inline float acc_result(go_acc* a)
{
    float t[GO_SUM_CLASSES];
    for (int c = 0; c < GO_SUM_CLASSES; ++c) t[c] = a->cls[c];
    for (int w = 1; w < GO_SUM_CLASSES; w *= 2)
        for (int c = 0; c < GO_SUM_CLASSES; c += 2 * w) t[c] = t[c] + t[c + w];
    return t[0];
}

void go_brute_cumulative(const go_ray* rays, size_t n_rays, const go_f4* s, size_t n,
                         float* out, double* out64, int blocks)
{
    if (blocks < 1) blocks = 1;
    #pragma omp parallel for schedule(dynamic, 16)
    for (size_t ri = 0; ri < n_rays; ++ri) {
        go_ray ray = rays[ri];
        go_acc acc; acc_init(&acc, n, blocks);
        double acc64 = 0.0; float b2, d;
        for (size_t si = 0; si < n; ++si)
            if (sphere_hit(&ray, &s[si], &b2, &d)) {
                float w = hit_integral(b2, s[si].w);
                acc_add(&acc, (long)si, w); acc64 += (double)w;
            }
        out[ri] = acc_result(&acc);
        if (out64) out64[ri] = acc64;
    }
}

/* Per-hit outputs of trace_sph pass 2 (trace_sph.cuh:143-167) for one ray range,
 * given exclusive offsets. */
void go_brute_hits(const go_ray* rays, size_t n_rays, const go_f4* s, size_t n,
                   const int* offsets, int* idx, float* integrals, float* dists)
{
    #pragma omp parallel for schedule(dynamic, 16)
    for (size_t ri = 0; ri < n_rays; ++ri) {
        go_ray ray = rays[ri];
        int o = offsets[ri]; float b2, d;
        for (size_t si = 0; si < n; ++si)
            if (sphere_hit(&ray, &s[si], &b2, &d)) {
                idx[o] = (int)si; integrals[o] = hit_integral(b2, s[si].w); dists[o] = d; ++o;
            }
    }
}

/* ------------------------------------------------------------------------- */
/* Tree traversal                                                             */
/* ------------------------------------------------------------------------- */

#define GO_STACK 1024

/* trace_kernel (include/grace/cuda/kernels/bintree_trace.cuh:52-197) for packets of
 * `width` consecutive rays sharing one stack, as a warp does (width 32 in the
 * reference; width 1 is a plain single-ray depth-first walk).  mode 0: hit counts
 * (OnHit_increment), mode 1: cumulative integral (OnHit_sphere_cumulate).
 * stats (optional, 4 x n_rays uint64): per RAY nodes visited, leaves visited, prims
 * tested, hits -- counted for that ray alone (a ray "visits" a node when every
 * ancestor's box test passed for that ray), the figure SURVEY.md section 8d's
 * algorithmic-bytes formula needs. */
int go_trace(const go_ray* rays, size_t n_rays, const go_f4* s, size_t n_prims,
             const float* nodes, size_t n_nodes, const go_i4* leaves, int root,
             int width, int mode, void* out, uint64_t* stats, int blocks)
{
    if (blocks < 1) blocks = 1;
    if (width < 1 || width > 64) return -2;
    int overflow = 0;
    size_t n_packets = (n_rays + width - 1) / width;
    #pragma omp parallel for schedule(dynamic, 4)
    for (size_t pk = 0; pk < n_packets; ++pk) {
        size_t r0 = pk * width;
        int w = (r0 + (size_t)width <= n_rays) ? width : (int)(n_rays - r0);
        int stack[GO_STACK];
        unsigned char* act = NULL; /* per stack slot, per lane: lane alone reaches it */
        if (stats) act = (unsigned char*)malloc((size_t)GO_STACK * w);
        float* invd = (float*)malloc(sizeof(float) * 3 * w);
        go_acc* acc = (go_acc*)malloc(sizeof(go_acc) * w);
        for (int l = 0; l < w; ++l) acc_init(&acc[l], n_prims, blocks);
        int* cnt = (int*)calloc(w, sizeof(int));
        for (int l = 0; l < w; ++l) {
            invd[3*l+0] = 1.f / rays[r0+l].dx;
            invd[3*l+1] = 1.f / rays[r0+l].dy;
            invd[3*l+2] = 1.f / rays[r0+l].dz;
        }
        int sp = 0;
        stack[0] = root;
        if (act) memset(act, 1, w);
        unsigned char* cur_act = act ? (unsigned char*)malloc(w) : NULL;
        while (sp >= 0) {
            int idx = stack[sp];
            if (act) memcpy(cur_act, act + (size_t)sp * w, w);
            --sp;
            if ((size_t)idx < n_nodes) {
                const float* nf = nodes + 16 * (size_t)idx;
                const int* ni = (const int*)nf;
                int any_r = 0, any_l = 0;
                unsigned char lr[64];
                for (int l = 0; l < w; ++l) {
                    const go_ray* r = &rays[r0+l];
                    float o[3] = { r->ox, r->oy, r->oz };
                    int h = aabbs_hit(invd + 3*l, o, r->length, nf);
                    lr[l] = (unsigned char)h;
                    any_r |= (h & 1); any_l |= (h >= 2);
                    if (stats && cur_act[l]) stats[4*(r0+l)+0]++;
                }
                if (any_r) {
                    if (sp + 1 >= GO_STACK) { overflow = 1; break; }
                    ++sp; stack[sp] = ni[1];
                    if (act) for (int l = 0; l < w; ++l) act[(size_t)sp*w+l] = cur_act[l] && (lr[l] & 1);
                }
                if (any_l) {
                    if (sp + 1 >= GO_STACK) { overflow = 1; break; }
                    ++sp; stack[sp] = ni[0];
                    if (act) for (int l = 0; l < w; ++l) act[(size_t)sp*w+l] = cur_act[l] && (lr[l] >= 2);
                }
            } else {
                go_i4 leaf = leaves[(size_t)idx - n_nodes];
                for (int l = 0; l < w; ++l) {
                    const go_ray* r = &rays[r0+l];
                    if (stats && cur_act[l]) { stats[4*(r0+l)+1]++; stats[4*(r0+l)+2] += leaf.y; }
                    for (int i = 0; i < leaf.y; ++i) {
                        float b2, d;
                        if (sphere_hit(r, &s[leaf.x + i], &b2, &d)) {
                            cnt[l]++;
                            if (mode == 1) acc_add(&acc[l], leaf.x + i, hit_integral(b2, s[leaf.x + i].w));
                        }
                    }
                }
            }
        }
        for (int l = 0; l < w; ++l) {
            if (mode == 0) ((int*)out)[r0+l] = cnt[l];
            else ((float*)out)[r0+l] = acc_result(&acc[l]);
            if (stats) stats[4*(r0+l)+3] = cnt[l];
        }
        free(invd); free(acc); free(cnt);
        if (act) { free(act); free(cur_act); }
    }
    return overflow ? -1 : 0;
}

/* ------------------------------------------------------------------------- */
/* Scans                                                                      */
/* ------------------------------------------------------------------------- */

/* thrust::exclusive_scan of hit counts (include/grace/cuda/trace_sph.cuh:135-137);
 * returns the total. */
long go_exclusive_scan_i32(const int* in, size_t n, int* out)
{
    long acc = 0;
    for (size_t i = 0; i < n; ++i) { int v = in[i]; out[i] = (int)acc; acc += v; }
    return acc;
}

/* Sequential per-segment exclusive scan: the host check of
 * tests/segmented_scan/segmented_scan.cu:126-136 (the contract of
 * grace::exclusive_segmented_scan, include/grace/cuda/scan.cuh:15-37). */
void go_segscan_f32(const int* offsets, size_t n_seg, const float* data, size_t n,
                    float* out)
{
    for (size_t row = 0; row < n_seg; ++row) {
        size_t b = offsets[row], e = (row + 1 < n_seg) ? (size_t)offsets[row + 1] : n;
        float x = 0;
        for (size_t i = b; i < e; ++i) { out[i] = x; x = x + data[i]; }
    }
}

void go_segscan_f64(const int* offsets, size_t n_seg, const double* data, size_t n,
                    double* out)
{
    for (size_t row = 0; row < n_seg; ++row) {
        size_t b = offsets[row], e = (row + 1 < n_seg) ? (size_t)offsets[row + 1] : n;
        double x = 0;
        for (size_t i = b; i < e; ++i) { out[i] = x; x = x + data[i]; }
    }
}

/* ------------------------------------------------------------------------- */
/* Ray inputs                                                                 */
/* ------------------------------------------------------------------------- */

/* HEALPix nested-scheme pixel centre (Gorski et al. 2005); the reference generates
 * its source-centred ray directions with pix2vec_nest
 * (RayVectorGeneration/src/generateRays.c:57-59).  Validated against oracle/_ref
 * (the reference's chealpix.c compiled as it lies). */
void go_healpix_pix2vec_nest(long nside, long ipix, double* vec)
{
    static const int jrll[12] = { 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4 };
    static const int jpll[12] = { 1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7 };
    const double halfpi = 1.570796326794896619231321691639751442099;
    long npface = nside * nside, npix = 12 * npface;
    int face = (int)(ipix / npface);
    long ipf = ipix % npface;
    long ix = 0, iy = 0;
    for (int b = 0; b < 31; ++b) {
        ix |= ((ipf >> (2 * b)) & 1) << b;
        iy |= ((ipf >> (2 * b + 1)) & 1) << b;
    }
    long nl4 = 4 * nside;
    long jr = jrll[face] * nside - ix - iy - 1;
    double fact2 = 4.0 / npix, fact1 = (nside << 1) * fact2;
    long nr, kshift; double z;
    if (jr < nside) { nr = jr; z = 1.0 - nr * nr * fact2; kshift = 0; }
    else if (jr > 3 * nside) { nr = nl4 - jr; z = nr * nr * fact2 - 1.0; kshift = 0; }
    else { nr = nside; z = (2 * nside - jr) * fact1; kshift = (jr - nside) & 1; }
    long jp = (jpll[face] * nr + ix - iy + 1 + kshift) / 2;
    if (jp > nl4) jp -= nl4;
    if (jp < 1) jp += nl4;
    double phi = (jp - (kshift + 1) * 0.5) * (halfpi / nr);
    double st = sqrt((1.0 - z) * (1.0 + z));
    vec[0] = st * cos(phi); vec[1] = st * sin(phi); vec[2] = z;
}

/* 30-bit direction key used to order isotropic rays
 * (include/grace/cuda/kernels/gen_rays.cuh:38-43). */
uint32_t go_ray_dir_morton_key(const go_ray* r)
{
    return go_morton_key30_unit((r->dx + 1) / 2.f, (r->dy + 1) / 2.f, (r->dz + 1) / 2.f);
}

/* orthographic_projection_rays (include/grace/cuda/kernels/gen_rays.cuh:319-360 and
 * :667-725) specialised by orthogonal_rays_z (tests/helper/rays.cuh:55-79): view
 * direction (0,0,-1), up (0,1,0).  All arithmetic in fp32 as Real = float. */
void go_orthogonal_rays_z(int n_side, const float* mins4, const float* maxs4,
                          go_ray* rays, float* area)
{
    float cx = (float)((mins4[0] + maxs4[0]) / 2.);
    float cy = (float)((mins4[1] + maxs4[1]) / 2.);
    float sx = maxs4[0] - mins4[0] + 2 * maxs4[3];
    float sy = maxs4[1] - mins4[1] + 2 * maxs4[3];
    float sz = maxs4[2] - mins4[2] + 2 * maxs4[3];
    if (sx > sy) sy = sx; else if (sy > sx) sx = sy;
    if (area) *area = (sx / n_side) * (sy / n_side);
    /* camera at (cx, cy, span.z) looking at the box centre: direction (0,0,-1)
     * after normalisation; v = normalize(cross(dir, up)) = (1,0,0),
     * u = normalize(cross(v, dir)) = (0,1,0); both scaled by extent / 2. */
    float cam[3] = { cx, cy, sz };
    float vdx = 0.f, vdy = 0.f, vdz = -1.f;
    float vert = sy, horiz = vert * 1.0f;
    float vx = (float)(1.f * (horiz / 2.));
    float uy = (float)(1.f * (vert / 2.));
    float length = 2 * sz;
    for (int t = 0; t < n_side * n_side; ++t) {
        int i = t % n_side, j = t / n_side;
        float x = (2 * ((i + 0.5f) / n_side) - 1) * 1.0f;
        float y = 1 - 2 * ((j + 0.5f) / n_side);
        go_ray r;
        r.dx = vdx; r.dy = vdy; r.dz = vdz;
        r.ox = cam[0] + (x * vx + y * 0.f + 1.f * 0.f);
        r.oy = cam[1] + (x * 0.f + y * uy + 1.f * 0.f);
        r.oz = cam[2] + (x * 0.f + y * 0.f + 1.f * 0.f);
        r.length = length;
        rays[t] = r;
    }
}

/* ------------------------------------------------------------------------- */
/* Triangle primitive path (tests/profile_trace_triangle)                     */
/* ------------------------------------------------------------------------- */

/* Triangle = {v, e1, e2}, 9 floats (tests/profile_trace_triangle/triangle.cuh:11-25). */
typedef struct { float v[3], e1[3], e2[3]; } go_tri;

/* TriangleCentroid (triangle.cuh:92-102): v + (1./3.) * (e1 + e2); the scalar binds to
 * operator*(float, float3) (tests/helper/vector_math.cuh:34-37), so all fp32. */
static inline void tri_centroid(const go_tri* t, float* c)
{
    const float third = (float)(1. / 3.);
    for (int k = 0; k < 3; ++k) c[k] = t->v[k] + third * (t->e1[k] + t->e2[k]);
}

void go_tri_centroid_bounds(const go_tri* tris, size_t n, float* bot, float* top)
{
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = 0; i < n; ++i) {
        float c[3]; tri_centroid(&tris[i], c);
        for (int k = 0; k < 3; ++k) { if (c[k] < lo[k]) lo[k] = c[k]; if (c[k] > hi[k]) hi[k] = c[k]; }
    }
    for (int k = 0; k < 3; ++k) { bot[k] = lo[k]; top[k] = hi[k]; }
}

/* grace::morton_keys(d_tris, d_keys, TriangleCentroid(), ...) (tris_tree.cuh:27,
 * kernels/morton.cuh:43-50,107-113) with 30-bit keys. */
void go_morton_keys30_tri(const go_tri* tris, size_t n, const float* bot, const float* top,
                          uint32_t* keys)
{
    const int span = (1u << 10) - 1;
    float sx = span / (top[0] - bot[0]), sy = span / (top[1] - bot[1]), sz = span / (top[2] - bot[2]);
    for (size_t i = 0; i < n; ++i) {
        float c[3]; tri_centroid(&tris[i], c);
        keys[i] = go_morton_key30((uint32_t)(sx * (c[0] - bot[0])), (uint32_t)(sy * (c[1] - bot[1])),
                                  (uint32_t)(sz * (c[2] - bot[2])));
    }
}

void go_tri_aabb(const go_tri* t, float* bot, float* top) { prim_aabb((const float*)t, 1, 0, bot, top); }

/* tests/helper/vector_math.cu:27-52: fp64 products and sums, results narrowed to float
 * where the reference assigns to float3 / float. */
static inline double tri_dot(const float* a, const float* b)
{
    double x = (double)a[0] * b[0], y = (double)a[1] * b[1], z = (double)a[2] * b[2];
    return x + y + z;
}
static inline void tri_cross(const float* a, const float* b, float* r)
{
    r[0] = (float)((double)a[1] * b[2] - (double)a[2] * b[1]);
    r[1] = (float)((double)a[2] * b[0] - (double)a[0] * b[2]);
    r[2] = (float)((double)a[0] * b[1] - (double)a[1] * b[0]);
}

#define GO_TRIANGLE_EPSILON 1E-14f

/* Moeller-Trumbore, back faces culled (triangle.cuh:54-88). */
static inline int tri_intersect(const go_ray* ray, const go_tri* tri, float* t)
{
    const float dir[3] = { ray->dx, ray->dy, ray->dz };
    float P[3]; tri_cross(dir, tri->e2, P);
    float det = (float)tri_dot(tri->e1, P);
    if (det < GO_TRIANGLE_EPSILON) return 0;
    float inv_det = (float)(1. / det);
    const float OV[3] = { ray->ox - tri->v[0], ray->oy - tri->v[1], ray->oz - tri->v[2] };
    float u = (float)(tri_dot(OV, P) * inv_det);
    if (u < 0.f || u > 1.f) return 0;
    float Q[3]; tri_cross(OV, tri->e1, Q);
    float v = (float)(tri_dot(dir, Q) * inv_det);
    if (v < 0.f || u + v > 1.f) return 0;
    *t = (float)(tri_dot(tri->e2, Q) * inv_det);
    return 1;
}

int go_tri_intersect(const go_ray* ray, const go_tri* tri, float* t) { return tri_intersect(ray, tri, t); }

/* trace_closest_tri (tris_trace.cu:43-62) by brute force: RayEntry_tri, RayIntersect_tri,
 * OnHit_tri (tris_trace.cuh:11-73) applied to every triangle in index order -- what the
 * packet traversal computes when it is conservative. */
void go_brute_closest_tri(const go_ray* rays, size_t n_rays, const go_tri* tris, size_t n, int* out,
                          float* t_out)
{
    #pragma omp parallel for schedule(dynamic, 16)
    for (size_t ri = 0; ri < n_rays; ++ri) {
        go_ray ray = rays[ri];
        int data = -1;
        float t_min = ray.length * (1.f + GO_TRI_AABB_EPS);
        for (size_t i = 0; i < n; ++i) {
            float t;
            if (tri_intersect(&ray, &tris[i], &t) && t <= t_min && t >= GO_TRIANGLE_EPSILON) {
                t_min = t; data = (int)i;
            }
        }
        out[ri] = data;
        if (t_out) t_out[ri] = t_min;
    }
}

/* pinhole_camera_rays (include/grace/cuda/kernels/gen_rays.cuh:362-395,727-789), Real =
 * float: basis on the host, image_plane_coord + normalisation per ray.  The reference
 * normalises with rnorm3d on the device; here 1/sqrt in fp32 (inputs, not parity-bound). */
void go_pinhole_rays(int res_x, int res_y, const float* cam, const float* look_at, const float* up,
                     float fovy, float length, go_ray* rays)
{
    float vd[3] = { look_at[0] - cam[0], look_at[1] - cam[1], look_at[2] - cam[2] };
    float c1[3] = { vd[1] * up[2] - vd[2] * up[1], vd[2] * up[0] - vd[0] * up[2], vd[0] * up[1] - vd[1] * up[0] };
    double N = 1. / sqrt((double)(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]));
    float v[3] = { (float)(c1[0] * N), (float)(c1[1] * N), (float)(c1[2] * N) };
    float c2[3] = { v[1] * vd[2] - v[2] * vd[1], v[2] * vd[0] - v[0] * vd[2], v[0] * vd[1] - v[1] * vd[0] };
    N = 1. / sqrt((double)(c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2]));
    float u[3] = { (float)(c2[0] * N), (float)(c2[1] * N), (float)(c2[2] * N) };
    N = 1. / sqrt((double)(vd[0] * vd[0] + vd[1] * vd[1] + vd[2] * vd[2]));
    float n[3] = { (float)(vd[0] * N), (float)(vd[1] * N), (float)(vd[2] * N) };
    float pre = (float)(1. / tan(fovy / 2.));
    for (int k = 0; k < 3; ++k) n[k] *= pre;
    float aspect = (float)res_x / res_y;
    for (int tid = 0; tid < res_x * res_y; ++tid) {
        int i = tid % res_x, j = tid / res_x;
        float x = (2 * ((i + 0.5f) / res_x) - 1) * aspect;
        float y = 1 - 2 * ((j + 0.5f) / res_y);
        float d[3];
        for (int k = 0; k < 3; ++k) d[k] = x * v[k] + y * u[k] + 1.f * n[k];
        float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        go_ray r = { d[0] * inv, d[1] * inv, d[2] * inv, cam[0], cam[1], cam[2], length };
        rays[tid] = r;
    }
}

/* orthographic_projection_rays (include/grace/cuda/gen_rays.cuh:264-329;
 * kernels/gen_rays.cuh:319-360,667-725), Real = float: normalize3 / cross as in
 * generic/vecmath.h:9-52 (products in float, norm in double), image_plane_coord with aspect
 * 1 and n = 0 (kernels/gen_rays.cuh:76-95). */
static void go_cross(const float* u, const float* v, float* out)
{
    out[0] = u[1] * v[2] - u[2] * v[1];
    out[1] = u[2] * v[0] - u[0] * v[2];
    out[2] = u[0] * v[1] - u[1] * v[0];
}
static void go_normalize3(float* v)
{
    double N = 1. / sqrt((double)(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
    for (int k = 0; k < 3; ++k) v[k] = (float)(v[k] * N);
}
void go_orthographic_projection_rays(int res_x, int res_y, const float* cam, const float* look_at,
                                     const float* up, float vertical_extent, float length,
                                     go_ray* rays)
{
    float aspect = (float)res_x / res_y;
    float horizontal_extent = vertical_extent * aspect;
    float vd[3] = { look_at[0] - cam[0], look_at[1] - cam[1], look_at[2] - cam[2] };
    go_normalize3(vd);
    float v[3], u[3];
    go_cross(vd, up, v); go_normalize3(v);
    go_cross(v, vd, u); go_normalize3(u);
    for (int k = 0; k < 3; ++k) {
        v[k] = (float)(v[k] * (horizontal_extent / 2.));
        u[k] = (float)(u[k] * (vertical_extent / 2.));
    }
    for (long tid = 0; tid < (long)res_x * res_y; ++tid) {
        int i = (int)(tid % res_x), j = (int)(tid / res_x);
        float x = (2 * ((i + 0.5f) / res_x) - 1) * 1.f;
        float y = 1 - 2 * ((j + 0.5f) / res_y);
        float z = 1.f;
        float p[3];
        for (int k = 0; k < 3; ++k) p[k] = x * v[k] + y * u[k] + z * 0.f;
        go_ray r = { vd[0], vd[1], vd[2], cam[0] + p[0], cam[1] + p[1], cam[2] + p[2], length };
        rays[tid] = r;
    }
}

/* one_to_many_rays_kernel (kernels/gen_rays.cuh:206-243), unsorted: points are `stride`
 * floats (is_double 0) or doubles (1) each.  The reference normalises with rnorm3d on the
 * device; here 1/sqrt in fp32 (ray inputs are not parity-bound, gen_rays.cuh:21-24). */
void go_one_to_many_rays(const void* points, int is_double, int stride, size_t n, float ox,
                         float oy, float oz, go_ray* rays)
{
    for (size_t t = 0; t < n; ++t) {
        float dx, dy, dz;
        if (is_double) {
            const double* q = (const double*)points + t * (size_t)stride;
            dx = (float)(q[0] - ox); dy = (float)(q[1] - oy); dz = (float)(q[2] - oz);
        } else {
            const float* q = (const float*)points + t * (size_t)stride;
            dx = q[0] - ox; dy = q[1] - oy; dz = q[2] - oz;
        }
        float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        go_ray r = { dx * inv, dy * inv, dz * inv, ox, oy, oz, (float)(1.0 / (double)inv) };
        rays[t] = r;
    }
}

/* ---- double4 spheres (Real4 = double4, Real = double; build_sph.cuh:84-126, trace_sph.cuh) -- */

/* DeltaEuclidean on double4 (generic/functors/albvh.h:44-74): differences and products in
 * double, the sum narrowed to the float it returns; +inf at both ends. */
void go_deltas_euclid_d4(const double* s, size_t n, float* out)
{
    out[0] = INFINITY; out[n] = INFINITY;
    for (size_t i = 0; i + 1 < n; ++i) {
        const double* a = s + 4 * i; const double* b = s + 4 * (i + 1);
        out[i + 1] = (float)((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1])
                             + (a[2] - b[2]) * (a[2] - b[2]));
    }
}

/* sphere_hit<double4, double> (generic/intersect.h:9-55): ray members are float, everything
 * else double. */
static inline int sphere_hit_d(const go_ray* ray, const double* s, double* b2, double* dot_p)
{
    double px = s[0] - ray->ox, py = s[1] - ray->oy, pz = s[2] - ray->oz;
    double rx = ray->dx, ry = ray->dy, rz = ray->dz;
    *dot_p = px * rx + py * ry + pz * rz;
    double bx = px - *dot_p * rx, by = py - *dot_p * ry, bz = pz - *dot_p * rz;
    *b2 = bx * bx + by * by + bz * bz;
    if (*b2 >= s[3] * s[3]) return 0;
    if (*dot_p < 0.0f) return 0;
    if (*dot_p >= ray->length) return 0;
    return 1;
}

/* OnHit_sphere_cumulate with Real = double (functors/trace.cuh:164-186) and lerp<double>
 * (generic/interpolate.h:11-39, device branch: fma). */
static inline double hit_integral_d(double b2, double w)
{
    double ir = 1.f / w;
    double x = (GO_N_TABLE - 1) * (sqrt(b2) * ir);
    int x_idx = (int)x;
    if (x_idx >= GO_N_TABLE - 1) { x = (double)(GO_N_TABLE - 1); x_idx = GO_N_TABLE - 2; }
    double y0 = go_table[x_idx], y1 = go_table[x_idx + 1];
    double t = x - x_idx;
    double integral = fma(t, y1 - y0, y0);
    integral *= (ir * ir);
    return integral;
}

void go_brute_hitcounts_d4(const go_ray* rays, size_t n_rays, const double* s, size_t n, int* counts)
{
    #pragma omp parallel for schedule(dynamic, 16)
    for (size_t ri = 0; ri < n_rays; ++ri) {
        go_ray ray = rays[ri];
        int hits = 0; double b2, d;
        for (size_t si = 0; si < n; ++si) hits += sphere_hit_d(&ray, s + 4 * si, &b2, &d);
        counts[ri] = hits;
    }
}

/* Column densities of double4 spheres.  The reference keeps one running double sum per ray in
 * ascending primitive index (RayData_sphere<double,double>); blocks = 1 gives that.  This
 * implementation's stated result (blocks > 1) is the float path's CLASS-ORDERED sum in double: class
 * (p >> 10) & 7, each class summed in ascending primitive order, the 8 class sums added pairwise --
 * so that a packet of double4 rays can be shared by up to 8 waves, like a float one.  The two
 * differ by a few ulp of double (1e-16 relative); the stated tolerance is 1e-5. */
void go_brute_cumulative_d4(const go_ray* rays, size_t n_rays, const double* s, size_t n, double* out, int blocks)
{
    #pragma omp parallel for schedule(dynamic, 16)
    for (size_t ri = 0; ri < n_rays; ++ri) {
        go_ray ray = rays[ri];
        double cls[GO_SUM_CLASSES], b2, d;
        for (int c = 0; c < GO_SUM_CLASSES; ++c) cls[c] = 0.0;
        for (size_t si = 0; si < n; ++si)
            if (sphere_hit_d(&ray, s + 4 * si, &b2, &d)) {
                const int c = blocks > 1 ? (int)((si >> GO_GRANULE_SHIFT) & (GO_SUM_CLASSES - 1)) : 0;
                cls[c] += hit_integral_d(b2, s[4 * si + 3]);
            }
        for (int w = 1; w < GO_SUM_CLASSES; w *= 2)
            for (int c = 0; c < GO_SUM_CLASSES; c += 2 * w) cls[c] = cls[c] + cls[c + w];
        out[ri] = cls[0];
    }
}

