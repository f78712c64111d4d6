// Component-wise minima / maxima of device arrays of small vectors, for gfx950.
//
// Replaces the thrust::minmax_element / thrust::reduce calls behind the reference's
// grace::min_max_x/y/z/w and min_vec2/3/4, max_vec2/3/4 (include/grace/cuda/util/extrema.cuh:
// 190-772) for every vector type whose components are float, double, int or unsigned int:
// the header include/grace/cuda/util/extrema.cuh passes the component type, the number of leading
// components wanted and the record stride.  One streaming pass (HBM-bound: stride bytes per
// record, read once) leaves a partial result per workgroup; a single workgroup folds those.
// min / max only, so the values are exactly the reference's (a NaN component is skipped here
// where thrust's comparison-based reduction would propagate an order-dependent result).
#include "common.hpp"

using namespace grace_hip;

namespace {

constexpr int EXT_BLOCK = 256;
constexpr int EXT_MAX_BLOCKS = 1024;

template <typename T> struct Limits;
template <> struct Limits<float> {
    static __device__ float lowest() { return -INFINITY; }
    static __device__ float highest() { return INFINITY; }
};
template <> struct Limits<double> {
    static __device__ double lowest() { return -double(INFINITY); }
    static __device__ double highest() { return double(INFINITY); }
};
template <> struct Limits<int> {
    static __device__ int lowest() { return int(0x80000000u); }
    static __device__ int highest() { return 0x7fffffff; }
};
template <> struct Limits<unsigned int> {
    static __device__ unsigned int lowest() { return 0u; }
    static __device__ unsigned int highest() { return 0xffffffffu; }
};

template <typename T>
__device__ __forceinline__ T pick_min(const T a, const T b) { return b < a ? b : a; }
template <typename T>
__device__ __forceinline__ T pick_max(const T a, const T b) { return a < b ? b : a; }

// Folds (lo, hi) over the workgroup; thread 0 ends up with the result.
template <typename T, int NC>
__device__ __forceinline__ void block_fold(T (&lo)[NC], T (&hi)[NC])
{
#pragma unroll
    for (int k = 0; k < NC; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = pick_min(lo[k], __shfl_xor(lo[k], off));
            hi[k] = pick_max(hi[k], __shfl_xor(hi[k], off));
        }
    }
    __shared__ T s_lo[EXT_BLOCK / 64][NC], s_hi[EXT_BLOCK / 64][NC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NC; ++k) { s_lo[wave][k] = lo[k]; s_hi[wave][k] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < EXT_BLOCK / 64; ++w) {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                lo[k] = pick_min(lo[k], s_lo[w][k]);
                hi[k] = pick_max(hi[k], s_hi[w][k]);
            }
        }
    }
}

// partial: [gridDim.x][2 * NC] = {mins, maxs} per workgroup.
template <typename T, int NC>
__global__ __launch_bounds__(EXT_BLOCK) void extrema_kernel(const char* __restrict__ data, size_t n,
                                                           size_t stride, T* __restrict__ partial)
{
    T lo[NC], hi[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { lo[k] = Limits<T>::highest(); hi[k] = Limits<T>::lowest(); }
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x) {
        const T* q = reinterpret_cast<const T*>(data + i * stride);
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const T v = q[k];
            lo[k] = pick_min(lo[k], v);
            hi[k] = pick_max(hi[k], v);
        }
    }
    block_fold<T, NC>(lo, hi);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            partial[size_t(blockIdx.x) * 2 * NC + k] = lo[k];
            partial[size_t(blockIdx.x) * 2 * NC + NC + k] = hi[k];
        }
    }
}

template <typename T, int NC>
__global__ __launch_bounds__(EXT_BLOCK) void extrema_fold_kernel(const T* __restrict__ partial,
                                                                int n_partial, T* __restrict__ out)
{
    T lo[NC], hi[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { lo[k] = Limits<T>::highest(); hi[k] = Limits<T>::lowest(); }
    for (int b = threadIdx.x; b < n_partial; b += blockDim.x) {
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            lo[k] = pick_min(lo[k], partial[size_t(b) * 2 * NC + k]);
            hi[k] = pick_max(hi[k], partial[size_t(b) * 2 * NC + NC + k]);
        }
    }
    block_fold<T, NC>(lo, hi);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NC; ++k) { out[k] = lo[k]; out[NC + k] = hi[k]; }
    }
}

template <typename T, int NC>
grace_status extrema_nc(const void* d_data, size_t n, size_t stride, void* h_mins, void* h_maxs,
                        hipStream_t stream)
{
    int grid = stream_grid(n, EXT_BLOCK, 4);
    if (grid > EXT_MAX_BLOCKS) grid = EXT_MAX_BLOCKS;
    FrameGuard frame;
    GRACE_TRY(frame.begin(Workspace::aligned(size_t(grid) * 2 * NC * sizeof(T)) + 512, stream));
    T* partial = Workspace::take<T>(size_t(grid) * 2 * NC);
    T* out = Workspace::take<T>(2 * NC);
    extrema_kernel<T, NC><<<grid, EXT_BLOCK, 0, stream>>>(static_cast<const char*>(d_data), n, stride,
                                                         partial);
    GRACE_CHECK_LAUNCH();
    extrema_fold_kernel<T, NC><<<1, EXT_BLOCK, 0, stream>>>(partial, grid, out);
    GRACE_CHECK_LAUNCH();
    T h[2 * NC];
    GRACE_TRY_HIP(hipMemcpyAsync(h, out, sizeof(h), hipMemcpyDeviceToHost, stream));
    GRACE_TRY_HIP(hipStreamSynchronize(stream));
    for (int k = 0; k < NC; ++k) {
        static_cast<T*>(h_mins)[k] = h[k];
        static_cast<T*>(h_maxs)[k] = h[NC + k];
    }
    return GRACE_OK;
}

template <typename T>
grace_status extrema_t(const void* d_data, size_t n, int n_comp, size_t stride, void* h_mins,
                       void* h_maxs, hipStream_t stream)
{
    switch (n_comp) {
    case 1: return extrema_nc<T, 1>(d_data, n, stride, h_mins, h_maxs, stream);
    case 2: return extrema_nc<T, 2>(d_data, n, stride, h_mins, h_maxs, stream);
    case 3: return extrema_nc<T, 3>(d_data, n, stride, h_mins, h_maxs, stream);
    default: return extrema_nc<T, 4>(d_data, n, stride, h_mins, h_maxs, stream);
    }
}

} // namespace

extern "C" {

grace_status grace_minmax_components(const void* d_data, size_t n, int elem_type, int n_comp,
                                     size_t stride_bytes, void* h_mins, void* h_maxs,
                                     grace_stream stream)
{
    GRACE_REQUIRE(d_data && n > 0, "min_max: empty input");
    GRACE_REQUIRE(h_mins && h_maxs, "min_max: null output");
    GRACE_REQUIRE(n_comp >= 1 && n_comp <= 4, "min_max: 1 to 4 components");
    const size_t elem = elem_type == GRACE_ELEM_F64 ? 8 : 4;
    GRACE_REQUIRE(stride_bytes >= elem * size_t(n_comp) && stride_bytes % elem == 0,
                  "min_max: record stride must hold the components and be a multiple of their size");
    hipStream_t s = as_stream(stream);
    switch (elem_type) {
    case GRACE_ELEM_F32: return extrema_t<float>(d_data, n, n_comp, stride_bytes, h_mins, h_maxs, s);
    case GRACE_ELEM_F64: return extrema_t<double>(d_data, n, n_comp, stride_bytes, h_mins, h_maxs, s);
    case GRACE_ELEM_I32: return extrema_t<int>(d_data, n, n_comp, stride_bytes, h_mins, h_maxs, s);
    case GRACE_ELEM_U32: return extrema_t<unsigned int>(d_data, n, n_comp, stride_bytes, h_mins, h_maxs, s);
    default: break;
    }
    return set_error(GRACE_INVALID_ARGUMENT, __FILE__, __LINE__, "min_max: unknown element type");
}

} // extern "C"
