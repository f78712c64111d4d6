// project_gadget over the GPUs of one node from ONE process (single process,
// one stream per device, communicator from ncclCommInitAll): one host thread per rank, each with a
// library context of its own (grace_context_create / grace_context_set_current); particles and BVH
// replicated by a deterministic per-rank build (no communication); rank r traces the contiguous
// ray range [r per, (r + 1) per), per = ceil(N_rays / P) rounded up to 64; one ncclAllGather of
// 4 bytes per ray (RCCL over xGMI) puts the image on every rank.  Rank 0 also traces the whole
// frame and compares: the sharded image must equal it bit for bit.
//
//   project_gadget_sharded <N_rays/32> <max_per_leaf> <gadget file> <out.f32> [ranks] [share]
//     ranks: default = the number of visible devices, rank r on device r.
//     share: every rank on device 0 -- the rehearsal of P > 1 on a one-GPU box: P threads, P
//            contexts, one GPU traced concurrently.  RCCL refuses two ranks on one device, so
//            the gather is then P device-to-device copies into rank 0's buffer.
#include "grace/cuda/build_sph.cuh"
#include "grace/cuda/nodes.h"
#include "grace/cuda/trace_sph.cuh"
#include "grace/cuda/util/extrema.cuh"
#include "grace/ray.h"
#include "helper/tree.cuh"
#include "helper/rays.cuh"
#include "grace/read_gadget.h"

#include <rccl/rccl.h>

#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#define NCCL_CHECK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
    std::fprintf(stderr, "RCCL error %s at %s:%d\n", ncclGetErrorString(r_), __FILE__, __LINE__); std::exit(5); } } while (0)

struct Shared {
    std::vector<float4> h_spheres;
    size_t N_side = 0, N_rays = 0, per = 0;
    int ranks = 1, max_per_leaf = 32;
    bool share = false;
    std::vector<ncclComm_t> comms;
    float* gather_on_rank0 = nullptr;        // share mode: rank 0's [ranks * per] buffer
    std::atomic<int> arrived{0};
    std::vector<float> image;                // rank 0: the gathered frame
    bool equal_to_unsharded = false;
};

static void rank_main(Shared* sh, const int rank)
{
    GRACE_HIP_CHECK(hipSetDevice(sh->share ? 0 : rank));
    grace_context ctx = NULL;
    GRACE_STATUS_CHECK(grace_context_create(&ctx));
    GRACE_STATUS_CHECK(grace_context_set_current(ctx));
    {
        // Replicated scene: every rank sorts and builds for itself -- same input, same tree.
        thrust::device_vector<float4> d_spheres(sh->h_spheres.begin(), sh->h_spheres.end());
        float4 mins, maxs;
        grace::min_vec4(d_spheres, &mins);
        grace::max_vec4(d_spheres, &maxs);
        mins.w = maxs.w = 0;
        grace::Tree d_tree(d_spheres.size(), sh->max_per_leaf);
        build_tree(d_spheres, mins, maxs, d_tree);

        thrust::device_vector<grace::Ray> d_rays;
        orthogonal_rays_z(sh->N_side, mins, maxs, d_rays);

        // This rank's contiguous shard (possibly short or empty at the end).
        const size_t lo = std::min(size_t(rank) * sh->per, sh->N_rays);
        const size_t hi = std::min(lo + sh->per, sh->N_rays);
        thrust::device_vector<grace::Ray> d_my_rays(hi - lo);
        if (hi > lo)
            GRACE_HIP_CHECK(hipMemcpy(thrust::raw_pointer_cast(d_my_rays.data()),
                                      thrust::raw_pointer_cast(d_rays.data()) + lo,
                                      (hi - lo) * sizeof(grace::Ray), hipMemcpyDeviceToDevice));
        thrust::device_vector<float> d_mine(hi - lo);
        grace::trace_cumulative_sph(d_my_rays, d_spheres, d_tree, d_mine);
        d_mine.resize(sh->per);                                    // padded to the gather's slot

        thrust::device_vector<float> d_full(size_t(sh->ranks) * sh->per);
        if (!sh->share) {
            NCCL_CHECK(ncclAllGather(thrust::raw_pointer_cast(d_mine.data()), thrust::raw_pointer_cast(d_full.data()),
                                     sh->per, ncclFloat, sh->comms[rank], 0));
            GRACE_HIP_CHECK(hipStreamSynchronize(0));
        } else {
            if (rank == 0) sh->gather_on_rank0 = thrust::raw_pointer_cast(d_full.data());
            sh->arrived.fetch_add(1);
            while (sh->arrived.load() < sh->ranks) std::this_thread::yield();      // rank 0's buffer exists
            GRACE_HIP_CHECK(hipMemcpy(sh->gather_on_rank0 + size_t(rank) * sh->per,
                                      thrust::raw_pointer_cast(d_mine.data()), sh->per * sizeof(float),
                                      hipMemcpyDeviceToDevice));
            sh->arrived.fetch_add(1);
            while (sh->arrived.load() < 2 * sh->ranks) std::this_thread::yield();  // every copy has landed
        }
        if (rank == 0) {
            sh->image.resize(sh->N_rays);
            GRACE_HIP_CHECK(hipMemcpy(sh->image.data(), thrust::raw_pointer_cast(d_full.data()),
                                      sh->N_rays * sizeof(float), hipMemcpyDeviceToHost));
            thrust::device_vector<float> d_whole(sh->N_rays);
            grace::trace_cumulative_sph(d_rays, d_spheres, d_tree, d_whole);
            thrust::host_vector<float> h_whole = d_whole;
            sh->equal_to_unsharded =
                std::memcmp(h_whole.data(), sh->image.data(), sh->N_rays * sizeof(float)) == 0;
        }
        if (sh->share) {   // rank 0's buffer must outlive the other ranks' copies: leave together
            sh->arrived.fetch_add(1);
            while (sh->arrived.load() < 3 * sh->ranks) std::this_thread::yield();
        }
    }
    GRACE_STATUS_CHECK(grace_context_set_current(NULL));
    GRACE_STATUS_CHECK(grace_context_destroy(ctx));
}

int main(int argc, char* argv[])
{
    if (argc < 5) { std::cerr << "usage: N_rays/32 max_per_leaf gadget_file out.f32 [ranks] [share]\n"; return 2; }
    Shared sh;
    size_t N_rays = 32 * size_t(std::strtol(argv[1], NULL, 10));
    sh.max_per_leaf = int(std::strtol(argv[2], NULL, 10));
    sh.N_side = size_t(std::floor(std::pow(double(N_rays), 0.500001)));
    sh.N_side = ((sh.N_side + 31) / 32) * 32;
    sh.N_rays = sh.N_side * sh.N_side;
    int n_dev = 0;
    GRACE_HIP_CHECK(hipGetDeviceCount(&n_dev));
    sh.ranks = argc > 5 ? int(std::strtol(argv[5], NULL, 10)) : n_dev;
    sh.share = argc > 6 && std::string(argv[6]) == "share";
    if (sh.ranks < 1 || (!sh.share && sh.ranks > n_dev)) { std::cerr << "bad rank count\n"; return 2; }
    sh.per = ((sh.N_rays + sh.ranks - 1) / sh.ranks + 63) / 64 * 64;
    read_gadget(argv[3], sh.h_spheres);
    std::cout << "Number of particles:     " << sh.h_spheres.size() << std::endl
              << "Number of rays:          " << sh.N_rays << std::endl
              << "Ranks:                   " << sh.ranks << (sh.share ? " (sharing device 0)" : "") << std::endl
              << "Rays per rank:           " << sh.per << std::endl;

    if (!sh.share) {
        std::vector<int> devs(sh.ranks);
        for (int r = 0; r < sh.ranks; ++r) devs[r] = r;
        sh.comms.resize(sh.ranks);
        NCCL_CHECK(ncclCommInitAll(sh.comms.data(), sh.ranks, devs.data()));
    }
    std::vector<std::thread> threads;
    for (int r = 0; r < sh.ranks; ++r) threads.emplace_back(rank_main, &sh, r);
    for (auto& t : threads) t.join();
    for (auto& c : sh.comms) NCCL_CHECK(ncclCommDestroy(c));

    double sum = 0.0;
    for (float v : sh.image) sum += v;
    std::cout << "Mean output " << sum / double(sh.N_rays) << std::endl
              << (sh.equal_to_unsharded ? "sharded image == unsharded image: PASSED" : "sharded image DIFFERS: FAILED")
              << std::endl;
    std::FILE* f = std::fopen(argv[4], "wb");
    if (!f || std::fwrite(sh.image.data(), sizeof(float), sh.N_rays, f) != sh.N_rays) return 3;
    std::fclose(f);
    return sh.equal_to_unsharded ? EXIT_SUCCESS : 4;
}
