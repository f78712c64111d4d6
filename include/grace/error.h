// grace/error.h -- error behaviour of the reference (include/grace/error.h:35-64) over the
// status codes of the C ABI: an invalid argument throws std::invalid_argument where the
// reference throws it (bintree_trace.cuh:231-238, albvh.cuh:795-799, gen_rays.cuh:124-129);
// any other failure prints the message and exit()s with the code, like GRACE_CUDA_CHECK.
#pragma once

#include "grace/types.h"
#include "grace_hip.h"

#include <assert.h>
#include <cstdlib>
#include <iostream>
#include <stdexcept>

#ifdef GRACE_DEBUG
#define GRACE_ASSERT(...) { assert((__VA_ARGS__)); }
#else
#define GRACE_ASSERT(...)
#endif

#define GRACE_GOT_TO() std::cerr << "At " << __FILE__ << "@" << __LINE__ << std::endl;

// Wrap around all calls into libgrace_hip.so / HIP to handle errors.
#define GRACE_HIP_CHECK(code) { grace::hip_error_check((code), __FILE__, __LINE__); }
#define GRACE_STATUS_CHECK(status) { grace::status_check((status), __FILE__, __LINE__); }

namespace grace {

GRACE_HOST void hip_error_check(hipError_t code, const char* file, int line, bool terminate = true)
{
    if (code != hipSuccess) {
        std::cerr << "**** GRACE HIP Error ****" << std::endl
                  << "File:  " << file << std::endl
                  << "Line:  " << line << std::endl
                  << "Error: " << hipGetErrorString(code) << std::endl;
        if (terminate)
            exit(code);
    }
}

GRACE_HOST void status_check(grace_status status, const char* file, int line)
{
    if (status == GRACE_OK)
        return;
    if (status == GRACE_INVALID_ARGUMENT)
        throw std::invalid_argument(grace_last_error());
    std::cerr << "**** GRACE HIP Error ****" << std::endl
              << "File:  " << file << std::endl
              << "Line:  " << line << std::endl
              << "Error: " << grace_last_error() << std::endl;
    exit(int(status));
}

} // namespace grace
