// The host path of grace::morton_key (BASELINE config 1, the reference's tests/morton_key):
// prints, for the integers x y z given on the command line, the spaced co-ordinates and the key --
// 30-bit forms for "30", 63-bit forms for "63" -- and the key of three reals in (0, 1) for
// "f" / "d".  tests/test_capi_cpu.py feeds it the reference's known-answer inputs
// (tests/golden/kat.json <- tests/morton_key/30bit_key.cu:20-26, 63bit_key.cu:20-26) and compares.
// Builds with plain g++ and with hipcc: the product headers are host-callable.
#include "grace/generic/bits.h"
#include "grace/generic/morton.h"

#include <cstdlib>
#include <iostream>
#include <string>

int main(int argc, char* argv[])
{
    if (argc != 5) {
        std::cerr << "usage: " << argv[0] << " 30|63|f|d x y z" << std::endl;
        return EXIT_FAILURE;
    }
    const std::string mode = argv[1];
    if (mode == "30") {
        const grace::uinteger32 x = std::strtoul(argv[2], NULL, 10), y = std::strtoul(argv[3], NULL, 10),
                                z = std::strtoul(argv[4], NULL, 10);
        std::cout << grace::detail::space_by_two_10bit(x) << " " << grace::detail::space_by_two_10bit(y)
                  << " " << grace::detail::space_by_two_10bit(z) << " " << grace::morton_key(x, y, z)
                  << std::endl;
    } else if (mode == "63") {
        const grace::uinteger64 x = std::strtoull(argv[2], NULL, 10), y = std::strtoull(argv[3], NULL, 10),
                                z = std::strtoull(argv[4], NULL, 10);
        std::cout << grace::detail::space_by_two_21bit(x) << " " << grace::detail::space_by_two_21bit(y)
                  << " " << grace::detail::space_by_two_21bit(z) << " " << grace::morton_key(x, y, z)
                  << std::endl;
    } else if (mode == "f") {
        std::cout << grace::morton_key(std::strtof(argv[2], NULL), std::strtof(argv[3], NULL),
                                       std::strtof(argv[4], NULL)) << std::endl;
    } else if (mode == "d") {
        std::cout << grace::morton_key(std::strtod(argv[2], NULL), std::strtod(argv[3], NULL),
                                       std::strtod(argv[4], NULL)) << std::endl;
    } else {
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
