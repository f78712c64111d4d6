// read_gadget.h -- Gadget-2 (format 1) reader for gas positions + smoothing lengths,
// the role of the reference's tests/helper/read_gadget.cuh:69-159 (block order POS, VEL, ID,
// [MASS], U, RHO, HSML; 256-byte header with npart[6], mass[6]; 4-byte block markers).
// Host-only; fills spheres {x, y, z, h} (any float4-like record with .x .y .z .w: HIP's float4
// for the thrust drop-in headers, grace::float4 for the HIP-free mirror).
#pragma once

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

template <typename Float4>
inline void read_gadget(const std::string& fname, std::vector<Float4>& h_spheres)
{
    std::FILE* f = std::fopen(fname.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open Gadget file " + fname);
    auto fail = [&](const char* what) {
        std::fclose(f);
        throw std::runtime_error(std::string("Gadget file ") + fname + ": " + what);
    };
    auto read_marker = [&]() {
        int32_t m = 0;
        if (std::fread(&m, 4, 1, f) != 1) fail("truncated block marker");
        return m;
    };
    auto skip_block = [&]() {
        const int32_t m = read_marker();
        if (std::fseek(f, m, SEEK_CUR) != 0) fail("truncated block");
        read_marker();
    };

    int32_t npart[6];
    double mass[6];
    read_marker();
    if (std::fread(npart, 4, 6, f) != 6 || std::fread(mass, 8, 6, f) != 6) fail("truncated header");
    std::fseek(f, 256 - 6 * 4 - 6 * 8, SEEK_CUR);
    read_marker();

    const int n_gas = npart[0];
    if (n_gas == 0) fail("has no gas particles!");
    long n_withmass = 0;
    for (int i = 0; i < 6; ++i)
        if (mass[i] == 0) n_withmass += npart[i];

    h_spheres.resize(n_gas);
    std::vector<float> buf(3 * size_t(n_gas));
    read_marker();                                    // POS: gas particles come first
    if (std::fread(buf.data(), 4, buf.size(), f) != buf.size()) fail("truncated POS block");
    long rest = 0;
    for (int i = 1; i < 6; ++i) rest += npart[i];
    std::fseek(f, rest * 12, SEEK_CUR);
    read_marker();
    for (int n = 0; n < n_gas; ++n) {
        h_spheres[n].x = buf[3 * n]; h_spheres[n].y = buf[3 * n + 1]; h_spheres[n].z = buf[3 * n + 2];
    }
    skip_block();                                     // VEL
    skip_block();                                     // ID
    if (n_withmass > 0) skip_block();                 // MASS
    skip_block();                                     // U
    skip_block();                                     // RHO
    read_marker();                                    // HSML
    buf.resize(n_gas);
    if (std::fread(buf.data(), 4, buf.size(), f) != buf.size()) fail("truncated HSML block");
    for (int n = 0; n < n_gas; ++n) h_spheres[n].w = buf[n];
    std::fclose(f);
}
