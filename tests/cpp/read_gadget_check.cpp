// Reads a Gadget-2 file with include/grace/read_gadget.h and prints N and column sums so
// that the pytest harness can compare them with the Python writer's inputs.
#include "grace/grace.h"
#include "grace/read_gadget.h"

#include <cstdio>

int main(int argc, char* argv[])
{
    if (argc < 2) return 2;
    std::vector<grace::float4> s;
    read_gadget(argv[1], s);
    double sx = 0, sy = 0, sz = 0, sw = 0;
    for (size_t i = 0; i < s.size(); ++i) { sx += s[i].x; sy += s[i].y; sz += s[i].z; sw += s[i].w; }
    std::printf("%zu %.17g %.17g %.17g %.17g %.9g %.9g\n", s.size(), sx, sy, sz, sw, s.front().x, s.back().w);
    return 0;
}
