"""CPU tests of the boundary: the C-ABI library loads without a GPU and exports every
symbol include/grace_hip.h declares; the HIP-free C++ header mirror compiles with plain g++;
the product path has no fallback; the N > 1 sharding logic runs under gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "grace-devel_amd", "lib", "libgrace_hip.so")


def _declared():
    hdr = open(os.path.join(ROOT, "include", "grace_hip.h")).read()
    return sorted(set(re.findall(r"\b(grace_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(LIB)
    names = _declared()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.grace_version.restype = ctypes.c_int
    assert lib.grace_version() >= 100          # no compute call: there is no GPU here


def test_every_entry_point_cites_the_reference():
    hdr = open(os.path.join(ROOT, "include", "grace_hip.h")).read()
    # each block of declarations is preceded by a comment naming a reference file:line
    assert len(re.findall(r"(cuh|\.h|\.cu|\.c):\d+", hdr)) >= 25


def test_cpp_header_mirror_is_hip_free(tmp_path):
    """include/grace/grace.h + tests/cpp/tree_traversal.cpp build with plain g++."""
    exe = tmp_path / "tree_traversal"
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "tree_traversal.cpp"), "-o", str(exe),
           "-L" + os.path.dirname(LIB), "-lgrace_hip", "-L" + os.path.join(ROOT, "oracle"),
           "-lgrace_oracle", "-Wl,-rpath," + os.path.dirname(LIB),
           "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.check_call(cmd)
    assert exe.exists()
    # the generator / double4 mirrors too (templates are only checked when instantiated)
    exe2 = tmp_path / "ray_generators"
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror",
                           "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "ray_generators.cpp"), "-o", str(exe2),
                           "-L" + os.path.dirname(LIB), "-lgrace_hip",
                           "-Wl,-rpath," + os.path.dirname(LIB)])
    assert exe2.exists()


def test_reference_header_paths_compile_with_hipcc(tmp_path):
    """The drop-in header set (include/grace/cuda/*.cuh, nodes.h, ray.h, types.h ... over
    thrust::device_vector) compiles for gfx950 with hipcc: three callers written against the
    reference's include lines (run on the GPU by tests/test_gpu_dropin.py)."""
    for name in ("dropin_project_gadget", "dropin_tree_traversal", "dropin_types", "dropin_triangles",
                 "dropin_generic"):
        exe = tmp_path / name
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17",
                               "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "tests", "cpp"),
                               os.path.join(ROOT, "tests", "cpp", name + ".hip"), "-o", str(exe),
                               "-L" + os.path.dirname(LIB), "-lgrace_hip",
                               "-Wl,-rpath," + os.path.dirname(LIB)])
        assert exe.exists()


def test_single_process_sharded_program_compiles_against_rccl(tmp_path):
    """tests/cpp/project_gadget_sharded.hip: one process, one thread + one library context per
    rank, ncclCommInitAll + ncclAllGather of 4 B/ray -- compiles against
    /opt/rocm/include/rccl/rccl.h and links librccl (run on the GPU by tests/test_gpu_round3.py)."""
    exe = tmp_path / "project_gadget_sharded"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17",
                           "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tests", "cpp"),
                           os.path.join(ROOT, "tests", "cpp", "project_gadget_sharded.hip"), "-o", str(exe),
                           "-L" + os.path.dirname(LIB), "-lgrace_hip", "-L/opt/rocm/lib", "-lrccl", "-pthread",
                           "-Wl,-rpath," + os.path.dirname(LIB)])
    assert exe.exists()


def test_host_morton_key_known_answers(tmp_path, oracle):
    """BASELINE config 1 (tests/morton_key: the CPU host path) on the PRODUCT headers
    include/grace/generic/{bits,morton}.h, built with plain g++: the reference's own known-answer
    vectors (tests/golden/kat.json <- tests/morton_key/30bit_key.cu:20-26, 63bit_key.cu:20-26),
    and the float / double forms against the oracle's keys on random points."""
    import json
    exe = str(tmp_path / "morton_key_kat")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "morton_key_kat.cpp"), "-o", exe])
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))
    for bits in ("30", "63"):
        k = kat["morton" + bits]
        out = subprocess.check_output([exe, bits, str(k["x"]), str(k["y"]), str(k["z"])], text=True).split()
        assert [int(v) for v in out] == [k["spaced_x"], k["spaced_y"], k["spaced_z"], k["key"]]
    # high bits beyond 10 / 21 are masked away (generic/bits.h:27,38)
    out = subprocess.check_output([exe, "30", str(309 + 1024 * 5), "942", "619"], text=True).split()
    assert int(out[3]) == kat["morton30"]["key"]
    rng = np.random.default_rng(5)
    pts = rng.random((64, 3))
    p4 = np.concatenate([pts, np.zeros((64, 1))], 1).astype(np.float32)
    ref30 = oracle.morton_keys30(p4, (0, 0, 0), (1, 1, 1))
    for i in range(64):
        x, y, z = (repr(float(v)) for v in p4[i, :3])
        assert int(subprocess.check_output([exe, "f", x, y, z], text=True)) == int(ref30[i])
    # morton_key(double, double, double): span * x in double, truncated, interleaved -- against
    # Python integers
    def spread(v, n):
        return sum(((v >> b) & 1) << (3 * b) for b in range(n))
    for i in range(16):
        x, y, z = (float(v) for v in pts[i])
        want = (spread(int(2097151 * z), 21) << 2) | (spread(int(2097151 * y), 21) << 1) | spread(int(2097151 * x), 21)
        assert int(subprocess.check_output([exe, "d", repr(x), repr(y), repr(z)], text=True)) == want


def test_product_path_does_not_touch_the_oracle():
    """No file of the product (package, headers, C ABI sources) mentions the oracle."""
    bad = []
    for base in ("grace-devel_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            if os.sep + "build" in d or os.sep + "lib" in d or "__pycache__" in d:
                continue
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".cuh")):
                    txt = open(os.path.join(d, f), errors="ignore").read()
                    if re.search(r"grace_oracle|import oracle|from oracle|go_[a-z]+\(", txt):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_binding_fails_loudly_without_the_library(tmp_path):
    """A copy of the binding with no lib/ next to it must refuse to import."""
    pkg = tmp_path / "pkg" / "grace_hip"
    pkg.mkdir(parents=True)
    src = os.path.join(ROOT, "grace-devel_amd", "grace_hip", "__init__.py")
    (pkg / "__init__.py").write_text(open(src).read())
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import grace_hip"
                        % str(tmp_path / "pkg")], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr.replace("NO CPU", "no CPU")


def test_shard_bounds_cover_every_ray_once():
    sys.path.insert(0, os.path.join(ROOT, "grace-devel_amd"))
    from grace_hip import sharding
    for n in (32, 64, 100000, 1024 * 1024, 1024 * 1024 + 32):
        for world in (1, 2, 3, 4, 8):
            per = sharding.shard_size(n, world)
            assert per % 64 == 0 and per * world >= n
            covered = np.zeros(n, np.int32)
            for r in range(world):
                lo, hi = sharding.shard_bounds(n, world, r)
                assert lo == min(r * per, n) and hi - lo <= per
                covered[lo:hi] += 1
            assert np.all(covered == 1)


def _gloo_worker(rank, world, n_rays, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "grace-devel_amd"))
    from grace_hip import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = sharding.shard_size(n_rays, world)
    lo, hi = sharding.shard_bounds(n_rays, world, rank)
    mine = torch.full((per,), -1.0)
    # stands in for trace_cumulative_sph on this rank's rays: a function of the ray index
    mine[: hi - lo] = torch.arange(lo, hi, dtype=torch.float32) * 0.5 + 1.0
    full = sharding.gather_results(mine, n_rays, world, dist)
    ok = bool(torch.equal(full, torch.arange(n_rays, dtype=torch.float32) * 0.5 + 1.0))
    # the double-buffered pipeline bench.py uses for N > 1: five steps, each step's results
    # (a function of ray index and step) gathered while the next step fills the other buffer
    pipe = sharding.GatherPipeline(per, world, n_rays, dist, "cpu")
    outs = []
    for k in range(5):
        buf = pipe.buffer(k)
        buf.fill_(-1.0)
        buf[: hi - lo] = torch.arange(lo, hi, dtype=torch.float32) * 0.5 + 1.0 + k
        outs.append((k, pipe.gather(k)))
        if k >= 1:      # step k - 1's gather may be read once step k + 1 asks for its buffer again
            pipe.buffer(k + 1)
            ok = ok and bool(torch.equal(outs[k - 1][1], torch.arange(n_rays, dtype=torch.float32) * 0.5 + 1.0 + (k - 1)))
    pipe.drain()
    ok = ok and bool(torch.equal(outs[4][1], torch.arange(n_rays, dtype=torch.float32) * 0.5 + 1.0 + 4))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("n_rays", [4096, 100000])
def test_ray_sharding_and_gather_under_gloo(n_rays):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_rays) % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, n_rays, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_gadget_roundtrip_python_and_cpp(tmp_path):
    """Synthetic Gadget-2 file (layout of tests/helper/read_gadget.cuh): Python writer ->
    Python reader and the C++ header reader (with and without a MASS block)."""
    sys.path.insert(0, os.path.join(ROOT, "grace-devel_amd"))
    from grace_hip import gadget
    rng = np.random.default_rng(3)
    pos = rng.random((4096, 3), dtype=np.float32); h = (0.01 + 0.02 * rng.random(4096)).astype(np.float32)
    exe = tmp_path / "rg"
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "read_gadget_check.cpp"), "-o", str(exe),
                           "-L" + os.path.dirname(LIB), "-lgrace_hip", "-Wl,-rpath," + os.path.dirname(LIB)])
    for in_header in (True, False):
        fname = str(tmp_path / ("snap_%d" % in_header))
        gadget.write_gadget(fname, pos, h, masses_in_header=in_header)
        s = gadget.read_gadget(fname)
        assert np.array_equal(s[:, :3], pos) and np.array_equal(s[:, 3], h)
        out = subprocess.check_output([str(exe), fname], text=True).split()
        assert int(out[0]) == 4096
        sums = [float(x) for x in out[1:5]]
        ref = [float(pos[:, k].astype(np.float64).sum()) for k in range(3)] + [float(h.astype(np.float64).sum())]
        assert np.allclose(sums, ref, rtol=1e-12)


def test_weak_scaling_shard_is_one_frame():
    """bench.py --scaling weak: the job is `world` frames of side^2 rays; each rank's contiguous
    shard must be exactly its own frame (side^2 is a multiple of 64)."""
    from grace_hip import sharding
    frame = 1024 * 1024
    for world in (1, 2, 4, 8):
        n = world * frame
        assert sharding.shard_size(n, world) == frame
        for r in range(world):
            assert sharding.shard_bounds(n, world, r) == (r * frame, (r + 1) * frame)
