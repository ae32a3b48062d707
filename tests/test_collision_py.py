"""Class-level tests of Collider on the GPU; mirrors tests/test_collision_py.py of the reference
(test_collision :49-97, test_random_collision :100-150, _resized :153-207, test_auto_program
:210-258, test_count_only :261-296) with the PyOpenCL plumbing replaced by collision_amd.hip."""
import numpy as np
import pytest

from collision_amd.collision import Collider, CollisionProgram
from tests.util import pair_set, run_collider

pytestmark = pytest.mark.gpu

DTYPES = [np.dtype("float32"), np.dtype("float64")]


@pytest.fixture(scope="module", params=DTYPES, ids=str)
def coord_dtype(request):
    return request.param


@pytest.fixture(scope="module")
def collision_programs(hip_env, coord_dtype):
    from collision_amd.bounds import BoundsProgram
    from collision_amd.radix import PrefixScanProgram, RadixProgram
    ctx, cq = hip_env
    return (CollisionProgram(ctx, coord_dtype), (RadixProgram(ctx), PrefixScanProgram(ctx)),
            BoundsProgram(ctx, (coord_dtype, 3)))


def _scene(generated, coord_dtype, size):
    key = "scene_%s_%d" % (coord_dtype.name, size)
    return generated[key + "_coords"], generated[key + "_radii"], pair_set(generated[key + "_pairs"])


def test_collision(hip_env, coord_dtype, collision_programs, vectors):
    ctx, cq = hip_env
    v = vectors["six_sphere_scene"]
    coords = np.array(v["coords"], dtype=coord_dtype)
    radii = np.array(v["radii"], dtype=coord_dtype)
    expected = pair_set(v["expected_pairs"])
    collider = Collider(ctx, len(coords), 3, 8, coord_dtype, *collision_programs)
    count, pairs = run_collider(ctx, cq, collider, coords, radii, len(expected))
    assert count == len(expected)
    assert pair_set(pairs) == expected          # orientation matters: no per-pair sort


@pytest.mark.parametrize("size,ngroups,group_size", [(120, 5, 8), (256, 4, 32), (317, 4, 16), (341, 4, 64)])
def test_random_collision(hip_env, coord_dtype, collision_programs, generated, size, ngroups, group_size):
    ctx, cq = hip_env
    coords, radii, expected = _scene(generated, coord_dtype, size)     # reference's find_collisions
    collider = Collider(ctx, size, ngroups, group_size, coord_dtype, *collision_programs)
    count, pairs = run_collider(ctx, cq, collider, coords, radii, len(expected))
    assert count == len(expected)
    assert pair_set(np.sort(pairs, axis=1)) == expected


@pytest.mark.parametrize("old_shape,new_shape", [((350, 8, 64), (351, 8, 64)), ((350, 8, 64), (351, None, None))])
def test_random_collision_resized(hip_env, coord_dtype, collision_programs, generated, old_shape, new_shape):
    ctx, cq = hip_env
    collider = Collider(ctx, *old_shape, coord_dtype, *collision_programs)
    # use it once at the old size so that the resize really has buffers to replace
    coords, radii, _ = _scene(generated, coord_dtype, 341)
    run_collider(ctx, cq, Collider(ctx, 341, 8, 64, coord_dtype, *collision_programs), coords, radii, 0)
    collider.resize(*new_shape)
    coords, radii, expected = _scene(generated, coord_dtype, 351)
    count, pairs = run_collider(ctx, cq, collider, coords, radii, len(expected))
    assert count == len(expected)
    assert pair_set(np.sort(pairs, axis=1)) == expected


def test_auto_program(hip_env, coord_dtype, generated):
    ctx, cq = hip_env
    coords, radii, expected = _scene(generated, coord_dtype, 8)
    collider = Collider(ctx, 8, 1, 8, coord_dtype)
    count, pairs = run_collider(ctx, cq, collider, coords, radii, max(len(expected), 1))
    assert count == len(expected)
    assert pair_set(np.sort(pairs, axis=1)) == expected


def test_count_only(hip_env, coord_dtype, collision_programs, generated):
    ctx, cq = hip_env
    coords, radii, expected = _scene(generated, coord_dtype, 100)
    collider = Collider(ctx, 100, 10, 8, coord_dtype, *collision_programs)
    count, _ = run_collider(ctx, cq, collider, coords, radii, 0)
    assert count == len(expected)


def test_repeated_calls_and_capacity_overflow(hip_env, coord_dtype, collision_programs, generated):
    # collision.cl:203-207: the counter keeps counting past the capacity; stored pairs are a subset
    ctx, cq = hip_env
    coords, radii, expected = _scene(generated, coord_dtype, 341)
    collider = Collider(ctx, 341, 4, 64, coord_dtype, *collision_programs)
    for cap in (len(expected) // 2, len(expected), len(expected) + 10):
        count, pairs = run_collider(ctx, cq, collider, coords, radii, cap)
        assert count == len(expected)
        got = pair_set(np.sort(pairs, axis=1))
        assert len(got) == min(cap, len(expected)) and got <= expected


def test_get_collisions_survives_graph_capture_and_replay(hip_env, oracle):
    """The whole path captured once into a hipGraph (torch.cuda.graph on the launch stream) and
    replayed: every replay must give the eager / oracle pair set.  20 k spheres, so the radix
    histogram spans several scan tiles (the single-launch scan must step aside under capture)."""
    torch = pytest.importorskip("torch")
    ctx, _ = hip_env
    from collision_amd import hip
    from tests.util import pad4
    n, cap = 20000, 1 << 16
    rng = np.random.RandomState(4)
    coords = rng.random_sample((n, 3)).astype("float32")
    radii = np.full(n, 0.008, dtype="float32")
    ref = oracle.collide(oracle.pad4(coords), radii, capacity=cap)
    expected = pair_set(ref["pairs"])
    assert 1000 < ref["count"] < cap
    c_t = torch.from_numpy(pad4(coords)).cuda()
    r_t = torch.from_numpy(radii).cuda()
    n_t = torch.zeros(1, dtype=torch.int32, device="cuda")
    p_t = torch.zeros((cap, 2), dtype=torch.int32, device="cuda")
    collider = Collider(ctx, n, 4, 64, np.dtype("float32"))
    bufs = [hip.Buffer.from_tensor(ctx, t) for t in (c_t, r_t, n_t, p_t)]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        cq = hip.CommandQueue(ctx, stream=side.cuda_stream)
        collider.get_collisions(cq, bufs[0], bufs[1], bufs[2], bufs[3], cap)      # eager warm-up (allocations)
        side.synchronize()
        assert int(n_t.item()) == ref["count"]
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            collider.get_collisions(cq, bufs[0], bufs[1], bufs[2], bufs[3], cap)
        for _ in range(3):
            p_t.zero_()
            n_t.fill_(12345)
            graph.replay()
            side.synchronize()
            count = int(n_t.item())
            assert count == ref["count"]
            assert pair_set(p_t[:count].cpu().numpy().view(np.uint32)) == expected


@pytest.mark.parametrize("size,r", [(2000, 0.005), (10000, 0.001)])
def test_config1_scene_matches_the_reference_output(hip_env, generated, size, r):
    """BASELINE config 1's scene (RandomState(4), r = 0.001, 10 000 spheres) through the HIP Collider against the pair
    set the REFERENCE's own find_collisions produced on it (tests/golden/make_golden.py -> generated.npz): the one
    fixture the reference itself computed at a BASELINE size."""
    ctx, cq = hip_env
    rng = np.random.RandomState(4)
    coords = rng.random_sample((size, 3)).astype(np.float32)
    radii = np.full(size, r, np.float32)
    expected = pair_set(generated["config1_%d_pairs" % size])
    collider = Collider(ctx, size, 16, 64)
    count, pairs = run_collider(ctx, cq, collider, coords, radii, max(len(expected), 16))
    assert count == len(expected)
    assert pair_set(np.sort(pairs, axis=1)) == expected
