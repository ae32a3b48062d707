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
