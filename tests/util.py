"""Small helpers shared by the GPU parity tests."""
import numpy as np

from collision_amd import hip


def pad4(coords):
    coords = np.asarray(coords)
    out = np.zeros((len(coords), 4), dtype=coords.dtype)
    out[:, :3] = coords[:, :3]
    out[:, 3] = 12345.0          # lane w is ignored by every kernel; make that visible
    return out


def upload(ctx, array):
    return hip.Buffer(ctx, hostbuf=np.ascontiguousarray(array))


def download(cq, buf, dtype, shape=None, wait_for=None):
    return hip.read_buffer(cq, buf, dtype, shape, wait_for=wait_for)


def pair_set(pairs):
    return set(map(tuple, np.asarray(pairs).tolist()))


def packed_pairs(pairs):
    """(count, 2) uint32 -> sorted uint64 keys (first << 32 | second): set comparison for millions of pairs."""
    p = np.asarray(pairs, dtype=np.uint64).reshape(-1, 2)
    return np.sort((p[:, 0] << np.uint64(32)) | p[:, 1])


def assert_same_pair_set(got, want):
    """Same pairs with the same orientation, any order, none reported twice."""
    g, w = packed_pairs(got), packed_pairs(want)
    assert len(g) == len(w), (len(g), len(w))
    np.testing.assert_array_equal(g, w)
    assert (np.diff(g) != 0).all()


def run_collider(ctx, cq, collider, coords, radii, capacity):
    """Upload, run get_collisions, read back (count, pairs[:min(count, capacity)])."""
    dt = collider.program.coord_dtype
    coords_buf = upload(ctx, pad4(np.asarray(coords, dtype=dt)))
    radii_buf = upload(ctx, np.asarray(radii, dtype=dt))
    n_buf = hip.Buffer(ctx, 4)
    pairs_buf = hip.Buffer(ctx, max(capacity, 1) * 8) if capacity else None
    e = collider.get_collisions(cq, coords_buf, radii_buf, n_buf, pairs_buf, capacity)
    count = int(download(cq, n_buf, np.uint32, 1, wait_for=[e])[0])
    got = min(count, capacity)
    pairs = download(cq, pairs_buf, np.uint32, (got, 2)) if got else np.empty((0, 2), np.uint32)
    return count, pairs


def collider_state(cq, collider):
    """Read back the internal arrays of a Collider after get_collisions."""
    from collision_amd.collision import Node
    n, p = collider.size, collider.padded_size
    dt = collider.program.coord_dtype
    return dict(
        codes=download(cq, collider._codes_bufs[1], np.uint32, p),
        ids=download(cq, collider._ids_bufs[1], np.uint32, p),
        nodes=download(cq, collider._nodes_buf, Node, 2 * n - 1),
        bounds=download(cq, collider._bounds_buf, dt, (2 * n - 1, 2, 4)),
    )
