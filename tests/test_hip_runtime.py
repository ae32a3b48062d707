"""The small HIP runtime layer that stands in for PyOpenCL's Context/CommandQueue/Buffer/Event
(collision_amd/hip.py over the col_* runtime entry points of the C ABI)."""
import numpy as np
import pytest

from collision_amd import hip

pytestmark = pytest.mark.gpu


def test_device_and_context(hip_env):
    ctx, cq = hip_env
    assert hip.device_count() >= 1
    assert "gfx950" in ctx.name
    assert ctx == hip.Context(0) and ctx != hip.Context(1)


def test_buffer_roundtrip_and_offsets(hip_env):
    ctx, cq = hip_env
    a = np.arange(1000, dtype=np.uint32)
    buf = hip.Buffer(ctx, hostbuf=a)
    np.testing.assert_array_equal(hip.read_buffer(cq, buf, np.uint32), a)
    np.testing.assert_array_equal(hip.read_buffer(cq, buf, np.uint32, 10, offset=400), a[100:110])
    other = hip.Buffer(ctx, a.nbytes)
    e = hip.enqueue_copy(cq, other, buf)
    np.testing.assert_array_equal(hip.read_buffer(cq, other, np.uint32, wait_for=[e]), a)
    hip.enqueue_copy(cq, other, buf, byte_count=40, src_offset=0, dst_offset=80)
    out = hip.read_buffer(cq, other, np.uint32)
    np.testing.assert_array_equal(out[20:30], a[:10])
    hip.write_buffer(cq, other, np.full(5, 7, np.uint32), offset=4)
    np.testing.assert_array_equal(hip.read_buffer(cq, other, np.uint32, 7), [0, 7, 7, 7, 7, 7, 6])
    with pytest.raises(ValueError):
        hip.Buffer(ctx)


@pytest.mark.parametrize("dtype,value", [("uint8", 7), ("uint16", 513), ("uint32", 0xDEADBEEF),
                                         ("uint64", 0x0123456789ABCDEF), ("float64", -2.5)])
def test_fill_patterns(hip_env, dtype, value):
    # cl.enqueue_fill_buffer with 1/2/4/8-byte patterns (collision.py:137-154, offset.py:41-45)
    ctx, cq = hip_env
    n = 1000
    item = np.dtype(dtype).itemsize
    buf = hip.Buffer(ctx, hostbuf=np.zeros(n, dtype=dtype))
    e = hip.enqueue_fill_buffer(cq, buf, np.array(value, dtype=dtype), 10 * item, 100 * item)
    out = hip.read_buffer(cq, buf, dtype, wait_for=[e])
    assert (out[10:110] == np.array(value, dtype=dtype)).all() and (out[:10] == 0).all() and (out[110:] == 0).all()
    with pytest.raises(ValueError):
        hip.enqueue_fill_buffer(cq, buf, np.zeros(1, dtype="uint32"), 0, 6)


def test_events_order_work_across_queues(hip_env):
    ctx, cq = hip_env
    cq2 = hip.CommandQueue(ctx)
    n = 1 << 22
    a = np.arange(n, dtype=np.uint32)
    src, mid, dst = hip.Buffer(ctx, hostbuf=a), hip.Buffer(ctx, a.nbytes), hip.Buffer(ctx, a.nbytes)
    e1 = hip.enqueue_copy(cq, mid, src)
    e2 = hip.enqueue_copy(cq2, dst, mid, wait_for=[e1])          # a different stream waits on the event
    np.testing.assert_array_equal(hip.read_buffer(cq2, dst, np.uint32, wait_for=[e2]), a)
    start = hip.Event(cq)
    hip.enqueue_copy(cq, mid, src)
    stop = hip.Event(cq)
    hip.wait_for_events([start, stop])
    assert 0.0 <= start.elapsed_ms(stop) < 1000.0


def test_wrapping_foreign_memory(hip_env):
    torch = pytest.importorskip("torch")
    ctx, cq = hip_env
    t = torch.arange(4096, dtype=torch.int32, device="cuda")
    buf = hip.Buffer.from_tensor(ctx, t)
    assert buf.size == 4096 * 4
    torch.cuda.synchronize()
    np.testing.assert_array_equal(hip.read_buffer(cq, buf, np.int32), np.arange(4096))
