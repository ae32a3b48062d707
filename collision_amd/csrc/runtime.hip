// Runtime plumbing of the C ABI: device memory, copies, fills, streams, events.
// These stand in for the PyOpenCL objects the reference's callers create
// (tests/conftest.py:4-12, cl.Buffer / cl.enqueue_copy / cl.enqueue_fill_buffer / cl.Event).
#include "col_common.h"
#include <string.h>

namespace {

template <typename T>
__global__ void k_fill(T *dst, T v, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) dst[i] = v;
}

template <typename T>
static int fill_launch(void *stream, void *dst, const void *pattern, size_t count) {
    T v;
    memcpy(&v, pattern, sizeof(T));
    size_t blocks = col_ceil_div(count, 256);
    if (blocks > 4096) blocks = 4096;
    k_fill<T><<<dim3((unsigned)blocks), dim3(256), 0, col_stream(stream)>>>((T *)dst, v, count);
    COL_LAUNCH_OK();
    return COL_OK;
}

// diagnostics: which XCD (HW_REG_XCC_ID) each workgroup of a 256-thread launch lands on
__global__ void k_xcc_census(unsigned *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF;
}

}  // namespace

extern "C" {

int col_debug_xcc_census(void *stream, uint32_t *out, uint32_t nblocks) {
    k_xcc_census<<<dim3(nblocks), dim3(256), 0, col_stream(stream)>>>(out);
    COL_LAUNCH_OK();
    return COL_OK;
}

const char *col_error_string(int code) {
    if (code == COL_OK) return "ok";
    if (code == COL_EINVAL) return "collision_hip: invalid argument";
    if (code == COL_ENOSCRATCH) return "collision_hip: missing scratch buffer";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "collision_hip: unknown error";
}

int col_version(void) { return 100; }

int col_device_count(int *count) { COL_HIP(hipGetDeviceCount(count)); return COL_OK; }
int col_set_device(int device) { COL_HIP(hipSetDevice(device)); return COL_OK; }
int col_get_device(int *device) { COL_HIP(hipGetDevice(device)); return COL_OK; }
int col_device_sync(void) { COL_HIP(hipDeviceSynchronize()); return COL_OK; }

int col_device_name(char *buf, int len) {
    int dev = 0;
    hipDeviceProp_t prop;
    COL_HIP(hipGetDevice(&dev));
    COL_HIP(hipGetDeviceProperties(&prop, dev));
    if (len <= 0) return COL_EINVAL;
    snprintf(buf, (size_t)len, "%s %s (%d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return COL_OK;
}

int col_malloc(void **ptr, size_t bytes) { COL_HIP(hipMalloc(ptr, bytes ? bytes : 4)); return COL_OK; }
int col_free(void *ptr) { COL_HIP(hipFree(ptr)); return COL_OK; }
int col_host_alloc(void **ptr, size_t bytes) { COL_HIP(hipHostMalloc(ptr, bytes ? bytes : 4, hipHostMallocCoherent)); return COL_OK; }
int col_host_free(void *ptr) { COL_HIP(hipHostFree(ptr)); return COL_OK; }

int col_memcpy_h2d(void *stream, void *dst, const void *src, size_t bytes) {
    if (bytes) COL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, col_stream(stream)));
    return COL_OK;
}
int col_memcpy_d2h(void *stream, void *dst, const void *src, size_t bytes) {
    if (bytes) COL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, col_stream(stream)));
    return COL_OK;
}
int col_memcpy_d2d(void *stream, void *dst, const void *src, size_t bytes) {
    if (bytes) COL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, col_stream(stream)));
    return COL_OK;
}

int col_fill(void *stream, void *dst, const void *pattern, size_t pattern_bytes, size_t count) {
    if (count == 0) return COL_OK;
    switch (pattern_bytes) {
    case 1: return fill_launch<uint8_t>(stream, dst, pattern, count);
    case 2: return fill_launch<uint16_t>(stream, dst, pattern, count);
    case 4: {
        uint32_t v;
        memcpy(&v, pattern, 4);
        COL_HIP(hipMemsetD32Async((hipDeviceptr_t)dst, (int)v, count, col_stream(stream)));
        return COL_OK;
    }
    case 8: return fill_launch<uint64_t>(stream, dst, pattern, count);
    case 16: return fill_launch<uint4>(stream, dst, pattern, count);
    default: return COL_EINVAL;
    }
}

int col_stream_create(void **stream) {
    hipStream_t s;
    COL_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return COL_OK;
}
int col_stream_destroy(void *stream) { COL_HIP(hipStreamDestroy(col_stream(stream))); return COL_OK; }
int col_stream_sync(void *stream) { COL_HIP(hipStreamSynchronize(col_stream(stream))); return COL_OK; }
int col_stream_wait_event(void *stream, void *event) {
    COL_HIP(hipStreamWaitEvent(col_stream(stream), (hipEvent_t)event, 0));
    return COL_OK;
}
int col_event_create(void **event) {
    hipEvent_t e;
    COL_HIP(hipEventCreate(&e));
    *event = (void *)e;
    return COL_OK;
}
int col_event_destroy(void *event) { COL_HIP(hipEventDestroy((hipEvent_t)event)); return COL_OK; }
int col_event_record(void *event, void *stream) { COL_HIP(hipEventRecord((hipEvent_t)event, col_stream(stream))); return COL_OK; }
int col_event_sync(void *event) { COL_HIP(hipEventSynchronize((hipEvent_t)event)); return COL_OK; }
int col_event_elapsed_ms(float *ms, void *start, void *stop) {
    COL_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return COL_OK;
}

}  // extern "C"
