// Reductions with an accumulator list that is not compiled in: the reference renders reduce.cl from a Jinja2 template
// with ANY list of (initial value, binary function name) pairs and hands it to the OpenCL compiler
// (/root/reference/collision/reduce.py:9-22).  The two compiled-in lists and the table-driven col_reduce_list
// (reduce.hip) cover what the reference itself uses; everything else takes this way: the host side
// (collision_amd/reduce.py) renders a HIP kernel pair with the structure of reduce.cl:5-58, this file has hiprtc compile
// it for the device's architecture and launches it with the CALLER's geometry (ngroups x group_size work-items, as
// reduce.py:62-76 does), so that the order in which a non-associative function meets its operands is the reference's.
// hiprtc is loaded on first use (dlopen): nothing else in the library depends on it.
#include <dlfcn.h>
#include <string.h>
#include <string>
#include <vector>
#include "col_common.h"

namespace {

typedef void *rtc_program;
struct Rtc {
    int (*create)(rtc_program *, const char *, const char *, int, const char **, const char **);
    int (*compile)(rtc_program, int, const char **);
    int (*log_size)(rtc_program, size_t *);
    int (*log)(rtc_program, char *);
    int (*code_size)(rtc_program, size_t *);
    int (*code)(rtc_program, char *);
    int (*destroy)(rtc_program *);
    bool ok;
};

const Rtc &rtc() {
    static Rtc r = [] {
        Rtc x;
        memset(&x, 0, sizeof(x));
        void *h = dlopen("libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return x;
        x.create = (decltype(x.create))dlsym(h, "hiprtcCreateProgram");
        x.compile = (decltype(x.compile))dlsym(h, "hiprtcCompileProgram");
        x.log_size = (decltype(x.log_size))dlsym(h, "hiprtcGetProgramLogSize");
        x.log = (decltype(x.log))dlsym(h, "hiprtcGetProgramLog");
        x.code_size = (decltype(x.code_size))dlsym(h, "hiprtcGetCodeSize");
        x.code = (decltype(x.code))dlsym(h, "hiprtcGetCode");
        x.destroy = (decltype(x.destroy))dlsym(h, "hiprtcDestroyProgram");
        x.ok = x.create && x.compile && x.log_size && x.log && x.code_size && x.code && x.destroy;
        return x;
    }();
    return r;
}

void put_log(char *log, size_t cap, const std::string &text) {
    if (!log || cap == 0) return;
    const size_t k = text.size() < cap - 1 ? text.size() : cap - 1;
    memcpy(log, text.data(), k);
    log[k] = 0;
}

// compile `source` for `arch`; code = the code object
int compile(const char *source, const char *arch, std::vector<char> &code, char *log, size_t log_cap) {
    const Rtc &r = rtc();
    if (!r.ok) { put_log(log, log_cap, "libhiprtc.so could not be loaded"); return COL_EINVAL; }
    rtc_program prog = nullptr;
    if (r.create(&prog, source, "reduce_list.hip", 0, nullptr, nullptr) != 0) { put_log(log, log_cap, "hiprtcCreateProgram failed"); return COL_EINVAL; }
    const std::string a = std::string("--offload-arch=") + arch;
    // the parity contract of the build (csrc/Makefile): IEEE semantics, no contraction
    const char *opts[] = {a.c_str(), "-O3", "-ffp-contract=off", "-fno-fast-math"};
    const int rc = r.compile(prog, 4, opts);
    size_t ls = 0;
    if (r.log_size(prog, &ls) == 0 && ls > 1) {
        std::string text(ls, '\0');
        r.log(prog, &text[0]);
        put_log(log, log_cap, text);
    } else put_log(log, log_cap, "");
    if (rc != 0) { r.destroy(&prog); return COL_EINVAL; }
    size_t cs = 0;
    if (r.code_size(prog, &cs) != 0 || cs == 0) { r.destroy(&prog); return COL_EINVAL; }
    code.resize(cs);
    const int rc2 = r.code(prog, code.data());
    r.destroy(&prog);
    return rc2 == 0 ? COL_OK : COL_EINVAL;
}

struct Handle {
    hipModule_t module;
    hipFunction_t stage1, stage2;
};

}  // namespace

extern "C" {

// compile only (no device needed): does `source` build for `arch`?  log: the compiler's messages
int col_reduce_rtc_check(const char *source, const char *arch, char *log, size_t log_cap) {
    if (!source || !arch) return COL_EINVAL;
    std::vector<char> code;
    return compile(source, arch, code, log, log_cap);
}

// compile `source` (two kernels, `bounds1` and `bounds2`, see collision_amd/reduce.py) for the current device
int col_reduce_rtc_create(const char *source, char *log, size_t log_cap, void **handle) {
    if (!source || !handle) return COL_EINVAL;
    int dev = 0;
    COL_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    COL_HIP(hipGetDeviceProperties(&prop, dev));
    std::vector<char> code;
    const int rc = compile(source, prop.gcnArchName, code, log, log_cap);
    if (rc) return rc;
    Handle *h = new Handle;
    if (hipModuleLoadData(&h->module, code.data()) != hipSuccess) { delete h; put_log(log, log_cap, "hipModuleLoadData failed"); return COL_EINVAL; }
    if (hipModuleGetFunction(&h->stage1, h->module, "bounds1") != hipSuccess ||
        hipModuleGetFunction(&h->stage2, h->module, "bounds2") != hipSuccess) {
        (void)hipModuleUnload(h->module);
        delete h;
        put_log(log, log_cap, "the module lacks bounds1 / bounds2");
        return COL_EINVAL;
    }
    *handle = h;
    return COL_OK;
}

int col_reduce_rtc_destroy(void *handle) {
    if (!handle) return COL_OK;
    Handle *h = (Handle *)handle;
    const hipError_t e = hipModuleUnload(h->module);
    delete h;
    return (int)e;
}

// reduce.py:62-76: bounds1 on ngroups x group_size work-items (acc_bytes of LDS per work-item), then bounds2 on one group of
// ngroups work-items.  partials: ngroups * acc_bytes bytes; out: acc_bytes bytes (one row per accumulator).
int col_reduce_rtc(void *stream, void *handle, const void *values, uint64_t n, uint32_t ngroups, uint32_t group_size,
                   uint32_t acc_bytes, void *partials, void *out) {
    if (!handle || !partials || !out || ngroups == 0 || group_size == 0) return COL_EINVAL;
    if (ngroups > 1024 || group_size > 1024 || (uint64_t)acc_bytes * (group_size > ngroups ? group_size : ngroups) > 65536) return COL_EINVAL;
    Handle *h = (Handle *)handle;
    unsigned long long nn = n;
    void *a1[] = {(void *)&values, (void *)&nn, (void *)&partials};
    COL_HIP(hipModuleLaunchKernel(h->stage1, ngroups, 1, 1, group_size, 1, 1, acc_bytes * group_size, col_stream(stream), a1, nullptr));
    void *a2[] = {(void *)&partials, (void *)&out};
    COL_HIP(hipModuleLaunchKernel(h->stage2, 1, 1, 1, ngroups, 1, 1, acc_bytes * ngroups, col_stream(stream), a2, nullptr));
    return COL_OK;
}

}  // extern "C"
