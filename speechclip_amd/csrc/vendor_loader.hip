// Comparator hook of sc_gemm_bf16.  The product library contains NO vendor GEMM and does not link hipBLASLt: the comparator lives in its own
// library (csrc/vendor/vendor_gemm.hip -> libspeechclip_vendor_cmp.so, `make vendor`), which this file dlopen()s from the directory of the
// product library the first time a comparator workspace is registered (speechclip_amd.ops.set_vendor_gemm(True) / SC_GEMM_VENDOR=1 --
// bench.py's `vendor_comparator` leg and tools/blas_compare.py).  Without that registration sc_vendor_gemm_try answers "not applicable" and
// every GEMM runs on the hand-written kernels of gemm.hip.
#include <dlfcn.h>
#include <string.h>
#include <string>
#include "common.h"
#include "vendor/vendor_abi.h"
#include "../../include/speechclip_hip.h"

namespace {
typedef scv_set_gemm_workspace_fn set_ws_fn;      // vendor/vendor_abi.h: the ONE declaration both libraries are compiled against
typedef scv_gemm_try_fn try_fn;
scv_stream_slot_fn g_slot = nullptr;
void* g_cmp = nullptr;
set_ws_fn g_set_ws = nullptr;
try_fn g_try = nullptr;
bool g_enabled = false;

bool load_comparator() {
    if (g_cmp) return true;
    Dl_info info;
    std::string dir = ".";
    if (dladdr((const void*)&load_comparator, &info) && info.dli_fname) {
        dir = info.dli_fname;
        const size_t slash = dir.rfind('/');
        dir = slash == std::string::npos ? "." : dir.substr(0, slash);
    }
    const std::string path = dir + "/libspeechclip_vendor_cmp.so";
    g_cmp = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!g_cmp) { sc_set_error("sc_set_gemm_workspace: the vendor comparator library is not built (%s: %s); run `make -C speechclip_amd/csrc vendor`", path.c_str(), dlerror()); return false; }
    g_set_ws = (set_ws_fn)dlsym(g_cmp, "scv_set_gemm_workspace");
    g_try = (try_fn)dlsym(g_cmp, "scv_gemm_try");
    g_slot = (scv_stream_slot_fn)dlsym(g_cmp, "scv_stream_slot");
    if (!g_set_ws || !g_try || !g_slot) { sc_set_error("sc_set_gemm_workspace: %s lacks the comparator entry points", path.c_str()); dlclose(g_cmp); g_cmp = nullptr; return false; }
    return true;
}
}  // namespace

// Register a device scratch buffer (caller-owned, must outlive every sc_gemm_bf16 call) that enables the comparator path; NULL / 0 disables it.
extern "C" int sc_set_gemm_workspace(void* ws, int64_t bytes) {
    if (!ws || bytes <= 0) {
        g_enabled = false;
        return g_cmp ? g_set_ws(nullptr, 0) : 0;
    }
    if (!load_comparator()) return -1;
    const int rc = g_set_ws(ws, bytes);
    g_enabled = rc == 0;
    return rc;
}

// returns 0 done, 1 not applicable (caller uses its own kernel), < 0 error
int sc_vendor_gemm_try(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias, const void* residual,
                       int64_t ldr, int64_t M, int N, int K, int out_f32, hipStream_t s) {
    if (!g_enabled) return 1;
    int st = 0;
    const int rc = g_try(A, lda, W, ldw, C, ldc, bias, residual, ldr, M, N, K, out_f32, s, &st);
    if (rc < 0) sc_set_error("sc_gemm_bf16: hipblasLtMatmul failed (%d)", st);
    return rc;
}

// Test hook (tests/test_gemm_gpu.py): the comparator workspace half stream `s` owns: 0 / 1, -1 none yet, -2 both taken, -3 comparator not loaded.
extern "C" int sc_debug_vendor_stream_slot(void* stream) { return (g_cmp && g_slot) ? g_slot((hipStream_t)stream) : -3; }
