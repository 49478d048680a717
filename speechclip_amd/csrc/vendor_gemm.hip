// COMPARATOR ONLY: plain GEMMs through hipBLASLt behind the same sc_gemm_bf16 entry, enabled by registering a workspace
// (sc_set_gemm_workspace; speechclip_amd.ops.set_vendor_gemm / SC_GEMM_VENDOR=1).  The product path never does: every GEMM of the hot path
// runs on the hand-written kernels of gemm.hip.  bench.py measures this path BESIDE the headline (`vendor_comparator`); round 2, same box,
// whole step: 46.0 ms hand-written vs 44.3 ms with the library on QKV / out-proj / fc2 / the ViT projections.
#include <hipblaslt/hipblaslt.h>
#include <map>
#include <mutex>
#include <tuple>
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr, ld = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws = 0;
    bool ok = false;
};

hipblasLtHandle_t g_handle = nullptr;
void* g_ws = nullptr;
size_t g_ws_bytes = 0;
std::mutex g_mu;
std::map<std::tuple<int64_t, int, int, int64_t, int64_t, int64_t, int, int>, Plan> g_plans;

bool build_plan(Plan& p, int64_t M, int N, int K, int64_t lda, int64_t ldc, int64_t ldr, bool has_res, bool out_f32) {
    const hipDataType ct = out_f32 ? HIP_R_32F : HIP_R_16BF;   // C (residual) and D (output) share the type, as in sc_gemm_bf16
    // Row-major C[M,N] = A[M,K] . W[N,K]^T   <=>   column-major C^T[N,M] = op_T(W_cm[K,N]) . A_cm[K,M]
    if (hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return false;
    const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta));
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb));
    const hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_BIAS;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep));
    const hipDataType bt = HIP_R_32F;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt));
    if (hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16BF, K, N, K) != HIPBLAS_STATUS_SUCCESS) return false;       // W as [K, N] col-major, ld = K
    if (hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_16BF, K, M, lda) != HIPBLAS_STATUS_SUCCESS) return false;     // A as [K, M] col-major, ld = lda
    if (hipblasLtMatrixLayoutCreate(&p.lc, ct, N, M, has_res ? ldr : ldc) != HIPBLAS_STATUS_SUCCESS) return false;
    if (hipblasLtMatrixLayoutCreate(&p.ld, ct, N, M, ldc) != HIPBLAS_STATUS_SUCCESS) return false;
    hipblasLtMatmulPreference_t pref;
    if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return false;
    const uint64_t maxws = g_ws_bytes / 2;
    hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &maxws, sizeof(maxws));
    hipblasLtMatmulHeuristicResult_t res[4];
    int n = 0;
    const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(g_handle, p.desc, p.la, p.lb, p.lc, p.ld, pref, 4, res, &n);
    hipblasLtMatmulPreferenceDestroy(pref);
    if (st != HIPBLAS_STATUS_SUCCESS || n <= 0) return false;
    for (int i = 0; i < n; ++i)
        if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= g_ws_bytes / 2) { p.algo = res[i].algo; p.ws = res[i].workspaceSize; return true; }
    return false;
}

}  // namespace

// Register a device scratch buffer (caller-owned, must outlive every sc_gemm_bf16 call) that enables the library path; NULL / 0 disables it.
extern "C" int sc_set_gemm_workspace(void* ws, int64_t bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_ws = ws;
    g_ws_bytes = ws ? (size_t)bytes : 0;
    g_plans.clear();          // plans were sized against the previous workspace (descriptors are leaked on purpose: a handful per process)
    return 0;
}

// returns 0 done, 1 not applicable (caller uses its own kernel), < 0 error
int sc_vendor_gemm_try(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias, const void* residual,
                       int64_t ldr, int64_t M, int N, int K, int out_f32, hipStream_t s) {
    if (!g_ws || !bias || ldw != K || M < 8192 || N < 256 || K < 256) return 1;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_handle && hipblasLtCreate(&g_handle) != HIPBLAS_STATUS_SUCCESS) { g_handle = nullptr; return 1; }
    const auto key = std::make_tuple(M, N, K, lda, ldc, residual ? ldr : (int64_t)-1, residual ? 1 : 0, out_f32);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        Plan p;
        p.ok = build_plan(p, M, N, K, lda, ldc, ldr, residual != nullptr, out_f32 != 0);
        it = g_plans.emplace(key, p).first;
    }
    Plan& p = it->second;
    if (!p.ok) return 1;
    if (hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS) return 1;
    const float alpha = 1.0f, beta = residual ? 1.0f : 0.0f;
    // two streams may run library GEMMs concurrently (the image tower beside the speech tower): each owns one half of the workspace;
    // a third stream does not get the library path (its GEMMs run on the hand-written kernels, which need no workspace)
    static hipStream_t slot_stream[2];
    static int n_slots = 0;
    int slot = -1;
    for (int i = 0; i < n_slots; ++i)
        if (slot_stream[i] == s) slot = i;
    if (slot < 0) {
        if (n_slots == 2) return 1;
        slot = n_slots;
        slot_stream[n_slots++] = s;
    }
    void* ws = (void*)((char*)g_ws + (size_t)slot * (g_ws_bytes / 2));
    const hipblasStatus_t st = hipblasLtMatmul(g_handle, p.desc, &alpha, W, p.la, A, p.lb, &beta, residual ? residual : C, p.lc, C, p.ld, &p.algo, ws,
                                               p.ws, s);
    if (st != HIPBLAS_STATUS_SUCCESS) { sc_set_error("sc_gemm_bf16: hipblasLtMatmul failed (%d)", (int)st); return -3; }
    return 0;
}
