// ctx.hip -- handle, error string, ABI version.
#include <stdarg.h>
#include <mutex>
#include <set>
#include <utility>
#include "common.h"

static thread_local char g_err[512] = "";
thread_local int g_uav_arith = 0;
thread_local unsigned g_uav_debug = 0;
int env_init_tables(uav_ctx* ctx);

void uav_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

hipError_t uav_dyn_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({dev, kernel});
    return e;
}

extern "C" {

int uav_abi_version(void) { return UAV_ABI_VERSION; }
const char* uav_last_error(void) { return g_err; }

int uav_create(uav_ctx** out, int device, size_t ws_bytes) {
    UAV_REQUIRE(out != nullptr, "uav_create: out is NULL");
    UAV_REQUIRE(ws_bytes >= (1u << 20), "uav_create: ws_bytes must be >= 1 MiB");
    UAV_CHECK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    UAV_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    UAV_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
                "libuavppo is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
    uav_ctx* c = new uav_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    c->ws_bytes = ws_bytes;
    c->pow075 = nullptr;
    c->wave = nullptr;
    c->ftab = nullptr;
    c->comm = nullptr;
    c->comm_rank = -1;
    c->comm_world = 0;
    // process-level override of the default arithmetic, read here once (never on a call path)
    c->lstm_arith = getenv("UAV_LSTM_F32_MFMA") ? UAV_ARITH_F32_MFMA : (getenv("UAV_LSTM_BF16X6") ? UAV_ARITH_BF16X6 : UAV_ARITH_FP16X3);
    c->debug = (getenv("UAV_LSTM_STEP_F32") ? UAV_DEBUG_STEP_F32 : 0u) | (getenv("UAV_LSTM_X_F32") ? UAV_DEBUG_X_F32 : 0u) |
               (getenv("UAV_LSTM_DG_F32") ? UAV_DEBUG_DG_F32 : 0u);
    if (hipMalloc(&c->ws, ws_bytes) != hipSuccess) {
        (void)hipFree(c->ws);
        delete c;
        uav_set_error("uav_create: hipMalloc(%zu) failed", ws_bytes);
        return 1;
    }
    if (env_init_tables(c) != 0) {
        (void)hipFree(c->ws);
        (void)hipFree(c->pow075);
        (void)hipFree(c->wave);
        (void)hipFree(c->ftab);
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

int uav_set_lstm_arith(uav_ctx* ctx, int mode) {
    UAV_REQUIRE(ctx && mode >= UAV_ARITH_FP16X3 && mode <= UAV_ARITH_F32_MFMA, "uav_set_lstm_arith: bad argument");
    ctx->lstm_arith = mode;
    return 0;
}
int uav_get_lstm_arith(const uav_ctx* ctx) { return ctx ? ctx->lstm_arith : -1; }

int uav_set_debug_flags(uav_ctx* ctx, unsigned flags) {
    UAV_REQUIRE(ctx && (flags & ~(UAV_DEBUG_STEP_F32 | UAV_DEBUG_X_F32 | UAV_DEBUG_DG_F32 | UAV_DEBUG_GEMM_TN_OFF)) == 0, "uav_set_debug_flags: bad argument");
    ctx->debug = flags;
    return 0;
}

void uav_destroy(uav_ctx* ctx) {
    if (!ctx) return;
    (void)uav_comm_destroy(ctx);
    for (auto& e : ctx->side_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& s : ctx->side)
        if (s) (void)hipStreamDestroy(s);
    (void)hipFree(ctx->ws);
    (void)hipFree(ctx->pow075);
    (void)hipFree(ctx->wave);
    (void)hipFree(ctx->ftab);
    delete ctx;
}

}  // extern "C"
