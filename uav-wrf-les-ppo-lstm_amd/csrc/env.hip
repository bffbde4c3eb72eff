// env.hip -- E1-E5 as stand-alone kernels: reset, one vectorised step with auto-reset, peek.
// One thread per environment; arithmetic in env_core.h.  (The fused rollout kernel in
// rollout.hip runs the same core inside its time loop.)
#include <math.h>
#include <vector>
#include "env_core.h"

// called once from uav_create: the table is the host libm's pow, i.e. the reference's own values
int env_init_tables(uav_ctx* ctx) {
    std::vector<double> t(POW_TABLE_N);
    for (int i = 0; i < POW_TABLE_N; ++i) t[i] = pow((double)i, 0.75);
    UAV_CHECK_HIP(hipMalloc(&ctx->pow075, sizeof(double) * POW_TABLE_N));
    UAV_CHECK_HIP(hipMemcpy(ctx->pow075, t.data(), sizeof(double) * POW_TABLE_N, hipMemcpyHostToDevice));
    // the deterministic ripple of environment.py:58, 0.3*sin(0.05*x)*cos(0.07*y): its two factors over the grid
    std::vector<double> wv(2 * GRID);
    for (int i = 0; i < GRID; ++i) {
        wv[i] = sin(0.05 * (double)i);
        wv[GRID + i] = cos(0.07 * (double)i);
    }
    UAV_CHECK_HIP(hipMalloc(&ctx->wave, sizeof(double) * 2 * GRID));
    UAV_CHECK_HIP(hipMemcpy(ctx->wave, wv.data(), sizeof(double) * 2 * GRID, hipMemcpyHostToDevice));
    // tables of the f64 log / cos-sin / exp of env_core.h (ft_*), host libm
    std::vector<double> ft(FT_N);
    for (int i = 0; i < FT_LOG_N; ++i) ft[FT_LOG + i] = log1p((double)i / FT_LOG_N);
    for (int i = 0; i < FT_CS_N; ++i) {
        const double a = 6.283185307179586477 * (double)i / FT_CS_N;
        ft[FT_CS + 2 * i] = cos(a);
        ft[FT_CS + 2 * i + 1] = sin(a);
    }
    ft[FT_CS] = 1.0; ft[FT_CS + 1] = 0.0;                                       // exact at the quarter turns
    ft[FT_CS + 2 * (FT_CS_N / 4)] = 0.0; ft[FT_CS + 2 * (FT_CS_N / 4) + 1] = 1.0;
    ft[FT_CS + 2 * (FT_CS_N / 2)] = -1.0; ft[FT_CS + 2 * (FT_CS_N / 2) + 1] = 0.0;
    ft[FT_CS + 2 * (3 * FT_CS_N / 4)] = 0.0; ft[FT_CS + 2 * (3 * FT_CS_N / 4) + 1] = -1.0;
    for (int j = 0; j < FT_EXP_N; ++j) ft[FT_EXP + j] = exp2((double)j / FT_EXP_N);
    UAV_CHECK_HIP(hipMalloc(&ctx->ftab, sizeof(double) * FT_N));
    UAV_CHECK_HIP(hipMemcpy(ctx->ftab, ft.data(), sizeof(double) * FT_N, hipMemcpyHostToDevice));
    return 0;
}

int env_params_from_cfg(const uav_ctx* ctx, const uav_env_cfg* cfg, int n_env, EnvParams& P) {
    UAV_REQUIRE(cfg, "env: cfg is NULL");
    UAV_REQUIRE(cfg->variant >= UAV_ENV_V20 && cfg->variant <= UAV_ENV_V11, "env: variant %d", cfg->variant);
    UAV_REQUIRE(cfg->field_mode == UAV_FIELD_PROCEDURAL || cfg->field_mode == UAV_FIELD_MATERIALISED,
                "env: field_mode %d", cfg->field_mode);
    if (cfg->field_mode == UAV_FIELD_MATERIALISED)
        UAV_REQUIRE(cfg->bank && cfg->bank_src && cfg->n_fields > 0, "env: materialised mode needs bank, bank_src, n_fields");
    UAV_REQUIRE(cfg->curriculum || cfg->radius > 0.0, "env: radius must be positive");
    P.variant = cfg->variant;
    P.field_mode = cfg->field_mode;
    P.n_fields = cfg->n_fields;
    P.bonus_is_f64 = cfg->bonus_is_f64;
    P.n_env_total = cfg->n_env_total > 0 ? cfg->n_env_total : n_env;
    P.env_offset = cfg->env_offset;
    P.max_steps = (cfg->variant == UAV_ENV_V11) ? 5000 : 1000;           // config.py:7
    UAV_REQUIRE(cfg->trend_k >= 0 && cfg->trend_k <= 2, "env: trend_k must be 0, 1 or 2");
    P.trend_k = cfg->trend_k;
    P.radius = cfg->radius;
    P.reach_bonus = fmin(500.0, 150.0 * (50.0 / cfg->radius));
    P.bonus = cfg->bonus;
    P.clip_hi = (cfg->variant == UAV_ENV_V11) ? (500.0 - 1e-6) : 499.0;  // environment.py:105
    const double sigma = (cfg->variant == UAV_ENV_V21) ? 15.0 : (500.0 / 16.0);
    P.two_sigma2 = 2.0 * sigma * sigma;
    P.seed = cfg->seed;
    P.bank = cfg->bank;
    P.bank_src = cfg->bank_src;
    P.pow075 = ctx->pow075;
    P.wave = ctx->wave;
    P.pow_near = nullptr;
    P.pow_near_n = 0;
    P.ftab = ctx->ftab;
    P.curr = reinterpret_cast<const double*>(cfg->curriculum);
    return 0;
}

__global__ __launch_bounds__(256) void env_reset_kernel(EnvParams P, EnvBlob b, int n, float* __restrict__ obs_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    EnvState s;
    s.episode = 0;
    unsigned short* vis = b.visited + (size_t)i * NVIS;
    env_begin_episode(P, P.env_offset + i, s, vis);
    env_store(b, i, s);
    float o[8];
    env_obs(P, s, vis, o);
    const int od = 6 + P.trend_k;
    for (int k = 0; k < od; ++k) obs_out[(size_t)i * od + k] = o[k];
}

__global__ __launch_bounds__(256) void env_step_kernel(EnvParams P, EnvBlob b, int n, const int32_t* __restrict__ act,
                                                       const double* __restrict__ noise, float* __restrict__ obs_out,
                                                       float* __restrict__ rew, float* __restrict__ done,
                                                       uint8_t* __restrict__ flags, float* __restrict__ info,
                                                       float* __restrict__ term_obs, double* __restrict__ rew64) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    env_params_refresh(P);
    EnvState s = env_load(b, i);
    unsigned short* vis = b.visited + (size_t)i * NVIS;
    const int eg = P.env_offset + i;
    double z0, z1;
    if (noise) { z0 = noise[2 * (size_t)i]; z1 = noise[2 * (size_t)i + 1]; }
    else env_step_noise(P, eg, s, z0, z1);
    int a = act[i];
    a = a < 0 ? 0 : (a > 4 ? 4 : a);
    StepOut o;
    double tx, ty;
    env_step_wind(s, z0, z1, tx, ty);
    env_step_core(P, eg, s, vis, a, tx, ty, o);
    const int od = 6 + P.trend_k;
    if (term_obs) for (int k = 0; k < od; ++k) term_obs[(size_t)i * od + k] = o.obs[k];
    if (info) for (int k = 0; k < 5; ++k) info[(size_t)i * 5 + k] = (float)o.info[k];
    rew[i] = (float)o.reward;
    if (rew64) rew64[i] = o.reward;
    done[i] = o.done ? 1.f : 0.f;
    flags[i] = (uint8_t)((o.done ? 1 : 0) | (o.reached ? 2 : 0));
    if (o.done) {                                   // the reset of train_ppo2.0.py:139
        s.episode += 1;
        env_begin_episode(P, eg, s, vis);
        env_obs(P, s, vis, o.obs);
    }
    env_store(b, i, s);
    for (int k = 0; k < od; ++k) obs_out[(size_t)i * od + k] = o.obs[k];
}

__global__ __launch_bounds__(256) void env_peek_kernel(EnvBlob b, int n, float* pos, double* source, int32_t* steps,
                                                       int32_t* episode) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (pos) { pos[2 * i] = b.px[i]; pos[2 * i + 1] = b.py[i]; }
    if (source) { source[2 * i] = b.sx[i]; source[2 * i + 1] = b.sy[i]; }
    if (steps) steps[i] = b.steps[i];
    if (episode) episode[i] = b.episode[i];
}

// the whole 500x500 field of ONE env's current episode, as the reference keeps it in
// env.conc_field / env.tke_field (environment.py:61-62): out[x][y] = (conc, tke)
__global__ __launch_bounds__(256) void env_materialise_kernel(EnvParams P, EnvBlob b, int env, double* __restrict__ out) {
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= GRID * GRID) return;
    EnvState s = env_load(b, env);
    double c, t;
    field_at(P, P.env_offset + env, s, cell / GRID, cell % GRID, c, t);
    out[2 * (size_t)cell] = c;
    out[2 * (size_t)cell + 1] = t;
}

extern "C" {

int uav_env_materialise(uav_ctx* ctx, const void* state, int n_env, const uav_env_cfg* cfg, int env_index,
                        double* field_out, uav_stream stream) {
    UAV_REQUIRE(ctx && state && field_out && env_index >= 0 && env_index < n_env, "uav_env_materialise: bad argument");
    EnvParams P;
    int rc = env_params_from_cfg(ctx, cfg, n_env, P);
    if (rc) return rc;
    hipLaunchKernelGGL(env_materialise_kernel, dim3((GRID * GRID + 255) / 256), dim3(256), 0, as_stream(stream), P,
                       env_blob_view(const_cast<void*>(state), n_env), env_index, field_out);
    UAV_LAUNCH_CHECK();
    return 0;
}

size_t uav_env_state_bytes(int n_env) { return n_env > 0 ? env_blob_bytes(n_env) : 0; }

int uav_env_reset(uav_ctx* ctx, void* state, int n_env, const uav_env_cfg* cfg, float* obs_out, uav_stream stream) {
    UAV_REQUIRE(ctx && state && obs_out && n_env > 0, "uav_env_reset: bad argument");
    EnvParams P;
    int rc = env_params_from_cfg(ctx, cfg, n_env, P);
    if (rc) return rc;
    hipLaunchKernelGGL(env_reset_kernel, dim3((n_env + 255) / 256), dim3(256), 0, as_stream(stream), P,
                       env_blob_view(state, n_env), n_env, obs_out);
    UAV_LAUNCH_CHECK();
    return 0;
}

int uav_env_step(uav_ctx* ctx, void* state, int n_env, const uav_env_cfg* cfg, const int32_t* act, const double* noise,
                 float* obs_out, float* rew, float* done, uint8_t* flags, float* info, float* term_obs, double* rew64,
                 uav_stream stream) {
    UAV_REQUIRE(ctx && state && act && obs_out && rew && done && flags && n_env > 0, "uav_env_step: bad argument");
    EnvParams P;
    int rc = env_params_from_cfg(ctx, cfg, n_env, P);
    if (rc) return rc;
    hipLaunchKernelGGL(env_step_kernel, dim3((n_env + 255) / 256), dim3(256), 0, as_stream(stream), P,
                       env_blob_view(state, n_env), n_env, act, noise, obs_out, rew, done, flags, info, term_obs, rew64);
    UAV_LAUNCH_CHECK();
    return 0;
}

int uav_env_peek(uav_ctx* ctx, const void* state, int n_env, float* pos, double* source, int32_t* steps,
                 int32_t* episode, uav_stream stream) {
    UAV_REQUIRE(ctx && state && n_env > 0, "uav_env_peek: bad argument");
    hipLaunchKernelGGL(env_peek_kernel, dim3((n_env + 255) / 256), dim3(256), 0, as_stream(stream),
                       env_blob_view(const_cast<void*>(state), n_env), n_env, pos, source, steps, episode);
    UAV_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
