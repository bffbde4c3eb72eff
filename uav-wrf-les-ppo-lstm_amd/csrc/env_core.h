// env_core.h -- device-side arithmetic of ONE plume environment (E2-E5), shared by the
// stand-alone step kernel (env.hip) and the fused persistent rollout kernel (rollout.hip).
//
// Reference: PPOV2.0/environment.py:41-49 (reset), :51-62 (field), :64-80 (obs), :82-169 (step);
// PPOV2.1/environment.py:56 (sigma=15); PPOV1.1/environment.py:105 (clip 500-1e-6).
// The reference computes in f64 with f32 casts at fixed points; the same types are used here
// (MI355X has full-rate-enough f64 VALU; the env step is a few hundred flops per env-step), so in
// materialised-field mode positions, observations and done flags are bit-identical to numpy's.
#pragma once
#include "common.h"
#include "philox.h"

constexpr int GRID = 500;
constexpr int CELLS = 10;
constexpr int CELL = GRID / CELLS;
constexpr int NVIS = CELLS * CELLS;

struct EnvParams {            // kernel-argument copy of uav_env_cfg + derived constants
    int variant, field_mode, n_fields, bonus_is_f64;
    int n_env_total, env_offset, max_steps, trend_k;   // trend_k: extra observation channels (0..2)
    double radius, bonus, clip_hi, two_sigma2;
    double reach_bonus;       // min(500, 150*(50/radius)), environment.py:151 -- a per-launch constant
    uint64_t seed;
    const double* bank;       // [F][GRID][GRID][2]
    const double* bank_src;   // [F][2]
    const double* pow075;     // [POW_TABLE_N] host-libm pow(i, 0.75)
    const double* wave;       // [2][GRID] host-libm sin(0.05 x) | cos(0.07 y) of environment.py:58
    const double* pow_near;   // optional copy of pow075[0 .. pow_near_n) in faster memory (the fused rollouts' LDS), else NULL
    int pow_near_n;
    const double* ftab;       // [FT_N] host-libm tables of the f64 log / cos-sin / exp below (global, or the rollouts' LDS copy)
    const double* curr;       // device-side curriculum state (uav_curriculum_*), or NULL: { radius, bonus, bonus_is_f64, .. }
};

// With a device-side curriculum the three curriculum values of a launch come from device memory (wave-uniform loads at kernel
// entry) instead of the host's copy in the kernel arguments: the host then never has to wait for the episode outcomes.
__device__ __forceinline__ void env_params_refresh(EnvParams& P) {
    if (P.curr) {
        const double r = P.curr[0];
        P.radius = r;
        P.bonus = P.curr[1];
        P.bonus_is_f64 = P.curr[2] != 0.0;
        P.reach_bonus = fmin(500.0, 150.0 * (50.0 / r));                  // environment.py:151, as env_params_from_cfg
    }
}

// ---- f64 log, cos/sin and exp for the procedural field and the Box-Muller normals, table-driven.  The arguments are not
// arbitrary doubles: log is needed of a 24-bit integer, cos/sin of a 24-bit fraction of a turn, exp of -d^2 / 2 sigma^2 in
// [-300, 0].  A table entry (host libm, correctly rounded) + a few polynomial terms of a remainder below 2^-12 / 2 pi / 1024 /
// ln 2 / 128 gives the value to an ABSOLUTE error of ~1e-15 (relative ~3e-16 except -ln u just past the near / far switch of
// ft_neglog_u24, where ~2.4e-4 is formed from terms of ~0.69: relative 5e-13, absolute still 1e-15) -- the oracle's numpy calls
// agree to 1e-15 -- in a third of the
// instructions of the general-purpose routines, which sat on the chain of the wave that paces the fused rollouts.
constexpr int FT_LOG = 0, FT_LOG_N = 4096;            // log(1 + i / 4096)
constexpr int FT_CS = FT_LOG + FT_LOG_N, FT_CS_N = 1024;   // cos, sin (2 pi i / 1024), interleaved
constexpr int FT_EXP = FT_CS + 2 * FT_CS_N, FT_EXP_N = 64;   // 2^(j / 64)
constexpr int FT_N = FT_EXP + FT_EXP_N;

__device__ __forceinline__ double ft_rcp(double x) {   // 1 / x to f64 accuracy for normal x: hardware seed + two Newton steps
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    return r * (2.0 - x * r);
}
// -ln(k / 2^24) for an integer 1 <= k <= 2^24 (= -ln u1 of the Box-Muller radius).  Near u1 = 1 the difference of two logs
// would cancel (absolute error 2e-15 against a value of 6e-8): there, k > 2^24 - 4096, the series of -log1p(-t), t < 2^-12.
__device__ __forceinline__ double ft_neglog_u24(uint32_t k, const double* __restrict__ ft) {
    const int e = 31 - __builtin_clz(k);
    const uint32_t m = k << (24 - e);                  // [2^24, 2^25): k = 2^(e - 24) m
    const uint32_t i = (m >> 12) & 0xfffu, lo = m & 0xfffu;
    const double d = (double)lo * ft_rcp((double)(m - lo));      // k = 2^e (1 + i / 4096)(1 + d), d < 2^-12
    const double l1p = d * (1.0 - d * (0.5 - d * (1.0 / 3.0 - d * 0.25)));
    const double far = ((double)(24 - e) * 0.6931471805599453094 - ft[FT_LOG + i]) - l1p;
    const double t = (double)(16777216u - k) * (1.0 / 16777216.0);
    const double near = t * (1.0 + t * (0.5 + t * (1.0 / 3.0 + t * 0.25)));
    return (16777216u - k) < 4096u ? near : far;
}
// cos and sin of 2 pi m / 2^24 for a 24-bit m
__device__ __forceinline__ void ft_sincos_u24(uint32_t m, const double* __restrict__ ft, double& sn, double& cs) {
    const uint32_t idx = m >> 14, rem = m & 0x3fffu;
    const double C = ft[FT_CS + 2 * idx], S = ft[FT_CS + 2 * idx + 1];
    const double d = (double)rem * (6.283185307179586477 / 16777216.0), d2 = d * d;
    const double cd = 1.0 - d2 * (0.5 - d2 * (1.0 / 24.0 - d2 * (1.0 / 720.0)));
    const double sd = d * (1.0 - d2 * (1.0 / 6.0 - d2 * (1.0 / 120.0)));
    cs = C * cd - S * sd;
    sn = S * cd + C * sd;
}
// exp(x) for -700 < x <= 0
__device__ __forceinline__ double ft_exp_neg(double x, const double* __restrict__ ft) {
    const double n = rint(x * 92.332482616893656877);                          // 64 / ln 2
    const double r = (x - n * 0.010830424695996044) - n * 2.5310172166650877e-13;       // ln 2 / 64 = hi (35 bits: n hi is exact) + lo
    const int ni = (int)n;
    const double p = 1.0 + r * (1.0 + r * (0.5 + r * (1.0 / 6.0 + r * (1.0 / 24.0 + r * (1.0 / 120.0)))));
    return __builtin_amdgcn_ldexp(ft[FT_EXP + (ni & 63)] * p, ni >> 6);
}

// The two small tables an env step reads AFTER its action is known -- pow(visit count, 0.75) and the ripple factors of the
// cell -- copied into LDS by the fused rollout kernels: from L2 each was a dependent ~700-cycle load on the chain of the
// wave that paces the step.
constexpr int ENV_POW_NEAR = 1024;                    // covers every visit count of a 1000-step episode
constexpr int ENV_LDS_TABLE_DOUBLES = ENV_POW_NEAR + 2 * 500 + FT_N;
__device__ __forceinline__ void env_tables_to_lds(EnvParams& P, double* tab, int tid, int nthreads) {
    for (int i = tid; i < ENV_POW_NEAR; i += nthreads) tab[i] = P.pow075[i];
    for (int i = tid; i < 2 * 500; i += nthreads) tab[ENV_POW_NEAR + i] = P.wave[i];
    for (int i = tid; i < FT_N; i += nthreads) tab[ENV_POW_NEAR + 2 * 500 + i] = P.ftab[i];
    P.pow_near = tab;
    P.pow_near_n = ENV_POW_NEAR;
    P.wave = tab + ENV_POW_NEAR;
    P.ftab = tab + ENV_POW_NEAR + 2 * 500;
}

struct EnvState {             // registers of one env
    float px, py;             // agent_pos (f32 after the first step; (0,0) at reset)
    double sx, sy;            // source_pos
    int steps, episode;
    double conc, tke;         // conc = field/100 (the value obs[2] and prev_conc both use), tke = raw field
    float q1, q2;             // obs[2] of the previous two observations (trend channels; reset value at episode start)
};

// SoA view of the caller-owned state blob
struct EnvBlob {
    float* px; float* py; double* sx; double* sy; int* steps; int* episode; double* conc; double* tke; float* q1; float* q2;
    unsigned short* visited;  // [n][NVIS]
};
__host__ __device__ inline size_t env_blob_bytes(int n) {
    return (size_t)n * (4 + 4 + 8 + 8 + 4 + 4 + 8 + 8 + 4 + 4 + 2 * NVIS) + 256;
}
__host__ __device__ inline EnvBlob env_blob_view(void* base, int n) {
    EnvBlob b;
    char* p = (char*)base;
    b.sx = (double*)p; p += (size_t)n * 8;
    b.sy = (double*)p; p += (size_t)n * 8;
    b.conc = (double*)p; p += (size_t)n * 8;
    b.tke = (double*)p; p += (size_t)n * 8;
    b.px = (float*)p; p += (size_t)n * 4;
    b.py = (float*)p; p += (size_t)n * 4;
    b.steps = (int*)p; p += (size_t)n * 4;
    b.episode = (int*)p; p += (size_t)n * 4;
    b.q1 = (float*)p; p += (size_t)n * 4;
    b.q2 = (float*)p; p += (size_t)n * 4;
    b.visited = (unsigned short*)p;
    return b;
}
__device__ __forceinline__ EnvState env_load(const EnvBlob& b, int i) {
    EnvState s;
    s.px = b.px[i]; s.py = b.py[i]; s.sx = b.sx[i]; s.sy = b.sy[i];
    s.steps = b.steps[i]; s.episode = b.episode[i]; s.conc = b.conc[i]; s.tke = b.tke[i];
    s.q1 = b.q1[i]; s.q2 = b.q2[i];
    return s;
}
__device__ __forceinline__ void env_store(const EnvBlob& b, int i, const EnvState& s) {
    b.px[i] = s.px; b.py[i] = s.py; b.sx[i] = s.sx; b.sy[i] = s.sy;
    b.steps[i] = s.steps; b.episode[i] = s.episode; b.conc[i] = s.conc; b.tke[i] = s.tke;
    b.q1[i] = s.q1; b.q2[i] = s.q2;
}

__device__ __forceinline__ int clipi(int v) { return v < 0 ? 0 : (v > GRID - 1 ? GRID - 1 : v); }

// pow(vc, 0.75) for vc = 0..5000 (environment.py:133), filled by the HOST's libm at uav_create so it
// is the very double the reference's `visit_count**0.75` produces; also keeps f64 pow out of the kernels.
constexpr int POW_TABLE_N = 5001;      // table pointer travels in EnvParams (owned by the uav_ctx)

// two standard normals from one Philox block: Box-Muller in f64 on 24-bit uniforms, u1 in (0,1], u2 in [0,1):
// z0 = sqrt(-2 ln u1) cos(2 pi u2), z1 = ... sin(2 pi u2).  oracle/procedural_oracle.py restates exactly this from the
// same draws (numpy f64), so the procedural mode is pinned to libm accuracy (1e-15), not statistically.
__device__ __forceinline__ double bm_radius(uint32_t a, const double* __restrict__ ft) {
    return sqrt(2.0 * ft_neglog_u24((a >> 8) + 1u, ft));          // u1 = k / 2^24 in (0,1], k = (a >> 8) + 1
}
__device__ __forceinline__ void normal2(const Philox4& r, const double* __restrict__ ft, double& z0, double& z1) {
    const double rad = bm_radius(r.x, ft);
    double sn, cs;
    ft_sincos_u24(r.y >> 8, ft, sn, cs);                                  // u2 = (r.y >> 8) / 2^24 in [0,1)
    z0 = rad * cs;
    z1 = rad * sn;
}

// E3: concentration and 'tke' at integer cell (x, y)
__device__ __forceinline__ void field_at(const EnvParams& P, int env_global, const EnvState& s, int x, int y,
                                         double& conc, double& tke) {
    if (P.field_mode == UAV_FIELD_MATERIALISED) {
        const long long f = ((long long)env_global + (long long)s.episode * P.n_env_total) % P.n_fields;
        const double2 v = *reinterpret_cast<const double2*>(P.bank + ((f * GRID + x) * (long long)GRID + y) * 2);
        conc = v.x;
        tke = v.y;
        return;
    }
    // procedural: |N(0,1)| and U[0,1) of this (env, episode, cell) from the counter RNG, then the reference's formula in
    // f64 with its own operation order (environment.py:52-61); sin(0.05 x), cos(0.07 y) from the host-libm tables
    const Philox4 r = philox4x32_10(P.seed, (uint32_t)(x * GRID + y), (uint32_t)env_global, (uint32_t)s.episode, RNG_FIELD);
    double gs, gc;
    ft_sincos_u24(r.y >> 8, P.ftab, gs, gc);
    const double g = bm_radius(r.x, P.ftab) * gc;
    const double u = (double)r.w * (1.0 / 4294967296.0);
    const double wave = (0.3 * P.wave[x]) * P.wave[GRID + y];
    tke = 3.0 * ((fabs(g) + wave) + 0.2 * u);                                       // :56-60
    const double dx = (double)x - s.sx, dy = (double)y - s.sy;
    const double dist = sqrt(dx * dx + dy * dy);                                    // :53
    const double base = 100.0 * ft_exp_neg(-(dist * dist) / P.two_sigma2, P.ftab);  // :54 (V2.1 :56)
    const double c = base + tke;
    conc = c < 0.0 ? 0.0 : (c > 100.0 ? 100.0 : c);                                  // :61
}

// E5 with the field values of the f32 cell already in s.conc (= conc/100) and s.tke
__device__ __forceinline__ void env_obs(const EnvParams& P, const EnvState& s, const unsigned short* vis, float* o) {
    const int x = clipi((int)s.px), y = clipi((int)s.py);
    const int vc = vis[(x / CELL) * CELLS + (y / CELL)];
    // min(vc/5.0, 1.0) rounded to f32: the six possible values are constants
    const float lvl = vc >= 5 ? 1.0f : (vc == 4 ? 0.8f : (vc == 3 ? 0.6f : (vc == 2 ? 0.4f : (vc == 1 ? 0.2f : 0.0f))));
    o[0] = s.px / 500.0f;                           // f32 / weak int (environment.py:74)
    o[1] = s.py / 500.0f;
    o[2] = (float)s.conc;
    o[3] = (float)(s.tke / 9.0);
    o[4] = (float)((double)s.steps / (double)P.max_steps);
    o[5] = lvl;
    // trend channels (BASELINE C5 'grad[CH4] trend obs'; build-defined): change of the
    // concentration feature against the previous one / two observations of the episode
    o[6] = o[2] - s.q1;
    o[7] = o[2] - s.q2;
}

// E2: start episode s.episode of env `env_global`
__device__ __forceinline__ void env_begin_episode(const EnvParams& P, int env_global, EnvState& s, unsigned short* vis) {
    if (P.field_mode == UAV_FIELD_MATERIALISED) {
        const long long f = ((long long)env_global + (long long)s.episode * P.n_env_total) % P.n_fields;
        s.sx = P.bank_src[2 * f];
        s.sy = P.bank_src[2 * f + 1];
    } else {
        const Philox4 r = philox4x32_10(P.seed, 0u, (uint32_t)env_global, (uint32_t)s.episode, RNG_SOURCE);
        s.sx = u01_f64(r.x, r.y) * 400.0 + 50.0;   // rand(2)*(500-100)+50, environment.py:42-43
        s.sy = u01_f64(r.z, r.w) * 400.0 + 50.0;
    }
    s.px = 0.f;
    s.py = 0.f;
    s.steps = 0;
    for (int k = 0; k < NVIS; ++k) vis[k] = 0;
    double c;
    field_at(P, env_global, s, 0, 0, c, s.tke);
    s.conc = c / 100.0;
    s.q1 = s.q2 = (float)s.conc;
}

struct StepOut {
    float obs[8];        // observation of the state AFTER the move (terminal obs if done); [6],[7] = trend channels
    double reward;
    bool done, reached;
    double info[5];      // concentration_reward, explore_reward, move_penalty, tke_penalty, boundary_penalty
};

// The wind displacement of a step from its two normals z0, z1 (environment.py:100-101).  It does not depend on the action,
// so the fused rollout computes it before the action is known.
__device__ __forceinline__ void env_step_wind(const EnvState& s, double z0, double z1, double& tx, double& ty) {
    constexpr double MOVE = GRID * 0.05;
    const double k = (MOVE * 0.2);
    tx = k * (z0 * s.tke / 9.0);
    ty = k * (z1 * s.tke / 9.0);
}

// E4.  tx, ty: env_step_wind() of the state before the step.
__device__ __forceinline__ void env_step_core(const EnvParams& P, int env_global, EnvState& s, unsigned short* vis,
                                              int action, double tx, double ty, StepOut& out) {
    s.steps += 1;
    const double prev_conc = s.conc;                                  // :86-88 (cell of the f32 position), already /100
    constexpr double MOVE = GRID * 0.05;                              // :91
    double dx = 0.0, dy = 0.0;
    if (action == 1) dy = MOVE; else if (action == 2) dy = -MOVE; else if (action == 3) dx = MOVE; else if (action == 4) dx = -MOVE;
    const double norm_d = (action == 0) ? 0.0 : MOVE;
    const double move_pen = (action == 0) ? -0.15 : -0.0;            // -0.15*(1 - |d|/25), :94-95
    double nx = ((double)s.px + dx) + tx, ny = ((double)s.py + dy) + ty;    // :104
    nx = fmin(fmax(nx, 0.0), P.clip_hi);                              // :105
    ny = fmin(fmax(ny, 0.0), P.clip_hi);
    s.px = (float)nx;                                                 // :106
    s.py = (float)ny;

    // field at the new f32 cell (obs, next prev_conc) and at the f64 cell (gradient) -- they differ
    // only when the f32 rounding crosses an integer
    const int fx = clipi((int)s.px), fy = clipi((int)s.py);
    double craw;
    const float o2_old = (float)s.conc;
    field_at(P, env_global, s, fx, fy, craw, s.tke);
    s.conc = craw / 100.0;
    s.q2 = s.q1;
    s.q1 = o2_old;
    const int cx = clipi((int)nx), cy = clipi((int)ny);
    double cur = s.conc;
    if (cx != fx || cy != fy) {
        double dummy;
        field_at(P, env_global, s, cx, cy, craw, dummy);
        cur = craw / 100.0;
    }
    const double grad = (cur - prev_conc) / (norm_d + 1e-6);          // :109-112
    // min(nx/500, (500-nx)/500, ny/500, (500-ny)/500): a correctly rounded division by a positive constant is
    // monotonic, so dividing the minimum gives the identical double (:114-119)
    const double bdist = fmin(fmin(nx, 500.0 - nx), fmin(ny, 500.0 - ny)) / 500.0;
    double bpen = 0.0;
    if (bdist < 0.15 && grad < -0.01) {
        const double t = 0.15 - bdist;
        bpen = -0.1 * (t * t);                                        // :121-124
    }
    // int(nx // 50) for 0 <= nx < 500: floor(nx/50) == floor(floor(nx)/50), integer arithmetic (:127-128)
    const int gx = (int)nx / CELL, gy = (int)ny / CELL;
    const int vi = gx * CELLS + gy;
    const int vc = (int)vis[vi] + 1;
    vis[vi] = (unsigned short)vc;                                     // :129-130

    env_obs(P, s, vis, out.obs);                                      // :133,136 (f32-position cell)
    const double den = (vc < P.pow_near_n ? P.pow_near[vc] : P.pow075[vc < POW_TABLE_N ? vc : POW_TABLE_N - 1]) + 1.0;
    const float conc_r = 2.0f * out.obs[2];                           // f32, :140
    const float tke_p = 0.4f * out.obs[3];                            // f32, :143
    double explore, total;
    if (P.bonus_is_f64) {                                             // np.float64 bonus (model.py:142)
        explore = (P.bonus * (double)(1.0f - out.obs[5])) / den;
        total = (double)conc_r + explore;
    } else {                                                          // weak python float: f32 expression
        const float e32 = ((float)P.bonus * (1.0f - out.obs[5])) / (float)den;
        explore = (double)e32;
        total = (double)(conc_r + e32);
    }
    total = total + move_pen;
    total = total - (double)tke_p;
    total = total + bpen;
    const double ddx = (double)s.px - s.sx, ddy = (double)s.py - s.sy;
    const double dist = sqrt(ddx * ddx + ddy * ddy);                  // :148
    out.reached = dist <= P.radius;
    if (out.reached) total = total + P.reach_bonus;                   // :150-151
    out.done = (s.steps >= P.max_steps) || out.reached;              // :153
    out.reward = total;
    out.info[0] = (double)conc_r;
    out.info[1] = explore;
    out.info[2] = move_pen;
    out.info[3] = -(double)tke_p;
    out.info[4] = bpen;
}

// noise of (env, episode, step) from the counter RNG
__device__ __forceinline__ void env_step_noise(const EnvParams& P, int env_global, const EnvState& s, double& z0, double& z1) {
    const Philox4 r = philox4x32_10(P.seed, (uint32_t)s.steps, (uint32_t)env_global, (uint32_t)s.episode, RNG_STEP);
    normal2(r, P.ftab, z0, z1);
}
