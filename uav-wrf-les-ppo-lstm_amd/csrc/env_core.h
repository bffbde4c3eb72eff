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
};

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
__device__ __forceinline__ double bm_radius(uint32_t a) {
    const double u1 = ((double)(a >> 8) + 1.0) * (1.0 / 16777216.0);      // (0,1]
    return sqrt(-2.0 * log(u1));
}
__device__ __forceinline__ void normal2(const Philox4& r, double& z0, double& z1) {
    const double rad = bm_radius(r.x);
    const double u2 = (double)(r.y >> 8) * (1.0 / 16777216.0);            // [0,1)
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
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
    const double g = bm_radius(r.x) * cospi(2.0 * ((double)(r.y >> 8) * (1.0 / 16777216.0)));
    const double u = (double)r.w * (1.0 / 4294967296.0);
    const double wave = (0.3 * P.wave[x]) * P.wave[GRID + y];
    tke = 3.0 * ((fabs(g) + wave) + 0.2 * u);                                       // :56-60
    const double dx = (double)x - s.sx, dy = (double)y - s.sy;
    const double dist = sqrt(dx * dx + dy * dy);                                    // :53
    const double base = 100.0 * exp(-(dist * dist) / P.two_sigma2);                 // :54 (V2.1 :56)
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
    const double den = P.pow075[vc < POW_TABLE_N ? vc : POW_TABLE_N - 1] + 1.0;
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
    normal2(r, z0, z1);
}
