// rollout.hip -- R1: fused persistent rollout for the LSTM actor-critic.
//
// Reference loop body: PPOV2.0/train_ppo2.0.py:157-198 (policy forward -> Categorical sample ->
// env.step -> buffer.store), here for N environments x T steps in ONE launch.
//
// One workgroup owns 16 environments for the whole horizon:
//   * H/16 "gate" waves keep their W_hh / W_ih slices in VGPRs (as lstm.hip) and the cell state in
//     registers; h_t lives in a padded LDS tile;
//   * wave 0 additionally plays the "env" role: it computes the actor/critic heads of h_t with one
//     MFMA chain (head weights as B-fragments from LDS), then lanes 0..15 each sample an action
//     (counter RNG, torch Categorical(probs) semantics), step their environment (env_core.h, f64)
//     and store the transition into the (env, T, feat) buffers;
//   * that VALU/f64 work for step t overlaps the other waves' recurrent MFMAs for step t+1 (both
//     only need h_t; wave 0's SIMD partner fills the matrix pipe meanwhile); two workgroup
//     barriers per step, no inter-workgroup communication.
// (Tried and measured: spreading the env role over all waves with opposite [env|MFMA] order between SIMD
//  partners.  Correct, but two inlined copies of the f64 env chain beside 128 weight VGPRs spill ~90 VGPRs
//  in every wave: 1.9 ms vs 1.2 ms.  The env role stays on wave 0.)
// Procedural fields make the whole rollout HBM-write-only apart from the policy parameters:
// 44 B per env-step (obs 24, act 4, rew 4, val 4, logp 4, done 4) + 5 B (keep, flags).
#include "env_core.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

int env_params_from_cfg(const uav_ctx* ctx, const uav_env_cfg* cfg, int n_env, EnvParams& P);

constexpr int RMT = 16;
constexpr float R_F32_EPS = 1.1920928955078125e-07f;

#define r_sigmoid fast_sigmoid
#define r_tanh fast_tanh

template <int H>
struct RGeom {
    static constexpr int NW = H / 16;
    static constexpr int KS = H / 4;
    static constexpr int SEG = KS + 4;
    static constexpr int S = 4 * SEG + 8;
};
template <int H>
__device__ __forceinline__ int rpos(int u) { return (u / (H / 4)) * RGeom<H>::SEG + (u % (H / 4)); }

struct RolloutBufs {
    float* cur_obs; float* h; float* c;
    float* obs; int32_t* act; float* rew; float* val; float* logp; float* done; uint8_t* flags; float* keep;
    float* last_val; const int32_t* forced_act; const double* noise; int32_t* nan_count;
    float* info;                // optional [N][T][6]: the 5 reward parts of environment.py:161-167 + obs[2] of the step
    float* stash; float* y;     // optional: BPTT stash [N][T][6H] + y [N][T][H], so PPO epoch 0 skips its forward pass
};

template <int H, int NA>
__global__ __launch_bounds__(H * 4) void rollout_lstm_kernel(EnvParams P, EnvBlob blob, int N, int T,
                                                                        uint64_t iter, const float* __restrict__ params,
                                                                        RolloutBufs B) {
    using G = RGeom<H>;
    constexpr int KS = G::KS, SEG = G::SEG, S = G::S, I = 6, NH = NA + 1;
    __shared__ __attribute__((aligned(16))) float hbuf[RMT * S];
    __shared__ __attribute__((aligned(16))) float xbuf[RMT * 8];
    __shared__ float kbuf[RMT];
    __shared__ float hd[RMT * 16];
    __shared__ __attribute__((aligned(16))) float wbuf[16 * S];   // head weights, row = head, padded like an h row
    __shared__ unsigned short vis[RMT * NVIS];
    // wave 0's W_hh slice, parked once: reloading it after the env block makes those 4*KS registers dead
    // across the f64 env chain, which then fits without scratch spills (it used to spill ~60 VGPRs)
    __shared__ float whpark[4 * KS * 64];
    __shared__ EnvState es_s[RMT];
    __shared__ float trs[3 * RMT * 8];                            // parked transitions (see the env block)                                // env registers parked in LDS between steps

    const float* w_ih = params;
    const float* w_hh = w_ih + 4 * H * I;
    const float* b_ih = w_hh + 4 * H * H;
    const float* b_hh = b_ih + 4 * H;
    const float* w_hd = b_hh + 4 * H;          // [NH][H]
    const float* b_hd = w_hd + NH * H;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * RMT;
    const bool is_env_wave = (w == 0);   // wave 0: gate wave AND env role

    // ------------------------------------------------------------------ per-role persistent registers
    float wh[4][KS], wx[4][2], bias[4];
    float c_reg[4] = {0.f, 0.f, 0.f, 0.f}, h_keep[4] = {0.f, 0.f, 0.f, 0.f};
    const int u = 16 * w + j;
    const int my_env = n0 + lane;              // env of lanes 0..15 of the env wave
    const bool env_lane = is_env_wave && lane < RMT && my_env < N;
    unsigned short* myvis = vis + (lane & 15) * NVIS;

    {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* src = w_hh + (size_t)(q * H + u) * H + kq * KS;
#pragma unroll
            for (int s = 0; s < KS; s += 4) {
                const float4 v = *reinterpret_cast<const float4*>(src + s);
                wh[q][s] = v.x; wh[q][s + 1] = v.y; wh[q][s + 2] = v.z; wh[q][s + 3] = v.w;
            }
            bias[q] = b_ih[q * H + u] + b_hh[q * H + u];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int k = 2 * kq + s;
                wx[q][s] = (k < I) ? w_ih[(size_t)(q * H + u) * I + k] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = 4 * kq + r;
            const int n = min(n0 + e, N - 1);
            c_reg[r] = B.c[(size_t)n * H + u];
            h_keep[r] = B.h[(size_t)n * H + u];
            hbuf[e * S + rpos<H>(u)] = h_keep[r];
        }
    }
    for (int idx = threadIdx.x; idx < 16 * H; idx += blockDim.x) {
        const int hdx = idx / H, uu = idx % H;
        wbuf[hdx * S + rpos<H>(uu)] = (hdx < NH) ? w_hd[(size_t)hdx * H + uu] : 0.f;
    }
    if (is_env_wave) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int sidx = 0; sidx < KS; ++sidx) whpark[(q * KS + sidx) * 64 + lane] = wh[q][sidx];
    }
    if (is_env_wave) {
        if (lane < RMT) {
            const int n = min(my_env, N - 1);
            es_s[lane] = env_load(blob, n);
            for (int k = 0; k < NVIS; ++k) myvis[k] = blob.visited[(size_t)n * NVIS + k];
#pragma unroll
            for (int f = 0; f < 8; ++f) xbuf[lane * 8 + f] = f < 6 ? B.cur_obs[(size_t)n * 6 + f] : 0.f;
            kbuf[lane] = 1.f;
        }
    }
    lds_barrier();

    // acc = bias + h_{t-1} W_hh^T of the step about to run (gate waves)
    f32x4 acc[4];
    auto recurrent = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{bias[q], bias[q], bias[q], bias[q]};
        const float* hrow = hbuf + j * S + kq * SEG;
#pragma unroll
        for (int s = 0; s < KS; s += 4) {
            const float4 a = *reinterpret_cast<const float4*>(hrow + s);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wh[q][s], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wh[q][s + 1], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wh[q][s + 2], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wh[q][s + 3], acc[q], 0, 0, 0);
        }
    };
    recurrent();
    lds_barrier();     // every wave has read h_{-1} before step 0 overwrites hbuf

    const int steps = T + (B.last_val ? 1 : 0);   // one extra value-only pass for V(s_T)
    for (int t = 0; t < steps; ++t) {
        const bool value_only = (t == T);
        // ---------------------------------------------------------------- phase 1: finish the cell of step t
        {
            const float2 ax = *reinterpret_cast<const float2*>(&xbuf[j * 8 + 2 * kq]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax.x, wx[q][0], acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax.y, wx[q][1], acc[q], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = 4 * kq + r;
                const float gi = r_sigmoid(acc[0][r]), gf = r_sigmoid(acc[1][r]);
                const float gg = r_tanh(acc[2][r]), go = r_sigmoid(acc[3][r]);
                const float c = gf * c_reg[r] + gi * gg;
                const float h = go * r_tanh(c);
                hbuf[e * S + rpos<H>(u)] = h;
                if (B.stash && !value_only && n0 + e < N) {
                    // the update's first epoch uses these parameters: its forward pass IS this rollout
                    const size_t row = (size_t)(n0 + e) * T + t;
                    float* sp = B.stash + row * (6 * H);
                    sp[u] = gi; sp[H + u] = gf; sp[2 * H + u] = gg; sp[3 * H + u] = go;
                    sp[4 * H + u] = c_reg[r];
                    B.y[row * H + u] = h;
                }
                if (!value_only) { c_reg[r] = c; h_keep[r] = h; }
            }
        }
        lds_barrier();                       // barrier 1: h_t visible
        // ---------------------------------------------------------------- phase 2 (overlapped roles)
        if (is_env_wave) {
            // heads of h_t: one MFMA chain, D[row = env][col = head]
            // four independent accumulation chains (a 16x16x4 f32 MFMA has a 40-cycle dependent latency)
            f32x4 ha = {0.f, 0.f, 0.f, 0.f}, hb2 = ha, hc2 = ha, hd2 = ha;
            const float* hrow = hbuf + j * S + kq * SEG;
            const float* wrow = wbuf + j * S + kq * SEG;
#pragma unroll
            for (int s = 0; s < KS; s += 4) {
                const float4 a = *reinterpret_cast<const float4*>(hrow + s);
                const float4 b = *reinterpret_cast<const float4*>(wrow + s);
                ha = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, ha, 0, 0, 0);
                hb2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, hb2, 0, 0, 0);
                hc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, hc2, 0, 0, 0);
                hd2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, hd2, 0, 0, 0);
            }
            ha = (ha + hb2) + (hc2 + hd2);
#pragma unroll
            for (int r = 0; r < 4; ++r) hd[(4 * kq + r) * 16 + j] = ha[r];
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the hd tile is written (single wave)
            __builtin_amdgcn_wave_barrier();
            if (lane < RMT) {
                float z[NA], p[NA];
#pragma unroll
                for (int a = 0; a < NA; ++a) z[a] = hd[lane * 16 + a] + b_hd[a];
                const float V = hd[lane * 16 + NA] + b_hd[NA];
                if (value_only) {
                    if (env_lane) B.last_val[my_env] = V;
                } else {
                    // softmax + Categorical(probs) sample / log_prob (train_ppo2.0.py:161-163,189)
                    float m = z[0];
#pragma unroll
                    for (int a = 1; a < NA; ++a) m = fmaxf(m, z[a]);
                    float ssum = 0.f;
#pragma unroll
                    for (int a = 0; a < NA; ++a) { p[a] = __expf(z[a] - m); ssum += p[a]; }
                    float psum = 0.f;
                    bool bad = false;
#pragma unroll
                    for (int a = 0; a < NA; ++a) { p[a] = p[a] / ssum; psum += p[a]; bad |= (p[a] != p[a]); }
                    if (bad && env_lane) atomicAdd(B.nan_count, 1);
                    const int eg = P.env_offset + my_env;
                    int a_sel;
                    const size_t row = (size_t)min(my_env, N - 1) * T + t;
                    if (B.forced_act) {
                        a_sel = B.forced_act[row];
                    } else {
                        const Philox4 rr = philox4x32_10(P.seed, (uint32_t)t, (uint32_t)eg, (uint32_t)iter, RNG_ACTION);
                        const float target = u01_f32(rr.x) * psum;
                        float cdf = 0.f;
                        a_sel = NA - 1;
                        bool found = false;
#pragma unroll
                        for (int a = 0; a < NA; ++a) {             // inverse CDF: first a with target < cdf
                            cdf += p[a];
                            if (!found && target < cdf) { a_sel = a; found = true; }
                        }
                    }
                    a_sel = a_sel < 0 ? 0 : (a_sel > NA - 1 ? NA - 1 : a_sel);
                    float qa = 0.f;
#pragma unroll
                    for (int a = 0; a < NA; ++a) if (a == a_sel) qa = p[a] / psum;
                    const float lp = __logf(fminf(fmaxf(qa, R_F32_EPS), 1.0f - R_F32_EPS));
                    // environment step (f64, env_core.h) + auto reset
                    EnvState es = es_s[lane];
                    double z0, z1;
                    if (B.noise) { z0 = B.noise[2 * row]; z1 = B.noise[2 * row + 1]; }
                    else env_step_noise(P, eg, es, z0, z1);
                    StepOut so;
                    env_step_core(P, eg, es, myvis, a_sel, z0, z1, so);
                    // park the transition in LDS; the global stores are issued at the very end of the env block so
                    // that no later scratch reload / load wait (vmcnt counts stores too) stalls on their HBM acks
                    float* tr = trs + lane * 8;
                    tr[0] = __int_as_float(a_sel);
                    tr[1] = (float)so.reward;
                    tr[2] = V;
                    tr[3] = lp;
                    tr[4] = so.done ? 1.f : 0.f;
                    tr[5] = __int_as_float((so.done ? 1 : 0) | (so.reached ? 2 : 0));
                    tr[6] = kbuf[lane];
                    if (B.info) {
#pragma unroll
                        for (int f = 0; f < 5; ++f) trs[2 * RMT * 8 + lane * 8 + f] = (float)so.info[f];
                        trs[2 * RMT * 8 + lane * 8 + 5] = so.obs[2];
                    }
                    float ob_old[6];
#pragma unroll
                    for (int f = 0; f < 6; ++f) ob_old[f] = xbuf[lane * 8 + f];
#pragma unroll
                    for (int f = 0; f < 6; ++f) trs[RMT * 8 + lane * 8 + f] = ob_old[f];
                    if (so.done) {
                        es.episode += 1;
                        env_begin_episode(P, eg, es, myvis);
                        env_obs(P, es, myvis, so.obs);
                    }
#pragma unroll
                    for (int f = 0; f < 6; ++f) xbuf[lane * 8 + f] = so.obs[f];
                    es_s[lane] = es;
                    kbuf[lane] = so.done ? 0.f : 1.f;
                    if (env_lane) {
                        const float* tq = trs + lane * 8;
#pragma unroll
                        for (int f = 0; f < 6; ++f) B.obs[row * 6 + f] = trs[RMT * 8 + lane * 8 + f];
                        B.act[row] = __float_as_int(tq[0]);
                        B.rew[row] = tq[1];
                        B.val[row] = tq[2];
                        B.logp[row] = tq[3];
                        B.done[row] = tq[4];
                        B.flags[row] = (uint8_t)__float_as_int(tq[5]);
                        B.keep[row] = tq[6];
                        if (B.info) {
#pragma unroll
                            for (int f = 0; f < 6; ++f) B.info[row * 6 + f] = trs[2 * RMT * 8 + lane * 8 + f];
                        }
                    }
                }
            }
            // all lanes of wave 0: bring the weight slice back (it was dead across the env block)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int sidx = 0; sidx < KS; ++sidx) wh[q][sidx] = whpark[(q * KS + sidx) * 64 + lane];
        }
        if (!value_only && t + 1 < steps) recurrent();              // bias + h_t W_hh^T for step t+1
        lds_barrier();                       // barrier 2: x_{t+1}, keep_{t+1} visible; recurrent reads of h_t done
        if (!value_only) {
            // episode ended at step t: the recurrent state restarts from zero (acc rows fall back to the bias)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float kp = kbuf[4 * kq + r];
                if (kp == 0.f) {
                    c_reg[r] = 0.f;
                    h_keep[r] = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q][r] = bias[q];
                }
            }
        }
    }
    // ------------------------------------------------------------------ write back persistent state
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * kq + r;
        if (n < N) {
            B.h[(size_t)n * H + u] = h_keep[r];
            B.c[(size_t)n * H + u] = c_reg[r];
        }
    }
    if (env_lane) {
        env_store(blob, my_env, es_s[lane]);
        for (int k = 0; k < NVIS; ++k) blob.visited[(size_t)my_env * NVIS + k] = myvis[k];
#pragma unroll
        for (int f = 0; f < 6; ++f) B.cur_obs[(size_t)my_env * 6 + f] = xbuf[lane * 8 + f];
    }
}

template <int H>
static int launch_rollout(const EnvParams& P, EnvBlob blob, int N, int T, uint64_t iter, const float* params,
                          const RolloutBufs& B, hipStream_t st) {
    const dim3 grid((N + RMT - 1) / RMT), block(H * 4);
    hipLaunchKernelGGL((rollout_lstm_kernel<H, 5>), grid, block, 0, st, P, blob, N, T, iter, params, B);
    UAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int uav_rollout(uav_ctx* ctx, void* env_state, int n_env, const uav_env_cfg* cfg, int policy_kind,
                           const float* params, int hidden, int horizon, uint64_t iter, float* cur_obs, float* h,
                           float* c, float* obs, int32_t* act, float* rew, float* val, float* logp, float* done,
                           uint8_t* flags, float* keep, float* last_val, const int32_t* forced_act,
                           const double* noise, int32_t* nan_count, float* stash, float* y_out, float* info,
                           uav_stream stream) {
    UAV_REQUIRE(ctx && env_state && params && cur_obs && obs && act && rew && val && logp && done && flags && nan_count,
                "uav_rollout: NULL argument");
    UAV_REQUIRE(n_env > 0 && horizon > 0, "uav_rollout: n_env=%d horizon=%d", n_env, horizon);
    UAV_REQUIRE(policy_kind == 1, "uav_rollout: only the LSTM policy (policy_kind 1) has a fused kernel; "
                                  "the MLP policy rolls out step by step (uav_mlp_fwd + uav_policy_sample + uav_env_step)");
    UAV_REQUIRE(h && c && keep, "uav_rollout: LSTM policy needs h, c, keep");
    EnvParams P;
    int rc = env_params_from_cfg(ctx, cfg, n_env, P);
    if (rc) return rc;
    UAV_REQUIRE((stash == nullptr) == (y_out == nullptr), "uav_rollout: stash and y_out go together");
    RolloutBufs B{cur_obs, h, c, obs, act, rew, val, logp, done, flags, keep, last_val, forced_act, noise, nan_count, info, stash, y_out};
    EnvBlob blob = env_blob_view(env_state, n_env);
    switch (hidden) {
        case 64: return launch_rollout<64>(P, blob, n_env, horizon, iter, params, B, as_stream(stream));
        case 128: return launch_rollout<128>(P, blob, n_env, horizon, iter, params, B, as_stream(stream));
    }
    UAV_REQUIRE(false, "uav_rollout: hidden=%d unsupported (64, 128)", hidden);
}
