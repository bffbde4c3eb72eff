// rollout.hip -- R1: fused persistent rollout for the LSTM actor-critic.
//
// Reference loop body: PPOV2.0/train_ppo2.0.py:157-198 (policy forward -> Categorical sample ->
// env.step -> buffer.store), here for N environments x T steps in ONE launch.
//
// One workgroup owns 16 environments for the whole horizon:
//   * H/16 "gate" waves keep the two fp16 pieces of their W_hh slice in VGPRs (as lstm_fwd_h3_kernel) and the
//     cell state in registers; h_t lives in two fp16 LDS planes;
//   * wave 0 additionally plays the "env" role: it computes the actor/critic heads of h_t with one
//     MFMA chain (head-weight pieces as A-fragments from LDS), then lanes 0..15 each sample an action
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
#include "loss_core.h"
#include "rows_dot_core.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

int env_params_from_cfg(const uav_ctx* ctx, const uav_env_cfg* cfg, int n_env, EnvParams& P);
int launch_rollout_mlp(uav_ctx* ctx, void* env_state, int n_env, const uav_env_cfg* cfg, const float* params, int horizon,
                       uint64_t iter, float* cur_obs, float* obs, int32_t* act, float* rew, float* val, float* logp,
                       float* done, uint8_t* flags, float* last_val, const int32_t* forced_act, const double* noise,
                       int32_t* nan_count, float* info, float* heads, hipStream_t st);

constexpr int RMT = 16;

#ifdef UAV_X6_PROFILE
// phase timing of wave 0 (instrumented build only, tools/build_prof.sh): cycles summed over the rollout by workgroup 0
__device__ unsigned long long g_roll_prof[8];
#define R_PROF_DECL unsigned long long pm_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pl_ = __builtin_readcyclecounter()
#define R_PROF_MARK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pm_[i] += n_ - pl_; pl_ = n_; } while (0)
#define R_PROF_FLUSH() do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) g_roll_prof[i_] = pm_[i_]; } while (0)
extern "C" int uav_roll_prof_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_roll_prof), sizeof(g_roll_prof)) == hipSuccess ? 0 : 1;
}
#else
#define R_PROF_DECL
#define R_PROF_MARK(i)
#define R_PROF_FLUSH()
#endif
constexpr float R_F32_EPS = 1.1920928955078125e-07f;

#define r_sigmoid fast_sigmoid
#define r_tanh fast_tanh


// LDS geometry of the split-fp16 rollout (dynamic part; the env role's small arrays are static)
template <int H>
struct RGeom {
    static constexpr int NW = H / 16;
    static constexpr int NS = H / 32;                 // K = 32 slabs over the hidden dimension
    static constexpr int RS = H + 8;                  // padded plane row (fp16): conflict-free ds_read_b128
    static constexpr int PLANE = RMT * RS;
    static constexpr int W0 = 4 * NS * 2 * 64 * 8;    // fp16: BOTH weight pieces of wave 0 (its VGPRs belong to the env role)
    static constexpr size_t LDS = (2 * PLANE /*h*/ + 2 * PLANE /*head weights*/ + W0) * sizeof(unsigned short) +
                                  (NW * 8 * 64 /*W_ih fragments*/ + 4 * H /*bias*/) * sizeof(float);
};

struct RolloutBufs {
    float* cur_obs; float* h; float* c;
    float* obs; int32_t* act; float* rew; float* val; float* logp; float* done; uint8_t* flags; float* keep;
    float* last_val; const int32_t* forced_act; const double* noise; int32_t* nan_count;
    float* info;                // optional [N][T][10]: 5 reward parts of environment.py:161-167, obs[2], agent_pos, source_pos of the step
    float* heads;               // optional [N][T][NA+1]: logits | value of the step
    float* stash; float* y;     // optional: BPTT stash [N][T][6H] + y [N][T][H], so PPO epoch 0 skips its forward pass
};

// The recurrent product h W_hh^T runs on the fp16 matrix pipe at f32 accuracy (two-piece operand split, three piece
// products into a main and a cross accumulator; common.h split2h, lstm.hip lstm_fwd_h3_kernel) in the weights-as-A
// orientation: lane (j, kq) owns env j and the four consecutive units uo..uo+3, so the stash leaves as dwordx4 stores and
// h_t is parked with one ds_write_b64 per piece.  Wave 0 (gate wave AND env role) keeps NO weights in registers -- both
// pieces of its slice sit in LDS and are read back as lane-contiguous b128 fragments -- so the f64 env chain has the
// register file to itself; the two roles run separate, barrier-matched time loops, which keeps the other waves' 128
// weight VGPRs out of wave 0's live set.
template <int H, int NA>
__global__ __launch_bounds__(H * 4) void rollout_lstm_kernel(EnvParams P_arg, EnvBlob blob, int N, int T,
                                                                        uint64_t iter, const float* __restrict__ params,
                                                                        RolloutBufs B) {
    using G = RGeom<H>;
    constexpr int NS = G::NS, RS = G::RS, PLANE = G::PLANE, W0 = G::W0, NW = G::NW;
    constexpr int I = 6, NH = NA + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* hpl = reinterpret_cast<unsigned short*>(smem);          // [2 pieces][RMT][RS] h_t
    unsigned short* whp = hpl + 2 * PLANE;                                  // [2 pieces][16 heads][RS] head weights
    unsigned short* w0p = whp + 2 * PLANE;                                  // [4 gates][NS][2 pieces][64 lanes][8]
    float* wxl = reinterpret_cast<float*>(w0p + W0);                        // [NW][4 gates][2 k-steps][64 lanes]
    float* bl = wxl + NW * 512;                                             // [4H] b_ih + b_hh
    __shared__ __attribute__((aligned(16))) float xbuf[RMT * 8];
    __shared__ float kbuf[RMT];
    __shared__ float hd[RMT * 16];
    __shared__ unsigned short vis[RMT * NVIS];
    __shared__ EnvState es_s[RMT];
    __shared__ float trs[4 * RMT * 8];                            // the step's transition, parked by wave 0, stored by the last gate wave
    __shared__ __attribute__((aligned(16))) f32x4 acc0[4 * 64];  // wave 0's next-step accumulators, computed by the gate waves
    __shared__ double env_tab[ENV_LDS_TABLE_DOUBLES];            // pow(vc, 0.75) | ripple factors (env_core.h)
    // one workgroup per CU at h = 128 (~145 KB with the 66 KB of env tables; h = 64: ~111 KB): any growth of a table or of the
    // geometry must fail HERE, not as a launch error on the GPU box
    static_assert(G::LDS + sizeof(xbuf) + sizeof(kbuf) + sizeof(hd) + sizeof(vis) + sizeof(es_s) + sizeof(trs) + sizeof(acc0) +
                      sizeof(env_tab) <= 160 * 1024, "rollout_lstm_kernel: LDS over 160 KB per workgroup");
    EnvParams P = P_arg;
    env_params_refresh(P);
    env_tables_to_lds(P, env_tab, threadIdx.x, H * 4);

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
    const int uw = 16 * w + j;                 // unit whose weight rows this lane holds / parks (A operand row)
    const int uo = 16 * w + 4 * kq;            // first of this lane's four units; its env is j
    const int nj = min(n0 + j, N - 1);
    const bool live = n0 + j < N;
    const int my_env = n0 + lane;              // env of lanes 0..15 of the env wave
    const bool env_lane = is_env_wave && lane < RMT && my_env < N;
    unsigned short* myvis = vis + (lane & 15) * NVIS;

    // ------------------------------------------------------------------ weights
    f16x8 wb[4][NS][2];                                                      // gate waves only
    f16x8* const w0f = reinterpret_cast<f16x8*>(w0p) + lane;                 // + ((q * NS + s) * 2 + piece) * 64
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float* src = w_hh + (size_t)(q * H + uw) * H + 32 * s + 8 * kq;
            const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            f16x8 p0v, p1v;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                _Float16 p0, p1;
                split2h(v[i], p0, p1);
                p0v[i] = p0; p1v[i] = p1;
            }
            if (is_env_wave) {
                w0f[((q * NS + s) * 2 + 0) * 64] = p0v;
                w0f[((q * NS + s) * 2 + 1) * 64] = p1v;
            } else {
                wb[q][s][0] = p0v; wb[q][s][1] = p1v;
            }
        }
    float* const wxw = wxl + w * 512 + lane;                                 // this lane's W_ih fragments: + (2 q + s) * 64
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k = 2 * kq + s;
            wxw[(2 * q + s) * 64] = (k < I) ? w_ih[(size_t)(q * H + uw) * I + k] : 0.f;
        }
    for (int idx = threadIdx.x; idx < 4 * H; idx += H * 4) bl[idx] = b_ih[idx] + b_hh[idx];
    for (int idx = threadIdx.x; idx < 16 * H; idx += H * 4) {                // head weights as two fp16 planes
        const int hdx = idx / H, uu = idx % H;
        _Float16 p0, p1;
        split2h((hdx < NH) ? w_hd[(size_t)hdx * H + uu] : 0.f, p0, p1);
        unsigned short* d = whp + hdx * RS + uu;
        d[0] = h_bits(p0); d[PLANE] = h_bits(p1);
    }
    auto put_h = [&](const float (&hv)[4]) {                                 // split and park h[env j][uo .. uo+3]
        unsigned short b[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            _Float16 p0, p1;
            split2h(hv[r], p0, p1);
            b[0][r] = h_bits(p0); b[1][r] = h_bits(p1);
        }
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            uint2 v;
            v.x = (unsigned)b[pc][0] | ((unsigned)b[pc][1] << 16);
            v.y = (unsigned)b[pc][2] | ((unsigned)b[pc][3] << 16);
            *reinterpret_cast<uint2*>(hpl + pc * PLANE + j * RS + uo) = v;
        }
    };
    float c_reg[4];
    {
        const float4 cv = *reinterpret_cast<const float4*>(B.c + (size_t)nj * H + uo);
        const float4 hv = *reinterpret_cast<const float4*>(B.h + (size_t)nj * H + uo);
        c_reg[0] = cv.x; c_reg[1] = cv.y; c_reg[2] = cv.z; c_reg[3] = cv.w;
        const float h_in[4] = {hv.x, hv.y, hv.z, hv.w};
        put_h(h_in);
    }
    if (is_env_wave && lane < RMT) {
        const int n = min(my_env, N - 1);
        es_s[lane] = env_load(blob, n);
        for (int k = 0; k < NVIS; ++k) myvis[k] = blob.visited[(size_t)n * NVIS + k];
#pragma unroll
        for (int f = 0; f < 8; ++f) xbuf[lane * 8 + f] = f < 6 ? B.cur_obs[(size_t)n * 6 + f] : 0.f;
        kbuf[lane] = 1.f;
    }
    lds_barrier();

    f32x4 acc[4];
    // One LDS address register for the four bias rows (+ q * H floats as the ds_read immediate): with a transparent lane offset
    // the compiler folds bl's > 64 KB offset into four separate address VGPRs, one of which the gate waves then spill and
    // reload inside the time loop -- and a scratch reload means s_waitcnt vmcnt(0), i.e. waiting for the stash stores.
    int bo = uo;
    asm volatile("" : "+v"(bo));
    const float* blu = bl + bo;
    auto bias_acc = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(blu + q * H);
            acc[q] = f32x4{v.x, v.y, v.z, v.w};
        }
    };
    // acc = bias + W_hh h^T of the step about to run: main products onto the bias, cross products apart, combined at the end
    auto recurrent = [&](auto&& wp0, auto&& wp1) {
        bias_acc();
        f32x4 acl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acl[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned short* hrow = hpl + j * RS + 8 * kq;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(hrow + 32 * s);
            const f16x8 a1 = *reinterpret_cast<const f16x8*>(hrow + PLANE + 32 * s);
#pragma unroll
            for (int q = 0; q < 4; ++q) acl[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp1(q, s), a0, acl[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp0(q, s), a0, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acl[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp0(q, s), a1, acl[q], 0, 0, 0);
            asm volatile("" ::: "memory");               // one slab's h fragments at a time
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = acc[q] + acl[q] * H3_LO;
    };
    auto recurrent_regs = [&]() {
        recurrent([&](int q, int s) { return wb[q][s][0]; }, [&](int q, int s) { return wb[q][s][1]; });
    };
    // Wave 0's share of the recurrent product (units 0..15 of the four gates), computed by the gate waves from the
    // LDS copy of its weights while they would otherwise wait at barrier 2 for the env step.  Tile q goes to wave
    // 1 + q % (NW - 1), except that wave NW/2 -- wave 0's SIMD partner, whose MFMAs would serialise with the env chain's
    // VALU work (tools/simd_overlap_probe.hip) -- hands its tile to the next wave.  Same operand fragments and
    // accumulation order as recurrent(), so the bits do not change.
    auto recurrent_for_wave0 = [&]() {
        const f16x8* const wf = reinterpret_cast<const f16x8*>(w0p) + lane;
        const unsigned short* hrow = hpl + j * RS + 8 * kq;
        const int q0 = (NW == 8) ? (w < 4 ? w - 1 : (w == 5 ? 3 : 4)) : w - 1;
        for (int q = q0; q < 4; q += NW - 1) {
            const float4 bv = *reinterpret_cast<const float4*>(bl + q * H + 4 * kq);
            f32x4 e = {bv.x, bv.y, bv.z, bv.w}, el = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(hrow + 32 * s);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(hrow + PLANE + 32 * s);
                const f16x8 p0 = wf[((q * NS + s) * 2 + 0) * 64], p1 = wf[((q * NS + s) * 2 + 1) * 64];
                el = __builtin_amdgcn_mfma_f32_16x16x32_f16(p1, a0, el, 0, 0, 0);
                e = __builtin_amdgcn_mfma_f32_16x16x32_f16(p0, a0, e, 0, 0, 0);
                el = __builtin_amdgcn_mfma_f32_16x16x32_f16(p0, a1, el, 0, 0, 0);
                asm volatile("" ::: "memory");
            }
            acc0[q * 64 + lane] = e + el * H3_LO;
        }
    };
    auto recurrent_lds = [&]() {
        recurrent([&](int q, int s) { return w0f[((q * NS + s) * 2 + 0) * 64]; },
                  [&](int q, int s) { return w0f[((q * NS + s) * 2 + 1) * 64]; });
    };
    // phase 1: input projection (two exact-f32 k-steps), gate pointwise, h_t parked, stash / y stored.
    // The recurrent state handed to the next rollout (B.h) is stored at the last real step; keep_fixup zeroes it
    // again if that step ended the episode.
    auto finish_cell = [&](int t, bool value_only) {
        const float2 ax = *reinterpret_cast<const float2*>(&xbuf[j * 8 + 2 * kq]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxw[(2 * q) * 64], ax.x, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxw[(2 * q + 1) * 64], ax.y, acc[q], 0, 0, 0);
        }
        const bool st = B.stash && !value_only && live;       // the update's first epoch reuses this forward pass
        const unsigned row = (unsigned)(n0 + j) * T + t;       // N*T*6H < 2^32 elements is checked at launch
        float* sp = B.stash + (size_t)row * (6 * H) + uo;
        float gi[4], gf[4], gg[4], cc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gi[r] = r_sigmoid(acc[0][r]); gf[r] = r_sigmoid(acc[1][r]); gg[r] = r_tanh(acc[2][r]);
            cc[r] = gf[r] * c_reg[r] + gi[r] * gg[r];
        }
        if (st) {
            *reinterpret_cast<float4*>(sp) = float4{gi[0], gi[1], gi[2], gi[3]};
            *reinterpret_cast<float4*>(sp + H) = float4{gf[0], gf[1], gf[2], gf[3]};
            *reinterpret_cast<float4*>(sp + 2 * H) = float4{gg[0], gg[1], gg[2], gg[3]};
            *reinterpret_cast<float4*>(sp + 4 * H) = float4{c_reg[0], c_reg[1], c_reg[2], c_reg[3]};
        }
        float go[4], hh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            go[r] = r_sigmoid(acc[3][r]);
            hh[r] = go[r] * r_tanh(cc[r]);
            if (!value_only) c_reg[r] = cc[r];
        }
        put_h(hh);
        if (st) {
            *reinterpret_cast<float4*>(sp + 3 * H) = float4{go[0], go[1], go[2], go[3]};
            *reinterpret_cast<float4*>(B.y + (size_t)row * H + uo) = float4{hh[0], hh[1], hh[2], hh[3]};
        }
        if (t == T - 1 && live)
            *reinterpret_cast<float4*>(B.h + (size_t)(n0 + j) * H + uo) = float4{hh[0], hh[1], hh[2], hh[3]};
    };
    // episode ended at step t: the recurrent state restarts from zero (acc falls back to the bias)
    auto keep_fixup = [&](int t) {
        if (kbuf[j] == 0.f) {
#pragma unroll
            for (int r = 0; r < 4; ++r) c_reg[r] = 0.f;
            bias_acc();
            if (t == T - 1 && live) *reinterpret_cast<float4*>(B.h + (size_t)(n0 + j) * H + uo) = float4{0.f, 0.f, 0.f, 0.f};
        }
    };

    // The transition of step t leaves for HBM from the last gate wave, not from wave 0 (whose heads -> action -> env chain
    // paces the kernel): wave 0 parks it in trs ([action, reward, V, logp, done, flags, keep, -] | [obs 6, source x, y] |
    // [info 8] | [logits 5]) and the last gate wave stores it right after barrier 2.  trs is rewritten only after barrier 1
    // of the next step, which that wave passes after its reads have completed.  (Spreading the stores over three gate
    // waves measured slower, 792 vs 773 us per rollout.)
    auto store_transition = [&](int t) {
        if (lane < RMT && n0 + lane < N) {
            const size_t row = (size_t)(n0 + lane) * T + t;
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
            if (B.heads) {
#pragma unroll
                for (int a = 0; a < NA; ++a) B.heads[row * NH + a] = trs[3 * RMT * 8 + lane * 8 + a];
                B.heads[row * NH + NA] = tq[2];
            }
            if (B.info) {
#pragma unroll
                for (int f = 0; f < 8; ++f) B.info[row * 10 + f] = trs[2 * RMT * 8 + lane * 8 + f];
                B.info[row * 10 + 8] = trs[RMT * 8 + lane * 8 + 6];
                B.info[row * 10 + 9] = trs[RMT * 8 + lane * 8 + 7];
            }
        }
    };

    const int steps = T + (B.last_val ? 1 : 0);   // one extra value-only pass for V(s_T)
    if (!is_env_wave) {
        // ------------------------------------------------------------------ gate waves
        recurrent_regs();
        lds_barrier();     // every wave has read h_{-1} before step 0 overwrites the planes
        for (int t = 0; t < steps; ++t) {
            const bool value_only = (t == T);
            finish_cell(t, value_only);
            lds_barrier();                       // barrier 1: h_t visible
            if (!value_only && t + 1 < steps) {
                recurrent_regs();                // bias + W_hh h_t for step t+1
                recurrent_for_wave0();
            }
            lds_barrier();                       // barrier 2: x_{t+1}, keep_{t+1} visible; reads of h_t done
            if (!value_only) {
                keep_fixup(t);
                if (w == NW - 1) store_transition(t);
            }
        }
    } else {
        // ------------------------------------------------------------------ wave 0: gate wave + env role
        recurrent_lds();
        lds_barrier();
        float bh[NH];                            // head biases: read once (a load inside the loop could not be hoisted past the stores)
#pragma unroll
        for (int a = 0; a < NH; ++a) bh[a] = b_hd[a];
        R_PROF_DECL;
        for (int t = 0; t < steps; ++t) {
            const bool value_only = (t == T);
            R_PROF_MARK(7);
            finish_cell(t, value_only);
            R_PROF_MARK(0);
            // the step's action-independent part (sampling uniform, the wind displacement from its two normals), computed
            // while the other waves finish their cells: off the heads -> action -> env chain that paces the kernel
            uint32_t u_act = 0;
            double wind_x = 0.0, wind_y = 0.0;
            if (lane < RMT && !value_only) {
                double z0, z1;
                const int eg = P.env_offset + my_env;
                const size_t row = (size_t)min(my_env, N - 1) * T + t;
                if (!B.forced_act) u_act = philox4x32_10(P.seed, (uint32_t)t, (uint32_t)eg, (uint32_t)iter, RNG_ACTION).x;
                if (B.noise) { z0 = B.noise[2 * row]; z1 = B.noise[2 * row + 1]; }
                else env_step_noise(P, eg, es_s[lane], z0, z1);
                env_step_wind(es_s[lane], z0, z1, wind_x, wind_y);
            }
            R_PROF_MARK(6);
            lds_barrier();                       // barrier 1: h_t visible
            R_PROF_MARK(1);
            // heads of h_t: D[head 4 kq + r][env j] = W_head h_t^T, three piece products per slab
            f32x4 ha = {0.f, 0.f, 0.f, 0.f}, hb2 = ha;
            {
                const unsigned short* hrow = hpl + j * RS + 8 * kq;
                const unsigned short* wrow = whp + j * RS + 8 * kq;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    f16x8 a[2], bw[2];
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        a[pc] = *reinterpret_cast<const f16x8*>(hrow + pc * PLANE + 32 * s);
                        bw[pc] = *reinterpret_cast<const f16x8*>(wrow + pc * PLANE + 32 * s);
                    }
                    hb2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bw[1], a[0], hb2, 0, 0, 0);
                    ha = __builtin_amdgcn_mfma_f32_16x16x32_f16(bw[0], a[0], ha, 0, 0, 0);
                    hb2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bw[0], a[1], hb2, 0, 0, 0);
                }
                ha = ha + hb2 * H3_LO;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) hd[j * 16 + 4 * kq + r] = ha[r];
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the hd tile is written (single wave)
            __builtin_amdgcn_wave_barrier();
            R_PROF_MARK(2);
            if (lane < RMT) {
                float z[NA], p[NA];
#pragma unroll
                for (int a = 0; a < NA; ++a) z[a] = hd[lane * 16 + a] + bh[a];
                const float V = hd[lane * 16 + NA] + bh[NA];
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
                        const float target = u01_f32(u_act) * psum;
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
                    float psel = 0.f;
#pragma unroll
                    for (int a = 0; a < NA; ++a) if (a == a_sel) psel = p[a];
                    const float qa = psel / psum;
                    const float lp = __logf(fminf(fmaxf(qa, R_F32_EPS), 1.0f - R_F32_EPS));
                    R_PROF_MARK(3);
                    // environment step (f64, env_core.h) + auto reset
                    EnvState es = es_s[lane];
                    StepOut so;
                    env_step_core(P, eg, es, myvis, a_sel, wind_x, wind_y, so);
                    R_PROF_MARK(4);
                    // park the transition in LDS for store_transition()
                    float* tr = trs + lane * 8;
                    tr[0] = __int_as_float(a_sel);
                    tr[1] = (float)so.reward;
                    tr[2] = V;
                    tr[3] = lp;
                    tr[4] = so.done ? 1.f : 0.f;
                    tr[5] = __int_as_float((so.done ? 1 : 0) | (so.reached ? 2 : 0));
                    tr[6] = kbuf[lane];
                    if (B.heads) {
#pragma unroll
                        for (int a = 0; a < NA; ++a) trs[3 * RMT * 8 + lane * 8 + a] = z[a];
                    }
                    if (B.info) {
#pragma unroll
                        for (int f = 0; f < 5; ++f) trs[2 * RMT * 8 + lane * 8 + f] = (float)so.info[f];
                        trs[2 * RMT * 8 + lane * 8 + 5] = so.obs[2];
                        trs[2 * RMT * 8 + lane * 8 + 6] = es.px;          // agent_pos after the move (before any auto-reset)
                        trs[2 * RMT * 8 + lane * 8 + 7] = es.py;
                    }
                    float ob_old[6];
#pragma unroll
                    for (int f = 0; f < 6; ++f) ob_old[f] = xbuf[lane * 8 + f];
#pragma unroll
                    for (int f = 0; f < 6; ++f) trs[RMT * 8 + lane * 8 + f] = ob_old[f];
                    trs[RMT * 8 + lane * 8 + 6] = (float)es.sx;           // source of the episode this step belongs to
                    trs[RMT * 8 + lane * 8 + 7] = (float)es.sy;
                    if (so.done) {
                        es.episode += 1;
                        env_begin_episode(P, eg, es, myvis);
                        env_obs(P, es, myvis, so.obs);
                    }
#pragma unroll
                    for (int f = 0; f < 6; ++f) xbuf[lane * 8 + f] = so.obs[f];
                    es_s[lane] = es;
                    kbuf[lane] = so.done ? 0.f : 1.f;
                }
            }
            R_PROF_MARK(5);
            lds_barrier();                       // barrier 2: the gate waves have left bias + W_hh h_t of this wave's units in acc0
            if (!value_only && t + 1 < steps) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = acc0[q * 64 + lane];
            }
            if (!value_only) keep_fixup(t);
        }
        R_PROF_FLUSH();
    }
    // ------------------------------------------------------------------ write back persistent state
    if (live) *reinterpret_cast<float4*>(B.c + (size_t)(n0 + j) * H + uo) = float4{c_reg[0], c_reg[1], c_reg[2], c_reg[3]};
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
    UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&rollout_lstm_kernel<H, 5>), (int)RGeom<H>::LDS));
    hipLaunchKernelGGL((rollout_lstm_kernel<H, 5>), grid, block, RGeom<H>::LDS, st, P, blob, N, T, iter, params, B);
    UAV_LAUNCH_CHECK();
    return 0;
}

// ---- the tail of ONE step of a step-wise rollout (policies the fused rollout kernels do not cover: stacked layers, h = 256)
// as one launch: policy heads of the top layer's y_t, action sample, environment step, PPOBuffer.store, next observation into
// the sequence array -- five launches (rows-dot GEMM 14.3 us, uav_policy_sample_at 4.3, uav_env_step 8.6, uav_store_transition
// 2.9, a copy 3.4 at 4096 envs, each mostly launch and memory latency) that all sit on the step's dependency chain.
// A wave takes four envs: the heads as rows_dot_kernel forms them (rows_dot_core.h: the same sums, so heads[:, t] is bit for
// bit uav_gemm_f32's), then lanes 0..3 each carry one env through uav_policy_sample_at's draw (loss_core.h) and
// env_step_kernel's body (env_core.h).  4096 envs = 1024 waves = one per SIMD.
template <int A>
__global__ __launch_bounds__(256) void rollout_tail_kernel(
    EnvParams P, EnvBlob b, int n, const float* __restrict__ y, int64_t ldy, int K, const float* __restrict__ w_head,
    const float* __restrict__ b_head, float* __restrict__ heads, int64_t ldh, int T, int t, uint64_t seed, uint64_t counter,
    int64_t index_offset, const int32_t* __restrict__ forced, const double* __restrict__ noise, int32_t* __restrict__ act_out,
    float* __restrict__ cur_obs, float* __restrict__ obs_seq, float* __restrict__ keep, int32_t* __restrict__ act_buf,
    float* __restrict__ val_buf, float* __restrict__ logp_buf, float* __restrict__ keep_buf, float* __restrict__ rew_buf,
    float* __restrict__ done_buf, uint8_t* __restrict__ flags_buf, int32_t* __restrict__ nan_count) {
    const int lane = threadIdx.x & 63;
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (m0 >= n) return;
    env_params_refresh(P);
    RowsDot<8, 1> rd;
    rd.load_w(w_head, K, A + 1, K, lane);
    float red[4];
    rd.rows4(y, ldy, m0, n, K, lane, red);
    float z[A + 1];
#pragma unroll
    for (int o = 0; o <= A; ++o) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float tot = RowsDot<8, 1>::total(red[r], o);
            if (lane == r) v = tot;
        }
        z[o] = v + b_head[o];
    }
    const int64_t i64 = m0 + lane;
    if (lane >= 4 || i64 >= n) return;
    const int i = (int)i64;
#pragma unroll
    for (int o = 0; o <= A; ++o) heads[i64 * ldh + o] = z[o];
    // ---- uav_policy_sample_at
    float p[A], qa;
    bool bad;
    int a = sample_categorical<A>(z, p, nullptr, seed, counter, index_offset + i, forced ? &forced[i] : nullptr, qa, bad);
    if (bad) atomicAdd(nan_count, 1);
    const int64_t col = i64 * T + t;
    act_out[i] = a;
    act_buf[col] = a;
    logp_buf[col] = logf(fminf(fmaxf(qa, F32_EPS), 1.0f - F32_EPS));
    val_buf[col] = z[A];
    // ---- uav_env_step (env_step_kernel's body)
    EnvState s = env_load(b, i);
    unsigned short* vis = b.visited + (size_t)i * NVIS;
    const int eg = P.env_offset + i;
    double z0, z1;
    if (noise) { z0 = noise[2 * (size_t)i]; z1 = noise[2 * (size_t)i + 1]; }
    else env_step_noise(P, eg, s, z0, z1);
    a = a < 0 ? 0 : (a > 4 ? 4 : a);
    StepOut o;
    double tx, ty;
    env_step_wind(s, z0, z1, tx, ty);
    env_step_core(P, eg, s, vis, a, tx, ty, o);
    const float d = o.done ? 1.f : 0.f;
    const uint8_t fl = (uint8_t)((o.done ? 1 : 0) | (o.reached ? 2 : 0));
    const float rw = (float)o.reward;
    if (o.done) {                                   // the reset of train_ppo2.0.py:139
        s.episode += 1;
        env_begin_episode(P, eg, s, vis);
        env_obs(P, s, vis, o.obs);
    }
    env_store(b, i, s);
    // ---- uav_store_transition, and the observation step t + 1 starts from
    keep_buf[col] = keep[i];
    rew_buf[col] = rw;
    done_buf[col] = d;
    flags_buf[col] = fl;
    keep[i] = 1.0f - d;
    const int od = 6 + P.trend_k;
    for (int k = 0; k < od; ++k) {
        cur_obs[(size_t)i * od + k] = o.obs[k];
        if (t + 1 < T) obs_seq[((size_t)i * T + t + 1) * od + k] = o.obs[k];
    }
}

extern "C" int uav_rollout_tail(uav_ctx* ctx, void* env_state, int n_env, const uav_env_cfg* cfg, const float* y, int64_t y_stride,
                                int hidden, const float* w_head, const float* b_head, int n_act, float* heads, int64_t heads_stride,
                                int T, int t, uint64_t seed, uint64_t counter, int64_t index_offset, const int32_t* forced_act,
                                const double* noise, int32_t* act_out, float* cur_obs, float* obs_seq, float* keep,
                                int32_t* act_buf, float* val_buf, float* logp_buf, float* keep_buf, float* rew_buf, float* done_buf,
                                uint8_t* flags_buf, int32_t* nan_count, uav_stream stream) {
    UAV_REQUIRE(ctx && env_state && y && w_head && b_head && heads && act_out && cur_obs && obs_seq && keep && act_buf && val_buf &&
                logp_buf && keep_buf && rew_buf && done_buf && flags_buf && nan_count, "uav_rollout_tail: NULL argument");
    UAV_REQUIRE(n_env > 0 && T > 0 && t >= 0 && t < T, "uav_rollout_tail: n_env=%d T=%d t=%d", n_env, T, t);
    UAV_REQUIRE(hidden >= 4 && hidden <= 256 && hidden % 4 == 0 && y_stride % 4 == 0 && y_stride >= hidden &&
                (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(w_head) & 15) == 0,
                "uav_rollout_tail: hidden %d (a multiple of 4 up to 256), y rows 16-byte aligned", hidden);
    UAV_REQUIRE(heads_stride > n_act, "uav_rollout_tail: heads_stride=%lld n_act=%d", (long long)heads_stride, n_act);
    EnvParams P;
    int rc = env_params_from_cfg(ctx, cfg, n_env, P);
    if (rc) return rc;
    const dim3 grid((unsigned)((n_env + 15) / 16));
#define LAUNCH_T(A_)                                                                                                              \
    hipLaunchKernelGGL(rollout_tail_kernel<A_>, grid, dim3(256), 0, as_stream(stream), P, env_blob_view(env_state, n_env), n_env, \
                       y, y_stride, hidden, w_head, b_head, heads, heads_stride, T, t, seed, counter, index_offset, forced_act,   \
                       noise, act_out, cur_obs, obs_seq, keep, act_buf, val_buf, logp_buf, keep_buf, rew_buf, done_buf,           \
                       flags_buf, nan_count)
    switch (n_act) {
        case 2: LAUNCH_T(2); break;
        case 3: LAUNCH_T(3); break;
        case 4: LAUNCH_T(4); break;
        case 5: LAUNCH_T(5); break;
        case 6: LAUNCH_T(6); break;
        default: UAV_REQUIRE(false, "uav_rollout_tail: n_act=%d unsupported (2..6)", n_act);
    }
#undef LAUNCH_T
    UAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int uav_rollout(uav_ctx* ctx, void* env_state, int n_env, const uav_env_cfg* cfg, int policy_kind,
                           const float* params, int hidden, int horizon, uint64_t iter, float* cur_obs, float* h,
                           float* c, float* obs, int32_t* act, float* rew, float* val, float* logp, float* done,
                           uint8_t* flags, float* keep, float* last_val, const int32_t* forced_act,
                           const double* noise, int32_t* nan_count, float* stash, float* y_out, float* info,
                           float* heads, uav_stream stream) {
    UAV_REQUIRE(ctx && env_state && params && cur_obs && obs && act && rew && val && logp && done && flags && nan_count,
                "uav_rollout: NULL argument");
    UAV_REQUIRE(n_env > 0 && horizon > 0, "uav_rollout: n_env=%d horizon=%d", n_env, horizon);
    UAV_REQUIRE(policy_kind == 0 || policy_kind == 1, "uav_rollout: policy_kind %d (0 = MLP, 1 = LSTM)", policy_kind);
    if (policy_kind == 0) {      // the reference's MLP policy: csrc/mlp_fused.hip
        UAV_REQUIRE(!stash && !y_out, "uav_rollout: stash / y_out belong to the LSTM policy");
        return launch_rollout_mlp(ctx, env_state, n_env, cfg, params, horizon, iter, cur_obs, obs, act, rew, val, logp, done,
                                  flags, last_val, forced_act, noise, nan_count, info, heads, as_stream(stream));
    }
    UAV_REQUIRE(h && c && keep, "uav_rollout: LSTM policy needs h, c, keep");
    EnvParams P;
    int rc = env_params_from_cfg(ctx, cfg, n_env, P);
    if (rc) return rc;
    UAV_REQUIRE((stash == nullptr) == (y_out == nullptr), "uav_rollout: stash and y_out go together");
    UAV_REQUIRE(!stash || (int64_t)n_env * horizon * 6 * hidden < (1ll << 32), "uav_rollout: stash rows exceed 32-bit offsets");
    RolloutBufs B{cur_obs, h, c, obs, act, rew, val, logp, done, flags, keep, last_val, forced_act, noise, nan_count, info, heads, stash, y_out};
    EnvBlob blob = env_blob_view(env_state, n_env);
    switch (hidden) {
        case 64: return launch_rollout<64>(P, blob, n_env, horizon, iter, params, B, as_stream(stream));
        case 128: return launch_rollout<128>(P, blob, n_env, horizon, iter, params, B, as_stream(stream));
    }
    UAV_REQUIRE(false, "uav_rollout: hidden=%d unsupported (64, 128)", hidden);
}
