// gae.hip -- G1 (GAE scan) and G2 (whole-buffer advantage normalisation + returns).
//
// Reference: PPOV2.0/train_ppo2.0.py:18-32 (scan), :35-40 (normalise, returns).
// HBM-bound: per env-step 12 B read + 4 B write (scan), 4 B read (stats), 8 B read + 8 B
// write (normalise).  Layout (env, T): one wavefront owns one env row, lane i <-> t = 64c+i,
// so every load/store of a chunk is one coalesced 256-B wave access.
#include "common.h"

// A[t] = b_t + a_t * A[t+1] is an affine map; the suffix composition over a 64-step chunk is a
// Hillis-Steele scan on (a, b) pairs with wave shuffles:  (a1,b1) o (a2,b2) = (a1*a2, b1 + a1*b2).
__global__ __launch_bounds__(256) void gae_scan_kernel(const float* __restrict__ rew,
                                                       const float* __restrict__ val,
                                                       const float* __restrict__ done,
                                                       const float* __restrict__ last_val, int n_env,
                                                       int T, float gamma, float gl, int mode,
                                                       float* __restrict__ adv) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int env = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (env >= n_env) return;   // whole wave exits together (env is wave-uniform)
    const size_t row = (size_t)env * T;
    const int nchunk = (T + 63) >> 6;
    float carry = 0.f;          // A at the first step of the chunk to the right
    float v_right = 0.f, d_right = 0.f;   // val/done of that step
    for (int c = nchunk - 1; c >= 0; --c) {
        const int t = c * 64 + lane;
        const bool in = t < T;
        const float r = in ? rew[row + t] : 0.f;
        const float v = in ? val[row + t] : 0.f;
        const float d = in ? done[row + t] : 0.f;
        // value / done of step t+1: lane+1, or the saved first element of the next chunk
        float v1 = __shfl_down(v, 1, 64);
        float d1 = __shfl_down(d, 1, 64);
        if (lane == 63) { v1 = v_right; d1 = d_right; }
        float a = 1.f, b = 0.f;
        if (in) {
            float nnt, nv;
            const bool last = (t == T - 1);
            if (mode == UAV_GAE_REFERENCE_EXACT) {
                // train_ppo2.0.py:23-28: mask from done[t+1]; the last step uses its own done/value
                nnt = 1.0f - (last ? d : d1);
                nv = (last ? v : v1) * nnt;
            } else {
                nnt = 1.0f - d;
                nv = (last ? (last_val ? last_val[env] : 0.f) : v1) * nnt;
            }
            b = (r + gamma * nv) - v;     // delta, train_ppo2.0.py:30
            a = gl * nnt;                 // gamma*lambda*next_non_terminal, :31
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float a2 = __shfl_down(a, o, 64);
            const float b2 = __shfl_down(b, o, 64);
            if (lane + o < 64) {
                b = b + a * b2;
                a = a * a2;
            }
        }
        const float A = b + a * carry;
        if (in) adv[row + t] = A;
        carry = __shfl(A, 0, 64);
        v_right = __shfl(v, 0, 64);
        d_right = __shfl(d, 0, 64);
    }
}

// ---- statistics: deterministic two-stage f64 reduction ------------------------------------------
constexpr int STATS_BLOCKS = 512;

__global__ __launch_bounds__(256) void adv_stats_partial(const float* __restrict__ x, int64_t n,
                                                         double* __restrict__ partial) {
    __shared__ double sm[4];
    double s = 0.0, q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = (double)x[i];
        s += v;
        q += v * v;
    }
    s = block256_sum(s, sm);
    q = block256_sum(q, sm);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = s;
        partial[2 * blockIdx.x + 1] = q;
    }
}

__global__ __launch_bounds__(256) void adv_stats_final(const double* __restrict__ partial, int nb,
                                                       int64_t n, double* __restrict__ stats3) {
    __shared__ double sm[4];
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) {
        s += partial[2 * i];
        q += partial[2 * i + 1];
    }
    s = block256_sum(s, sm);
    q = block256_sum(q, sm);
    if (threadIdx.x == 0) {
        stats3[0] = s;
        stats3[1] = q;
        stats3[2] = (double)n;
    }
}

__global__ __launch_bounds__(256) void adv_normalise_kernel(const float* __restrict__ adv,
                                                            const float* __restrict__ val, int64_t n,
                                                            const double* __restrict__ stats3,
                                                            float* __restrict__ adv_out,
                                                            float* __restrict__ ret_out) {
#pragma clang fp contract(off)
    const double cnt = stats3[2];
    const double mean_d = stats3[0] / cnt;
    // unbiased variance (torch .std()); cnt==1 gives 0/0 = NaN, caught by the reference's guard
    const double var = (stats3[1] - cnt * mean_d * mean_d) / (cnt - 1.0);
    const float mean = (float)mean_d;
    float sd = (float)sqrt(var > 0.0 ? var : (var == var ? 0.0 : var));
    if (sd < 1e-6f || sd != sd) sd = 1.0f;            // train_ppo2.0.py:37-38
    const float den = sd + 1e-6f;                     // :39
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float a = (adv[i] - mean) / den;
        adv_out[i] = a;
        ret_out[i] = a + val[i];                      // :40 returns from the NORMALISED advantage
    }
}

// ---- T1 input: the success bits of the episodes that ended in a rollout, compacted in (env, time) order.
// flags u8 [n] (bit0 done, bit1 reached; non-zero exactly where an episode ended) -> msg u8 [4 + cap + 1]:
// count (4 bytes, little endian) | bit1 of the k-th ended episode at msg[4 + k], k < cap | one spare byte.
// One block: every thread counts its contiguous chunk, a block scan gives its write offset (order preserving, no atomics).
__global__ __launch_bounds__(1024) void pack_success_kernel(const uint8_t* __restrict__ flags, int64_t n, int cap,
                                                            uint8_t* __restrict__ msg) {
    __shared__ int part[32];
    const int tid = threadIdx.x;
    int64_t per = (n + 1023) / 1024;
    per = (per + 15) / 16 * 16;                         // 16-byte loads; ended episodes are rare, so whole zero words are skipped
    const int64_t lo = (int64_t)tid * per, hi = (lo + per < n) ? lo + per : n;
    const bool vec = (reinterpret_cast<uintptr_t>(flags) & 15) == 0;
    auto visit = [&](auto&& f) {                        // f(byte) for every non-zero flag of this thread's chunk, in order
        int64_t i = lo;
        if (vec)
            for (; i + 16 <= hi; i += 16) {
                const uint4 w = *reinterpret_cast<const uint4*>(flags + i);
                if ((w.x | w.y | w.z | w.w) == 0u) continue;
                const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint8_t v = (ww[q] >> (8 * k)) & 255;
                        if (v) f(v);
                    }
            }
        for (; i < hi; ++i)
            if (flags[i]) f(flags[i]);
    };
    int cnt = 0;
    visit([&](uint8_t) { ++cnt; });
    for (int i = tid; i < 4 + cap + 1; i += 1024) msg[i] = 0;
    // exclusive prefix of cnt over the 1024 threads: wave scan by shuffles, then the 16 wave totals by the first wave
    const int lane = tid & 63, wv = tid >> 6;
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) part[wv] = inc;
    __syncthreads();
    if (wv == 0) {
        int t = lane < 16 ? part[lane] : 0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int v = __shfl_up(t, o, 64);
            if (lane >= o) t += v;
        }
        if (lane < 16) part[16 + lane] = t;             // inclusive totals of waves 0..lane
    }
    __syncthreads();
    int pos = inc - cnt + (wv ? part[16 + wv - 1] : 0);
    if (tid == 0) {
        const unsigned total = (unsigned)part[31];
        msg[0] = total & 255; msg[1] = (total >> 8) & 255; msg[2] = (total >> 16) & 255; msg[3] = (total >> 24) & 255;
    }
    if (cnt) visit([&](uint8_t v) {
        if (pos < cap) msg[4 + pos] = (v >> 1) & 1;
        ++pos;
    });
}

// ---- T1 on the device.  PPOTrainer.update (model.py:131-164), one finished episode at a time in the order of the messages
// (rank, then (env, time) inside a rank = global (env, time) order), f64 throughout like the reference's Python / numpy scalars.
struct CurriculumState {
    double radius, bonus, bonus_is_f64, overflow;      // read by the env kernels (env_params_refresh): the first three
    long long episodes, successes;                     // finished episodes / successes seen since init
    int hist_len, win_succ;                            // the 120-episode window: its length and the successes in it
    int pad_[2];
};
constexpr double CUR_MIN_RADIUS = 5.0, CUR_INITIAL_RADIUS = 50.0, CUR_RADIUS_DECAY = 0.9;       // config.py:27-29
constexpr double CUR_SUCCESS_THRESHOLD = 0.6, CUR_DECAY_FACTOR = 0.999;                         // config.py:30, 22
constexpr int CUR_WINDOW = 120;                                                                // config.py:31

__global__ void curriculum_init_kernel(CurriculumState* s, double radius, double bonus, int is_f64) {
    if (threadIdx.x != 0) return;
    s->radius = radius; s->bonus = bonus; s->bonus_is_f64 = is_f64 ? 1.0 : 0.0; s->overflow = 0.0;
    s->episodes = 0; s->successes = 0; s->hist_len = 0; s->win_succ = 0;
}

// N1: one thread walks one env row through the rollout (f64 running sums, sequential in t: what a host loop over the buffers
// does), and appends a row for every episode that ends.  64-thread blocks: 4096 envs = 64 workgroups on a side stream.
__global__ __launch_bounds__(64) void episode_rows_kernel(const float* __restrict__ rew, const float* __restrict__ info,
                                                          const uint8_t* __restrict__ flags, int n, int T, int env_offset,
                                                          double* __restrict__ carry, double* __restrict__ rows, int cap,
                                                          int32_t* __restrict__ count) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n) return;
    double s[6], steps;
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = carry[(size_t)e * 8 + k];
    steps = carry[(size_t)e * 8 + 6];
    const float* r = rew + (size_t)e * T;
    const float* in = info + (size_t)e * T * 10;
    const uint8_t* f = flags + (size_t)e * T;
    for (int t = 0; t < T; ++t) {
        s[0] += (double)r[t];
#pragma unroll
        for (int k = 0; k < 5; ++k) s[1 + k] += (double)in[(size_t)t * 10 + k];
        steps += 1.0;
        const uint8_t fl = f[t];
        if (fl & 1) {
            const int slot = atomicAdd(count, 1);
            if (slot < cap) {
                double* o = rows + (size_t)slot * 12;
                o[0] = (double)(env_offset + e); o[1] = (double)t;
#pragma unroll
                for (int k = 0; k < 6; ++k) o[2 + k] = s[k];
                o[8] = steps;
                o[9] = (fl & 2) ? 1.0 : 0.0;
                o[10] = (fl & 2) ? (double)in[(size_t)t * 10 + 5] * 100.0 : 0.0;           // train_ppo2.0.py:203
                o[11] = 0.0;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) s[k] = 0.0;
            steps = 0.0;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) carry[(size_t)e * 8 + k] = s[k];
    carry[(size_t)e * 8 + 6] = steps;
}

// One wave: the 64 lanes stage a message chunk in LDS (a byte load from global memory costs the single walking thread a full
// round trip: ~0.3 us per episode, 265 us for a C3 rollout's ~800 episodes), lane 0 walks it.
__global__ __launch_bounds__(64) void curriculum_update_kernel(CurriculumState* st, const uint8_t* __restrict__ msgs, int world, int cap) {
    constexpr int CH = 4096;
    __shared__ uint8_t chunk[CH];
    __shared__ CurriculumState sh;
    const int lane = threadIdx.x;
    if (lane == 0) sh = *st;
    __syncthreads();
    const size_t len = (size_t)4 + cap + 1;
    for (int r = 0; r < world; ++r) {
        const uint8_t* m = msgs + (size_t)r * len;
        long long cnt = (long long)m[0] | ((long long)m[1] << 8) | ((long long)m[2] << 16) | ((long long)m[3] << 24);
        if (cnt > cap) {       // (a rank ended more episodes than its message holds: flagged, the rest dropped)
            if (lane == 0) sh.overflow = 1.0;
            cnt = cap;
        }
        for (long long base = 0; base < cnt; base += CH) {
            const int nb = (int)((cnt - base < CH) ? cnt - base : CH);
            __syncthreads();
            for (int i = lane; i < nb; i += 64) chunk[i] = m[4 + base + i];
            __syncthreads();
            // Between two window closes an episode only adds to four counters, so the 64 lanes count the successes of
            // the events up to the next close together (ballot + popcount) and every lane then carries the same state
            // through the close's f64 arithmetic: one serial step per 120 episodes instead of one per episode.
            CurriculumState s = sh;
            for (int k = 0; k < nb;) {
                const int need = CUR_WINDOW - s.hist_len;                                   // >= 1: the window is cleared when full
                const int n = (nb - k < need) ? nb - k : need;
                int succ = 0;
                for (int j = 0; j < n; j += 64) {
                    const int i = j + lane;
                    succ += __popcll(__ballot(i < n && chunk[k + i] != 0));
                }
                k += n;
                const double env_radius = s.radius;                                         // :132
                s.episodes += n; s.successes += succ;
                s.hist_len += n; s.win_succ += succ;                                        // :135-137
                const bool full = s.hist_len >= CUR_WINDOW;
                double rate = 0.0;
                if (full) {
                    rate = (double)s.win_succ / (double)s.hist_len;                         // np.mean of the window
                    s.bonus = s.bonus * pow(CUR_DECAY_FACTOR, 1.0 + rate);                  // :140-142 (an np.float64 from here on)
                    s.bonus_is_f64 = 1.0;
                }
                s.bonus = fmax(s.bonus, 0.1);                                               // :144 (idempotent: once per group is once per episode)
                if (full) {
                    if (rate > CUR_SUCCESS_THRESHOLD)                                       // :148-152
                        s.radius = fmax(CUR_MIN_RADIUS, s.radius * pow(CUR_RADIUS_DECAY, 2.0 + 3.0 * (rate - CUR_SUCCESS_THRESHOLD)));
                    else if (rate < 0.25)                                                   // :153-157
                        s.radius = fmin(CUR_INITIAL_RADIUS, s.radius * 1.1);
                    const double d = s.radius - env_radius;
                    if (fabs(d) > 5.0) s.radius = env_radius + 5.0 * (d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0));      // :160-161
                    s.hist_len = 0; s.win_succ = 0;                                         // :164
                }
            }
            __syncthreads();
            if (lane == 0) sh = s;
        }
    }
    __syncthreads();
    if (lane == 0) *st = sh;
}

extern "C" {

int uav_gae(uav_ctx* ctx, const float* rew, const float* val, const float* done,
            const float* last_val, int n_env, int horizon, float gamma, float lam, int mode,
            float* adv, uav_stream stream) {
    UAV_REQUIRE(ctx && rew && val && done && adv, "uav_gae: NULL argument");
    UAV_REQUIRE(n_env > 0 && horizon > 0, "uav_gae: n_env=%d horizon=%d", n_env, horizon);
    UAV_REQUIRE(mode == UAV_GAE_REFERENCE_EXACT || mode == UAV_GAE_STANDARD, "uav_gae: mode %d", mode);
    const float gl = (float)((double)gamma * (double)lam);
    // the reference multiplies the python floats first (GAMMA * LAMBDA, f64) then rounds to f32
    const int blocks = (n_env + 3) / 4;
    hipLaunchKernelGGL(gae_scan_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), rew, val, done,
                       last_val, n_env, horizon, gamma, gl, mode, adv);
    UAV_LAUNCH_CHECK();
    return 0;
}

int uav_adv_stats(uav_ctx* ctx, const float* adv, int64_t n, double* stats3, uav_stream stream) {
    UAV_REQUIRE(ctx && adv && stats3 && n > 0, "uav_adv_stats: bad argument");
    int nb = (int)((n + 1023) / 1024);
    if (nb > STATS_BLOCKS) nb = STATS_BLOCKS;
    double* partial = (double*)ctx->ws;
    hipLaunchKernelGGL(adv_stats_partial, dim3(nb), dim3(256), 0, as_stream(stream), adv, n, partial);
    hipLaunchKernelGGL(adv_stats_final, dim3(1), dim3(256), 0, as_stream(stream), partial, nb, n, stats3);
    UAV_LAUNCH_CHECK();
    return 0;
}

int uav_adv_normalise(uav_ctx* ctx, const float* adv, const float* val, int64_t n,
                      const double* stats3, float* adv_out, float* ret_out, uav_stream stream) {
    UAV_REQUIRE(ctx && adv && val && stats3 && adv_out && ret_out && n > 0, "uav_adv_normalise: bad argument");
    int nb = (int)((n + 1023) / 1024);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(adv_normalise_kernel, dim3(nb), dim3(256), 0, as_stream(stream), adv, val, n, stats3,
                       adv_out, ret_out);
    UAV_LAUNCH_CHECK();
    return 0;
}


int uav_episode_rows(uav_ctx* ctx, const float* rew, const float* info, const uint8_t* flags, int n_env, int T, int env_offset,
                     double* carry, double* rows, int cap, int32_t* count, uav_stream stream) {
    UAV_REQUIRE(ctx && rew && info && flags && carry && rows && count && n_env > 0 && T > 0 && cap > 0, "uav_episode_rows: bad argument");
    hipLaunchKernelGGL(episode_rows_kernel, dim3((n_env + 63) / 64), dim3(64), 0, as_stream(stream), rew, info, flags, n_env, T, env_offset,
                       carry, rows, cap, count);
    UAV_LAUNCH_CHECK();
    return 0;
}

size_t uav_curriculum_state_bytes(void) { return sizeof(CurriculumState); }

int uav_curriculum_init(uav_ctx* ctx, void* state, double radius, double bonus, int bonus_is_f64, uav_stream stream) {
    UAV_REQUIRE(ctx && state && radius > 0.0, "uav_curriculum_init: bad argument");
    hipLaunchKernelGGL(curriculum_init_kernel, dim3(1), dim3(64), 0, as_stream(stream), (CurriculumState*)state, radius, bonus, bonus_is_f64);
    UAV_LAUNCH_CHECK();
    return 0;
}

int uav_curriculum_update(uav_ctx* ctx, void* state, const uint8_t* msgs, int world, int cap, uav_stream stream) {
    UAV_REQUIRE(ctx && state && msgs && world > 0 && cap > 0, "uav_curriculum_update: bad argument");
    hipLaunchKernelGGL(curriculum_update_kernel, dim3(1), dim3(64), 0, as_stream(stream), (CurriculumState*)state, msgs, world, cap);
    UAV_LAUNCH_CHECK();
    return 0;
}

int uav_pack_success_bits(uav_ctx* ctx, const uint8_t* flags, int64_t n, int cap, uint8_t* msg, uav_stream stream) {
    UAV_REQUIRE(ctx && flags && msg && n > 0 && cap > 0, "uav_pack_success_bits: bad argument");
    hipLaunchKernelGGL(pack_success_kernel, dim3(1), dim3(1024), 0, as_stream(stream), flags, n, cap, msg);
    UAV_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
