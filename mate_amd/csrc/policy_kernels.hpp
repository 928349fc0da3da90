// policy_kernels.hpp -- on-device rule-based policies: the reference's GreedyCameraAgent and
// GreedyTargetAgent (mate/agents/greedy.py:13-227, :229-365) for every environment of the batch, so that
// scenario C3 (Greedy vs Greedy) never leaves the GPU.
//
// One wave per environment (4 per workgroup), lanes = agents or (sender, recipient) pairs.  What an agent
// may know is exactly what its observation row holds: its own private state, the opponents' public states
// gated by the view masks of the previous step, and its teammates' messages (exchanged here through LDS:
// the reference's mailbox is a host-side dict, mate/environment.py:836-892).
// Random draws (tape or Philox): camera "re-sample when blind" Bernoulli(0.1) + action sample
// (greedy.py:93-96), message delays randint(6, 50) (greedy.py:184-185); target goal choice (greedy.py:298),
// noise Bernoulli + sample (greedy.py:315-319), initial noise (greedy.py:277).
#pragma once
#include "engine_kernels.hpp"

namespace mate {

// Per step every lane makes ONE unconditional Philox call (no divergent re-draws), stream S_POL_STEP, sub = lane:
//   word 0, low half  -> message delay of the (sender, recipient) pair `lane`  (randint(6, 50), greedy.py:184)
//   word 0, high half -> the target's warehouse choice                         (np_random.choice, greedy.py:298)
//   word 1            -> the agent's Bernoulli uniform                          (greedy.py:93, 315)
//   words 2, 3        -> the two uniforms of the agent's action / noise sample (greedy.py:95, 319)
// (camera c is lane c, target t is lane 32 + t).  32-bit uniforms: the thresholds are 0.05 .. 0.75 and the
// samples feed a clipped action.  S_POL_TGT_RESET is keyed by the episode counter.
enum PolicyStream : uint32_t { S_POL_STEP = 16, S_POL_TGT_RESET = 19 };

struct PolicyTape {          // all optional (NULL = Philox); device pointers
    const double *cam_binom_u;        // [N][Nc]
    const double *cam_sample_u;       // [N][Nc][2]
    const int32_t *cam_delay;         // [N][Nc][Nc]  value drawn for the message sender->recipient (-1 = none)
    const double *tgt_choice_u;       // [N][Nt]
    const double *tgt_binom_u;        // [N][Nt]
    const double *tgt_sample_u;       // [N][Nt][2]
    const double *tgt_reset_sample_u; // [N][Nt][2]
};

struct PolicyPtrs {
    double *pol;               // [N][PW] agent memory record
    const uint32_t *masks;     // [N][MW] view masks of the previous step / reset
    double *cam_act, *tgt_act; // [N][Nc][2], [N][Nt][2] f64 outputs
    PolicyTape tape;
    int32_t TW, PW;            // 8-byte words of the target agents' section / of the whole record (pol_target_words, pol_record_words)
    int32_t lds_bytes;         // per wave
    int32_t memory_period;     // 25 (greedy.py:21)
    double noise_scale;        // 0.5 (greedy.py:236)
    const double *zoom_tab;    // the zoom solve tabulated over K (zoom_table_value; built by mate_engine_policy_enable): zoom_n values at
    double zoom_inv_h;         // K = i / zoom_inv_h
    int32_t zoom_n;
    int32_t caller_team;       // fused rollout: -1 both teams are the agents; 0 / 1 the camera / target team repeats the caller's
                               // joint action (Ptrs::cam_act / tgt_act) for every step of the launch (FrameSkip over MultiCamera / MultiTarget)
};

// The agents' memory record of one environment, one section per TEAM (round 5: a team the caller plays does not act -- MultiCamera /
// MultiTarget -- and step_greedy_kernel then neither loads nor stores, nor holds in LDS, that team's section):
//   target section  f64: tgt_prev_xy[t][2] | tgt_prev_noise[t][2]      i32: tgt_goal[t] | tgt_nonempty[t] | tgt_need[t] | episode
//   camera section  f64: mem[c][t][2] | prev_action[c][2]              i32: t2f[c][t] | delay[s][c] | neighbor[c][s] | has_state[c] | episode
// `episode`: the episode whose first call has run the team's agent.reset(observation) (greedy.py:43-61, 262-283) -- per team, so
// that a team that starts to act in the middle of an episode (the caller alternating step_greedy and step_versus_greedy) resets
// at ITS first call.
__host__ __device__ constexpr int pol_target_ints(int Nt) { return 3 * Nt + 1; }
__host__ __device__ constexpr int pol_target_words(int Nt) { return 4 * Nt + (pol_target_ints(Nt) + 1) / 2; }
__host__ __device__ constexpr int pol_camera_ints(int Nc, int Nt) { return Nc * Nt + 2 * Nc * Nc + Nc + 1; }
__host__ __device__ constexpr int pol_camera_words(int Nc, int Nt) { return 2 * Nc * Nt + 2 * Nc + (pol_camera_ints(Nc, Nt) + 1) / 2; }
__host__ __device__ constexpr int pol_record_words(int Nc, int Nt) { return pol_target_words(Nt) + pol_camera_words(Nc, Nt); }

template <typename ObsT>
struct PolCtx {
    const Params &p;
    const PolicyPtrs &q;
    double *f;                 // the record: [target section | camera section], then the camera agents' per-step scratch
    double *cf;                // camera section (f + TW)
    int32_t *ti, *ci, *si;     // the sections' integers; the scratch's integers (f + PW)
    // `base`: [target section | camera section | scratch] -- or, `cameras` false (step_greedy_kernel with the cameras played by the
    // caller), the target section alone: no camera accessor is touched then
    __device__ PolCtx(const Params &p_, const PolicyPtrs &q_, unsigned char *base, bool cameras = true) : p(p_), q(q_) {
        f = reinterpret_cast<double *>(base);
        ti = reinterpret_cast<int32_t *>(f + 4 * p.Nt);
        cf = f + q.TW;
        ci = reinterpret_cast<int32_t *>(cf + 2 * p.Nc * p.Nt + 2 * p.Nc);
        si = reinterpret_cast<int32_t *>(f + q.PW);
        (void)cameras;
    }
    __device__ double &tgt_prev(int t, int k) { return f[t * 2 + k]; }
    __device__ double &tgt_noise(int t, int k) { return f[p.Nt * 2 + t * 2 + k]; }
    __device__ int32_t &tgt_goal(int t) { return ti[t]; }
    __device__ int32_t &tgt_nonempty(int t) { return ti[p.Nt + t]; }
    __device__ int32_t &tgt_need(int t) { return ti[2 * p.Nt + t]; }
    __device__ int32_t &tgt_episode() { return ti[3 * p.Nt]; }
    __device__ double &mem(int c, int t, int k) { return cf[(c * p.Nt + t) * 2 + k]; }
    __device__ double &prev_action(int c, int k) { return cf[p.Nc * p.Nt * 2 + c * 2 + k]; }
    __device__ int32_t &t2f(int c, int t) { return ci[c * p.Nt + t]; }
    __device__ int32_t &delay(int s, int c) { return ci[p.Nc * p.Nt + s * p.Nc + c]; }
    __device__ int32_t &neighbor(int c, int s) { return ci[p.Nc * p.Nt + p.Nc * p.Nc + c * p.Nc + s]; }
    __device__ int32_t &has_state(int c) { return ci[p.Nc * p.Nt + 2 * p.Nc * p.Nc + c]; }
    __device__ int32_t &cam_episode() { return ci[p.Nc * p.Nt + 2 * p.Nc * p.Nc + p.Nc]; }
    // scratch past the record (policy_staging_words): message staging, per-camera masks, per-pair distances
    __device__ int32_t &send_bits(int s, int c) { return si[s * p.Nc + c]; }            // bit 31: 'state', bits 0..Nt-1: targets
    __device__ int32_t &near_bits(int c) { return si[p.Nc * p.Nc + c]; }                // targets within 110 % of the range of camera c
    __device__ double &pair_dist(int c, int t) { return f[q.PW + (p.Nc * p.Nc + p.Nc) / 2 + 2 + c * p.Nt + t]; }
};
// 8-byte words of that scratch: Nc*Nc + Nc ints (possibly starting in the upper half of the record's last word), Nc*Nt doubles
__host__ __device__ constexpr int policy_staging_words(int Nc, int Nt) { return (Nc * Nc + Nc) / 2 + 2 + Nc * Nt; }

// sin of an angle in degrees, 0 <= deg <= 90: Taylor series to x^21 on the un-reduced argument
// (remainder (pi/2)^23 / 23! = 1.2e-18), a third of the instructions of sincos_deg.
__device__ __forceinline__ double sin_deg_0_90(double deg) {
    const double x = deg * kDeg2Rad;
    const double z = x * x;
    double ps = 1.95729410633912612308e-20;                       // 1/21!
    ps = FMA_SC(ps, z, -8.22063524662432971696e-18);              // -1/19!
    ps = FMA_SC(ps, z, 2.81145725434552075980e-15);               // 1/17!
    ps = FMA_SC(ps, z, -7.64716373181981647590e-13);              // -1/15!
    ps = FMA_SC(ps, z, 1.60590438368216145994e-10);               // 1/13!
    ps = FMA_SC(ps, z, -2.50521083854417187751e-08);              // -1/11!
    ps = FMA_SC(ps, z, 2.75573192239858906526e-06);               // 1/9!
    ps = FMA_SC(ps, z, -1.98412698412698412698e-04);              // -1/7!
    ps = FMA_SC(ps, z, 8.33333333333333333333e-03);               // 1/5!
    ps = FMA_SC(ps, z, -1.66666666666666666667e-01);              // -1/3!
    return fma(x * z, ps, x);
}

// The zoom solve of GreedyCameraAgent.act: b <- area_product / (distance (1 + sin(b/2)))^2, 20 times from 180 (greedy.py:139-145),
// with Kc = area_product / distance^2.  The map is a contraction (|f'| <= 0.65), so last-place differences do not grow: the
// quotient is taken as Kc * (1/(1+sin))^2 with a Newton-refined reciprocal.  (An Estrin-form polynomial -- half the dependent
// depth -- was no faster: the solving wave shares its SIMD with three others and is issue-bound, not latency-bound.)
__device__ __forceinline__ double zoom_fixed_point(double Kc) {
    double b = 180.0;
    for (int it = 0; it < MATE_ZOOM_ITERATIONS; ++it) {
        const double half = b * 0.5;
        const double y = 1.0 + sin_deg_0_90(half < 90.0 ? half : 90.0);
        double r = __builtin_amdgcn_rcp(y);
        r = fma(r, fma(-y, r, 1.0), r);
        r = fma(r, fma(-y, r, 1.0), r);
        b = Kc * (r * r);
    }
    return b;
}

// The same quantity from a table.  The twenty iterations are a pure function of ONE scalar, K in (0, 720) (the branch that
// solves has distance > sqrt(area_product / 180) / 2, i.e. K < 720), smooth in K (the clamp at 90 degrees is never active
// there: b_1 = K / 4 <= 180): mate_engine_policy_enable tabulates it on the host with the iteration above at K = i / 40 and
// the agents read four neighbours and interpolate cubically (Lagrange).  Measured against the iteration over 2e6 random K:
// |error| <= 1.5e-13 degrees, the size of the iteration's own rounding -- and 460 dependent f64 instructions, two workgroup
// barriers and the lockstep of the four waves of a workgroup (17 % of a fused Greedy step) become one 32-byte load.
__device__ __forceinline__ double zoom_lookup(const PolicyPtrs &q, double K) {
    const double x = K * q.zoom_inv_h;
    int i = (int)x;
    if (!(x >= 1.0) || i > q.zoom_n - 3) return zoom_fixed_point(K);     // outside the table (never in the solving branch): iterate
    const double t = x - (double)i;
    const double *f = q.zoom_tab + (i - 1);
    const double fm1 = f[0], f0 = f[1], f1 = f[2], f2 = f[3];
    const double tp1 = t + 1.0, tm1 = t - 1.0, tm2 = t - 2.0;
    const double wm1 = t * tm1 * tm2 * (-1.0 / 6.0), w0 = tp1 * tm1 * tm2 * 0.5, w1 = tp1 * t * tm2 * (-0.5), w2 = tp1 * t * tm1 * (1.0 / 6.0);
    return fma(wm1, fm1, fma(w0, f0, fma(w1, f1, w2 * f2)));
}

// One step of both teams' agents of ONE environment on LDS-resident data: `a` the agents' memory record, `st` / `dy` /
// `di` the static and dynamic records, `mk` the packed view masks of the previous step.  One wave, one environment.  Joint
// actions go to q.cam_act / q.tgt_act when `active` and `publish`, and to lds_cam_act / lds_tgt_act when those are given (the
// fused rollout steps from them and publishes the last executed step's once per launch).
// `L` lanes per environment (Ctx): camera c is lane c of the environment's group, target t lane L / 2 + t (32 + t in a whole wave);
// `shift` the group's first hardware lane.  The draws are keyed by the lane a role has in a WHOLE wave (camera c: c, target t: 32 + t,
// pair k: k), so every L draws the same numbers.
template <typename ObsT, int L = 64>
__device__ __forceinline__ void greedy_policy_body(const Params &p, const PolicyPtrs &q, PolCtx<ObsT> &a, const double *st, const double *dy,
                                                   const int32_t *di, const uint32_t *mk,
                                                   int wave, int lane, int64_t env, bool active, double *lds_cam_act, double *lds_tgt_act,
                                                   long long *acc = nullptr, long long *t_prev = nullptr, bool publish = true, int shift = 0) {
    constexpr int TB = L / 2;                       // the target agents' first lane
    auto group_ballot = [&](bool x) -> unsigned long long {
        if constexpr (L == 64) return __ballot(x);
        else return (__ballot(x) >> shift) & ((1ull << L) - 1ull);
    };
    // Which teams' agents act (wave-uniform, `cams` / `tgts` below): both under step_greedy / rollout_greedy; under MultiCamera /
    // MultiTarget (q.caller_team = the learner's team, single_team.py:245-264) ONLY THE OPPONENTS -- the reference's wrapper holds no
    // agents for the learner's team, and its joint action is the caller's.  A team that does not act keeps its memory as it is and
    // runs its agent.reset at its own first call of an episode (`fresh_c` / `fresh_t`), so what the acting team does never depends
    // on it: every draw is keyed by (environment, tick, stream, lane).  The camera agents are three quarters of this function's
    // chain (11.5 k of a 32 k-cycle learner-versus-greedy step at 4096 x MATE-4v8-9 before; the camera learner's step skips them).
#ifdef MATE_PHASE_CLOCKS
#define POL_STAMP(i) do { if (acc) { const long long t_now = (long long)__builtin_amdgcn_s_memtime(); acc[i] += t_now - *t_prev; *t_prev = t_now; } } while (0)
#elif defined(MATE_ISA_MARKS)
#define POL_STAMP(i) asm volatile("; ==== MATE_GREEDY_PHASE " #i)
#else
#define POL_STAMP(i) do { } while (0)
#endif
    const int Nc = p.Nc, Nt = p.Nt;
    auto cam_x = [&](int c) { return st[c]; };
    auto cam_y = [&](int c) { return st[Nc + c]; };
    auto tx = [&](int t) { return dy[2 * Nc + t]; };
    auto ty = [&](int t) { return dy[2 * Nc + Nt + t]; };
    auto sees = [&](int c, int t) { const int b = c * Nt + t; return (mk[b >> 5] >> (b & 31)) & 1u; };
    const int32_t *env_i = di + Nt * TI_STRIDE;
    const uint32_t tick = (uint32_t)env_i[EI_TICK];
    const uint32_t env_global = p.first_env + (uint32_t)env;
    const bool cams = q.caller_team != 0, tgts = q.caller_team != 1;      // (which teams act: see below)
    // a team's first call of a new episode: agent.reset(observation)
    const bool fresh_c = cams && a.cam_episode() != env_i[EI_EPISODE], fresh_t = tgts && a.tgt_episode() != env_i[EI_EPISODE];
    // this step's draws (see PolicyStream); skipped when every draw comes from the tape
    double u_bern = 0.0, u_s0 = 0.0, u_s1 = 0.0;
    uint32_t w_delay = 0, w_choice = 0;
    if (!(q.tape.cam_binom_u && q.tape.cam_sample_u && q.tape.cam_delay && q.tape.tgt_choice_u && q.tape.tgt_binom_u && q.tape.tgt_sample_u)) {
        const U4 r = philox(p.seed_lo, p.seed_hi, env_global, tick, S_POL_STEP, (uint32_t)(lane < TB ? lane : 32 + (lane - TB)));
        w_delay = r.x & 0xffffu; w_choice = r.x >> 16;
        u_bern = (double)r.y * 2.3283064365386963e-10;
        u_s0 = (double)r.z * 2.3283064365386963e-10; u_s1 = (double)r.w * 2.3283064365386963e-10;
    }
    const uint64_t capword = reinterpret_cast<const uint64_t *>(st)[3 * Nc + 3 * p.No];

    // ------------------------------------------------------------------ reset + observe
    // The camera agents' bookkeeping runs on PAIR lanes (camera c, target t) -- Nc*Nt of them instead of Nc lanes looping over
    // the targets -- and on (sender, recipient) lanes for the messages; per-camera lanes only scan what those left in LDS.
    auto seen_mask = [&](int c) -> uint32_t {              // row c of camera_target_view_mask: bits [c Nt, c Nt + Nt) of the packed words
        const int b = c * Nt;
        const uint64_t w = (uint64_t)mk[b >> 5] | ((uint64_t)mk[(b >> 5) + 1] << 32);
        return (uint32_t)(w >> (b & 31)) & ((1u << Nt) - 1u);
    };
    const double threshold = 1.1 * p.rmax;                  // filterout_beyond_range / the tracking reach: 110 % of the range
    if (cams && lane < Nc) a.near_bits(lane) = 0;
    if (fresh_c) {                                          // GreedyCameraAgent.reset (greedy.py:43-61)
        if (lane < Nc) { a.prev_action(lane, 0) = 0.0; a.prev_action(lane, 1) = 0.0; a.has_state(lane) = 1; }
        for (int k = lane; k < Nc * Nc; k += L) { a.ci[Nc * Nt + k] = 0; a.ci[Nc * Nt + Nc * Nc + k] = 0; }   // delay, neighbor
    }
    wave_sync();
    if (cams)
    for (int k = lane; k < Nc * Nt; k += L) {              // process_messages of the observation (greedy.py:100-113)
        const int c = (int)(((float)k + 0.5f) * p.inv_Nt), t = k - c * Nt;
        const bool s = sees(c, t);
        int left = 0;
        if (fresh_c) { a.mem(c, t, 0) = 0.0; a.mem(c, t, 1) = 0.0; }      // hidden rows of the first observation are zeros
        else { left = a.t2f(c, t) - 1; if (left < 0) left = 0; }
        const double x = tx(t), y = ty(t);
        if (s) { left = q.memory_period; a.mem(c, t, 0) = x; a.mem(c, t, 1) = y; }
        a.t2f(c, t) = left;
        if (norm2(x - cam_x(c), y - cam_y(c)) < threshold) atomicOr(&a.near_bits(c), 1 << t);
    }
    const int tl = lane - TB;
    if (tgts && tl >= 0 && tl < Nt) {                       // GreedyTargetAgent.reset / process_messages (greedy.py:262-283,326-332)
        const int t = tl;
        const int gw = di[t * TI_STRIDE + TI_GW];
        const int state_goal = (gw & 0xff) - 1;
        const double step_size = ((capword >> t) & 1ull) ? p.tgt_step * 0.5 : p.tgt_step;
        if (fresh_t) {
            a.tgt_prev(t, 0) = tx(t); a.tgt_prev(t, 1) = ty(t);
            double u0, u1;
            if (q.tape.tgt_reset_sample_u) { u0 = q.tape.tgt_reset_sample_u[(env * Nt + t) * 2]; u1 = q.tape.tgt_reset_sample_u[(env * Nt + t) * 2 + 1]; }
            else { const U4 r = philox(p.seed_lo, p.seed_hi, env_global, (uint32_t)env_i[EI_EPISODE], S_POL_TGT_RESET, (uint32_t)t); u0 = u53(r.x, r.y); u1 = u53(r.z, r.w); }
            a.tgt_noise(t, 0) = 0.5 * (-step_size + (2.0 * step_size) * u0);               // 0.5 * action_space.sample()
            a.tgt_noise(t, 1) = 0.5 * (-step_size + (2.0 * step_size) * u1);
            a.tgt_goal(t) = state_goal;
            a.tgt_nonempty(t) = 0xf;
            a.tgt_need(t) = 0;
        }
        const int seen_empty = (gw >> 16) & 0xf;
        if (seen_empty & a.tgt_nonempty(t)) { a.tgt_nonempty(t) &= ~seen_empty; a.tgt_need(t) = 1; }
    }
    if (lane == 0 && fresh_c) a.cam_episode() = env_i[EI_EPISODE];
    if (lane == 0 && fresh_t) a.tgt_episode() = env_i[EI_EPISODE];
    wave_sync();
    POL_STAMP(8);

    // ------------------------------------------------------------------ communicate
    // cameras: send_responses (greedy.py:158-190), one lane per (sender, recipient) -- one round of pairs up to 8 cameras, up to
    // four for the 16 the engine takes (the pair's message delay: the lane's own Philox word in the first round, one more block
    // keyed by the pair's index beyond)
    const bool one_round = Nc * Nc <= L;
    // Serial loops of the agents turned into loops over BALLOT bits (round 5; the fused Greedy rollouts are VALU-bound and a latency-bound
    // per-step launch waits for every LDS round trip): a pair lane ORs the messages of the senders that DID send to its camera (bits
    // s Nc + c of `sent`: a message goes out every ~28 steps per pair) instead of reading all Nc staging words; a camera lane scans the
    // targets that ARE candidates (bits c Nt + t of `candidates`) instead of all Nt distances; the targets AND the warehouse sets of the
    // targets that DO broadcast (`needers`: a target that has just seen an empty warehouse) instead of testing all Nt.  Same order, same values.
    unsigned long long sent = ~0ull, candidates = ~0ull;
    if (cams) {
    auto send_pair = [&](int k) -> int {
        const int s = (int)(((float)k + 0.5f) * p.inv_Nc), c = k - s * Nc;
        int bits = 0;
        int d = a.delay(s, c) - 1;
        if (d < 0) d = 0;
        // message2send is non-empty iff it still holds 'state' or a target was seen this step
        const uint32_t seen_now = seen_mask(s);
        const bool has_state = a.has_state(s) != 0;
        if ((has_state || seen_now) && s != c && d == 0) {
            // filterout_beyond_range: of the targets seen this step, those within 110 % of the teammate's range
            const uint32_t list = a.neighbor(s, c) ? (seen_now & (uint32_t)a.near_bits(c)) : 0u;
            bits = (int)list | (has_state ? (int)0x80000000u : 0);
            if (bits) {
                int v;
                if (q.tape.cam_delay) v = q.tape.cam_delay[(env * Nc + s) * Nc + c];
                else { const int lo = q.memory_period / 4, hi = 2 * q.memory_period;        // randint(6, 50)
                       uint32_t w = w_delay;
                       if (k >= (L == 64 ? 64 : TB)) w = philox(p.seed_lo, p.seed_hi, env_global, tick, S_POL_STEP, (uint32_t)k).x & 0xffffu;      // (pair k's word is lane k's of a whole wave; this lane drew it itself only below L / 2)
                       v = lo + (int)((w * (uint32_t)(hi - lo)) >> 16); }
                d = v;
            }
        }
        a.delay(s, c) = d;
        a.send_bits(s, c) = bits;
        // receive_responses: the recipient learns its neighbour (greedy.py:192-226).  One round of pairs: every lane has read
        // neighbor(s, c) above before any lane writes here (one wave, one instruction stream, LDS operations in order)
        if (one_round && bits < 0) a.neighbor(c, s) = 1;
        return bits;
    };
    int my_bits = 0;
    if (lane < Nc * Nc) my_bits = send_pair(lane);
    if (one_round) sent = group_ballot(my_bits != 0);
    if (!one_round)
        for (int k = lane + L; k < Nc * Nc; k += L) send_pair(k);
    if (!one_round) {                                       // several rounds: behind EVERY round's reads
        wave_sync();
        for (int k = lane; k < Nc * Nc; k += L) {
            const int s = (int)(((float)k + 0.5f) * p.inv_Nc), c = k - s * Nc;
            if (a.send_bits(s, c) < 0) a.neighbor(c, s) = 1;
        }
    }
    wave_sync();
    // ... and the positions of the targets it was told about; then the tracking candidates: distance camera -> remembered
    // position, +inf when forgotten or out of reach (greedy.py:115-127)
    unsigned long long col_mask = 0ull;                                  // bit s Nc for every sender s (one round of pairs)
    if (one_round) for (int s = 0; s < Nc; ++s) col_mask |= 1ull << (s * Nc);
    double dn_lane = INFINITY;
    for (int k = lane; k < Nc * Nt; k += L) {
        const int c = (int)(((float)k + 0.5f) * p.inv_Nt), t = k - c * Nt;
        int told = 0;
        if (one_round) {
            for (unsigned long long m = (sent >> c) & col_mask; m != 0ull; m &= m - 1ull) {
                const int sb = __ffsll((long long)m) - 1;                // = s Nc
                told |= a.si[sb + c];                                    // send_bits(s, c)
            }
        } else
        for (int s = 0; s < Nc; ++s) told |= a.send_bits(s, c);
        double mx = a.mem(c, t, 0), my = a.mem(c, t, 1);
        int left = a.t2f(c, t);
        if ((told >> t) & 1) { mx = tx(t); my = ty(t); left = q.memory_period; a.mem(c, t, 0) = mx; a.mem(c, t, 1) = my; a.t2f(c, t) = left; }
        double dn = INFINITY;
        if (left > 0) {
            const double dnorm = norm2(mx - cam_x(c), my - cam_y(c));
            if (dnorm < threshold) dn = dnorm;
        }
        a.pair_dist(c, t) = dn;
        if (k == lane) dn_lane = dn;
    }
    if (Nc * Nt <= L) candidates = group_ballot(dn_lane < INFINITY);
    if (lane < Nc && (a.has_state(lane) || seen_mask(lane))) a.has_state(lane) = 0;   // message2send.clear()
    }
    // targets: broadcast non-empty warehouse sets (greedy.py:334-358).  Every lane's reads precede every lane's writes: one wave,
    // one instruction stream, LDS operations in order.
    const bool broadcasts = tgts && tl >= 0 && tl < Nt && a.tgt_need(tl) != 0;
    const uint32_t needers = (uint32_t)(group_ballot(broadcasts) >> TB);      // (target t on lane L / 2 + t; nobody: the sets stay as they are)
    if (needers != 0u && tgts && tl >= 0 && tl < Nt) {
        int set = a.tgt_nonempty(tl);
        for (uint32_t m = needers; m != 0u; m &= m - 1u) set &= a.tgt_nonempty(__ffs((int)m) - 1);
        a.tgt_nonempty(tl) = set;
        a.tgt_need(tl) = 0;
    }
    wave_sync();

    POL_STAMP(11);
    // ------------------------------------------------------------------ act
    // GreedyCameraAgent.act (greedy.py:69-156), part 1: pick the target, decide which viewing-angle rule applies
    int best = -1;
    double theta = 0.0, orientation = 0.0, min_va = 0.0, best_orientation = 0.0, best_va = 0.0, K = 0.0;
    bool solve = false;
    if (cams && lane < Nc) {
        const int c = lane;
        const double phi = dy[c];
        theta = dy[Nc + c];
        // The agent reconstructs its sight range and orientation from the (sr cos phi, sr sin phi) pair of its
        // observation row (agents/utils.py:206-255: norm and arctan2 of that pair); that round trip returns
        // sr and phi to within a few ulp, so the state values are used directly.
        const double sight = sqrt_pos(div_nz(p.area, theta));
        orientation = phi;
        const double q2 = div_nz(sight, p.rmax);
        min_va = theta * (q2 * q2);
        double best_d = INFINITY;
        if (Nc * Nt <= L) {                                         // ascending t over the candidates, first minimum wins, like the reference's loop
            for (uint32_t m = (uint32_t)(candidates >> (c * Nt)) & ((1u << Nt) - 1u); m != 0u; m &= m - 1u) {
                const int t = __ffs((int)m) - 1;
                const double dnorm = a.pair_dist(c, t);
                if (dnorm < best_d) { best = t; best_d = dnorm; }
            }
        } else
        for (int t = 0; t < Nt; ++t) {
            const double dnorm = a.pair_dist(c, t);
            if (dnorm < best_d) { best = t; best_d = dnorm; }
        }
        if (best >= 0) {
            const double rx = a.mem(c, best, 0) - cam_x(c), ry = a.mem(c, best, 1) - cam_y(c);
            best_orientation = atan2_deg(ry, rx);
            const double distance = best_d;
            if (distance * (1.0 + sin_deg_0_90(min_va / 2.0)) >= p.rmax) best_va = min_va;
            else {
                const double area_product = theta * (sight * sight);
                if (distance <= sqrt_pos(area_product / 180.0) / 2.0) best_va = 180.0;
                else { solve = true; K = div_nz(area_product, distance * distance); }
            }
        }
    }
    POL_STAMP(12);
    if (cams && lane < Nc && solve) best_va = clipd(zoom_lookup(q, K), min_va, 180.0);      // greedy.py:139-146, tabulated (zoom_lookup)
    POL_STAMP(9);
    if (cams && lane < Nc) {                                // part 2: the action
        const int c = lane;
        double a0, a1;
        if (best >= 0) {
            a0 = clip_uniform(normalize_angle(best_orientation - orientation), -p.rot, p.rot);
            a1 = clip_uniform(best_va - theta, -p.zoom, p.zoom);
        } else {
            double u;
            if (q.tape.cam_binom_u) u = q.tape.cam_binom_u[env * Nc + c];
            else u = u_bern;
            if (u > 1.0 - 0.1) {                            // np_random.binomial(1, 0.1)
                double u0, u1;
                if (q.tape.cam_sample_u) { u0 = q.tape.cam_sample_u[(env * Nc + c) * 2]; u1 = q.tape.cam_sample_u[(env * Nc + c) * 2 + 1]; }
                else { u0 = u_s0; u1 = u_s1; }
                a0 = -p.rot + (2.0 * p.rot) * u0; a1 = -p.zoom + (2.0 * p.zoom) * u1;
            } else { a0 = a.prev_action(c, 0); a1 = a.prev_action(c, 1); }
        }
        a.prev_action(c, 0) = a0; a.prev_action(c, 1) = a1;
        if (lds_cam_act) { lds_cam_act[2 * c] = a0; lds_cam_act[2 * c + 1] = a1; }
        if (active && publish) { q.cam_act[(env * Nc + c) * 2] = a0; q.cam_act[(env * Nc + c) * 2 + 1] = a1; }
    }
    if (tgts && tl >= 0 && tl < Nt) {                       // GreedyTargetAgent.act (greedy.py:285-324)
        const int t = tl;
        const int gw = di[t * TI_STRIDE + TI_GW];
        const int state_goal = (gw & 0xff) - 1;
        const double step_size = ((capword >> t) & 1ull) ? p.tgt_step * 0.5 : p.tgt_step;
        int goal = a.tgt_goal(t);
        if (state_goal >= 0) goal = state_goal;
        const int nonempty = a.tgt_nonempty(t);
        if (goal < 0 || (state_goal < 0 && !((nonempty >> goal) & 1))) {
            goal = -1;
            const int k = __popc(nonempty);
            if (k > 0) {
                double u;
                if (q.tape.tgt_choice_u) u = q.tape.tgt_choice_u[env * Nt + t];
                else u = (double)w_choice * 1.52587890625e-05;
                int j = (int)(u * (double)k);
                if (j >= k) j = k - 1;
                for (int w = 0, seen = 0; w < 4; ++w) if ((nonempty >> w) & 1) { if (seen == j) goal = w; ++seen; }
            }
        }
        a.tgt_goal(t) = goal;
        const double x = tx(t), y = ty(t);
        const double pax = x - a.tgt_prev(t, 0), pay = y - a.tgt_prev(t, 1);
        double ax = 0.0, ay = 0.0;
        if (goal >= 0) {
            const double wx = (goal == 0 || goal == 3) ? kWarehouseCenter : -kWarehouseCenter;
            const double wy = (goal < 2) ? kWarehouseCenter : -kWarehouseCenter;
            ax = wx - x; ay = wy - y;
        }
        const double len = norm2(ax, ay);
        if (len > step_size) { const double k2 = div_nz(step_size, len); ax *= k2; ay *= k2; }
        const double prob = norm2(pax, pay) > 0.2 * step_size ? 0.05 : 0.75;
        double u;
        if (q.tape.tgt_binom_u) u = q.tape.tgt_binom_u[env * Nt + t];
        else u = u_bern;
        double nx = a.tgt_noise(t, 0), ny = a.tgt_noise(t, 1);
        if ((prob <= 0.5) ? (u > 1.0 - prob) : (u <= prob)) {
            double u0, u1;
            if (q.tape.tgt_sample_u) { u0 = q.tape.tgt_sample_u[(env * Nt + t) * 2]; u1 = q.tape.tgt_sample_u[(env * Nt + t) * 2 + 1]; }
            else { u0 = u_s0; u1 = u_s1; }
            nx = q.noise_scale * (-step_size + (2.0 * step_size) * u0);
            ny = q.noise_scale * (-step_size + (2.0 * step_size) * u1);
        }
        const double outx = clipd(ax + nx, -step_size, step_size), outy = clipd(ay + ny, -step_size, step_size);
        a.tgt_prev(t, 0) = x; a.tgt_prev(t, 1) = y;
        a.tgt_noise(t, 0) = nx; a.tgt_noise(t, 1) = ny;
        if (lds_tgt_act) { lds_tgt_act[2 * t] = outx; lds_tgt_act[2 * t + 1] = outy; }
        if (active && publish) { q.tgt_act[(env * Nt + t) * 2] = outx; q.tgt_act[(env * Nt + t) * 2 + 1] = outy; }
    }
}

template <typename ObsT, typename Shape>
__global__ __launch_bounds__(256) void greedy_policy_kernel(const Params *__restrict__ pp, const Ptrs g, const PolicyPtrs q) {
    const Shape shape(pp);
    const Params &p = shape.get();
    extern __shared__ __align__(16) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // the four waves of a workgroup never synchronise: each owns one environment
    const int64_t env = (int64_t)blockIdx.x * 4 + wave;
    if (env >= g.N) return;
    unsigned char *base = smem + wave * q.lds_bytes;
    // LDS: [policy record + staging][static record][dynamic record][mask words]
    PolCtx<ObsT> a(p, q, base);
    double *st = reinterpret_cast<double *>(base + (size_t)(q.PW + policy_staging_words(p.Nc, p.Nt)) * 8);
    double *dy = st + p.SW;
    int32_t *di = reinterpret_cast<int32_t *>(dy + p.DF);
    uint32_t *mk = reinterpret_cast<uint32_t *>(dy + p.DW);
    {
        const double *src = q.pol + env * q.PW;
        for (int k = lane; k < q.PW; k += 64) a.f[k] = src[k];
        const double *s = g.stat + env * p.SW;
        for (int k = lane; k < p.SW; k += 64) st[k] = s[k];
        const double *d = g.dyn + env * p.DW;
        for (int k = lane; k < p.DW; k += 64) dy[k] = d[k];
        const uint32_t *m = q.masks + env * p.MW;
        for (int k = lane; k < p.MW; k += 64) mk[k] = m[k];
    }
    wave_sync();
    if (g.freeze_done && (di + p.Nt * TI_STRIDE)[EI_DONE] != 0) return;          // finished, waiting for the batched reset
    greedy_policy_body<ObsT>(p, q, a, st, dy, di, mk, wave, lane, env, true, nullptr, nullptr);
    wave_sync();
    double *dst = q.pol + env * q.PW;
    for (int k = lane; k < q.PW; k += 64) dst[k] = a.f[k];
}

// =============================================================================================
// K fused steps of Greedy cameras vs Greedy targets: the agents' step (greedy_policy_body) and the environment's step
// in one loop, with the environment records, the agents' memory and the view masks resident in LDS for the whole
// launch and the joint actions handed over in LDS.  Same results as K x (greedy_policy_kernel + step_kernel), bit for
// bit (tested).  No
// wave of a workgroup depends on another (the agents' zoom solve is a table lookup, not a shared iteration any more): a wave
// past the end of the batch, or whose episode has ended, skips agents and step alike.
__host__ __device__ constexpr int policy_slice_bytes(int PW, int Nc, int Nt) { return shape_round_up((PW + policy_staging_words(Nc, Nt) + 2 * (Nc + Nt)) * 8, 16); }

// The caller's team of a fused rollout: its joint action, decoded as step() would (f32 / f64 pairs, or grid indices:
// DiscreteCamera.action / DiscreteTarget.action, discrete_action_spaces.py:71-73, 177-179), over the agents' in LDS.
template <typename ObsT, int L>
__device__ __forceinline__ void load_caller_actions(Ctx<ObsT, L> &c, int team, double *act_cam, double *act_tgt) {
    const Params &p = c.p;
    const Ptrs &g = c.g;
    const int k = c.lane;
    if (team == 0) {
        if (k < p.Nc) {
            double da, dz;
            if (g.act_discrete & 1) {
                int idx = reinterpret_cast<const int32_t *>(g.cam_act)[c.env * p.Nc + k];
                idx = idx < 0 ? 0 : (idx >= g.n_cam_grid ? g.n_cam_grid - 1 : idx);
                const double2 gxy = g.cam_grid[idx];
                da = p.rot * gxy.x; dz = p.zoom * gxy.y;
            } else if (g.act_f64 & 1) {
                const double *a = reinterpret_cast<const double *>(g.cam_act) + (c.env * p.Nc + k) * 2;
                da = a[0]; dz = a[1];
            } else {
                const float2 a = reinterpret_cast<const float2 *>(g.cam_act)[c.env * p.Nc + k];
                da = (double)a.x; dz = (double)a.y;
            }
            act_cam[2 * k] = da; act_cam[2 * k + 1] = dz;
        }
    } else if (k < p.Nt) {
        double ax, ay;
        if (g.act_discrete & 2) {
            int idx = reinterpret_cast<const int32_t *>(g.tgt_act)[c.env * p.Nt + k];
            idx = idx < 0 ? 0 : (idx >= g.n_tgt_grid ? g.n_tgt_grid - 1 : idx);
            const double2 gxy = g.tgt_grid[idx];
            const double high = ((c.capword() >> k) & 1ull) ? p.tgt_step * 0.5 : p.tgt_step;   // loaded targets halve their step
            ax = high * gxy.x; ay = high * gxy.y;
        } else if (g.act_f64 & 2) {
            const double *a = reinterpret_cast<const double *>(g.tgt_act) + (c.env * p.Nt + k) * 2;
            ax = a[0]; ay = a[1];
        } else {
            const float2 a = reinterpret_cast<const float2 *>(g.tgt_act)[c.env * p.Nt + k];
            ax = (double)a.x; ay = (double)a.y;
        }
        act_tgt[2 * k] = ax; act_tgt[2 * k + 1] = ay;
    }
}

// `q` as it lies in the kernel-argument segment (third argument: behind the 8-byte `pp` and `g`), read where it is used instead
// of from the ~40 SGPRs the compiler preloads it into at entry and keeps for the whole kernel (see kernarg_ptrs): the fused
// Greedy rollout spilled 77 scalar registers into vector lanes, and read them back ~100 times per step.
__device__ __forceinline__ const PolicyPtrs &kernarg_policy_ptrs(const PolicyPtrs &q) {
    static_assert(alignof(PolicyPtrs) == 8 && sizeof(Ptrs) % 8 == 0, "q follows pp and g in the kernel arguments");
#if defined(__HIP_DEVICE_COMPILE__)
    (void)q;
    return *(const PolicyPtrs *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + 8 + sizeof(Ptrs));
#else
    return q;
#endif
}

// 16-byte row chunks per lane when a group of L lanes packs the shape's f32 row blocks (camera block / target block)
template <typename Shape, int L>
constexpr int sub_held_chunks(bool camera) { return ((camera ? Shape::kChunksC : Shape::kChunksT) + L - 1) / L; }

// `E` ENVIRONMENTS PER WAVE (round 6; engine_kernels.hpp, Ctx): 1 = the mapping above; 2 / 4 = sub-wave groups of L = 64 / E lanes, one
// environment each -- the small scenarios, whose agents and visibility pairs fill a quarter of a wave: every vector instruction then
// advances E environments.  A workgroup holds 4 E environments (wave w, group s: environment (4 block + w) E + s); the groups of a wave
// share nothing but the instruction stream.  E > 1 runs the phase functions written for any L (no held roles, no carried collision
// screen); same results as E = 1, bit for bit (tests/test_gpu_subwave.py).
#ifndef MATE_SUB_HOLD
#define MATE_SUB_HOLD 1        // (experiments: 0 = the sub-wave groups fetch their row descriptors at every step)
#endif
#ifndef MATE_SUB_BLOCKS
#define MATE_SUB_BLOCKS 4      // (experiments: workgroups per CU the sub-wave kernels' register budget is set for)
#endif
template <typename ObsT, typename Shape, int E = 1>
__global__ __launch_bounds__(256, E == 1 ? Shape::kGreedyBlocks : MATE_SUB_BLOCKS) void rollout_greedy_kernel(const Params *__restrict__ pp, const Ptrs g, const PolicyPtrs q_arg) {
    constexpr int L = 64 / E;
    static_assert(E == 1 || E == 2 || E == 4 || E == 8, "environments per wave");
    const PolicyPtrs &q = kernarg_policy_ptrs(q_arg);
    const Shape shape(pp, true);
    const Params &p = shape.get();
    extern __shared__ __align__(16) unsigned char smem[];
    // tick and list parity: launch arguments, or the device-resident counter (one-step launches replayed from a HIP graph:
    // see step_kernel -- the host keeps dev_tick = dev_group = 0 in the record while it counts itself)
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.done_count) {
        const int32_t parity = (int32_t)((p.dev_group + (uint32_t)g.parity) & 1u);
        if (!g.pipelined) g.done_count[parity ^ 1] = 0;      // (pipelined restarts: the other list is being consumed right now; its reset clears it)
        g.ctrl[0] = parity;
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int hw_lane = threadIdx.x & 63, lane = hw_lane & (L - 1), shift = hw_lane & ~(L - 1);      // lane inside the environment's group; the group's first lane
    const int slot = E == 1 ? wave : wave * E + (hw_lane >> (E == 8 ? 3 : E == 4 ? 4 : 5));                       // the environment's slice of the workgroup's LDS
    const int64_t env_raw = (int64_t)blockIdx.x * (4 * E) + slot;
    const bool in_batch = env_raw < g.N;
    const int64_t env = in_batch ? env_raw : g.N - 1;
    bool untouched = false;          // pipelined restarts: not live at entry, see below (uniform per environment)
    const Ptrs &gk = kernarg_ptrs(g);
    const int pol_bytes = policy_slice_bytes(q.PW, p.Nc, p.Nt);
    unsigned char *pol_base = smem + 4 * E * p.lds_wave_bytes + slot * pol_bytes;
    PolCtx<ObsT> a(p, q, pol_base);
    double *act_cam = a.f + (q.PW + policy_staging_words(p.Nc, p.Nt)), *act_tgt = act_cam + 2 * p.Nc;
    {
        Ctx<ObsT, L> c(p, gk, smem + slot * p.lds_wave_bytes, lane, env, FLOW_GREEDY);
        c.shift = shift;
        // pipelined restarts (Ptrs::pipelined): an environment tagged for this launch's list parity goes live; one that is not
        // live at entry -- tagged for the other parity, or finished and in the hands of the reset running under this launch --
        // is left alone: no step and, at the end, no store (the reset may be rewriting its records right now).  Whether it is
        // live is decided from ONE 4-byte load of the record's `done` word, ahead of everything else: every value the concurrent
        // reset can leave there (finished, listed, tagged for the other parity) reads "not mine", so the verdict does not depend
        // on how far that reset has come, and nothing else of such an environment -- records, masks, agents' memory -- is read.
        int d_entry = 0;
        bool mine = false;
        const int32_t parity = (int32_t)((p.dev_group + (uint32_t)g.parity) & 1u);
        if (g.pipelined) {
            // (read while a reset on another stream may be writing it: an agent-scope atomic load, not a plain one the compiler may keep or split)
            const int32_t *done_word = reinterpret_cast<const int32_t *>(g.dyn + env * p.DW + p.DF) + p.Nt * TI_STRIDE + EI_DONE;
            d_entry = __hip_atomic_load(done_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if constexpr (E == 1) d_entry = __builtin_amdgcn_readfirstlane(d_entry);
            mine = (d_entry & kDoneTag) && ((d_entry >> 3) & 1) == parity;
            untouched = d_entry != 0 && !mine;      // (a wave past the end of the batch mirrors environment N - 1: the same rule)
        }
        if (!untouched) {
            load_records(c);
            wave_sync();
            const uint32_t *m = q.masks + env * p.MW;                 // the view the previous step / reset left
            for (int i = lane; i < p.MW; i += L) c.mask[i] = m[i];
            const double *src = q.pol + env * q.PW;
            for (int k = lane; k < q.PW; k += L) a.f[k] = src[k];
            build_entities(c);
        } else {
            // nothing of this environment is loaded: the launch prologue below (image statics, lane roles, collision seeds) still runs
            // over the wave's record slice, so the slice is zeros, not whatever the LDS held -- its results are discarded, but no
            // address may ever be formed from an unwritten field
            uint32_t *slice = reinterpret_cast<uint32_t *>(smem + slot * p.lds_wave_bytes);
            for (int i = lane; i < (p.lds_wave_bytes >> 2); i += L) slice[i] = 0u;
            wave_sync();
            if (lane == 0) c.ei(EI_DONE) = d_entry;                  // (all the step loop reads of it: "not live")
        }
        if (g.pipelined) {
            if (d_entry == 1 && in_batch && lane == 0 && g.done_count) {      // finished under auto_reset = 0 earlier, never listed: list it, and say so in the record itself
                const int slot = atomicAdd(g.done_count + parity, 1);
                g.done_list[(int64_t)parity * g.N + slot] = (int32_t)env;
                reinterpret_cast<int32_t *>(g.dyn + env * p.DW + p.DF)[p.Nt * TI_STRIDE + EI_DONE] = 3;
            }
            wave_sync();
            if (mine && lane == 0) c.ei(EI_DONE) = 0;
        }
        else if (in_batch) list_finished_at_entry(c);
        wave_sync();
    }
    constexpr bool IMAGE = E == 1 && Shape::kImage;   // row-image mode (engine_kernels.hpp: image_statics)
    // the row chunks a lane holds descriptors of: all of them where they are few (E = 1: Shape::kHeldGC / GT; a group of L lanes: L per round)
    constexpr int kGC = E == 1 ? Shape::kHeldGC : sub_held_chunks<Shape, L>(true), kGT = E == 1 ? Shape::kHeldGT : sub_held_chunks<Shape, L>(false);
    constexpr bool HOLD = Shape::kGreedyHeld && (E == 1 || (MATE_SUB_HOLD && kGC + kGT <= 12));
    PackDescriptorsT<HOLD ? kGC : kPackGC, HOLD ? kGT : kPackGT> held;
    RangeRoles roles;
    {
        Ctx<ObsT, L> c(p, gk, smem + slot * p.lds_wave_bytes, lane, env, FLOW_GREEDY);
        c.shift = shift;
        if constexpr (IMAGE) { range_roles(c, roles); pin_roles(roles); image_statics(c); }
        else if constexpr (HOLD) { if (packs_rows_f32(c)) load_pack_descriptors(c, held); }
    }
    // the lane's range-test roles held in registers, and with them the collision screen carried from step to step (NearCarry)
#ifdef MATE_NO_GREEDY_ROLES      // (experiment: what the held lane roles and the carried collision screen are worth)
    constexpr bool ROLES = false;
#else
    constexpr bool ROLES = E == 1 && Shape::kGreedyRoles;
#endif
    NearCarry near{};
    if constexpr (ROLES) {
        Ctx<ObsT> c(p, gk, smem + wave * p.lds_wave_bytes, lane, env, FLOW_GREEDY);
        if constexpr (!IMAGE) { range_roles(c, roles); pin_roles(roles); }
        near_seed(c, roles, near);
    }
    int last_gw = -1;                                              // image_targets: the goal word behind a target's goal / cargo slots
    uint32_t hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    const int wave_slot = (int)(hw_id & 15u);
    bool stepped = false;                                          // statics written (see Ctx::statics_done)
    DrawCarry carry{0u, 0u, 0xffffffffu};
#ifdef MATE_PHASE_CLOCKS
    long long acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // 0-6 as in rollout_kernel, 7 loop, 8 agents observe, 11 communicate, 12 choose, 9 the zoom solve, 10 the actions
    long long t_prev = (long long)__builtin_amdgcn_s_memtime();
    const long long t_first = t_prev, r_first = (long long)__builtin_amdgcn_s_memrealtime();
#define GREEDY_STAMP(i) do { const long long t_now = (long long)__builtin_amdgcn_s_memtime(); acc[i] += t_now - t_prev; t_prev = t_now; } while (0)
#define GREEDY_ACC acc, &t_prev
#elif defined(MATE_ISA_MARKS)      // tools/isa_phases.py: phase boundaries as comments in the -S output (no instruction is emitted)
#define GREEDY_STAMP(i) asm volatile("; ==== MATE_GREEDY_PHASE " #i)
#define GREEDY_ACC nullptr, nullptr
#else
#define GREEDY_STAMP(i) do { } while (0)
#define GREEDY_ACC nullptr, nullptr
#endif
#pragma clang loop unroll(disable)
    for (int r = 0; r < g.rollout_steps; ++r) {
        int lane_r = lane, wave_r = wave, slot_r = slot;          // opaque per iteration, see rollout_kernel
        asm volatile("" : "+v"(lane_r));
        asm volatile("" : "+s"(wave_r));
        if constexpr (E == 1) slot_r = wave_r; else asm volatile("" : "+v"(slot_r));
        const Params *pr = pp;
        asm volatile("" : "+s"(pr));
        const Shape shape_r(pr, true);
        const Params &p = shape_r.get();
        const int64_t env_w = (int64_t)blockIdx.x * (4 * E) + slot_r;
        const int64_t env_r = env_w < g.N ? env_w : g.N - 1;
        if constexpr (ROLES) pin_roles(roles, p.range_rounds, IMAGE, p.sector_rounds == 1);
        Ctx<ObsT, L> c(p, gk, smem + slot_r * p.lds_wave_bytes, lane_r, env_r, FLOW_GREEDY);
        c.shift = shift;
        c.out = (int64_t)r * g.N + env_r;
        c.act_cam = act_cam; c.act_tgt = act_tgt;
        c.statics_done = stepped;
        c.pivots = false;                    // (at the register limit: the quarter path's two round trips)
        const bool active = in_batch && c.ei(EI_DONE) == 0;
        if (g.rotate_prio) {
            const int turn = (r + wave_slot) & 3;
            if (turn & 2) { if (turn & 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
            else { if (turn & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        }
        GREEDY_STAMP(7);
        if (!active) {                                             // past the end of the batch, or the episode has ended: no agents, no step
            if (in_batch && lane_r == 0) {
                if (g.scalars) { float *o = g.scalars + c.out * 8; o[0] = 0.f; o[1] = 0.f; o[2] = 2.f; o[3] = o[4] = o[5] = o[6] = o[7] = 0.f; }
                if (g.idle_steps) g.idle_steps[env_r] += 1;
            }
            continue;
        }
        // (the joint actions stay in LDS; the last executed step's are published once, at the end of the launch)
        MATE_PHASE(128, greedy_policy_body<ObsT, L>(p, q, a, c.st, c.dy, c.di, c.mask, wave_r, lane_r, env_r, true, act_cam, act_tgt, GREEDY_ACC, false, shift));
        wave_sync();
        GREEDY_STAMP(10);
        if (q.caller_team >= 0) {
            load_caller_actions(c, q.caller_team, act_cam, act_tgt);
            wave_sync();
        }
        const uint32_t tick = p.dev_tick + g.tick + (uint32_t)r;
        StepDraws draws{0.0, 0.0};
        MATE_PHASE(1, draws = step_draws(c, tick, &carry));       // see-through uniforms only (mode() is MODE_STEP)
        GREEDY_STAMP(0);
        MATE_PHASE(2, simulate_cameras(c, draws, true));
        GREEDY_STAMP(1);
        MATE_PHASE(4, simulate_targets(c, draws, ROLES ? &near : nullptr));
        GREEDY_STAMP(2);
        uint32_t seen = 0u;
        MATE_PHASE(8,
            if constexpr (ROLES) update_view<true, true>(c, tick, S_TRANSMIT, true, roles, seen, &near);
            else { RangeRoles none; update_view<false, true>(c, tick, S_TRANSMIT, true, none); });
        GREEDY_STAMP(3);
        MATE_PHASE(16, assign_and_score(c, tick, g.scalars));
        GREEDY_STAMP(4);
        if constexpr (IMAGE) {
            MATE_PHASE(32, image_targets(c, last_gw); image_blocks(c, roles, seen));
            GREEDY_STAMP(5);
            MATE_PHASE(64, image_store(c); store_masks(c));
        } else {
        MATE_PHASE(32, fill_scratch(c));
        GREEDY_STAMP(5);
        MATE_PHASE(64,
            if constexpr (HOLD) pack_observations<true, Shape::kGreedyHeld>(c, held);
            else { PackDescriptors now; pack_observations<false, Shape::kGreedyHeld>(c, now); });
        }
        wave_sync();
        stepped = true;
        GREEDY_STAMP(6);
    }
#ifdef MATE_PHASE_CLOCKS
    if (in_batch && lane == 0 && g.phase_clocks) {
        for (int i = 0; i < 13; ++i) g.phase_clocks[env * kClockStride + i] = acc[i];
        g.phase_clocks[env * kClockStride + 14] = (long long)__builtin_amdgcn_s_memtime() - t_first;
        g.phase_clocks[env * kClockStride + 15] = (long long)__builtin_amdgcn_s_memrealtime() - r_first;
    }
#endif
    if (in_batch && !untouched) {
        Ctx<ObsT, L> c(p, gk, smem + slot * p.lds_wave_bytes, lane, env, FLOW_GREEDY);
        store_dynamic(c);
        double *dst = q.pol + env * q.PW;
        for (int k = lane; k < q.PW; k += L) dst[k] = a.f[k];
        if (stepped) {                                             // what mate_engine_policy_actions reads: the last executed step's joint actions
            for (int k = lane; k < 2 * p.Nc; k += L) q.cam_act[env * 2 * p.Nc + k] = act_cam[k];
            for (int k = lane; k < 2 * p.Nt; k += L) q.tgt_act[env * 2 * p.Nt + k] = act_tgt[k];
        }
    }
}

// =============================================================================================
// ONE (agents act, environment steps) iteration per launch: mate_engine_step_greedy / _step_versus_greedy (MultiCamera / MultiTarget,
// mate/wrappers/single_team.py:245-264 -- the flow of every examples/*/config.py: the learner plays one team at every step).
//
// Round 3 ran these flows on rollout_greedy_kernel with a single step: one launch, but a launch that pays the fused rollout's
// prologue -- lane roles, the seed of the carried collision screen, up to twelve held descriptor chunks: work that amortises over
// tens of steps -- in front of its only step, at four waves per SIMD (98-128 registers): 24 us at 4096 x MATE-4v8-9 where the
// agents' kernel and step_kernel take 9 + 15 as two launches, 91 us at 16 384.  This kernel is step_kernel's own sequence -- the
// records, the agents' memory and the previous view loaded under the Philox draws, the light packer, eight waves per SIMD where
// the registers allow -- with greedy_policy_body between the entity table and the kinematics and the joint actions handed over in
// LDS (FLOW_STEP_GREEDY).  Same phase functions, same bytes as the two-launch form and as the fused rollout with one step (tested).
// LDS per workgroup: 4 step slices, then 4 x (agents' memory + staging + joint actions + the previous step's mask words).
// (compiled for every shipped shape but MATE-1v2-*: with one camera and two targets the register allocator leaves 36-48 bytes of
// private scratch, which costs ~5 us per launch -- mate_amd/build.py refuses such a kernel; those two scenarios keep the one-step
// rollout_greedy_kernel form)
constexpr bool step_greedy_compiled(int Nc, int Nt, int /*No*/) { return !(Nc == 1 && Nt == 2); }
// (`cameras` false: the caller plays the cameras -- MultiCamera, the flow of every examples/*/camera/config.py --, so only the target
// agents act: their section of the memory record, the joint actions and the mask words: 0.7 KB per environment instead of 1.8, and seven
// workgroups per CU instead of six at MATE-4v8-9)
__host__ __device__ constexpr int step_greedy_slice_bytes(int PW, int TW, int Nc, int Nt, int MW, bool cameras) {
    return cameras ? policy_slice_bytes(PW, Nc, Nt) + shape_round_up(MW * 4 + 4, 16) : shape_round_up((TW + 2 * (Nc + Nt)) * 8 + MW * 4 + 4, 16);
}

template <typename ObsT, typename Shape>
__global__ __launch_bounds__(256, 4) __attribute__((amdgpu_num_sgpr(96)))   // (eight waves per SIMD, as step_kernel)
void step_greedy_kernel(const Params *__restrict__ pp, const Ptrs g, const PolicyPtrs q_arg) {
    const PolicyPtrs &q = kernarg_policy_ptrs(q_arg);
    const Shape shape(pp);
    const Params &p = shape.get();
    extern __shared__ __align__(16) unsigned char smem[];
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.done_count) {      // (tick and list parity: launch arguments or the device-resident counter, see step_kernel)
        const int32_t parity = (int32_t)((p.dev_group + (uint32_t)g.parity) & 1u);
        g.done_count[parity ^ 1] = 0;
        g.ctrl[0] = parity;
    }
    const uint32_t tick = p.dev_tick + g.tick;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int64_t env = (int64_t)blockIdx.x * 4 + wave;
    if (env >= g.N) return;                                          // (the four waves of a workgroup never synchronise)
    const Ptrs &gk = kernarg_ptrs(g);
    Ctx<ObsT> c(p, gk, smem + wave * p.lds_wave_bytes, lane, env, FLOW_STEP_GREEDY);
    // which teams' agents act (wave-uniform): the acting teams' sections of the memory record are loaded, held and stored -- words
    // [w_lo, w_hi) of [target section | camera section]; with the cameras played by the caller the LDS slice holds nothing else
    const bool cams = q.caller_team != 0, tgts = q.caller_team != 1;
    const int w_lo = tgts ? 0 : q.TW, w_hi = cams ? q.PW : q.TW, w_n = w_hi - w_lo;
    const int pol_bytes = step_greedy_slice_bytes(q.PW, q.TW, p.Nc, p.Nt, p.MW, cams);
    unsigned char *pol_base = smem + 4 * p.lds_wave_bytes + wave * pol_bytes;
    PolCtx<ObsT> a(p, q, pol_base, cams);
    double *act_cam = a.f + (cams ? q.PW + policy_staging_words(p.Nc, p.Nt) : q.TW), *act_tgt = act_cam + 2 * p.Nc;
    uint32_t *mk = reinterpret_cast<uint32_t *>(act_tgt + 2 * p.Nt);      // the view the previous step / reset left (the agents' observation gate)
    // the agents' memory and the previous view on their way together with the records (load_records_with_draws issues those and
    // runs the step's Philox draws under all of it)
    const double *pol_src = q.pol + env * q.PW + w_lo;
    double pw0 = pol_src[lane < w_n ? lane : 0], pw1 = 0.0, pw2 = 0.0;
    if (w_n > 64) pw1 = pol_src[lane + 64 < w_n ? lane + 64 : 0];
    if (w_n > 128) pw2 = pol_src[lane + 128 < w_n ? lane + 128 : 0];
    uint32_t mw0 = q.masks[env * p.MW + (lane < p.MW ? lane : 0)];
    asm volatile("" : "+v"(pw0), "+v"(pw1), "+v"(pw2), "+v"(mw0));
#ifdef MATE_PHASE_CLOCKS      // per-wave stamps (tools/versus_phases.py): 0 begin, 1 records, 2 entity table, 3 agents, 4 kinematics, 5 view, 6 goals, 7 rows, 8 end;
    const long long r_begin = (long long)__builtin_amdgcn_s_memrealtime();      // 9-13: the agents' sub-phases (observe, zoom, actions, communicate, choose)
    long long pol_acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pol_prev = 0;
#define SG_STAMP(i) do { if (lane == 0 && g.phase_clocks) g.phase_clocks[env * kClockStride + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define SG_ACC pol_acc, &pol_prev
#else
#define SG_STAMP(i) do { } while (0)
#define SG_ACC nullptr, nullptr
#endif
    SG_STAMP(0);
    const StepDraws draws = load_records_with_draws(c, tick, true);
    double *pol_lds = a.f + w_lo;
    if (lane < w_n) pol_lds[lane] = pw0;
    if (lane + 64 < w_n) pol_lds[lane + 64] = pw1;
    if (lane + 128 < w_n) pol_lds[lane + 128] = pw2;
    for (int k = lane + 192; k < w_n; k += 64) pol_lds[k] = pol_src[k];
    if (lane < p.MW) mk[lane] = mw0;
    for (int k = lane + 64; k < p.MW; k += 64) mk[k] = q.masks[env * p.MW + k];
    if (lane == 0) mk[p.MW] = 0u;                                        // (seen_mask reads two words)
    wave_sync();
    if (c.ei(EI_DONE) != 0) {        // finished: waiting for the reset launch (immediate: behind this one; batched: at the interval's end) -- no agents, no step
        if (lane == 0 && g.scalars) { float *o = g.scalars + c.out * 8; o[0] = 0.f; o[1] = 0.f; o[2] = 2.f; o[3] = o[4] = o[5] = o[6] = o[7] = 0.f; }
        if (lane == 0 && g.idle_steps) g.idle_steps[env] += 1;
        if (lane == 0 && g.done_count && c.ei(EI_DONE) == 1) {          // finished under auto_reset = 0 earlier: not on the list yet
            const int parity = c.list_parity();
            const int slot = atomicAdd(g.done_count + parity, 1);
            g.done_list[(int64_t)parity * g.N + slot] = (int32_t)env;
            reinterpret_cast<int32_t *>(g.dyn + env * p.DW + p.DF)[p.Nt * TI_STRIDE + EI_DONE] = 3;
        }
        return;
    }
    SG_STAMP(1);
    build_entities(c);
    wave_sync();
    SG_STAMP(2);
#ifdef MATE_PHASE_CLOCKS
    pol_prev = (long long)__builtin_amdgcn_s_memtime();
#endif
    greedy_policy_body<ObsT>(p, q, a, c.st, c.dy, c.di, mk, wave, lane, env, true, act_cam, act_tgt, SG_ACC, false);
    wave_sync();
    if (q.caller_team >= 0) {
        load_caller_actions(c, q.caller_team, act_cam, act_tgt);
        wave_sync();
    }
    c.act_cam = act_cam; c.act_tgt = act_tgt;
    SG_STAMP(3);
    simulate_cameras(c, draws, true);
    simulate_targets(c, draws);
    SG_STAMP(4);
    const bool reg_tail = p.sector_rounds <= 1;
    int tracked_reg = 0, inside_reg = -1;
    PackDescriptors pack_desc;
    constexpr int GCE = Shape::kHeldGC, GTE = Shape::kHeldGT;
    const bool early_desc = Shape::kGreedyHeld && GCE + GTE <= 8 && packs_rows_f32(c);
    if (reg_tail) {
        RangeRoles none;
        uint32_t seen_unused;
        unsigned long long sector_ballot = 0ull;
        update_view<false, false>(c, tick, S_TRANSMIT, true, none, seen_unused, nullptr, &sector_ballot);
        view_tail_regs(c, sector_ballot, tracked_reg, inside_reg);
    } else update_view(c, tick, S_TRANSMIT, true);
    SG_STAMP(5);
    if (early_desc) load_pack_descriptors(c, pack_desc);
    if (reg_tail) assign_and_score(c, tick, g.scalars, &tracked_reg, &inside_reg);
    else assign_and_score(c, tick, g.scalars);
    SG_STAMP(6);
    // the records, the agents' memory and the joint actions are final: out before the packer (see step_kernel)
    store_dynamic(c);
    { double *dst = q.pol + env * q.PW + w_lo; for (int k = lane; k < w_n; k += 64) dst[k] = pol_lds[k]; }
    for (int k = lane; k < 2 * p.Nc; k += 64) q.cam_act[env * 2 * p.Nc + k] = act_cam[k];      // what mate_engine_policy_actions reads: this step's joint actions (the caller's team's as decoded)
    for (int k = lane; k < 2 * p.Nt; k += 64) q.tgt_act[env * 2 * p.Nt + k] = act_tgt[k];
    fill_scratch(c);
    if (early_desc) pack_observations<true>(c, pack_desc); else pack_observations<false>(c, pack_desc);
    SG_STAMP(7);
    SG_STAMP(8);
#ifdef MATE_PHASE_CLOCKS
    if (lane == 0 && g.phase_clocks) {
        for (int i = 8; i < 13; ++i) g.phase_clocks[env * kClockStride + 1 + i] = pol_acc[i];
        g.phase_clocks[env * kClockStride + 15] = (long long)__builtin_amdgcn_s_memrealtime() - r_begin;
    }
#endif
}

}  // namespace mate
