// engine_kernels.hpp -- gfx950 kernels of the batched MultiAgentTracking step engine.
//
// Mapping: a group of L lanes per environment (Ctx<ObsT, L>) -- ONE WAVE (64 lanes), four environments per 256-thread
// workgroup.  An environment's state is two small contiguous records (static geometry,
// dynamic state) that the wave loads with one coalesced 8-byte-per-lane read, stages in LDS,
// and then works on with lanes mapped to (entity, entity) pairs; wave ballots produce the
// visibility masks; the observation rows are gathered from an LDS scratch through a
// per-scenario descriptor table and stored as contiguous 16-byte-per-lane rows.
#pragma once
#include "device_math.hpp"

// Measurement hooks (phases compiled out or executed twice, a fake occlusion record, plain row stores): experiment builds only --
// experiments.hpp says what each does.  The shipped library sees the defaults below: every phase once, the real record,
// non-temporal row stores.
#if defined(MATE_ABLATE) || defined(MATE_DOUBLE) || defined(MATE_LUT_FAKE) || defined(MATE_STORE_PLAIN)
#include "experiments.hpp"
#else
#define MATE_PHASE(bit, ...) do { __VA_ARGS__; } while (0)
#define MATE_PHASE_AGAIN(bit, ...) do { } while (0)
#define MATE_ZOOM_ITERATIONS 20
#define MATE_LUT_TABLE_OF(lc) (lc)
#define MATE_ROW_STORE(v, dst) __builtin_nontemporal_store((v), (dst))
#endif

namespace mate {

enum Mode : int32_t { MODE_STEP = 0, MODE_STEP_RANDOM = 1, MODE_OBSERVE = 2 };

// dynamic int record, per target (5 ints) then per environment
enum { TI_BOUNTY = 0, TI_FREIGHT = 1, TI_GW = 2, TI_TSTEPS = 3, TI_TRSTEPS = 4, TI_STRIDE = 5 };
enum { EI_REMAINING = 0, EI_AWAITING = 16, EI_DELIVERED = 20, EI_EPSTEP = 21, EI_TICK = 22, EI_EPISODE = 23, EI_DONE = 24, EI_PAD = 25, EI_COUNT = 26 };
// TI_GW packing: bits 0-7 goal+1 (0 = no goal), 8-15 cargo weight, 16-19 empty bits, 24 colliding

struct Params {
    int32_t Nc, Nt, No, NK, NJ;
    int32_t Dc, Dt, cam_elems, tgt_elems;
    int32_t SW, DF, NI, DW;
    int32_t n_sector, n_range, sector_rounds, range_rounds;
    int32_t bit_cc, bit_range, bit_camobs, bit_always, MW;
    // Visibility FLAGS (one ObsT-wide word each, all-ones or zero, in LDS) are numbered compactly -- the packed mask
    // words pad every group of bits to 64, which as flags wasted half of the largest LDS region: sector flags
    // [0, n_sector) coincide with their mask bits, then the range tests, camera->obstacle, the always-true flag and
    // the Nt + No + Nc + No team-shared flags of the SharedFieldOfView mode.
    int32_t fs_range, fs_camobs, fs_always, fs_shared, nflags;
    int32_t nscratch, sc_cam, sc_tgt, sc_obs;
    int32_t tgt_table_off;   // first target descriptor (cam_elems rounded up to 4)
    int32_t kmax, nbucket;
    int32_t max_episode_steps, sparse_reward, num_cargoes_per_target, shuffle, start_with_cargoes, n_high;
    int32_t obs_f64;
    float inv_Nt, inv_NK, inv_NJ, inv_Nc;
    double tau, cam_radius, theta_min, rmax, rot, zoom, area, tgt_step, tgt_sight;
    double freight_scale, bounty_scale, reward_scale, max_team_reward;
    double obs_r_lo, obs_r_hi;
    uint32_t seed_lo, seed_hi, first_env;
    int32_t desc_table_bytes, lds_wave_bytes, off_st, off_dy, off_tmp, off_scratch, off_mask, off_misc, off_flags, off_ent, off_list;
    float inv_No;
    int32_t export_width;
    // Device-resident step counter (mate_engine_device_tick): while `dev_tick_on` the step kernels take the Philox
    // tick and the ping-pong parity of the finished-episode lists from here instead of from the launch arguments, and
    // the auto-reset launch that follows every step advances it -- a (step, auto-reset) pair then has the same
    // arguments at every step and a captured HIP graph of K pairs can be replayed.
    uint32_t dev_tick;            // written only by the auto-reset launch, through Ptrs::dev_tick_ptr
    uint32_t dev_group;           // auto-reset launches so far (its low bit is the list parity); follows dev_tick in memory
    int32_t dev_tick_on;
    // "Row image" mode of the fused rollouts (fill_shape(..., image = true), see image_statics): the observation rows of the
    // environment live in LDS for the whole launch, next to a table of the entities' public states; there are no flag words,
    // no gather scratch and no descriptors.
    int32_t image, off_pub, off_img;
};

// Everything in Params that follows from the entity counts and the observation type alone.  Shared by the
// host (mate_engine_create) and by the shape-specialised kernels, which overwrite the loaded record with
// these compile-time values so that index arithmetic, loop bounds and LDS offsets fold to literals.
__host__ __device__ constexpr int shape_round_up(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ constexpr void fill_shape(Params &p, int Nc, int Nt, int No, bool obs_f64, bool image = false) {
    p.Nc = Nc; p.Nt = Nt; p.No = No; p.NK = No + Nc; p.NJ = Nc + No + Nt;
    p.Dc = 13 + 9 + 5 * Nt + 4 * No + 7 * Nc;    // constants.py:267-282
    p.Dt = 13 + 14 + 7 * Nc + 4 * No + 5 * Nt;   // constants.py:285-300
    p.cam_elems = Nc * p.Dc; p.tgt_elems = Nt * p.Dt;
    p.tgt_table_off = shape_round_up(p.cam_elems, 4);
    p.SW = 3 * Nc + 3 * No + 1;
    p.DF = 2 * Nc + 2 * Nt + 2;
    p.NI = Nt * TI_STRIDE + EI_COUNT; if (p.NI & 1) p.NI += 1;
    p.DW = p.DF + p.NI / 2;
    p.n_sector = Nc * Nt + Nc * Nc; p.n_range = Nt * p.NJ;
    p.sector_rounds = (p.n_sector + 63) / 64; p.range_rounds = (p.n_range + 63) / 64;
    p.bit_cc = Nc * Nt;
    p.bit_range = p.sector_rounds * 64;
    p.bit_camobs = p.bit_range + p.range_rounds * 64;
    p.bit_always = p.bit_camobs + Nc * 64;
    p.MW = p.bit_always / 32 + 1;
    p.fs_range = p.n_sector; p.fs_camobs = p.fs_range + p.n_range; p.fs_always = p.fs_camobs + Nc * No;
    p.fs_shared = p.fs_always + 1; p.nflags = p.fs_shared + Nt + 2 * No + Nc;
    p.sc_cam = 30; p.sc_tgt = p.sc_cam + 10 * Nc; p.sc_obs = p.sc_tgt + 14 * Nt; p.nscratch = shape_round_up(p.sc_obs + 3 * No, 4);
    p.kmax = shape_round_up(360 + No * 185 + 2, 8);
    p.nbucket = 368;
    p.obs_f64 = obs_f64;
    p.inv_Nt = 1.0f / (float)Nt; p.inv_NK = p.NK > 0 ? 1.0f / (float)p.NK : 0.f; p.inv_NJ = 1.0f / (float)p.NJ;
    p.inv_Nc = Nc > 0 ? 1.0f / (float)Nc : 0.f; p.inv_No = No > 0 ? 1.0f / (float)No : 0.f;
    p.export_width = 2 * Nc + 3 * No + Nt + Nc * No + 2 * Nc + 2 * Nt + Nt + 4 * Nt + 4 * Nt + 5 * Nt + 20 + 7;
    // LDS carve of one environment-wave
    const int obs_size = obs_f64 ? 8 : 4;
    p.desc_table_bytes = shape_round_up((p.tgt_table_off + shape_round_up(p.tgt_elems, 4)) * 4, 16);
    int off = 0;
    p.off_st = off; off += p.SW * 8;                                 // (8-byte words, read as such: the dynamic record follows without padding)
    p.off_dy = off; off = shape_round_up(off + p.DW * 8, 16);
    // (sight^2 | step lengths | the predrawn transmittance uniforms, one per camera->target pair up to 64: only where there are obstacles to see through)
    p.off_tmp = off; off += shape_round_up((Nc + Nt + ((Nc > 0 && No > 0) ? (Nc * Nt < 64 ? Nc * Nt : 64) : 0)) * 8, 16);
    p.off_scratch = off; off += image ? 0 : shape_round_up(p.nscratch * obs_size, 16);
    p.off_mask = off; off += shape_round_up(p.MW * 4, 16);
    p.off_misc = off; off += shape_round_up((4 * Nt + 8) * 4, 16);
    p.off_flags = off; off += image ? 0 : shape_round_up(p.nflags * obs_size, 16);
    p.off_ent = off; off += shape_round_up(3 * p.NJ * 8 + 3 * p.NJ * 4, 16);   // f64 table + its f32 shadow
    p.off_list = off; off += (p.sector_rounds >= 2 && p.n_sector <= 256) ? shape_round_up(p.n_sector, 16) : 0;   // compacted sector candidates, one byte each (update_view)
    // row-image mode: public states [cameras 8 | targets 8 | obstacles 4 floats each] (+ 16 bytes a block read may overrun),
    // then the environment's observation rows exactly as they lie in the output buffers (camera block, target block)
    p.image = image ? 1 : 0;
    p.off_pub = off; off += image ? (8 * Nc + 8 * Nt + 4 * No) * 4 + 16 : 0;
    p.off_img = off; off += image ? (p.cam_elems + p.tgt_elems) * 4 : 0;
    p.lds_wave_bytes = off;
}
// Shapes the row-image mode is compiled for: f32 rows in whole 16-byte chunks -- or, a small target block (at most 512 floats), in
// whole 8-byte chunks: MATE-4v2-9's two target rows are 202 floats, so an environment's block begins on an 8-byte boundary only --,
// one round of sector pairs, at most three of range pairs (the lane roles are held in registers), and 16 environments per CU still
// resident (160 KiB of LDS).
constexpr bool image_fits(int Nc, int Nt, int No) {
    Params p{};
    fill_shape(p, Nc, Nt, No, false, true);
    const bool tgt16 = p.tgt_elems % 4 == 0 && p.tgt_elems <= 4 * 441, tgt8 = p.tgt_elems % 2 == 0 && p.tgt_elems <= 512;
    // (one round of sector pairs, or none: MATE-Navigation has no camera -- eight targets among 32 obstacles, 320 range pairs in five rounds)
    return p.cam_elems % 4 == 0 && (tgt16 || tgt8) && p.cam_elems <= 4 * 128 &&      // (image_store's unrolled rounds: 3 x 64 camera chunks, 7 x 64 - 7 target chunks)
           p.sector_rounds <= 1 && p.range_rounds <= 5 && p.lds_wave_bytes <= 10 * 1024;
}

// Environments per wave the fused rollouts of a compiled shape CAN run with (Ctx, `L` = 64 / E lanes per environment): 4 where the
// agents fit eight lanes per team (the greedy agents' lane roles: cameras from lane 0, targets from lane L / 2) and the pairs of a
// visibility test fit a few rounds of 16 lanes -- every shape with at most four cameras and four targets; MATE-4v8-9 and MATE-8v8-*
// fill half a wave and keep their register-resident kernels.  Which launches DO is the host's choice (sub_wave_of_launch).
__host__ __device__ constexpr int sub_wave_of(int Nc, int Nt, int No) {
    return (Nc <= 4 && Nt <= 4 && Nc + Nt <= 8 && Nt * (Nc + No + Nt) <= 16 * 8 && Nc * (Nt + Nc) <= 16 * 4) ? 4
         // two per wave: MATE-4v8-0 under the Greedy flows (x1.5 .. 1.6; with its nine obstacles MATE-4v8-9 measured x0.99 .. 1.09 and keeps one)
         : (Nc == 4 && Nt == 8 && No == 0) ? 2
         : 1;
}
// (experiment, -DMATE_SUB_EIGHT: eight per wave where the agents fit eight lanes and the agents' (sender, recipient) pairs one round)
__host__ __device__ constexpr int sub_wave_eight(int Nc, int Nt, int No) {
#ifdef MATE_SUB_EIGHT
    return (Nc <= 2 && Nt <= 4 && Nc * Nt <= 8) ? 8 : sub_wave_of(Nc, Nt, No);
#elif defined(MATE_SUB_TWO)      // (experiment: two per wave where four are the rule)
    return sub_wave_of(Nc, Nt, No) == 4 ? 2 : sub_wave_of(Nc, Nt, No);
#else
    return sub_wave_of(Nc, Nt, No);
#endif
}

// Kernel shape policies: AnyShape reads every constant from the device-resident Params on demand;
// FixedShape<Nc, Nt, No> is compiled for one scenario shape (the host picks it when the counts match).
struct AnyShape {
    static constexpr int kHeldGC = 2, kHeldGT = 6;
    static constexpr int kChunksC = 0, kChunksT = 0;      // 16-byte chunks of the shape's f32 row blocks (unknown here)
    static constexpr int kSubWave = 1;
    static constexpr bool kImage = false;
    static constexpr bool kHoldRoles = false;      // rollout kernel: lane roles in registers (needs compile-time round counts)
    static constexpr bool kGreedyRoles = false;
    static constexpr int kGreedyBlocks = 4;        // rollout_greedy_kernel: workgroups per CU the register budget is set for
    static constexpr bool kGreedyHeld = false;     // ... and whether it keeps the observation descriptors in registers across steps
    static constexpr bool kEarlyStateStore = true; // step_kernel: the state record leaves ahead of the packer (see there)
    const Params *pp;
    __device__ __forceinline__ explicit AnyShape(const Params *q, bool /*through_constant*/ = false) : pp(q) {}
    __device__ __forceinline__ const Params &get() const { return *pp; }
};
constexpr int shape_range_rounds(int Nc, int Nt, int No) { return (Nt * (Nc + No + Nt) + 63) / 64; }
struct ParamsWords { uint64_t w[sizeof(Params) / 8]; };
static_assert(sizeof(ParamsWords) == sizeof(Params), "Params is a whole number of 8-byte words");
__device__ __forceinline__ Params load_params_constant(const Params *q) {
    const __attribute__((address_space(4))) uint64_t *src = (const __attribute__((address_space(4))) uint64_t *)q;
    ParamsWords t;
#pragma unroll
    for (unsigned i = 0; i < sizeof(Params) / 8; ++i) t.w[i] = src[i];
    return __builtin_bit_cast(Params, t);
}
template <int NC, int NT, int NO, bool F64, bool IMAGE = false>
struct FixedShape {
    static constexpr bool kImage = IMAGE;
    static_assert(!IMAGE || (!F64 && image_fits(NC, NT, NO)), "row-image mode: f32 observations of a shape that fits");
    static constexpr bool kHoldRoles = shape_range_rounds(NC, NT, NO) <= (IMAGE ? 5 : 3);      // 5 words per round: three rounds fit beside the held descriptors, five where the row image needs none
    static constexpr bool kGreedyRoles = shape_range_rounds(NC, NT, NO) <= 4;    // rollout_greedy_kernel: MATE-8v8-9's four rounds too
    // observation descriptors the fused rollouts hold per lane: all chunks of the shape's rows (up to twelve uint4)
    static constexpr int kRowsC = (NC * (13 + 9 + 5 * NT + 4 * NO + 7 * NC) / 4 + 63) / 64, kRowsT = (NT * (13 + 14 + 7 * NC + 4 * NO + 5 * NT) / 4 + 63) / 64;
    static constexpr int kHeldGC = (kRowsC + kRowsT <= 12) ? kRowsC : 2, kHeldGT = (kRowsC + kRowsT <= 12) ? kRowsT : 6;
    static constexpr int kChunksC = NC * (13 + 9 + 5 * NT + 4 * NO + 7 * NC) / 4, kChunksT = NT * (13 + 14 + 7 * NC + 4 * NO + 5 * NT) / 4;
    // Environments per wave of the fused rollouts (sub_wave_of below): the scenarios whose agents and visibility pairs fill a quarter of a wave
    static constexpr int kSubWave = sub_wave_eight(NC, NT, NO);
    static constexpr int kGreedyBlocks = 4;
    static constexpr bool kGreedyHeld = true;
    // (MATE-8v8-9 sits at 63 of the 64 registers of full occupancy: with the state's stores ahead of the packer it needs 65)
    static constexpr bool kEarlyStateStore = !(NC == 8 && NT == 8 && NO == 9);
    Params local;
    // `through_constant` (the fused rollouts): the record is copied through the CONSTANT address space.  Nothing writes it
    // while a kernel runs (the host between launches, the auto-reset launch's dev_tick for the next one), and only then
    // may the compiler fetch its fields with scalar loads wherever the copy is made.  Through the generic pointer the copy
    // inside the rollout loop -- behind the previous step's observation stores, which might alias -- was ten flat_load per
    // step into 26 VGPRs, and their s_waitcnt vmcnt(0) made every wave wait for the HBM acknowledgement of those stores
    // before its next step (rollout_kernel -1.5 %, rollout_greedy_kernel -3 %).  The single-step kernels copy at their
    // top, where the compiler emits scalar loads anyway, and live at the 96-SGPR limit: they keep the plain copy.
    __device__ __forceinline__ explicit FixedShape(const Params *q, bool through_constant = false)
        : local(through_constant ? load_params_constant(q) : *q) { fill_shape(local, NC, NT, NO, F64, IMAGE); }
    __device__ __forceinline__ const Params &get() const { return local; }
};

struct Ptrs {
    double *stat;                 // [N][SW]
    double *dyn;                  // [N][DW] (8-byte words: DF doubles then NI ints)
    double2 *lut_knots;           // [N][Nc][kmax]  (phi, rho)
    uint16_t *lut_bucket;         // [N][Nc][nbucket]
    double2 *lut_deg;             // [N][Nc][kLutCells][kDegWords] per-cell segment records (fast lookup path; kCellsPerDegree cells per degree)
    int32_t *lut_count;           // [N][Nc]
    double2 *lut_knots_outer;     // optional (mate_engine_enable_outer_boundary): [N][Nc][kmax_outer] knots of Camera.boundary_outer
    int32_t *lut_count_outer;     // [N][Nc]
    int32_t kmax_outer;
    const uint32_t *desc;         // [cam_elems + tgt_elems]  src | bit << 16
    const uint2 *xdesc;           // optional fused post-processing: per element (descriptor, LDS offset of the row's own x or y)
    const void *xab;              // ... and (scale, bias) as ObsT pairs; NULL = plain observations
    const void *scratch_init;     // [nscratch] ObsT
    const double *reset_ranges;   // [Nc+No+Nt][4] cameras, obstacles, targets
    const void *cam_act, *tgt_act;
    const double *tape_ct, *tape_goal;
    void *cam_obs, *tgt_obs;
    float *scalars;
    uint32_t *masks;
    int32_t *idle_steps;          // [N] steps an environment spent idle waiting for a batched reset
    uint32_t *own_masks;          // engine-owned copy of the packed masks (input of the on-device policies), or NULL
    int32_t *done_count;          // [2] ping-pong counters
    int32_t *done_list;           // [2][N]
    int32_t *flag_count;          // [1] environments selected by a batched (flagged) reset ...
    int32_t *flag_list;           // [N] ... and their indices
    int32_t *lut_overflow;        // [1 + N Nc] count, then environment * Nc + camera of the occlusion tables a small-LDS table launch deferred
    int32_t lut_defer_above;      // that launch: tables with more rays than this go on the list (0 = build everything here)
    const uint8_t *reset_mask;    // optional
    const double *reset_tape;     // tape mode of reset (mate_engine_reset_tape): [N][reset_tape_len] uniforms, or NULL = Philox
    int32_t reset_tape_len;
    int32_t *reset_draws;         // optional [N]: uniforms each reset consumed (-1: the tape ran out)
    double *ep_stats;             // optional [5] accumulators over finished episodes: count, return, length, coverage, delivered
    double *sort_scratch;         // occlusion-table sort arrays of scenarios too large for the LDS (ResetLds::sort_in_hbm)
    uint32_t *dev_tick_ptr;       // &Params::dev_tick of the device-resident parameter block (advanced by the auto-reset launch)
    int32_t *ctrl;                // [0]: list parity the last step launch used (device-resident step counter mode)
    long long *phase_clocks;      // [N][16] s_memtime stamps (debug builds with -DMATE_PHASE_CLOCKS)
    int32_t debug_skip;           // phase-ablation mask (debug builds only)
    int64_t N;
    int32_t mode, act_f64, parity, reset_kind;     // act_f64: bit 0 camera actions, bit 1 target actions are f64 (else f32)
    int32_t rollout_steps;        // steps per launch of rollout_kernel
    int32_t freeze_done;          // batched auto-reset: finished environments idle (scalar done = 2) until the next reset launch
    uint32_t tick;                // host step counter: Philox tick of this launch (all environments step together); with a
                                  // device-resident counter: this launch's offset from it (steps since the last auto-reset launch)
    uint32_t tick_advance;        // device-resident counter: what the auto-reset launch adds to it (steps per reset interval)
    int32_t stagger;              // per-phase wave priorities (see phase_prio)
    const double2 *cam_grid, *tgt_grid;   // normalised discrete-action grids (or NULL)
    int32_t n_cam_grid, n_tgt_grid;
    int32_t act_discrete;         // bit 0 camera actions, bit 1 target actions are int32 grid indices
    int32_t obs_mode;             // bits 0-1 camera team, bits 2-3 target team: 0 plain, 1 EnhancedObservation, 2 SharedFieldOfView
    int32_t rotate_prio;          // rollout kernel: rotate the wave priorities (fair SIMD shares, see rollout_kernel)
    int32_t store_shifted;        // row-image rollouts: the line-aligned form of the row stores (image_store_form; mate_engine_set_store_form)
    // Pipelined restarts of the fused Greedy rollouts (mate_engine_rollout_greedy, auto_reset = MATE_RESET_PIPELINED): the reset of
    // what launch n finished runs on a side stream UNDER launch n + 1, and a restarted environment joins launch n + 2.  The
    // record's `done` word carries the hand-over: kDoneTag | parity << 3 = "restarted, live from the next launch of this list
    // parity on"; any launch of the other parity -- the one the reset runs under -- leaves such an environment alone (no step, no
    // store), so what a launch does never depends on how far the concurrent reset has come.
    int32_t pipelined;
    // The sub-wave rollout kernel as ONE step of the per-step flows (mate_engine_step / _step_random of the small scenarios, launch_step): step()'s
    // semantics instead of a rollout's -- a finished environment idles only under a batched restart (freeze_done), the tick and the list parity
    // may live on the device (mate_engine_device_tick).
    int32_t per_step;
};
constexpr int32_t kDoneTag = 4;

#ifdef MATE_PHASE_CLOCKS
#define PHASE_STAMP(i) do { if (lane == 0 && g.phase_clocks) g.phase_clocks[env * kClockStride + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define SUB_STAMP(c, i) do { if ((c).lane == 0 && (c).g.phase_clocks) (c).g.phase_clocks[(c).env * kClockStride + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#elif defined(MATE_ISA_MARKS)      // tools/isa_phases.py: phase boundaries of step_kernel as comments in the -S output (no instruction is emitted)
#define PHASE_STAMP(i) asm volatile("; ==== MATE_STEP_PHASE " #i)
#define SUB_STAMP(c, i) asm volatile("; ==== MATE_STEP_SUB " #i)
#else
#define PHASE_STAMP(i) do { } while (0)
#define SUB_STAMP(c, i) do { } while (0)
#endif
#ifdef MATE_SUB_CLOCKS
constexpr int kClockStride = 32;      // stamps per environment in Ptrs::phase_clocks
#else
constexpr int kClockStride = 16;
#endif
// Sub-phase accumulators of the fused rollout (python -m mate_amd.build --variant sub -DMATE_PHASE_CLOCKS -DMATE_SUB_CLOCKS;
// tools/archive/rollout_subphases.py): cycles summed over a launch's steps into Ctx::sub[0..7] (sub[15]: the previous stamp), events counted
// in the stamp buffer itself.
#if defined(MATE_PHASE_CLOCKS) && defined(MATE_SUB_CLOCKS)
#define SUB_ACC(c, i) do { if ((c).sub) { const long long t_sub = (long long)__builtin_amdgcn_s_memtime(); (c).sub[i] += t_sub - (c).sub[15]; (c).sub[15] = t_sub; } } while (0)
// (slots 24..31 of the environment's 32: callable under divergent control flow)
#define SUB_COUNT(c, i, cond) do { const unsigned long long b_sub = __ballot(cond); \
    if ((c).sub && b_sub != 0ull && (c).lane == __ffsll((long long)b_sub) - 1 && (c).g.phase_clocks) \
        atomicAdd(reinterpret_cast<unsigned long long *>((c).g.phase_clocks) + (c).env * kClockStride + 24 + (i), 1ull); } while (0)
#else
#define SUB_ACC(c, i) do { } while (0)
#define SUB_COUNT(c, i, cond) do { } while (0)
#endif
#define SUB_START(c) do { if ((c).sub) (c).sub[15] = (long long)__builtin_amdgcn_s_memtime(); } while (0)

// Phase-keyed issue priority.  The SIMD arbiter serves its oldest wave first, so the four co-resident
// environment-waves of a SIMD finish one after the other and the last one runs its tail alone, with nobody
// to hide its LDS/HBM round trips (measured: wave lifetimes 27k..41k cycles by wave slot).  Lowering a
// wave's priority as it advances lets the ones behind catch up: lifetimes 30k..34k, kernel -6 %.
// One priority (0-3) per phase boundary (start, after load/draws, before view, before goals, before pack);
// the host packs them from the decimal digits of MATE_STAGGER (default 33210, measured best; 0 disables).
__device__ __forceinline__ void phase_prio(int mode, int phase) {
    if (mode == 0) return;                              // host-packed: bit 30 = enabled, 2 bits per phase
    const int k = (mode >> (2 * phase)) & 3;
    if (k == 0) __builtin_amdgcn_s_setprio(0);
    else if (k == 1) __builtin_amdgcn_s_setprio(1);
    else if (k == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

// `g` as it lies in the kernel-argument segment (second argument, after the 8-byte `pp`).
__device__ __forceinline__ const Ptrs &kernarg_ptrs(const Ptrs &g) {
    static_assert(alignof(Ptrs) == 8 && sizeof(const Params *) == 8, "g follows pp at byte 8 of the kernel arguments");
#if defined(__HIP_DEVICE_COMPILE__)
    (void)g;
    return *(const Ptrs *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + 8);
#else
    return g;
#endif
}

// Wave-level LDS hand-off: all 64 lanes run in lockstep, so draining this wave's LDS queue and
// pinning the compiler's order is a complete producer->consumer fence inside the wave.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Launch-flag specialisation of the step kernel.  A launch carries a dozen wave-uniform switches (policy mode, action
// type, tapes, observation post-processing, optional outputs); read on demand each costs a scalar load, a wait and a
// branch on the critical path of every wave.  The two flows every training / benchmark loop runs are compiled with the
// switches fixed (the host picks the kernel per launch, launch_step in mate_engine.hip); FLOW_ANY reads them all.
enum Flow : int {
    FLOW_ANY = 0,
    FLOW_RANDOM = 1,    // step_random: on-device uniform policy, Philox draws, plain observations, all outputs present
    FLOW_ACT_F32 = 2,   // step: real-valued joint actions (f32 or f64, per team: Ptrs::act_f64), Philox draws, plain observations, all outputs present
    FLOW_GREEDY = 3,    // rollout_greedy_kernel: joint actions of the on-device Greedy agents, handed over in LDS
    FLOW_STEP_GREEDY = 4,   // step_greedy_kernel (policy_kernels.hpp): ONE (agents act, environment steps) iteration per launch with the per-step flows' semantics
};

// `L`: LANES PER ENVIRONMENT.  64 = one wave per environment, the mapping of everything above; 32 / 16 = two / four environments
// per wave (round 6, the small scenarios: MATE-2v4-0 has 6 agents and 12 + 24 visibility pairs, so a whole wave spent its
// life on a handful of live lanes while every vector instruction cost its four cycles).  A sub-wave group of L lanes owns one
// environment: `lane` is the lane INSIDE the group, `shift` the group's first hardware lane, every pointer of the context is the
// group's own LDS slice (a per-lane value: the compiler keeps one base VGPR and folds the field offsets into the DS
// instructions), `ballot()` returns the group's own L bits, and whatever was wave-uniform per environment (episode over, a
// target in a warehouse) is merely group-uniform: a branch on it diverges between the groups of a wave and the compiler's
// EXEC masking does the rest.  The phase functions that take `Ctx<ObsT, L>` are written for any L; the register-resident
// variants (held roles, the row image, the held state, the compacted sector list) exist for L = 64 only.
template <typename ObsT, int L = 64>
struct Ctx {
    static_assert(L == 64 || L == 32 || L == 16 || L == 8, "lanes per environment: a wave, a half, a quarter, an eighth");
    static constexpr int W = L;
    const Params &p;
    const Ptrs &g;
    const int flow;               // a compile-time constant of the kernel instantiation (folds once Ctx is scalarised)
    int lane;                     // lane inside the environment's group, [0, L)
    int shift = 0;                // the group's first hardware lane (0 when L == 64)
    int64_t env;
    int64_t out;                  // row of this environment in the output buffers (env, or step*N + env in rollouts)
    double *st, *dy, *tmp;
    int32_t *di;
    ObsT *scratch;
    uint32_t *mask;
    int32_t *misc;
    const uint32_t *table;
    unsigned char *base;
    double *ex, *ey, *er;         // unified entity table [cameras | obstacles | targets]: x, y, radius
    float *exf, *eyf, *erf;       // ... rounded to f32: the screens that only have to be conservative, or that fall back to
                                  // the f64 table inside an explicit error rim, read these (an f64 operation costs two f32 issue slots)
    const double *act_cam = nullptr, *act_tgt = nullptr;   // FLOW_GREEDY: this step's joint actions in LDS ([Nc][2], [Nt][2])
    float *pub = nullptr, *img = nullptr;                  // row-image mode: public-state table, observation rows (Params::off_pub / off_img)

    long long *sub = nullptr;     // (profiling variant: SUB_ACC / SUB_COUNT)
    bool pivots = true;           // sector_resolve: use the pivot angles of an overflow record (kernels at their register limit -- the generic
                                  // fused rollouts, the fused Greedy rollouts -- take the two-trip quarter path instead: same bracket, same limit)
    bool statics_done = false;    // fused rollouts, steps after the first: what never changes inside an episode (static
                                  // mask words and flags, obstacle / capacity slots of the scratch) is in LDS already

    __device__ __forceinline__ Ctx(const Params &p_, const Ptrs &g_, unsigned char *wave_base, int lane_, int64_t env_, int flow_ = FLOW_ANY)
        : p(p_), g(g_), flow(flow_), lane(lane_), env(env_), out(env_) {
        st = reinterpret_cast<double *>(wave_base + p.off_st);
        dy = reinterpret_cast<double *>(wave_base + p.off_dy);
        di = reinterpret_cast<int32_t *>(dy + p.DF);
        tmp = reinterpret_cast<double *>(wave_base + p.off_tmp);
        scratch = reinterpret_cast<ObsT *>(wave_base + p.off_scratch);
        mask = reinterpret_cast<uint32_t *>(wave_base + p.off_mask);
        misc = reinterpret_cast<int32_t *>(wave_base + p.off_misc);
        table = g_.desc;   // read through L1: every wave on the CU gathers through the same 6 KB table
        base = wave_base;
        ex = reinterpret_cast<double *>(wave_base + p.off_ent); ey = ex + p.NJ; er = ey + p.NJ;
        exf = reinterpret_cast<float *>(er + p.NJ); eyf = exf + p.NJ; erf = eyf + p.NJ;
        pub = reinterpret_cast<float *>(wave_base + p.off_pub); img = reinterpret_cast<float *>(wave_base + p.off_img);
    }
    // row-image mode (a compile-time constant of the shape-specialised fused rollouts; zero in the device record otherwise)
    __device__ __forceinline__ bool image() const { return L == 64 && p.image != 0; }
    // the verdicts of THIS environment's lanes: bit i = lane i of the group
    __device__ __forceinline__ unsigned long long ballot(bool x) const {
        if constexpr (L == 64) return __ballot(x);
        else return (__ballot(x) >> shift) & ((1ull << L) - 1ull);
    }
    // L verdicts into the packed mask words at bit `bit` (a multiple of L) -- by the group's lane 0
    __device__ __forceinline__ void put_bits(int bit, unsigned long long b) const {
        if constexpr (L == 64) { if (lane == 0) { mask[bit >> 5] = (uint32_t)b; mask[(bit >> 5) + 1] = (uint32_t)(b >> 32); } }
        else if (lane == 0) {
            if constexpr (L == 32) mask[bit >> 5] = (uint32_t)b;
            else if constexpr (L == 16) reinterpret_cast<uint16_t *>(mask)[bit >> 4] = (uint16_t)b;
            else reinterpret_cast<uint8_t *>(mask)[bit >> 3] = (uint8_t)b;
        }
    }
    // rounds of L pairs that cover n pairs (Params::sector_rounds / range_rounds count rounds of 64: they lay out the mask BITS)
    __device__ __forceinline__ static constexpr int rounds_of(int n) { return (n + L - 1) / L; }
    __device__ __forceinline__ float *pub_cam(int c) const { return pub + 8 * c; }
    __device__ __forceinline__ float *pub_tgt(int t) const { return pub + 8 * p.Nc + 8 * t; }
    __device__ __forceinline__ float *pub_obs(int o) const { return pub + 8 * (p.Nc + p.Nt) + 4 * o; }
    __device__ __forceinline__ float *img_cam_row(int c) const { return img + c * p.Dc; }
    __device__ __forceinline__ float *img_tgt_row(int t) const { return img + p.cam_elems + t * p.Dt; }
    // which of the two finished-episode lists this launch appends to: a launch argument, or the low bit of the device-resident
    // reset-launch counter (Params::dev_group).  Re-derived where it is used (an episode ends: rare) instead of held in an
    // SGPR for the whole kernel -- the step kernels live at the 96-SGPR limit of full occupancy.
    // (one of the two summands is always zero: the host keeps dev_group = 0 while it counts itself and passes parity = 0 otherwise)
    __device__ __forceinline__ int32_t list_parity() const { return (int32_t)((p.dev_group + (uint32_t)g.parity) & 1u); }
    // launch switches (constants in the specialised flows)
    __device__ __forceinline__ int mode() const { return flow == FLOW_RANDOM ? (int)MODE_STEP_RANDOM : (flow == FLOW_ACT_F32 || flow == FLOW_GREEDY || flow == FLOW_STEP_GREEDY) ? (int)MODE_STEP : g.mode; }
    __device__ __forceinline__ bool act_from_lds() const { return flow == FLOW_GREEDY || flow == FLOW_STEP_GREEDY; }
    // FLOW_ACT_F32 (the single-step kernels): an agent's lane fetches its own action together with the records, into the StepDraws
    // the kinematics read anyway (prefetch_action) -- read where it is used, behind the LDS commit, the load's round trip to
    // L2 / HBM lay bare on every wave's critical path (1.4 us of a 14 us step)
    __device__ __forceinline__ bool act_prefetched() const { return flow == FLOW_ACT_F32; }
    // camera->target pairs whose transmittance draw step_draws makes ahead of the visibility phase: one lane each -- the lanes
    // behind the agents', and the agents' own wherever those draw no actions (every flow but the on-device random policy)
    __device__ __forceinline__ int predrawn_pairs() const { return mode() == MODE_STEP_RANDOM ? L - p.Nc - p.Nt : L; }
    __device__ __forceinline__ int act_f64() const { return g.act_f64; }     // (a launch argument in every flow: f32 and f64 joint actions, per team, run the specialised kernel)
    __device__ __forceinline__ int act_discrete() const { return flow != FLOW_ANY ? 0 : g.act_discrete; }
    __device__ __forceinline__ const double *tape_ct() const { return flow != FLOW_ANY ? nullptr : g.tape_ct; }
    __device__ __forceinline__ const double *tape_goal() const { return flow != FLOW_ANY ? nullptr : g.tape_goal; }
    __device__ __forceinline__ int obs_mode() const { return flow != FLOW_ANY ? 0 : g.obs_mode; }
    __device__ __forceinline__ const uint2 *xdesc() const { return flow != FLOW_ANY ? nullptr : g.xdesc; }
    __device__ __forceinline__ bool freeze_done() const { return (flow == FLOW_GREEDY || flow == FLOW_STEP_GREEDY) ? false : g.freeze_done != 0; }
    __device__ __forceinline__ bool has_scratch_init() const { return flow != FLOW_ANY ? true : g.scratch_init != nullptr; }
    __device__ __forceinline__ bool has_cam_obs() const { return flow != FLOW_ANY ? true : g.cam_obs != nullptr; }
    __device__ __forceinline__ bool has_tgt_obs() const { return flow != FLOW_ANY ? true : g.tgt_obs != nullptr; }
    __device__ __forceinline__ bool has_scalars() const { return flow != FLOW_ANY ? true : g.scalars != nullptr; }
    // static record
    __device__ __forceinline__ double cam_x(int c) const { return st[c]; }
    __device__ __forceinline__ double cam_y(int c) const { return st[p.Nc + c]; }
    __device__ __forceinline__ double obs_x(int o) const { return st[2 * p.Nc + o]; }
    __device__ __forceinline__ double obs_y(int o) const { return st[2 * p.Nc + p.No + o]; }
    __device__ __forceinline__ double obs_r(int o) const { return st[2 * p.Nc + 2 * p.No + o]; }
    __device__ __forceinline__ uint64_t camobs(int c) const { return reinterpret_cast<const uint64_t *>(st)[2 * p.Nc + 3 * p.No + c]; }
    __device__ __forceinline__ uint64_t capword() const { return reinterpret_cast<const uint64_t *>(st)[3 * p.Nc + 3 * p.No]; }
    __device__ __forceinline__ void circle(int k, double &x, double &y, double &r) const {  // obstacles, then cameras (Target.add_obstacles, environment.py:743)
        const int j = k < p.No ? p.Nc + k : k - p.No;
        x = ex[j]; y = ey[j]; r = er[j];
    }
    __device__ __forceinline__ int tgt_slot(int t) const { return p.Nc + p.No + t; }
    // dynamic record
    __device__ __forceinline__ double &phi(int c) { return dy[c]; }
    __device__ __forceinline__ double &theta(int c) { return dy[p.Nc + c]; }
    __device__ __forceinline__ double &tx(int t) { return dy[2 * p.Nc + t]; }
    __device__ __forceinline__ double &ty(int t) { return dy[2 * p.Nc + p.Nt + t]; }
    __device__ __forceinline__ double &ep_reward() { return dy[2 * p.Nc + 2 * p.Nt]; }
    __device__ __forceinline__ double &ep_delayed() { return dy[2 * p.Nc + 2 * p.Nt + 1]; }
    __device__ __forceinline__ int32_t &ti(int t, int f) { return di[t * TI_STRIDE + f]; }
    __device__ __forceinline__ int32_t &ei(int f) { return di[p.Nt * TI_STRIDE + f]; }
    // temporaries
    __device__ __forceinline__ double &sight2(int c) { return tmp[c]; }          // a camera's SQUARED sight range: area / viewing angle
    __device__ __forceinline__ double &snorm(int t) { return tmp[p.Nc + t]; }
    __device__ __forceinline__ double &udraw(int pair) { return tmp[p.Nc + p.Nt + pair]; }
    __device__ __forceinline__ int32_t &near(int t) { return misc[t]; }
    __device__ __forceinline__ int32_t &inside(int t) { return misc[2 * p.Nt + t]; }
    __device__ __forceinline__ int32_t &tracked(int t) { return misc[3 * p.Nt + t]; }
    __device__ __forceinline__ int32_t &xch(int i) { return misc[4 * p.Nt + i]; }
    __device__ __forceinline__ bool mask_bit(int b) const { return (mask[b >> 5] >> (b & 31)) & 1u; }
    __device__ __forceinline__ uint32_t env_global() const { return p.first_env + (uint32_t)env; }
    // A tick-keyed draw.  One Philox-4x32 block holds two 64-bit draws: ticks 2k and 2k + 1 share the block with counter k and
    // take its first / second half -- the fused rollouts compute a block once per TWO steps (DrawCarry).
    __device__ __forceinline__ double draw(uint32_t tick, uint32_t stream, uint32_t sub) const {
        const U4 r = philox(p.seed_lo, p.seed_hi, env_global(), tick >> 1, stream, sub);
        return (tick & 1u) ? u53(r.z, r.w) : u53(r.x, r.y);
    }
};

template <typename ObsT> struct Bits;
template <> struct Bits<float> { using type = uint32_t; };
template <> struct Bits<double> { using type = uint64_t; };
template <typename ObsT, int L>
__device__ __forceinline__ void set_flag(const Ctx<ObsT, L> &c, int bit, bool on) {
    using U = typename Bits<ObsT>::type;
    reinterpret_cast<U *>(c.base + c.p.off_flags)[bit] = on ? ~(U)0 : (U)0;
}

// ---------------------------------------------------------------------------------------------
// record <-> LDS
template <typename ObsT, int L>
__device__ __forceinline__ void load_records(Ctx<ObsT, L> &c) {
    const double *s = c.g.stat + c.env * c.p.SW;
    const double *d = c.g.dyn + c.env * c.p.DW;
    for (int i = c.lane; i < c.p.SW; i += L) c.st[i] = s[i];
    for (int i = c.lane; i < c.p.DW; i += L) c.dy[i] = d[i];
    if (c.has_scratch_init() && !c.image()) {
        const ObsT *si = reinterpret_cast<const ObsT *>(c.g.scratch_init);
        for (int i = c.lane; i < c.p.nscratch; i += L) c.scratch[i] = si[i];
    }
    for (int i = c.lane; i < c.p.MW; i += L) c.mask[i] = 0u;
}

// The step kernel's version: the record loads are issued into registers first, the step's Philox draws run
// while they are in flight (all co-resident waves of a SIMD start together: without this they idle through the
// HBM latency together and then contend for the VALU together), then the data is committed to LDS.
struct StepDraws { double a0, a1; };
constexpr int kNearWords = 5;                     // (= kRoleRounds, declared further down with the range-test roles)
struct DrawCarry { uint32_t z, w, block; };      // second half of the lane's Philox block of the even tick, for the odd tick behind it
// What a lane draws is the same at every step of a launch: its stream and index, whether it draws at all, and -- an agent -- the
// bounds of its two action components.  The fused rollout derives it once and holds it (22 vector instructions per step).
struct DrawRole { uint32_t stream, sub; int32_t kind; double m0, m1; };      // kind: 0 none, 1 agent (action sample), 2 pair (transmittance draw)
template <typename ObsT, int L> __device__ StepDraws step_draws(Ctx<ObsT, L> &c, uint32_t tick, DrawCarry *carry = nullptr, const DrawRole *held = nullptr);

// The caller's joint action of this lane's agent (lanes [0, Nc): cameras, [Nc, Nc + Nt): targets), f32 or f64 per team, as issued loads
template <typename ObsT, int L>
__device__ __forceinline__ StepDraws prefetch_action(const Ctx<ObsT, L> &c) {
    const Params &p = c.p;
    const int lane = c.lane, t = lane - p.Nc;
    StepDraws a{0.0, 0.0};
    if (lane < p.Nc) {
        if (c.act_f64() & 1) { const double *q = reinterpret_cast<const double *>(c.g.cam_act) + (c.env * p.Nc + lane) * 2; a.a0 = q[0]; a.a1 = q[1]; }
        else { const float2 q = reinterpret_cast<const float2 *>(c.g.cam_act)[c.env * p.Nc + lane]; a.a0 = (double)q.x; a.a1 = (double)q.y; }
    } else if (t < p.Nt) {
        if (c.act_f64() & 2) { const double *q = reinterpret_cast<const double *>(c.g.tgt_act) + (c.env * p.Nt + t) * 2; a.a0 = q[0]; a.a1 = q[1]; }
        else { const float2 q = reinterpret_cast<const float2 *>(c.g.tgt_act)[c.env * p.Nt + t]; a.a0 = (double)q.x; a.a1 = (double)q.y; }
    }
    return a;
}

template <typename ObsT>
__device__ __forceinline__ StepDraws load_records_with_draws(Ctx<ObsT> &c, uint32_t tick, bool draw) {
    const Params &p = c.p;
    const int lane = c.lane;
    const double *s = c.g.stat + c.env * p.SW;
    const double *d = c.g.dyn + c.env * p.DW;
    const ObsT *si = c.has_scratch_init() ? reinterpret_cast<const ObsT *>(c.g.scratch_init) : nullptr;
    double s0 = s[lane < p.SW ? lane : 0], d0 = d[lane < p.DW ? lane : 0], s1 = 0.0, d1 = 0.0;
    if (p.SW > 64) s1 = s[lane + 64 < p.SW ? lane + 64 : 0];
    if (p.DW > 64) d1 = d[lane + 64 < p.DW ? lane + 64 : 0];
    ObsT q[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (si && p.nscratch > 64 * k) q[k] = si[lane + 64 * k < p.nscratch ? lane + 64 * k : 0];
    asm volatile("" : "+v"(s0), "+v"(s1), "+v"(d0), "+v"(d1));     // keep the loads up here: the optimiser would sink each into its use
    asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
    StepDraws draws{0.0, 0.0};
    if (c.act_prefetched()) {       // (the agents' lanes draw nothing of their own in this flow: their draws carry the caller's action)
        StepDraws act = prefetch_action(c);
        asm volatile("" : "+v"(act.a0), "+v"(act.a1));
        if (draw) (void)step_draws(c, tick);
        draws = act;
    } else
    if (draw) draws = step_draws(c, tick);
    if (lane < p.SW) c.st[lane] = s0;
    if (lane < p.DW) c.dy[lane] = d0;
    if (lane + 64 < p.SW) c.st[lane + 64] = s1;
    if (lane + 64 < p.DW) c.dy[lane + 64] = d1;
    for (int i = lane + 128; i < p.SW; i += 64) c.st[i] = s[i];
    for (int i = lane + 128; i < p.DW; i += 64) c.dy[i] = d[i];
    if (si) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (lane + 64 * k < p.nscratch) c.scratch[lane + 64 * k] = q[k];
        for (int i = lane + 256; i < p.nscratch; i += 64) c.scratch[i] = si[i];
    }
    for (int i = lane; i < p.MW; i += 64) c.mask[i] = 0u;
    return draws;
}

// unified entity table (after the records are visible in LDS)
template <typename ObsT, int L>
__device__ __forceinline__ void build_entities(Ctx<ObsT, L> &c) {
    const Params &p = c.p;
    for (int j = c.lane; j < p.NJ; j += L) {
        double x, y, r;
        if (j < p.Nc) { x = c.cam_x(j); y = c.cam_y(j); r = p.cam_radius; }
        else if (j < p.Nc + p.No) { const int o = j - p.Nc; x = c.obs_x(o); y = c.obs_y(o); r = c.obs_r(o); }
        else { const int t = j - p.Nc - p.No; x = c.tx(t); y = c.ty(t); r = 0.0; }
        c.ex[j] = x; c.ey[j] = y; c.er[j] = r;
        c.exf[j] = (float)x; c.eyf[j] = (float)y; c.erf[j] = (float)r;
    }
}

template <typename ObsT, int L>
__device__ __forceinline__ void store_dynamic(Ctx<ObsT, L> &c) {
    double *d = c.g.dyn + c.env * c.p.DW;
    for (int i = c.lane; i < c.p.DW; i += L) d[i] = c.dy[i];
}

// ---------------------------------------------------------------------------------------------
// Random numbers of one step, one Philox call per lane, all lanes at once: lanes [0, Nc) camera
// actions, [Nc, Nc+Nt) target actions, the remaining lanes pre-draw the see-through uniforms of the
// first camera->target pairs (cheaper than a second divergent Philox in the visibility phase).
template <typename ObsT, int L>
__device__ __forceinline__ DrawRole draw_role(const Ctx<ObsT, L> &c) {
    const Params &p = c.p;
    const int lane = c.lane, nact = p.Nc + p.Nt;
    const bool random_policy = c.mode() == MODE_STEP_RANDOM;
    const bool need_draws = !c.tape_ct() && p.Nc > 0 && p.No > 0;      // (without obstacles nothing is ever seen THROUGH one)
    DrawRole r{0u, 0u, 0, 0.0, 0.0};
    if (lane < nact && !random_policy) {       // an agent's lane with no action to draw: the pair behind the last pair lane's (predrawn_pairs)
        const int pair = L - nact + lane;
        if (pair < p.Nc * p.Nt) { r.stream = S_TRANSMIT; r.sub = (uint32_t)pair; r.kind = need_draws ? 2 : 0; }
    }
    else if (lane < p.Nc) { r.stream = S_ACT_CAM; r.sub = (uint32_t)lane; r.kind = 1; r.m0 = p.rot; r.m1 = p.zoom; }
    else if (lane < nact) { r.stream = S_ACT_TGT; r.sub = (uint32_t)(lane - p.Nc); r.kind = 1; r.m0 = p.tgt_step; r.m1 = p.tgt_step; }
    else if (lane - nact < p.Nc * p.Nt) { r.stream = S_TRANSMIT; r.sub = (uint32_t)(lane - nact); r.kind = need_draws ? 2 : 0; }
    return r;
}
__device__ __forceinline__ void pin_draw_role(DrawRole &r) {
    asm volatile("" : "+v"(r.stream)); asm volatile("" : "+v"(r.sub)); asm volatile("" : "+v"(r.kind)); asm volatile("" : "+v"(r.m0)); asm volatile("" : "+v"(r.m1));
}
template <typename ObsT, int L>
__device__ __forceinline__ StepDraws step_draws(Ctx<ObsT, L> &c, uint32_t tick, DrawCarry *carry, const DrawRole *held) {
    const Params &p = c.p;
    const int lane = c.lane;
    StepDraws d{0.0, 0.0};
    const int nact = p.Nc + p.Nt;
    const DrawRole role = held ? *held : draw_role(c);
    const uint32_t stream = role.stream, sub = role.sub;
    const bool active = role.kind != 0;
    // (a lane's role -- stream, sub -- is the same at every step of a launch, so the words it carries are its own)
    const uint32_t block = tick >> 1;
    uint32_t w0 = 0, w1 = 0;
    if (carry && (tick & 1u) && carry->block == block) { w0 = carry->z; w1 = carry->w; }        // wave-uniform: no Philox at all on this step
    else if (active) {
        const U4 r = philox(p.seed_lo, p.seed_hi, c.env_global(), block, stream, sub);
        if (tick & 1u) { w0 = r.z; w1 = r.w; }
        else { w0 = r.x; w1 = r.y; if (carry) { carry->z = r.z; carry->w = r.w; } }
    }
    if (carry && !(tick & 1u)) carry->block = block;
    if (active) {
        if (role.kind == 1) { d.a0 = action_component(w0, role.m0); d.a1 = action_component(w1, role.m1); }      // one instruction stream for both kinds of agent
        else c.udraw((int)sub) = u53(w0, w1);
    }
    return d;
}

// Phase A: kinematics.  Camera.simulate (entities.py:347-360), Target.simulate (entities.py:645-668).
template <typename ObsT, int L>
__device__ __forceinline__ void simulate_cameras(Ctx<ObsT, L> &c, const StepDraws &draws, bool advance) {
    const Params &p = c.p;
    const int lane = c.lane;
    if (lane < p.Nc) {
        double ph = c.phi(lane), th = c.theta(lane);
        if (advance) {
            double da, dz;
            if (c.mode() == MODE_STEP_RANDOM || c.act_prefetched()) { da = draws.a0; dz = draws.a1; }
            else if (c.act_from_lds()) { da = c.act_cam[2 * lane]; dz = c.act_cam[2 * lane + 1]; }
            else if (c.act_discrete() & 1) {                 // DiscreteCamera.action, discrete_action_spaces.py:71-73
                int idx = reinterpret_cast<const int32_t *>(c.g.cam_act)[c.env * p.Nc + lane];
                idx = idx < 0 ? 0 : (idx >= c.g.n_cam_grid ? c.g.n_cam_grid - 1 : idx);
                const double2 gxy = c.g.cam_grid[idx];
                da = p.rot * gxy.x; dz = p.zoom * gxy.y;
            } else if (c.act_f64() & 1) {
                const double *a = reinterpret_cast<const double *>(c.g.cam_act) + (c.env * p.Nc + lane) * 2;
                da = a[0]; dz = a[1];
            } else {
                const float2 a = reinterpret_cast<const float2 *>(c.g.cam_act)[c.env * p.Nc + lane];
                da = (double)a.x; dz = (double)a.y;
            }
            da = clip_uniform(da, -p.rot, p.rot);
            dz = clip_uniform(dz, -p.zoom, p.zoom);
            ph = normalize_angle(ph + da);
            th = clip_uniform(th + dz, p.theta_min, kMaxViewingAngle);
            c.phi(lane) = ph; c.theta(lane) = th;
        }
        // sight_range = sqrt(area / viewing_angle) (entities.py:360).  The visibility phase compares SQUARED distances with
        // the quotient and takes square roots only inside the rounding rim (sector_eval), so the f64 root is needed for
        // the f64 observation mirror only; f32 observations take an f32 root of the quotient (relative error 1e-7).
        const double sr2 = div_nz(p.area, th);
        c.sight2(lane) = sr2;
        ObsT *sc = c.scratch + p.sc_cam + lane * 10;
        if (!c.statics_done && !c.image()) { sc[0] = (ObsT)c.cam_x(lane); sc[1] = (ObsT)c.cam_y(lane); }
        // Camera.state: polar2cartesian(sight_range, orientation), entities.py:318
        if constexpr (sizeof(ObsT) == 4) {         // f32 observations: f64 argument reduction, f32 polynomials (+8 % on the fused rollout:
            float sn, cs;                          // the camera lanes' f64 sincos was the longest dependent chain of the phase)
            sincos_deg_f32(ph, sn, cs);
            const float srf = sqrt_f32_1ulp((float)sr2);
            if (c.image()) {                       // the camera's public state (read by every block that shows it) and its own row
                float *pc = c.pub_cam(lane), *row = c.img_cam_row(lane) + 13;
                const float x = srf * cs, y = srf * sn, t = (float)th;
                pc[3] = x; pc[4] = y; pc[5] = t; row[3] = x; row[4] = y; row[5] = t;
            } else {
            sc[3] = srf * cs; sc[4] = srf * sn; sc[5] = (float)th;
            }
        } else {
            const double sr = sqrt_pos(sr2);
            double sn, cs;
            sincos_deg(ph, sn, cs);
            sc[3] = (ObsT)(sr * cs); sc[4] = (ObsT)(sr * sn); sc[5] = (ObsT)th;
        }
    }
}

// Target.simulate on lanes [Nc, Nc+Nt).  Fast path: a (target, circle) pair whose circle provably
// cannot touch the step (entities.py:163: `relative.norm >= norm + radius`) is a no-op, so all pairs are
// screened in parallel with a sqrt-free conservative test; only flagged circles are walked, in index
// order.  A hit never lengthens the step (|v'|^2 = |v|^2 - a s^2 (2|v| - a) <= |v|^2 for penetration a and
// s = half_chord/r), so a circle out of reach of the original step stays out of reach.
// `carried` (the fused rollouts with held lane roles): the screen was made by the PREVIOUS step's range tests -- the
// (target, camera | obstacle) pairs of Sensor.perceive are the pairs of this screen, on the very positions this step starts
// from -- and arrives as the ballots of their rounds (NearCarry, update_view); no screen pass, no LDS hand-off here.
struct NearCarry { unsigned long long w[kNearWords]; };       // bit q = t * NJ + j of the range rounds: circle j may be touched by target t's next step
// target t's NK bits out of the string w3:w2:w1:w0, moved from entity order (cameras, obstacles) to the order the circles are
// walked in (obstacles, then cameras: Target.add_obstacles, environment.py:743)
__device__ __forceinline__ uint64_t near_field(const Params &p, const NearCarry &carried, int t) {
    // four steps in five no target of the environment is within a step of any circle: wave-uniform, three scalar instructions
    // instead of the 25 vector ones that cut a lane's bits out of the words
    unsigned long long any = 0ull;
#pragma unroll
    for (int k = 0; k < kNearWords; ++k) any |= carried.w[k];
    if (any == 0ull) return 0ull;
    const int first = t * p.NJ, word = first >> 6, sh = first & 63;
    unsigned long long lo = carried.w[0], hi = carried.w[1];
#pragma unroll
    for (int k = 1; k < kNearWords; ++k)
        if (word == k) { lo = carried.w[k]; hi = k + 1 < kNearWords ? carried.w[k + 1] : 0ull; }
    const unsigned long long field = ((lo >> sh) | (sh ? hi << (64 - sh) : 0ull)) & ((1ull << p.NK) - 1ull);
    return (field >> p.Nc) | ((field & ((1ull << p.Nc) - 1ull)) << p.No);
}
// `collide_word` (the two-wave step, whose target wave does not own the targets' integer words): the colliding bits of all
// targets as one word, bit t, instead of bit 24 of each TI_GW.
template <typename ObsT, int L>
__device__ __forceinline__ void simulate_targets(Ctx<ObsT, L> &c, const StepDraws &draws, const NearCarry *carried = nullptr, int32_t *collide_word = nullptr) {
    const Params &p = c.p;
    const int lane = c.lane;
    const int t = lane - p.Nc;
    const bool is_target = t >= 0 && t < p.Nt;
    // (sub-wave groups, L < 64: the verdicts of all rounds in ONE 64-bit word -- every such shape has at most 64 (target, circle) pairs)
    const bool ballot_screen = L == 64 && !carried && p.Nt * p.NK <= 64 * kNearWords && p.NK <= 64;
    const bool group_screen = L < 64 && !carried && p.Nt * p.NK <= 64;
    double ox = 0.0, oy = 0.0, vx = 0.0, vy = 0.0, n = 0.0, desx = 0.0, desy = 0.0;
    if (is_target) {
        double ax, ay;
        if (c.mode() == MODE_STEP_RANDOM || c.act_prefetched()) { ax = draws.a0; ay = draws.a1; }
        else if (c.act_from_lds()) { ax = c.act_tgt[2 * t]; ay = c.act_tgt[2 * t + 1]; }
        else if (c.act_discrete() & 2) {                     // DiscreteTarget.action, discrete_action_spaces.py:177-179
            int idx = reinterpret_cast<const int32_t *>(c.g.tgt_act)[c.env * p.Nt + t];
            idx = idx < 0 ? 0 : (idx >= c.g.n_tgt_grid ? c.g.n_tgt_grid - 1 : idx);
            const double2 gxy = c.g.tgt_grid[idx];
            const double high = ((c.capword() >> t) & 1ull) ? p.tgt_step * 0.5 : p.tgt_step;
            ax = high * gxy.x; ay = high * gxy.y;
        } else if (c.act_f64() & 2) {
            const double *a = reinterpret_cast<const double *>(c.g.tgt_act) + (c.env * p.Nt + t) * 2;
            ax = a[0]; ay = a[1];
        } else {
            const float2 a = reinterpret_cast<const float2 *>(c.g.tgt_act)[c.env * p.Nt + t];
            ax = (double)a.x; ay = (double)a.y;
        }
        const double step_size = ((c.capword() >> t) & 1ull) ? p.tgt_step * 0.5 : p.tgt_step;   // entities.py:612-615
        ox = c.tx(t); oy = c.ty(t);
        vx = ax; vy = ay;
        n = norm2(ax, ay);
        if (n > step_size) {
            // `step.norm = step_size` (entities.py:649-650) goes through the polar form in the reference
            // (s*(cos, sin) of atan2(a)); rescaling the vector is the same quantity to within the
            // last-place noise a device atan2/sincos would add anyway, at a tenth of the instructions.
#ifdef MATE_POLAR_CLAMP
            set_norm_polar(vx, vy, step_size); n = step_size;
#else
            const double k = div_nz(step_size, n);
            vx = ax * k; vy = ay * k; n = step_size;
#endif
        }
        desx = ox + vx; desy = oy + vy;
        if (!carried) { c.snorm(t) = n; if (!ballot_screen && !group_screen) { c.near(t) = 0; c.near(p.Nt + t) = 0; } }
    }
    uint64_t todo_carried = 0;
    if (carried) {
        if (is_target) {
            // the target's NK bits out of the 192-bit string w2:w1:w0, then from entity order (cameras, obstacles) to the
            // order the circles are walked in (obstacles, cameras: Target.add_obstacles, environment.py:743)
            todo_carried = near_field(p, *carried, t);
        }
    } else if (ballot_screen) {
        // every shipped scenario: the screen's verdicts as ballots (one 64-bit word per round of pairs, in scalar registers), a
        // target's lane cuts its NK bits out of them -- no LDS atomics, no second hand-off
        wave_sync();
        SUB_STAMP(c, 10);
        const int npairs = p.Nt * p.NK;
        unsigned long long hit[kNearWords];
#pragma unroll
        for (int round = 0; round < kNearWords; ++round) {
            hit[round] = 0ull;
            if (round * 64 < npairs) {
                const int q = round * 64 + lane, qq = q < npairs ? q : 0;
                const int tt = (int)(((float)qq + 0.5f) * p.inv_NK);
                const int k = qq - tt * p.NK;
                const int j = k < p.No ? p.Nc + k : k - p.No, tj = c.tgt_slot(tt);
                const float dx = c.exf[j] - c.exf[tj], dy = c.eyf[j] - c.eyf[tj];
                const float d2 = fmaf(dy, dy, dx * dx);
                const float nn = (float)c.snorm(tt);
                const float reach = nn + c.erf[j] + 1e-3f;         // (the same conservative f32 test as the LDS form below)
                hit[round] = __ballot(q < npairs && nn != 0.0f && !(d2 > reach * reach));
            }
        }
        if (is_target) {
            const int first = t * p.NK, word = first >> 6, sh = first & 63;
            unsigned long long lo = hit[0], hi = kNearWords > 1 ? hit[1] : 0ull;
#pragma unroll
            for (int k = 1; k < kNearWords; ++k)
                if (word == k) { lo = hit[k]; hi = k + 1 < kNearWords ? hit[k + 1] : 0ull; }
            todo_carried = ((lo >> sh) | (sh ? hi << (64 - sh) : 0ull)) & (p.NK >= 64 ? ~0ull : ((1ull << p.NK) - 1ull));
        }
    } else if (group_screen) {
        wave_sync();
        const int npairs = p.Nt * p.NK;
        unsigned long long bits = 0ull;
        for (int base = 0; base < npairs; base += L) {
            const int q = base + lane, qq = q < npairs ? q : 0;
            const int tt = (int)(((float)qq + 0.5f) * p.inv_NK);
            const int k = qq - tt * p.NK;
            const int j = k < p.No ? p.Nc + k : k - p.No, tj = c.tgt_slot(tt);
            const float dx = c.exf[j] - c.exf[tj], dy = c.eyf[j] - c.eyf[tj];
            const float d2 = fmaf(dy, dy, dx * dx);
            const float nn = (float)c.snorm(tt);
            const float reach = nn + c.erf[j] + 1e-3f;             // (the same conservative f32 test as the other two forms)
            bits |= c.ballot(q < npairs && nn != 0.0f && !(d2 > reach * reach)) << base;
        }
        if (is_target) todo_carried = (bits >> (t * p.NK)) & ((1ull << p.NK) - 1ull);
    } else {
    wave_sync();
    SUB_STAMP(c, 10);
    const int npairs = p.Nt * p.NK;
    for (int base = 0; base < npairs; base += L) {
        const int q = base + lane;
        if (q < npairs) {
            const int tt = (int)(((float)q + 0.5f) * p.inv_NK);
            const int k = q - tt * p.NK;
            // a SCREEN: it only has to keep every circle the step can touch, so f32 with an absolute margin does (positions
            // rounded to f32 are off by <= 3.1e-5, the distance by <= 1.3e-4, the f32 arithmetic by <= 1e-5: 1e-3 covers it
            // eight times); the walk below repeats the test exactly
            const int j = k < p.No ? p.Nc + k : k - p.No, tj = c.tgt_slot(tt);
            const float dx = c.exf[j] - c.exf[tj], dy = c.eyf[j] - c.eyf[tj];
            const float d2 = fmaf(dy, dy, dx * dx);
            const float nn = (float)c.snorm(tt);
            const float reach = nn + c.erf[j] + 1e-3f;
            const bool far = d2 > reach * reach;
            if (nn != 0.0f && !far) atomicOr(&c.near(k < 32 ? tt : p.Nt + tt), 1 << (k & 31));
        }
    }
    wave_sync();
    }
    SUB_STAMP(c, 11);
    if (is_target) {
        uint64_t todo = (carried || ballot_screen || group_screen) ? todo_carried : ((uint64_t)(uint32_t)c.near(t) | ((uint64_t)(uint32_t)c.near(p.Nt + t) << 32));
        bool n_known = true;
        while (todo) {
            const int k = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            double cx, cy, cr;
            c.circle(k, cx, cy, cr);
            if (!n_known) { n = norm2(vx, vy); n_known = true; }
            // sqrt-free screen against the current step: definitely out of reach -> Obstacle.obstruct returns the ray as is
            const double dx = cx - ox, dy = cy - oy;
            const double reach = n + cr;
            if (n != 0.0 && fma(dy, dy, dx * dx) > reach * reach * (1.0 + 1e-12)) continue;
            obstruct_tangential(ox, oy, vx, vy, n, n_known, cx, cy, cr);
        }
        const double nx = clip_uniform(ox + vx, -kTerrain, kTerrain);   // entities.py:664-666
        const double ny = clip_uniform(oy + vy, -kTerrain, kTerrain);
        const bool colliding = (fabs(nx - desx) > 1e-6) || (fabs(ny - desy) > 1e-6);  // entities.py:668
        c.tx(t) = nx; c.ty(t) = ny;
        c.ex[c.tgt_slot(t)] = nx; c.ey[c.tgt_slot(t)] = ny;
        c.exf[c.tgt_slot(t)] = (float)nx; c.eyf[c.tgt_slot(t)] = (float)ny;
        if (collide_word) {
            const unsigned long long b = c.ballot(colliding);
            if (t == 0) *collide_word = (int32_t)(b >> p.Nc);
        } else {
        int gw = c.ti(t, TI_GW) & ~(1 << 24);
        c.ti(t, TI_GW) = gw | ((int)colliding << 24);
        }
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------
// Occlusion-table lookup: Camera.sight_range_at (entities.py:507-511) == np.interp on the knots.
// General path: bucket[d] = index of the last knot with angle <= d - 180 (every integer degree is a
// knot), then a short search.
__device__ __noinline__ double lut_lookup(const double2 *knots, const uint16_t *bucket, int n, double x) {
    int d = (int)floor(x + 180.0);
    d = d < 1 ? 1 : (d > 359 ? 359 : d);
    int lo = bucket[d - 1];                           // knots[lo].x <= d-1-180 <= x  (x >= -180)
    int hi = bucket[d + 2 > 360 ? 360 : d + 2];       // knots[hi].x = d+2-180 > x
    if (hi > n - 1) hi = n - 1;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (knots[mid].x <= x) lo = mid; else hi = mid; }
    if (knots[hi].x <= x) lo = hi;
    const int j = lo;
    const double2 k0 = knots[j];
    if (j >= n - 1 || k0.x == x) return k0.y;
    const double2 k1 = knots[j + 1];
    const double slope = (k1.y - k0.y) / (k1.x - k0.x);
    double res = slope * (x - k0.x) + k0.y;
    if (res != res) {
        res = slope * (x - k1.x) + k1.y;
        if (res != res && k0.y == k1.y) res = k0.y;
    }
    return res;
}

// Fast path: one 96-byte record per (camera, cell of 1 / kCellsPerDegree degrees): the up to four knots of that cell -- from the last
// one at or below its start on -- as SEGMENTS (angle, range, slope to the next knot -- the last one's to the first knot of the
// next cell), +inf angles in the unused slots: one dependent memory round
// trip, and np.interp (`slope * (x - xp[j]) + fp[j]`, slope = (fp[j+1] - fp[j]) / (xp[j+1] - xp[j])) without its division --
// the builders make it once per table, with the same IEEE division, so the product and the sum see the same bits.  (Angles
// of a table increase strictly, so the slopes are finite and np.interp's NaN repairs never apply; a query below the next
// integer degree -- degree_of guarantees it -- never selects that knot, which is why it needs no slot.)  A degree
// with more knots than fit (the arc of a small, distant obstacle) is marked by a NaN first angle and
// carries (index of its first knot, number of knots incl. the next integer degree) instead: three pivot
// angles pick the quarter of that degree's knots holding the query, and that quarter is fetched into the
// same registers as kDegSlots plain knots (two more round trips, no extra registers; up to 4 (kDegSlots - 1) + 1 knots).
// Only beyond that the general path with its dependent binary search runs (8+ round trips: it used to set
// the slowest wave of a launch).
constexpr int kDegWords = 6;      // double2 per record: [x0 y0][s0 x1][y1 s1][x2 y2][s2 x3][y3 s3]
__device__ __forceinline__ void degree_record_set(double2 *rec, int i, double x, double y, double slope) {
    double *w = reinterpret_cast<double *>(rec) + 3 * i;
    w[0] = x; w[1] = y; w[2] = slope;
}
__device__ __forceinline__ double segment_interp(const double2 (&w)[kDegWords], double x, bool &overflow) {
    overflow = w[0].x != w[0].x;
    // (every word read into a value first: selecting between the loads themselves keeps the record in memory)
    const double xa = w[0].x, ya = w[0].y, sa = w[1].x, xb = w[1].y, yb = w[2].x, sb = w[2].y;
    const double xc = w[3].x, yc = w[3].y, sc = w[4].x, xd = w[4].y, yd = w[5].x, sd = w[5].y;
    double x0 = xa, y0 = ya, sl = sa;
    if (xb <= x) { x0 = xb; y0 = yb; sl = sb; }
    if (xc <= x) { x0 = xc; y0 = yc; sl = sc; }
    if (xd <= x) { x0 = xd; y0 = yd; sl = sd; }
    return sl * (x - x0) + y0;
}
constexpr int kDegSlots = 5;
// Overflow records (first angle NaN): [NaN, first knot] [knot count, pivot stride q] [pivot 1, pivot 2] ... [pivot 7, pivot 8]: eight
// pivot angles = the angles of the degree's knots q, 2q, ... 8q (+inf past the last), q = ceil((count - 1) / 9) <= kDegSlots - 1, so
// that the query's bracket lies among the five knots from (number of pivots <= query) * q on; stride 0 = no pivots (general path).
constexpr int kPivotKnots = 9 * (kDegSlots - 1) + 1;      // 37
__host__ __device__ constexpr int pivot_stride(int count) { return (count - 1 + 8) / 9; }
constexpr int kQuarterKnots = 4 * (kDegSlots - 1) + 1;
// The records' cells: kCellsPerDegree per degree (round 4: two -- a cell holds the knots from the last one at or below its start on,
// so one in 115 cells of a MATE-4v8-9 table overflows its record instead of one in 24, and the slowest waves of a launch are the
// ones that overflow at almost every step, tools/archive/rollout_subphases.py).  Cell starts c / kCellsPerDegree - 180 are exact in f64.
constexpr int kCellsPerDegree = 2;
constexpr int kLutCells = 360 * kCellsPerDegree;
__host__ __device__ __forceinline__ double cell_start(int c) { return (double)c * (1.0 / kCellsPerDegree) - 180.0; }
__device__ __forceinline__ int degree_of(double x) {      // the cell of angle x
    int d = (int)floor((x + 180.0) * (double)kCellsPerDegree);
    d = d < 0 ? 0 : (d > kLutCells - 1 ? kLutCells - 1 : d);
    if (x < cell_start(d)) d -= 1;                    // x + 180 rounded up across a cell boundary
    return d < 0 ? 0 : d;
}
__device__ __forceinline__ double degree_interp(const double2 (&w)[kDegWords], double x, bool &overflow) {      // (plain knots in w[0 .. kDegSlots))
    overflow = w[0].x != w[0].x;
    double2 k0 = w[0], k1 = w[1];
#pragma unroll
    for (int i = 1; i < kDegSlots - 1; ++i)
        if (w[i].x <= x) { k0 = w[i]; k1 = w[i + 1]; }
    if (k0.x == x) return k0.y;
    const double slope = (k1.y - k0.y) / (k1.x - k0.x);
    double res = slope * (x - k0.x) + k0.y;
    if (res != res) {
        res = slope * (x - k1.x) + k1.y;
        if (res != res && k0.y == k1.y) res = k0.y;
    }
    return res;
}

// Phase B: _update_view (environment.py:1356-1388).
struct SectorEval { bool seen, need; double rn, x; int64_t lc; };      // rn: the SQUARED distance camera -> other

// What a lane's range tests look like is the same at every step: which (target, other) pair it holds in each round,
// whether that is the diagonal, and the squared limit (sight + the other's radius)^2 -- all static inside an episode.
// The fused rollout, which is VALU-bound, computes them once per launch and keeps them in registers (`HELD`); the
// single-step kernel derives them in place.
constexpr int kRoleRounds = 5;       // (five: MATE-Navigation's row-image rollout; the fused random-policy rollout of the other shapes holds up to three -- Shape::kHoldRoles: a fourth, MATE-8v8-9's 200 range pairs,
                                     // measured no faster there at 126 of the 128 registers a wave may hold -- the fused Greedy rollout all four)
// Sensor.perceive (entities.py:229-232), `distance <= sight_range + radius`, is first tried on the f32 shadow of the entity
// table: with positions rounded to f32 the distance is off by at most 1.3e-4 and the squared distance by at most
// 2 * lim * 1.3e-4 + 3e-7 * lim^2 near the limit; outside a band of 2e-3 * lim + 2e-5 * lim^2 (eight times that) the f32
// comparison IS the verdict, inside it the f64 test below decides (a pair in 10^5).
__device__ __forceinline__ float range_rim(float lim) { return lim * (lim * 2e-5f + 2e-3f); }
template <typename ObsT, int L>
__device__ __forceinline__ bool range_exact(const Ctx<ObsT, L> &c, int tj, int j) {
    const double dx = c.ex[tj] - c.ex[j], dy = c.ey[tj] - c.ey[j];
    const double d2 = fma(dy, dy, dx * dx);
    const double lim = c.p.tgt_sight + c.er[j], lim2 = lim * lim;
    if (d2 < lim2 * (1.0 - 1e-14)) return true;        // decided on squares unless within rounding of the rim
    if (d2 > lim2 * (1.0 + 1e-14)) return false;
    return sqrt_pos(d2) <= lim;
}
// A value in the wave's LDS slice through a 32-bit LDS address the lane HOLDS (roles below): the address goes into the ds_read
// as it is, the byte offset into the instruction -- no shift / add per access.
typedef __attribute__((address_space(3))) const double lds_cf64;
typedef __attribute__((address_space(3))) const float lds_cf32;
__device__ __forceinline__ uint32_t lds_addr(const void *q) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)q; }
__device__ __forceinline__ double lds_f64(uint32_t a, int byte_off = 0) { return *(lds_cf64 *)(uintptr_t)(a + (uint32_t)byte_off); }
__device__ __forceinline__ float lds_f32(uint32_t a, int byte_off = 0) { return *(lds_cf32 *)(uintptr_t)(a + (uint32_t)byte_off); }
constexpr uint32_t kPass1Diag = 0xfffffffeu, kPass1None = 0xffffffffu;
constexpr int kSectorDiag = -3;      // RangeRoles::sector of a camera and itself (always seen, environment.py:1383-1384)
struct RangeRoles {
    uint32_t at[kRoleRounds], aj[kRoleRounds];   // LDS addresses of the f32 shadow x of the round's target / of the other (y: + 4 NJ)
    // (target sight range + other's radius)^2 MINUS the band's half-width, f32: below, the pair is seen; ... PLUS the half-width:
    // above, it is hidden; between the two the f64 test decides.  A target and itself: +inf, +inf (always seen); a lane without
    // a pair in this round: -inf, -inf (never) -- no separate diagonal / validity tests.
    float lim2[kRoleRounds];
    float rim[kRoleRounds];
    int32_t sector;                  // the lane's pair in the last sector round: sector_role, kSectorDiag, or -1
    uint32_t sec_cam, sec_other, sec_draw;       // ... LDS addresses: slice + 8 cam, slice + 8 (the other's entity slot), slice + 8 pair
    // shapes with two sector rounds (the compacted list): the lane's pair of each round for the range test that fills the list --
    // 8 cam | 8 (the other's entity slot) << 16, kPass1Diag for a camera and itself, kPass1None without a pair
    uint32_t pass1[2];
    // row-image mode: the (viewer, other) block each of the lane's pairs owns in the observation rows -- slot 0 the sector pair,
    // slots 1.. the range rounds: LDS byte offset of the other's public state | of the block in the viewer's row << 16
    uint32_t block[1 + kRoleRounds];
    uint32_t block_bits;             // 2 bits per slot: 0 none, 1 = 4 floats (obstacle), 2 = 5 (target), 3 = 7 (camera)
    // the collision screen of the NEXT step rides on the range tests (simulate_targets, NearCarry): (step size + the circle's
    // radius + 1e-3)^2 for a camera or an obstacle, negative for a pair that is no (target, circle) pair
    float reach2[kRoleRounds];
};
// LDS byte offsets (inside the wave's slice) of an entity's public state and of its block in a viewer's row; width code as above
__device__ __forceinline__ void image_block_of(const Params &p, bool viewer_is_camera, int viewer, int j, uint32_t &word, uint32_t &code) {
    // j: entity slot, cameras | obstacles | targets (the order of the entity table)
    int src, col;
    if (j < p.Nc) { src = p.off_pub + 32 * j; code = 3u; }
    else if (j < p.Nc + p.No) { src = p.off_pub + 32 * (p.Nc + p.Nt) + 16 * (j - p.Nc); code = 1u; }
    else { src = p.off_pub + 32 * p.Nc + 32 * (j - p.Nc - p.No); code = 2u; }
    if (viewer_is_camera) {                      // camera row: [preserved 13 | private 9 | targets 5 each | obstacles 4 each | cameras 7 each]
        col = 22 + (j < p.Nc ? 5 * p.Nt + 4 * p.No + 7 * j : j < p.Nc + p.No ? 5 * p.Nt + 4 * (j - p.Nc) : 5 * (j - p.Nc - p.No));
        word = (uint32_t)src | ((uint32_t)(p.off_img + 4 * (viewer * p.Dc + col)) << 16);
    } else {                                     // target row: [preserved 13 | private 14 | cameras 7 each | obstacles 4 each | targets 5 each]
        col = 27 + (j < p.Nc ? 7 * j : j < p.Nc + p.No ? 7 * p.Nc + 4 * (j - p.Nc) : 7 * p.Nc + 4 * p.No + 5 * (j - p.Nc - p.No));
        word = (uint32_t)src | ((uint32_t)(p.off_img + 4 * (p.cam_elems + viewer * p.Dt + col)) << 16);
    }
}
// The roles are pure functions of the lane id: left alone, the compiler re-derives them inside the step loop instead of
// holding them (15 instructions per step for the sector role alone).  Passing each word through an empty asm makes it opaque.
// (only what the kernel at hand reads is pinned -- a pinned word is a live register: `rounds` range rounds, the block words of the
// row-image mode, the sector pair of the shapes with one sector round)
__device__ __forceinline__ void pin_roles(RangeRoles &r, int rounds = kRoleRounds, bool image = true, bool sector = true) {
#pragma unroll
    for (int i = 0; i < kRoleRounds; ++i)
        if (i < rounds) { asm volatile("" : "+v"(r.at[i])); asm volatile("" : "+v"(r.aj[i])); asm volatile("" : "+v"(r.lim2[i])); asm volatile("" : "+v"(r.rim[i])); asm volatile("" : "+v"(r.reach2[i])); }
    if (image) {
#pragma unroll
        for (int i = 0; i < 1 + kRoleRounds; ++i)
            if (i <= rounds) asm volatile("" : "+v"(r.block[i]));
        asm volatile("" : "+v"(r.block_bits));
    }
    if (sector) { asm volatile("" : "+v"(r.sector)); asm volatile("" : "+v"(r.sec_cam)); asm volatile("" : "+v"(r.sec_other)); asm volatile("" : "+v"(r.sec_draw)); }
    else { asm volatile("" : "+v"(r.pass1[0])); asm volatile("" : "+v"(r.pass1[1])); }
}
template <typename ObsT>
__device__ __forceinline__ void range_roles(const Ctx<ObsT> &c, RangeRoles &roles) {
    const Params &p = c.p;
    roles.block_bits = 0;
    roles.sector = sector_role(p, (p.sector_rounds - 1) * 64 + c.lane);
#pragma unroll
    for (int round = 0; round < kRoleRounds; ++round) {
        const int q = round * 64 + c.lane;
        const int qq = q < p.n_range ? q : 0;
        const int t = (int)(((float)qq + 0.5f) * p.inv_NJ);
        const int j = qq - t * p.NJ;
        const int tj = c.tgt_slot(t);
        const float lim = (float)(p.tgt_sight + c.er[j]);
        const bool valid = q < p.n_range && round < p.range_rounds;
        roles.at[round] = lds_addr(c.exf + tj); roles.aj[round] = lds_addr(c.exf + j);
        roles.lim2[round] = !valid ? -INFINITY : j == tj ? INFINITY : lim * lim - range_rim(lim);
        roles.rim[round] = !valid ? -INFINITY : j == tj ? INFINITY : lim * lim + range_rim(lim);
        {   // a SCREEN: conservative by the whole step size (the walk repeats the test on the actual step, exactly), f32 with
            // an absolute margin eight times the worst rounding of the f32 shadow (see simulate_targets)
            const float step = (float)(((c.capword() >> t) & 1ull) ? p.tgt_step * 0.5 : p.tgt_step);
            const float reach = step + (float)c.er[j] + 1e-3f;
            roles.reach2[round] = (q < p.n_range && round < p.range_rounds && j < p.Nc + p.No) ? reach * reach : -1.0f;
        }
        roles.block[1 + round] = 0u;
        if (c.image() && q < p.n_range && round < p.range_rounds) {
            uint32_t code;
            image_block_of(p, false, t, j, roles.block[1 + round], code);
            roles.block_bits |= code << (2 * (1 + round));
        }
    }
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int role = sector_role(p, round * 64 + c.lane);
        roles.pass1[round] = kPass1None;
        if (role >= 0) {
            const int cam = role & 0xff, other = (role >> 8) & 0xff;
            const bool is_target = (role >> 16) & 1;
            roles.pass1[round] = !is_target && cam == other ? kPass1Diag : (uint32_t)(8 * cam) | ((uint32_t)(8 * (is_target ? c.tgt_slot(other) : other)) << 16);
        }
    }
    roles.block[0] = 0u;
    roles.sec_cam = roles.sec_other = roles.sec_draw = lds_addr(c.base);
    if (roles.sector >= 0) {
        const int cam = roles.sector & 0xff, other = (roles.sector >> 8) & 0xff;
        const bool is_target = (roles.sector >> 16) & 1;
        const int oj = is_target ? c.tgt_slot(other) : other;
        if (c.image()) {
            uint32_t code;
            image_block_of(p, true, cam, oj, roles.block[0], code);
            roles.block_bits |= code;
        }
        roles.sec_cam += 8u * (uint32_t)cam; roles.sec_other += 8u * (uint32_t)oj;
        if (is_target) roles.sec_draw += 8u * (uint32_t)(cam * p.Nt + other);
        if (!is_target && cam == other) roles.sector = kSectorDiag;
    }
}

// The first step of a launch has no previous step: the same comparison on the positions the launch starts from.
template <typename ObsT>
__device__ __forceinline__ void near_seed(const Ctx<ObsT> &c, const RangeRoles &roles, NearCarry &near) {
#pragma unroll
    for (int round = 0; round < kRoleRounds; ++round) {
        near.w[round] = 0ull;
        if (round < c.p.range_rounds) {
            const float dx = lds_f32(roles.at[round]) - lds_f32(roles.aj[round]);
            const float dy = lds_f32(roles.at[round], 4 * c.p.NJ) - lds_f32(roles.aj[round], 4 * c.p.NJ);
            near.w[round] = __ballot(fmaf(dy, dy, dx * dx) <= roles.reach2[round]);
        }
    }
}

// Camera.perceive (entities.py:491-505) up to the occlusion lookup.
__device__ __forceinline__ int sector_role(const Params &p, int q) {      // camera | other << 8 | is_target << 16, or -1
    if (q < 0 || q >= p.n_sector) return -1;      // (q < 0: the "last sector round" of a scenario without cameras)
    int cam, other; bool is_target;
    if (q < p.bit_cc) { cam = (int)(((float)q + 0.5f) * p.inv_Nt); other = q - cam * p.Nt; is_target = true; }
    else { const int r = q - p.bit_cc; cam = (int)(((float)r + 0.5f) * p.inv_Nc); other = r - cam * p.Nc; is_target = false; }
    return cam | (other << 8) | ((int)is_target << 16);
}
// `role`: sector_role of the pair, or kNoRole = derive it here (the fused rollout holds the last round's in a register)
constexpr int kNoRole = -2;
// `relative.norm > self.sight_range` (entities.py:495-496) on squares; inside the rounding rim, on the roots themselves
__device__ __forceinline__ bool sector_out_of_range(double d2, double s2) {
    if (d2 > s2 * (1.0 + 1e-14)) return true;
    return !(d2 < s2 * (1.0 - 1e-14)) && sqrt_pos(d2) > sqrt_pos(s2);
}
template <bool RANGED = false, typename ObsT, int L>
__device__ __forceinline__ SectorEval sector_eval(Ctx<ObsT, L> &c, int q, uint32_t tick, uint32_t stream, bool predrawn, int role = kNoRole) {
    const Params &p = c.p;
    SectorEval e;                    // rn, x, lc: meaningful under `need` only -- left undefined on the early exits (zeroing them on
    e.seen = false; e.need = false;  // every exit path was 18 vector moves per step)
    if (role == kNoRole) role = sector_role(p, q);
    if (role < 0) return e;
    const int cam = role & 0xff, other = (role >> 8) & 0xff;
    const bool is_target = (role >> 16) & 1;
    if (!is_target && cam == other) { e.seen = true; return e; }                   // environment.py:1383-1384
    const int oj = is_target ? c.tgt_slot(other) : other;
    const double rx = c.ex[oj] - c.ex[cam], ry = c.ey[oj] - c.ey[cam];
    const double d2 = fma(ry, ry, rx * rx);
    if (!RANGED && sector_out_of_range(d2, c.sight2(cam))) return e;              // RANGED: the caller has made this test
    const double ang = atan2_sector(ry, rx) * kRad2Deg;
    double ra = fabs(c.phi(cam) - ang);
    const double alt = 360.0 - ra;
    if (alt < ra) ra = alt;
    if (ra * 2.0 > c.theta(cam)) return e;
    // no obstacles: every knot of the occlusion table is the full range, and the pair is inside the sight range already, so
    // `norm <= sight_range_at(angle) (1 + 1e-6)` (entities.py:505) holds whatever the transmittance draw says
    if (p.No == 0) { e.seen = true; return e; }
    if (is_target) {                                                               // np_random.binomial(1, tau), entities.py:503
        const int pair = cam * p.Nt + other;
        double u;
        if (c.tape_ct()) u = c.tape_ct()[c.env * p.Nc * p.Nt + pair];
        else if (predrawn && pair < c.predrawn_pairs()) u = c.udraw(pair);
        else u = c.draw(tick, stream, (uint32_t)pair);
        if ((p.tau <= 0.5) ? (u > 1.0 - p.tau) : (u <= p.tau)) { e.seen = true; return e; }
    }
    e.need = true; e.rn = d2; e.x = normalize_angle(ang); e.lc = c.env * p.Nc + cam;
    return e;
}

// The same test for the pair a lane of the fused rollouts HOLDS (RangeRoles: the last sector round): its operands through the
// held LDS addresses, the pair's kind from the held role word.  Same arithmetic, same verdicts.
template <typename ObsT>
__device__ __forceinline__ SectorEval sector_eval_held(Ctx<ObsT> &c, const RangeRoles &h, uint32_t tick, uint32_t stream, bool predrawn) {
    const Params &p = c.p;
    SectorEval e;
    e.seen = false; e.need = false;
    const int role = h.sector;
    if (role < 0) { e.seen = role == kSectorDiag; return e; }
    const int ent = p.off_ent, yoff = 8 * p.NJ;
    const double rx = lds_f64(h.sec_other, ent) - lds_f64(h.sec_cam, ent), ry = lds_f64(h.sec_other, ent + yoff) - lds_f64(h.sec_cam, ent + yoff);
    const double d2 = fma(ry, ry, rx * rx);
    if (sector_out_of_range(d2, lds_f64(h.sec_cam, p.off_tmp))) return e;
    const double ang = atan2_sector(ry, rx) * kRad2Deg;
    double ra = fabs(lds_f64(h.sec_cam, p.off_dy) - ang);
    const double alt = 360.0 - ra;
    if (alt < ra) ra = alt;
    if (ra * 2.0 > lds_f64(h.sec_cam, p.off_dy + 8 * p.Nc)) return e;
    if (p.No == 0) { e.seen = true; return e; }
    if ((role >> 16) & 1) {                                                        // np_random.binomial(1, tau), entities.py:503
        double u;
        if (predrawn && p.Nc * p.Nt <= c.predrawn_pairs()) u = lds_f64(h.sec_draw, p.off_tmp + 8 * (p.Nc + p.Nt));
        else {
            const int pair = (role & 0xff) * p.Nt + ((role >> 8) & 0xff);
            if (predrawn && pair < c.predrawn_pairs()) u = c.udraw(pair);
            else u = c.draw(tick, stream, (uint32_t)pair);
        }
        if ((p.tau <= 0.5) ? (u > 1.0 - p.tau) : (u <= p.tau)) { e.seen = true; return e; }
    }
    e.need = true; e.rn = d2; e.x = normalize_angle(ang); e.lc = c.env * p.Nc + (role & 0xff);
    return e;
}

template <typename ObsT, int L>
__device__ __forceinline__ void sector_fetch(const Ctx<ObsT, L> &c, const SectorEval &e, double2 (&w)[kDegWords]) {
    if (e.need) {
        const double2 *rec = c.g.lut_deg + (MATE_LUT_TABLE_OF(e.lc) * kLutCells + degree_of(e.x)) * kDegWords;
#pragma unroll
        for (int i = 0; i < kDegWords; ++i) w[i] = rec[i];
    }
}

template <typename ObsT, int L>
__device__ __forceinline__ bool sector_resolve(const Ctx<ObsT, L> &c, const SectorEval &e, double2 (&w)[kDegWords]) {
    if (!e.need) return e.seen;
    bool overflow;
    double limit = segment_interp(w, e.x, overflow);
    SUB_COUNT(c, 1, overflow);                           // ... with a lookup in a degree that overflows its record
    if (overflow) {
        int start = (int)w[0].y, count = (int)w[1].x;
        const double2 *knots = c.g.lut_knots + e.lc * c.p.kmax;
        // A degree with more knots than the quarter scheme takes (two or more obstacles' arcs in one degree): the same three-pivot
        // selection first narrows it -- one round trip per level, 65 knots in one, 257 in two -- instead of the general path's
        // dependent binary search (ten round trips).  A launch of 4096 one-step waves lasts as long as its slowest wave, and with
        // one such lookup in a thousand there is one in (almost) every launch: it WAS the slowest wave, 14 k cycles in this phase
        // against 5.4 k (profiles/HISTORY.md 3.1d).
        // Eight PIVOT ANGLES ride in the overflow record itself (kPivotKnots; the builders write the angles of knots q, 2q, ... 8q of
        // the degree, q = ceil((count - 1) / 9), +inf beyond the last): up to 37 knots -- every degree but the rarest -- the
        // bracketing five are fetched at once, ONE round trip behind the record's instead of two (pivots, then knots).  A step in
        // which any pair overflows waits for the slowest lane: 30 % of the steps of MATE-4v8-9, nearly all of the heaviest
        // environments', which set the length of a short launch.
        if (c.pivots && count >= 2 && count <= kPivotKnots && w[1].y > 0.0) {
            const int q = (int)w[1].y, last = count - 1;
            const double pv[8] = {w[2].x, w[2].y, w[3].x, w[3].y, w[4].x, w[4].y, w[5].x, w[5].y};
            int nle = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) nle += pv[i] <= e.x ? 1 : 0;
            const int base = nle * q;
#pragma unroll
            for (int i = 0; i < kDegSlots; ++i) w[i] = knots[start + (base + i < last ? base + i : last)];
            limit = degree_interp(w, e.x, overflow);
        } else {
        while (count > kQuarterKnots && count <= 4 * 256 + 1) {
            const int q = (count + 2) / 4, last = count - 1;
            const double a1 = knots[start + (q < last ? q : last)].x;
            const double a2 = knots[start + (2 * q < last ? 2 * q : last)].x;
            const double a3 = knots[start + (3 * q < last ? 3 * q : last)].x;
            const int base = e.x >= a3 ? 3 * q : (e.x >= a2 ? 2 * q : (e.x >= a1 ? q : 0));
            start += base;
            count = (q + 1 < count - base) ? q + 1 : count - base;
        }
        if (count >= 2 && count <= kQuarterKnots) {
            const int q = (count + 2) / 4, last = count - 1;                          // q = ceil((count - 1) / 4) <= kDegSlots - 1
            const double a1 = knots[start + (q < last ? q : last)].x;
            const double a2 = knots[start + (2 * q < last ? 2 * q : last)].x;
            const double a3 = knots[start + (3 * q < last ? 3 * q : last)].x;
            const int base = e.x >= a3 ? 3 * q : (e.x >= a2 ? 2 * q : (e.x >= a1 ? q : 0));
#pragma unroll
            for (int i = 0; i < kDegSlots; ++i) w[i] = knots[start + (base + i < last ? base + i : last)];
            limit = degree_interp(w, e.x, overflow);
        } else {
            limit = lut_lookup(knots, c.g.lut_bucket + e.lc * c.p.nbucket, c.g.lut_count[e.lc], e.x);
        }
        }
    }
    // `relative.norm <= sight_range_at(angle) * (1 + 1e-6)` (entities.py:505) on squares, the root inside the rounding rim
    const double lim = limit * (1.0 + 1e-6), lim2 = lim * lim;
    if (e.rn < lim2 * (1.0 - 1e-14)) return lim > 0.0;
    if (e.rn > lim2 * (1.0 + 1e-14)) return false;
    return sqrt_pos(e.rn) <= lim;
}

// COMPACT (the fused rollouts): shapes with two or more rounds of sector pairs run the long part of the sector test on a
// compacted list of the pairs in sight range.  The single-step kernel keeps the round-by-round form: it lives on a 64-VGPR /
// 96-SGPR budget where the list bookkeeping spills, and measured 3 % slower with it.
// `seen_out` (row-image mode, which writes no flag words): bit 0 = the lane's sector pair is seen, bit 1 + r = its pair of
// range round r (image_blocks turns them into the pair's block of the observation rows).
// `sector_ballot` (the register-resident step of the fused rollout, HeldState): the packed camera->target / camera->camera
// word as the ballot that made it, and NO tail here -- tracked bits and warehouse membership are derived from it and from the
// positions in registers (view_tail_held), without the two LDS hand-offs of the tail below.
template <bool HELD, bool COMPACT = false, typename ObsT, int L>
__device__ __forceinline__ void update_view(Ctx<ObsT, L> &c, uint32_t tick, uint32_t stream, bool predrawn, const RangeRoles &held, uint32_t &seen_out,
                                            NearCarry *near_next = nullptr, unsigned long long *sector_ballot = nullptr) {
    const Params &p = c.p;
    const int lane = c.lane;
    seen_out = 0u;
    double2 w[kDegWords];
    int n_cand = 0;
    uint8_t *cand = c.base + p.off_list;
    static_assert(L == 64 || !HELD, "held lane roles: one wave per environment");
    const bool compact = L == 64 && COMPACT && p.sector_rounds >= 2 && p.n_sector <= 256;
    // rounds of L pairs (the Params' round counts are rounds of 64: they lay out the mask bits, which stay where they are) -- expressions, not
    // locals: a local that the range_tests lambda captures cost the MATE-8v8-9 step kernel its 64-register budget (65: seven waves per SIMD)
#define MATE_N_SR (L == 64 ? p.sector_rounds : c.rounds_of(p.n_sector))
#define MATE_N_RR (L == 64 ? p.range_rounds : c.rounds_of(p.n_range))
    if (compact) {
        // ---- two or more rounds of sector pairs (8 cameras: 128): most are out of sight range, and only the others need the
        // long part of Camera.perceive (atan2, the transmittance draw, the occlusion lookup).  Pass 1 makes the range test
        // for every pair, settles the pairs it decides (out of range: hidden; a camera and itself: seen) and compacts the
        // rest into a list; pass 2 runs the long part on the list, 64 candidates at a time (usually once).
        for (int round = 0; round < p.sector_rounds; ++round) {
            const int q = round * 64 + lane;
            bool diag = false, in_range = false;
            if (HELD && p.sector_rounds == 2) {       // the pair's LDS offsets held since the launch began (RangeRoles::pass1)
                const uint32_t held_pair = held.pass1[round & 1];
                if (held_pair != kPass1None) {
                    diag = held_pair == kPass1Diag;
                    if (!diag) {
                        const uint32_t a_cam = lds_addr(c.base) + (held_pair & 0xffffu), a_oj = lds_addr(c.base) + (held_pair >> 16);
                        const double rx = lds_f64(a_oj, p.off_ent) - lds_f64(a_cam, p.off_ent), ry = lds_f64(a_oj, p.off_ent + 8 * p.NJ) - lds_f64(a_cam, p.off_ent + 8 * p.NJ);
                        in_range = !sector_out_of_range(fma(ry, ry, rx * rx), lds_f64(a_cam, p.off_tmp));
                    }
                    set_flag(c, q, diag);
                }
            } else {
            const int role = sector_role(p, q);
            if (role >= 0) {
                const int cam = role & 0xff, other = (role >> 8) & 0xff;
                const bool is_target = (role >> 16) & 1;
                diag = !is_target && cam == other;
                if (!diag) {
                    const int oj = is_target ? c.tgt_slot(other) : other;
                    const double rx = c.ex[oj] - c.ex[cam], ry = c.ey[oj] - c.ey[cam];
                    in_range = !sector_out_of_range(fma(ry, ry, rx * rx), c.sight2(cam));
                }
                set_flag(c, q, diag);
            }
            }
            const unsigned long long b = __ballot(diag), m = __ballot(in_range);
            if (lane == 0) { c.mask[2 * round] = (uint32_t)b; c.mask[2 * round + 1] = (uint32_t)(b >> 32); }
            if (in_range) cand[n_cand + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint8_t)q;
            n_cand += __popcll(m);
        }
        wave_sync();
    }
    // ---- range tests: Sensor.perceive (entities.py:229-232) target -> camera / obstacle / target.
    // Two passes: (1) all rounds' LDS reads and arithmetic back to back (independent chains overlap their
    // latency), results collected in a per-lane bit set; (2) flags, ballots and mask words.
    unsigned long long range_ballot[kRoleRounds] = {};       // HELD: the rounds' mask words, written by lane 0 in one go
    auto write_range_ballots = [&]() {
        if constexpr (HELD) {
            const int rbase = p.bit_range >> 5;
            if (lane == 0)
#pragma unroll
                for (int round = 0; round < kRoleRounds; ++round)
                    if (round < p.range_rounds) { c.mask[rbase + 2 * round] = (uint32_t)range_ballot[round]; c.mask[rbase + 2 * round + 1] = (uint32_t)(range_ballot[round] >> 32); }
        }
    };
    auto range_tests = [&]() {
    const int rbase = p.bit_range >> 5;
    uint32_t seen_bits = 0;
    if constexpr (HELD) {
#pragma unroll
        for (int round = 0; round < kRoleRounds; ++round) {
            if (round < p.range_rounds) {
                const float dx = lds_f32(held.at[round]) - lds_f32(held.aj[round]);
                const float dy = lds_f32(held.at[round], 4 * p.NJ) - lds_f32(held.aj[round], 4 * p.NJ);
                const float d2 = fmaf(dy, dy, dx * dx);
                // (held as the two ends of the band: lim2[] = limit^2 - rim, rim[] = limit^2 + rim; the diagonal and the lanes
                // without a pair are decided by infinite ends)
                bool seen = d2 < held.lim2[round];
                if (!seen && !(d2 > held.rim[round])) {
                    const uint32_t first = lds_addr(c.exf);
                    seen = range_exact(c, (int)((held.at[round] - first) >> 2), (int)((held.aj[round] - first) >> 2));
                }
                seen_bits |= (uint32_t)seen << round;
                if (near_next) near_next->w[round] = __ballot(d2 <= held.reach2[round]);      // the next step's collision screen
            }
        }
    } else {
#pragma unroll 4
    for (int round = 0; round < MATE_N_RR; ++round) {
        const int q = round * L + lane;
        const int qq = q < p.n_range ? q : 0;
        const int t = (int)(((float)qq + 0.5f) * p.inv_NJ);
        const int j = qq - t * p.NJ;
        const int tj = c.tgt_slot(t);
        const bool diag = (j == tj);
        const float dx = c.exf[tj] - c.exf[j], dy = c.eyf[tj] - c.eyf[j];
        const float d2 = fmaf(dy, dy, dx * dx);
        const float lim = (float)p.tgt_sight + c.erf[j], lim2 = lim * lim, rim = range_rim(lim);
        bool seen = d2 < lim2 - rim;
        if (!seen && !(d2 > lim2 + rim)) seen = range_exact(c, tj, j);
        seen_bits |= (uint32_t)((seen || diag) && q < p.n_range) << round;
    }
    }
    for (int round = 0; round < MATE_N_RR; ++round) {
        const int q = round * L + lane;
        const bool seen = (seen_bits >> round) & 1u;
        if (!c.image() && q < p.n_range) set_flag(c, p.fs_range + q, seen);
        const unsigned long long b = c.ballot(seen);
        if (HELD && round < kRoleRounds) range_ballot[round] = b;      // (written with the sector word, below)
        else if constexpr (L == 64) { if (lane == 0) { c.mask[rbase + 2 * round] = (uint32_t)b; c.mask[rbase + 2 * round + 1] = (uint32_t)(b >> 32); } }
        else c.put_bits(p.bit_range + round * L, b);
    }
    seen_out |= seen_bits << 1;
    };
    if (compact) {
        // pass 2: the long part on the compacted list, 64 candidates at a time (usually one chunk).  The occlusion records of the
        // FIRST chunk travel while the range tests run -- not into registers, as in the one-round path below (24 VGPRs the
        // fused rollouts of these shapes do not have), but into the caches: one dword of each of the record's two possible
        // lines is touched before the range tests, the record itself is fetched behind them (a hit instead of a round trip
        // to HBM: the lookup's latency was a fifth of this phase).
        SectorEval first;
        first.seen = false; first.need = false;
        int q_first = 0;
        uint32_t touch0 = 0u, touch1 = 0u;
        if (lane < n_cand) {
            q_first = cand[lane];
            first = sector_eval<true>(c, q_first, tick, stream, predrawn);
            if (first.need) {
                const double2 *rec = c.g.lut_deg + (first.lc * kLutCells + degree_of(first.x)) * kDegWords;
                asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %2, off offset:92" : "+v"(touch0), "+v"(touch1) : "v"(rec) : "memory");
            }
        }
        range_tests();
        write_range_ballots();
        if (lane < n_cand) {
            sector_fetch(c, first, w);
            if (sector_resolve(c, first, w)) { set_flag(c, q_first, true); atomicOr(&c.mask[q_first >> 5], 1u << (q_first & 31)); }
        }
        // (the two touched words land whenever they land -- in order, so before the record behind them: their registers stay
        // reserved until here, or a late arrival would overwrite whatever the compiler had put there)
        asm volatile("" :: "v"(touch0), "v"(touch1));
        for (int k = lane + 64; k < n_cand; k += 64) {
            const int q = cand[k];
            const SectorEval e = sector_eval<true>(c, q, tick, stream, predrawn);
            sector_fetch(c, e, w);
            if (sector_resolve(c, e, w)) { set_flag(c, q, true); atomicOr(&c.mask[q >> 5], 1u << (q & 31)); }
        }
    } else {
        // ---- sector tests for camera->target and camera->camera; all rounds but the last resolve at once, the last round's
        // occlusion records travel while the range tests run
        // Sub-wave groups with TWO rounds of sector pairs (MATE-4v2-*: 24 pairs, MATE-4v4-*: 32, in rounds of 16): both rounds' occlusion
        // records travel while the range tests run -- resolved one after the other, the first round's record was a bare round trip to
        // L2 / HBM on every step (the view phase of the four-per-wave MATE-4v2-9 rollout: 9.5 k of a step's 24 k cycles).
        // (MATE-4v2-* only: the MATE-4v4-9 kernels have no 24 registers left for a second record)
        const bool both_in_flight = L < 64 && MATE_N_SR == 2 && p.Nc * p.Nt < 16;
        SectorEval early;
        early.seen = false; early.need = false;
        double2 w_early[kDegWords];
        if (both_in_flight) {
            early = sector_eval(c, lane, tick, stream, predrawn);
            sector_fetch(c, early, w_early);
        }
        for (int round = 0; round + 1 < MATE_N_SR && !both_in_flight; ++round) {
            const SectorEval e = sector_eval(c, round * L + lane, tick, stream, predrawn);
            sector_fetch(c, e, w);
            const bool seen = sector_resolve(c, e, w);
            if (round * L + lane < p.n_sector) set_flag(c, round * L + lane, seen);
            const unsigned long long b = c.ballot(seen);
            if constexpr (L == 64) { if (lane == 0) { c.mask[2 * round] = (uint32_t)b; c.mask[2 * round + 1] = (uint32_t)(b >> 32); } }
            else c.put_bits(round * L, b);
        }
        const int last = MATE_N_SR - 1;
        SectorEval pending;
        pending.seen = false; pending.need = false;
        if (last >= 0) {
            // (a branch-free form of this test -- every lane computing everything, verdicts combined at the end -- measured
            // no faster: the early exits cost scalar instructions, which issue beside the other waves' vector work)
            if constexpr (HELD) pending = sector_eval_held(c, held, tick, stream, predrawn);
            else pending = sector_eval(c, last * L + lane, tick, stream, predrawn);
            sector_fetch(c, pending, w);
        }
        SUB_STAMP(c, 13);
        SUB_ACC(c, 0);                                   // sector geometry + fetch issued
        SUB_COUNT(c, 0, pending.need);                   // steps with an occlusion lookup
        range_tests();
        SUB_ACC(c, 1);                                   // range tests
        if (both_in_flight) {
            const bool seen = sector_resolve(c, early, w_early);
            if (lane < p.n_sector) set_flag(c, lane, seen);
            c.put_bits(0, c.ballot(seen));
        }
        if (last >= 0) {
            const bool seen = sector_resolve(c, pending, w);
            SUB_ACC(c, 2);                               // wait for the record + interpolation (+ overflow trips)
            if (!c.image() && last * L + lane < p.n_sector) set_flag(c, last * L + lane, seen);
            seen_out |= (uint32_t)seen;
            const unsigned long long b = c.ballot(seen);
            if constexpr (L == 64) { if (lane == 0) { c.mask[2 * last] = (uint32_t)b; c.mask[2 * last + 1] = (uint32_t)(b >> 32); } }
            else c.put_bits(last * L, b);
            if (sector_ballot) *sector_ballot = b;
        }
        write_range_ballots();
    }
    SUB_STAMP(c, 14);
    // ---- static camera->obstacle bits (environment.py:752-755) and the always-true bit
    if (!c.statics_done) {
        if (lane < p.Nc) {
            const uint64_t m = c.camobs(lane);
            c.mask[(p.bit_camobs >> 5) + 2 * lane] = (uint32_t)m;
            c.mask[(p.bit_camobs >> 5) + 2 * lane + 1] = (uint32_t)(m >> 32);
        }
        if (lane == 0) { c.mask[p.bit_always >> 5] = 1u; if (!c.image()) set_flag(c, p.fs_always, true); }
        if (!c.image())
        for (int q = lane; q < p.Nc * p.No; q += L) {
            const int cam = (int)(((float)q + 0.5f) * p.inv_No);
            const int o = q - cam * p.No;
            set_flag(c, p.fs_camobs + cam * p.No + o, (c.camobs(cam) >> o) & 1ull);
        }
    }
    if (sector_ballot) return;
    wave_sync();
    // ---- tracked_bits = camera_target_view_mask.any(axis=0) (environment.py:1388); which warehouse holds the target
    if (lane < p.Nt) {
        int any = 0;
        for (int cam = 0; cam < p.Nc; ++cam) any |= (int)c.mask_bit(cam * p.Nt + lane);
        c.tracked(lane) = any;
        // warehouses (constants.py:70-72): 0 (+,+), 1 (-,+), 2 (-,-), 3 (+,-), Chebyshev radius 75 around (+-925, +-925)
        // (environment.py:1283).  Only the warehouse of the target's own quadrant can hold it (any other centre is at
        // least 925 away in one coordinate), so one test with that centre is the reference's loop over the four.
        const double x = c.tx(lane), y = c.ty(lane);
        const bool px = x > 0.0, py = y > 0.0;
        const double wx = px ? kWarehouseCenter : -kWarehouseCenter, wy = py ? kWarehouseCenter : -kWarehouseCenter;
        const double sup = fmax(fabs(x - wx), fabs(y - wy));
        c.inside(lane) = sup <= kWarehouseRadius ? (px ? (py ? 0 : 3) : (py ? 1 : 2)) : -1;
    }
    wave_sync();
#undef MATE_N_SR
#undef MATE_N_RR
}

template <bool HELD, bool COMPACT = false, typename ObsT, int L>
__device__ __forceinline__ void update_view(Ctx<ObsT, L> &c, uint32_t tick, uint32_t stream, bool predrawn, const RangeRoles &held) {
    uint32_t seen;
    update_view<HELD, COMPACT>(c, tick, stream, predrawn, held, seen);
}
template <typename ObsT, int L>
__device__ __forceinline__ void update_view(Ctx<ObsT, L> &c, uint32_t tick, uint32_t stream, bool predrawn) {
    RangeRoles none;
    update_view<false>(c, tick, stream, predrawn, none);
}

// The order-dependent part of _assign_goals (environment.py:1278-1318: the warehouses' remaining cargo is shared), on ONE lane,
// for the targets standing in a warehouse (c.inside); dense and delayed rewards of the deliveries are added to `reward` / `delayed`.
template <typename ObsT, int L>
__device__ __forceinline__ void goal_logistics(Ctx<ObsT, L> &c, uint32_t tick, double &reward, double &delayed, uint32_t inside_mask) {
    const Params &p = c.p;
    // `inside_mask`: bit t = target t stands in a warehouse (the callers' ballot, wave-uniform): the loop visits those targets only, in
    // ascending order as the reference's does -- one LDS round trip per target in a warehouse (usually one) instead of one per target
    for (uint32_t todo = inside_mask; todo != 0u; todo &= todo - 1u) {
        const int t = __ffs((int)todo) - 1;
        const int w = c.inside(t);
        if (w < 0) continue;
        int gw = c.ti(t, TI_GW);
        const int goal = (gw & 0xff) - 1;
        const int weight0 = (gw >> 8) & 0xff;
        bool proceed = true;
        if (goal >= 0) {
            if (goal == w) {                            // delivery, environment.py:1286-1293
                const double total_bounty = (double)weight0 * p.bounty_scale;
                const double r = (double)(c.ti(t, TI_FREIGHT) + c.ti(t, TI_BOUNTY));
                reward += r;
                delayed += r - (total_bounty - (double)c.ti(t, TI_BOUNTY));
                c.ei(EI_DELIVERED) += weight0;
                c.ei(EI_AWAITING + goal) -= weight0;
            } else {
                proceed = false;                        // environment.py:1294-1295
            }
        }
        if (proceed) {
            c.ti(t, TI_FREIGHT) = 0; c.ti(t, TI_BOUNTY) = 0;
            c.ti(t, TI_TSTEPS) = 0; c.ti(t, TI_TRSTEPS) = 0;
            gw &= ~0xffff;                              // goal := none, weight := 0
            int *row = &c.ei(EI_REMAINING + 4 * w);
            int k = 0;
            for (int gq = 0; gq < 4; ++gq) k += row[gq] > 0;
            if (k > 0) {                                // environment.py:1302-1315
                const double u = c.tape_goal() ? c.tape_goal()[c.env * p.Nt + t] : c.draw(tick, S_GOAL, (uint32_t)t);
                int j = (int)(u * (double)k);
                if (j >= k) j = k - 1;
                int new_goal = 0;
                for (int gq = 0, seen = 0; gq < 4; ++gq) if (row[gq] > 0) { if (seen == j) new_goal = gq; ++seen; }
                const int cap = 1 + (int)((c.capword() >> t) & 1ull);
                const int rem = row[new_goal];
                const int weight = cap < rem ? cap : rem;
                row[new_goal] -= weight;
                c.ti(t, TI_FREIGHT) = (int)((double)weight * p.freight_scale);
                c.ti(t, TI_BOUNTY) = (int)((double)weight * p.bounty_scale);
                gw |= (new_goal + 1) | (weight << 8);
            }
        }
        // empty bits of the warehouse the target stands in (environment.py:1317-1318)
        const int *row = &c.ei(EI_REMAINING + 4 * w);
        const bool empty = !(row[0] || row[1] || row[2] || row[3]);
        gw = (gw & ~(1 << (16 + w))) | ((int)empty << (16 + w));
        c.ti(t, TI_GW) = gw;
    }
}

// update_view's tail for lanes [0, Nt) from the ballot of the (only) sector round: tracked_bits = camera_target_view_mask.any(axis=0)
// (environment.py:1388), and which warehouse holds the target (see update_view)
template <typename ObsT>
__device__ __forceinline__ void view_tail_regs(Ctx<ObsT> &c, unsigned long long sector_ballot, int &tracked, int &inside) {
    const Params &p = c.p;
    const int lane = c.lane;
    unsigned long long any = 0ull;
    for (int cam = 0; cam < p.Nc; ++cam) any |= sector_ballot >> (cam * p.Nt);
    tracked = lane < p.Nt ? (int)((any >> (lane & 63)) & 1ull) : 0;
    inside = -1;
    if (lane < p.Nt) {
        const double x = c.tx(lane), y = c.ty(lane);
        const bool px = x > 0.0, py = y > 0.0;
        const double wx = px ? kWarehouseCenter : -kWarehouseCenter, wy = py ? kWarehouseCenter : -kWarehouseCenter;
        const double sup = fmax(fabs(x - wx), fabs(y - wy));
        inside = sup <= kWarehouseRadius ? (px ? (py ? 0 : 3) : (py ? 1 : 2)) : -1;
    }
}

// Phase C: _assign_goals + the bookkeeping of step() (environment.py:1271-1324, 613-632).
// `tracked_reg` / `inside_reg` (the single-step kernel, shapes with one sector round): the lane's tracked bit and warehouse straight
// from the sector ballot and the position (view_tail_regs) instead of through update_view's tail and its two LDS hand-offs.
template <typename ObsT, int L>
__device__ __forceinline__ void assign_and_score(Ctx<ObsT, L> &c, uint32_t tick, float *scalars_out, const int *tracked_reg = nullptr, const int *inside_reg = nullptr) {
    const Params &p = c.p;
    const int lane = c.lane;
    bool penal = false;
    if (tracked_reg && lane < p.Nt) { c.tracked(lane) = *tracked_reg; c.inside(lane) = *inside_reg; }      // (for goal_logistics and the team-wide flags)
    const int tracked_lane = lane < p.Nt ? (tracked_reg ? *tracked_reg : c.tracked(lane)) : 0;
    const int inside_lane = lane < p.Nt ? (inside_reg ? *inside_reg : c.inside(lane)) : -1;
    if (lane < p.Nt) {
        const int b = c.ti(lane, TI_BOUNTY);
        const int tr = tracked_lane;
        penal = tr && b > 0;                                  // environment.py:1275
        const int nb = b - tr;
        c.ti(lane, TI_BOUNTY) = nb > 0 ? nb : 0;               // environment.py:1276
    }
    const int n_penal = __popcll(c.ballot(penal));
    const uint32_t inside_mask = (uint32_t)c.ballot(lane < p.Nt && inside_lane >= 0);      // (target t on lane t; at most 16 targets)
    const bool any_inside = inside_mask != 0u;
    wave_sync();
    double reward = -(double)n_penal, delayed = 0.0;
    if (any_inside) {
        if (lane == 0) {
            goal_logistics(c, tick, reward, delayed, inside_mask);
            c.xch(0) = __double2hiint(reward); c.xch(1) = __double2loint(reward);
            c.xch(2) = __double2hiint(delayed); c.xch(3) = __double2loint(delayed);
        }
        wave_sync();
        reward = __hiloint2double(c.xch(0), c.xch(1));
        delayed = __hiloint2double(c.xch(2), c.xch(3));
    }
    // metrics (environment.py:966-979), counters (environment.py:626-632)
    bool with_bounty = false, tr = false;
    if (lane < p.Nt) {
        with_bounty = c.ti(lane, TI_BOUNTY) > 0;
        tr = tracked_lane != 0;
        c.ti(lane, TI_TSTEPS) += 1;
        c.ti(lane, TI_TRSTEPS) += (int)tr;
    }
    const int n_tracked = __popcll(c.ballot(tr));
    const int n_bounty = __popcll(c.ballot(with_bounty));
    const int n_both = __popcll(c.ballot(tr && with_bounty));
    if (lane == 0) {
        const double epr = c.ep_reward() + reward;
        const double epd = c.ep_delayed() + delayed;
        c.ep_reward() = epr; c.ep_delayed() = epd;
        const int delivered = c.ei(EI_DELIVERED);
        const double coverage = div_by_count((double)n_tracked, p.Nt);
        const double real_cov = n_bounty > 0 ? div_nz((double)n_both, (double)n_bounty) : 0.0;
        const double transport = delivered > 0 ? div_nz(epd, p.reward_scale * (double)delivered) : 0.0;
        const double r = p.sparse_reward ? delayed : reward;
        const int ep_step = c.ei(EI_EPSTEP) + 1;
        c.ei(EI_EPSTEP) = ep_step;
        const bool awaiting = c.ei(EI_AWAITING) || c.ei(EI_AWAITING + 1) || c.ei(EI_AWAITING + 2) || c.ei(EI_AWAITING + 3);
        const int done = !(ep_step <= p.max_episode_steps && awaiting);
        // 1 = finished; 3 = finished AND on the list of the next reset launch (so that nobody lists it twice)
        c.ei(EI_DONE) = done ? (c.g.done_count ? 3 : 1) : 0;
        c.ei(EI_TICK) = (int)(tick + 1u);
        if (c.has_scalars() && scalars_out) {
            float *o = scalars_out + c.out * 8;
            o[0] = (float)(-r); o[1] = (float)r; o[2] = (float)done; o[3] = (float)coverage;
            o[4] = (float)real_cov; o[5] = (float)transport; o[6] = (float)delivered; o[7] = (float)div_nz(r, p.max_team_reward);
        }
        if (done && c.g.done_count) {
            const int parity = c.list_parity();
            const int slot = atomicAdd(c.g.done_count + parity, 1);
            c.g.done_list[(int64_t)parity * c.g.N + slot] = (int32_t)c.env;
        }
        if (done && c.g.ep_stats) {      // episode statistics for logging (the record SURVEY.md 8e all-gathers); rare
            double *es = c.g.ep_stats;
            atomicAdd(es + 0, 1.0); atomicAdd(es + 1, epr); atomicAdd(es + 2, (double)ep_step);
            atomicAdd(es + 3, coverage); atomicAdd(es + 4, (double)delivered);
        }
    }
    wave_sync();
}

// metrics only (after reset / observe): no counters advance
template <typename ObsT, int L>
__device__ __forceinline__ void score_only(Ctx<ObsT, L> &c, float *scalars_out) {
    const Params &p = c.p;
    bool with_bounty = false, tr = false;
    if (c.lane < p.Nt) { with_bounty = c.ti(c.lane, TI_BOUNTY) > 0; tr = c.tracked(c.lane) != 0; }
    const int n_tracked = __popcll(c.ballot(tr));
    const int n_bounty = __popcll(c.ballot(with_bounty));
    const int n_both = __popcll(c.ballot(tr && with_bounty));
    if (c.lane == 0 && scalars_out) {
        float *o = scalars_out + c.out * 8;
        const int delivered = c.ei(EI_DELIVERED);
        o[0] = 0.f; o[1] = 0.f; o[2] = c.ei(EI_DONE) != 0 ? 1.f : 0.f; o[3] = (float)((double)n_tracked / (double)p.Nt);
        o[4] = n_bounty > 0 ? (float)((double)n_both / (double)n_bounty) : 0.f;
        o[5] = delivered > 0 ? (float)(c.ep_delayed() / (p.reward_scale * (double)delivered)) : 0.f;
        o[6] = (float)delivered; o[7] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// Phase D: joint_observation (environment.py:908-964): fill the per-environment scratch, then gather.
// `last_gw`: the packed goal word (goal, cargo weight, known-empty warehouses) this lane's target had when its slots were
// last written -- the fused rollout carries it from step to step and rewrites the ten slots it determines only when
// it changes (a pick-up, a delivery, a newly seen empty warehouse); -1 = write.
template <typename ObsT, int L>
__device__ __forceinline__ void fill_scratch(Ctx<ObsT, L> &c, int &last_gw) {
    const Params &p = c.p;
    const int lane = c.lane;
    const int tgt_mode = (c.obs_mode() >> 2) & 3;
    if (lane < p.Nt) {                        // Target.state(private=True), entities.py:631-637
        ObsT *sc = c.scratch + p.sc_tgt + lane * 14;
        const int gw = c.ti(lane, TI_GW) & 0xffffff;            // bit 24 (colliding) is not part of the observation
        sc[0] = (ObsT)c.tx(lane); sc[1] = (ObsT)c.ty(lane);
        if (!c.statics_done) {                                   // step_size / capacity, exact
            const int cap = 1 + (int)((c.capword() >> lane) & 1ull);
            sc[4] = (ObsT)(cap == 2 ? p.tgt_step * 0.5 : p.tgt_step); sc[5] = (ObsT)cap;
        }
      if (!c.statics_done || tgt_mode != 0 || gw != last_gw) {
        last_gw = gw;
        const int goal = (gw & 0xff) - 1, weight = (gw >> 8) & 0xff;
        sc[3] = (ObsT)(goal >= 0 && weight > 0 ? 1.0 : 0.0);
        int empty = (gw >> 16) & 0xf;
        if (tgt_mode == 1) {                  // EnhancedObservation: the true state of every warehouse (enhanced_observation.py:110-112)
            empty = 0;
            for (int w = 0; w < 4; ++w) {
                const int *row = &c.ei(EI_REMAINING + 4 * w);
                empty |= (int)!(row[0] || row[1] || row[2] || row[3]) << w;
            }
        } else if (tgt_mode == 2) {           // SharedFieldOfView: what any teammate knows (shared_field_of_view.py:122-127)
            for (int t = 0; t < p.Nt; ++t) empty |= (c.ti(t, TI_GW) >> 16) & 0xf;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            sc[6 + w] = (ObsT)(goal == w ? weight : 0);
            sc[10 + w] = (ObsT)((empty >> w) & 1);
        }
      }
    }
    if (!c.statics_done)
        for (int o = lane; o < p.No; o += L) {   // Obstacle.state, entities.py:147-148
            ObsT *sc = c.scratch + p.sc_obs + o * 3;
            sc[0] = (ObsT)c.obs_x(o); sc[1] = (ObsT)c.obs_y(o); sc[2] = (ObsT)c.obs_r(o);
        }
    // SharedFieldOfView: an entity is visible to the whole team when any member sees it
    // (shared_field_of_view.py:97-100, 117-120); flags live behind the mask flags, see build_descriptors
    if ((c.obs_mode() & 3) == 2) {
        if (lane < p.Nt) set_flag(c, p.fs_shared + lane, c.tracked(lane) != 0);
        for (int o = lane; o < p.No; o += L) {
            bool any = false;
            for (int cam = 0; cam < p.Nc; ++cam) any = any || ((c.camobs(cam) >> o) & 1ull);
            set_flag(c, p.fs_shared + p.Nt + o, any);
        }
    }
    if (tgt_mode == 2) {
        for (int j = lane; j < p.Nc + p.No; j += L) {
            bool any = false;
            for (int t = 0; t < p.Nt; ++t) any = any || c.mask_bit(p.bit_range + t * p.NJ + j);
            set_flag(c, p.fs_shared + p.Nt + p.No + j, any);
        }
    }
    wave_sync();
}

template <typename ObsT, int L>
__device__ __forceinline__ void fill_scratch(Ctx<ObsT, L> &c) {
    int always = -1;
    fill_scratch(c, always);
}

// The observation rows leave through this store: non-temporal (a write-once stream; -9 % kernel time against plain stores, profiles/HISTORY.md 3.1)
template <typename V>
__device__ __forceinline__ void stream_store(V v, V *dst) { MATE_ROW_STORE(v, dst); }
template <typename V>
__device__ __forceinline__ void stream_store(V v, __attribute__((address_space(1))) V *dst) { MATE_ROW_STORE(v, dst); }
template <typename ObsT> struct Vec;
template <> struct Vec<float> { using type = float4; static constexpr int W = 4; };
template <> struct Vec<double> { using type = double2; static constexpr int W = 2; };

// descriptor = byte offset of the source slot | byte offset of the visibility word << 16 (both inside the
// wave's LDS slice); the word is all-ones or zero, so a bitwise AND is the masked copy (exact +0.0 when hidden)
template <typename ObsT, int L>
__device__ __forceinline__ ObsT gather_one(const Ctx<ObsT, L> &c, uint32_t d) {
    using U = typename Bits<ObsT>::type;
    const U v = *reinterpret_cast<const U *>(c.base + (d & 0xffffu));
    const U m = *reinterpret_cast<const U *>(c.base + (d >> 16));
    const U r = v & m;
    ObsT out;
    __builtin_memcpy(&out, &r, sizeof(out));
    return out;
}

template <typename ObsT, int L>
__device__ __forceinline__ void pack_block(const Ctx<ObsT, L> &c, ObsT *dst, const uint32_t *table, int elems) {
    constexpr int W = Vec<ObsT>::W;
    using V = typename Vec<ObsT>::type;
    if ((elems % W) == 0) {                       // row block is 16-byte aligned for every environment
        V *out = reinterpret_cast<V *>(dst);
        const int nvec = elems / W;
        for (int i = c.lane; i < nvec; i += L) {
            if constexpr (W == 4) {
                const uint4 d = reinterpret_cast<const uint4 *>(table)[i];
                float4 v;
                v.x = gather_one(c, d.x); v.y = gather_one(c, d.y); v.z = gather_one(c, d.z); v.w = gather_one(c, d.w);
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4 nv = {v.x, v.y, v.z, v.w};
                stream_store(nv, reinterpret_cast<f32x4 *>(&out[i]));   // write-once stream: keep it out of the caches
            } else {
                const uint2 d = reinterpret_cast<const uint2 *>(table)[i];
                double2 v;
                v.x = gather_one(c, d.x); v.y = gather_one(c, d.y);
                typedef double f64x2 __attribute__((ext_vector_type(2)));
                const f64x2 nv = {v.x, v.y};
                stream_store(nv, reinterpret_cast<f64x2 *>(&out[i]));
            }
        }
    } else if (sizeof(ObsT) == 4 && (elems % 2) == 0) {      // blocks that are only 8-byte aligned (4v2: 202 floats per environment)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 *out = reinterpret_cast<f32x2 *>(dst);
        const uint2 *tab = reinterpret_cast<const uint2 *>(table);
        const int nvec = elems / 2;
        for (int base = c.lane; base < nvec; base += 4 * L) {
            uint2 d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int i = base + L * k; d[k] = tab[i < nvec ? i : 0]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(d[k].x), "+v"(d[k].y));   // descriptors before the first store
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = base + L * k;
                if (i < nvec) { const f32x2 v = {(float)gather_one(c, d[k].x), (float)gather_one(c, d[k].y)}; stream_store(v, &out[i]); }
            }
        }
    } else {
        for (int i = c.lane; i < elems; i += L) dst[i] = gather_one(c, table[i]);
    }
}

// Fused observation post-processing (RelativeCoordinates = agents/utils.py:40-94, RescaledObservation =
// agents/utils.py:97-137 of the reference): out = ((value - own coordinate) if visible else 0) * scale + bias.
template <typename ObsT, int L>
__device__ __forceinline__ void pack_block_xf(const Ctx<ObsT, L> &c, ObsT *dst, const uint2 *xdesc, const ObsT *xab, int elems) {
    using U = typename Bits<ObsT>::type;
    for (int i = c.lane; i < elems; i += L) {
        const uint2 d = xdesc[i];
        const ObsT v = *reinterpret_cast<const ObsT *>(c.base + (d.x & 0xffffu));
        const ObsT o = *reinterpret_cast<const ObsT *>(c.base + d.y);
        const U m = *reinterpret_cast<const U *>(c.base + (d.x >> 16));
        const ObsT rel = v - o;
        U bits;
        __builtin_memcpy(&bits, &rel, sizeof(bits));
        bits &= m;
        ObsT gated;
        __builtin_memcpy(&gated, &bits, sizeof(gated));
        dst[i] = gated * xab[2 * i] + xab[2 * i + 1];
    }
}

// Both row blocks of an f32 environment.  The descriptors of the first GC camera chunks and GT target chunks per
// lane (all of them for the shipped scenario shapes) are loaded BEFORE the first store: loads and stores share one
// in-order counter on gfx9, so a descriptor load issued behind an observation store can only be waited for
// together with that store's HBM acknowledgement -- seven such waits per step in a chunk-by-chunk loop.
// (GC / GT per kernel: the fused rollouts of a compiled shape hold exactly the chunks its rows have -- MATE-8v8-9 has five
// camera and five target chunks per lane; with the default two + six its other three camera chunks fetched their
// descriptors INSIDE the step loop, behind the stores: the pack phase of the fused Greedy rollout was a fifth of its step.)
constexpr int kPackGC = 2, kPackGT = 6;
template <int GC_, int GT_> struct PackDescriptorsT { static constexpr int GC = GC_, GT = GT_; uint4 dc[GC_ > 0 ? GC_ : 1], dt[GT_]; };
using PackDescriptors = PackDescriptorsT<kPackGC, kPackGT>;
constexpr int pack_chunks_per_lane(int elems) { return (elems / 4 + 63) / 64; }

template <typename ObsT, int L, typename D>
__device__ __forceinline__ void load_pack_descriptors(const Ctx<ObsT, L> &c, D &d) {
    constexpr int kPackGC = D::GC, kPackGT = D::GT;
    const Params &p = c.p;
    const int nvc = p.cam_elems / 4, nvt = p.tgt_elems / 4;
    const uint4 *tabc = reinterpret_cast<const uint4 *>(c.table);
    const uint4 *tabt = reinterpret_cast<const uint4 *>(c.table + p.tgt_table_off);
#pragma unroll
    for (int k = 0; k < kPackGC; ++k) { const int s = c.lane + L * k; d.dc[k] = tabc[s < nvc ? s : 0]; }
#pragma unroll
    for (int k = 0; k < kPackGT; ++k) { const int s = c.lane + L * k; d.dt[k] = tabt[s < nvt ? s : 0]; }
#pragma unroll
    for (int k = 0; k < kPackGC; ++k) asm volatile("" : "+v"(d.dc[k].x), "+v"(d.dc[k].y), "+v"(d.dc[k].z), "+v"(d.dc[k].w));
#pragma unroll
    for (int k = 0; k < kPackGT; ++k) asm volatile("" : "+v"(d.dt[k].x), "+v"(d.dt[k].y), "+v"(d.dt[k].z), "+v"(d.dt[k].w));
}

// `PART`: 3 both teams' rows (default), 1 the camera rows only, 2 the target rows only
template <int PART = 3, typename ObsT, int L, typename D>
__device__ __forceinline__ void pack_rows_f32(const Ctx<ObsT, L> &c, const D &d) {
    if constexpr (sizeof(ObsT) == 4) {
        const Params &p = c.p;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        constexpr int GC = D::GC, GT = D::GT;
        const int nvc = p.cam_elems / 4, nvt = p.tgt_elems / 4;
        const uint4 *tabc = reinterpret_cast<const uint4 *>(c.table);
        const uint4 *tabt = reinterpret_cast<const uint4 *>(c.table + p.tgt_table_off);
        f32x4 *cam = reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(c.g.cam_obs) + c.out * p.cam_elems);
        f32x4 *tgt = reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(c.g.tgt_obs) + c.out * p.tgt_elems);
        auto chunk = [&](const uint4 &d) { return f32x4{gather_one(c, d.x), gather_one(c, d.y), gather_one(c, d.z), gather_one(c, d.w)}; };
        if constexpr ((PART & 1) != 0) {
#pragma unroll
        for (int k = 0; k < GC; ++k) { const int s = c.lane + L * k; if (s < nvc) stream_store(chunk(d.dc[k]), &cam[s]); }
        for (int s = c.lane + L * GC; s < nvc; s += L) stream_store(chunk(tabc[s]), &cam[s]);   // larger scenarios
        }
        if constexpr ((PART & 2) != 0) {
#pragma unroll
        for (int k = 0; k < GT; ++k) { const int s = c.lane + L * k; if (s < nvt) stream_store(chunk(d.dt[k]), &tgt[s]); }
        for (int s = c.lane + L * GT; s < nvt; s += L) stream_store(chunk(tabt[s]), &tgt[s]);
        }
    }
}

template <typename ObsT, int L>
__device__ __forceinline__ bool packs_rows_f32(const Ctx<ObsT, L> &c) {
    const Params &p = c.p;
    return !c.xdesc() && sizeof(ObsT) == 4 && (p.cam_elems % 4) == 0 && (p.tgt_elems % 4) == 0 && c.has_tgt_obs() && (c.has_cam_obs() || p.cam_elems == 0);
}

template <typename ObsT, int L>
__device__ __forceinline__ void store_masks(const Ctx<ObsT, L> &c) {
    const Params &p = c.p;
    // (one round when the words fit a wave -- every shipped scenario: the general loop's trip-count arithmetic is a dozen
    // vector instructions per step)
    if (c.g.masks) {
        uint32_t *m = c.g.masks + c.out * p.MW;
        if (p.MW <= L) { if (c.lane < p.MW) m[c.lane] = c.mask[c.lane]; }
        else for (int i = c.lane; i < p.MW; i += L) m[i] = c.mask[i];
    }
    if (c.g.own_masks) {
        uint32_t *m = c.g.own_masks + c.env * p.MW;
        if (p.MW <= L) { if (c.lane < p.MW) m[c.lane] = c.mask[c.lane]; }
        else for (int i = c.lane; i < p.MW; i += L) m[i] = c.mask[i];
    }
}

// Both f32 row blocks of a shape whose rows are NOT whole 16-byte chunks everywhere (MATE-4v2-*: two target rows of 101 floats; the
// 1vN shapes), with EVERY descriptor of the lane in registers before the lane's first store.  pack_block fetches its descriptors inside
// its loop: on gfx9 loads and stores retire through one in-order counter, so a descriptor load issued behind a row store is waited
// for together with that store's HBM acknowledgement -- the PMC passes of the four-per-wave MATE-4v2-9 rollout showed waves waiting
// 62 % of their cycles with 231 vector instructions per environment-step (profiles/r06_pmc_summary.json).  Compiled shapes only
// (the counts are literals, the arrays registers); a block takes 16-byte chunks when its element count is a multiple of 4, 8-byte
// chunks when it is even.  load_odd_descriptors fetches them at the top of every step's pack, ahead of the step's stores
// (pack_blocks_prefetched); holding them for the whole launch instead -- 38 registers on MATE-4v2-9 in sixteen-lane groups -- measured
// no faster (0.493 against 0.496 of the roofline at 16 384 environments) and was dropped.  False = not applicable, the caller runs
// pack_block's loops.
constexpr int kOddCap = 8;                                       // descriptors of a block per lane
struct OddDescriptors { uint4 dc[kOddCap], dt[kOddCap]; };
template <typename ObsT, int L>
__device__ __forceinline__ bool odd_rows_apply(const Ctx<ObsT, L> &c, int &wc, int &wt, int &nc, int &nt) {
    const Params &p = c.p;
    wc = p.cam_elems == 0 ? 4 : (p.cam_elems % 4 == 0 ? 4 : (p.cam_elems % 2 == 0 ? 2 : 0));
    wt = p.tgt_elems % 4 == 0 ? 4 : (p.tgt_elems % 2 == 0 ? 2 : 0);
    if (sizeof(ObsT) != 4 || wc == 0 || wt == 0) return false;
    nc = p.cam_elems / wc; nt = p.tgt_elems / wt;
    return nc <= kOddCap * L && nt <= kOddCap * L && c.has_tgt_obs() && (p.cam_elems == 0 || c.has_cam_obs()) && !c.xdesc();
}
__device__ __forceinline__ void pin_odd_descriptors(OddDescriptors &d, int nc, int nt, int L) {
#pragma unroll
    for (int k = 0; k < kOddCap; ++k) {
        if (L * k < nc) asm volatile("" : "+v"(d.dc[k].x), "+v"(d.dc[k].y), "+v"(d.dc[k].z), "+v"(d.dc[k].w));
        if (L * k < nt) asm volatile("" : "+v"(d.dt[k].x), "+v"(d.dt[k].y), "+v"(d.dt[k].z), "+v"(d.dt[k].w));
    }
}
template <typename ObsT, int L>
__device__ __forceinline__ bool load_odd_descriptors(const Ctx<ObsT, L> &c, OddDescriptors &d) {
    int wc, wt, nc, nt;
    if (!odd_rows_apply(c, wc, wt, nc, nt)) return false;
    const uint32_t *tabc = c.table, *tabt = c.table + c.p.tgt_table_off;
#pragma unroll
    for (int k = 0; k < kOddCap; ++k) {
        const int i = c.lane + L * k, ic = i < nc ? i : 0, it = i < nt ? i : 0;
        d.dc[k] = make_uint4(0u, 0u, 0u, 0u); d.dt[k] = make_uint4(0u, 0u, 0u, 0u);
        if (L * k < nc) { if (wc == 4) d.dc[k] = reinterpret_cast<const uint4 *>(tabc)[ic]; else { const uint2 q = reinterpret_cast<const uint2 *>(tabc)[ic]; d.dc[k].x = q.x; d.dc[k].y = q.y; } }
        if (L * k < nt) { if (wt == 4) d.dt[k] = reinterpret_cast<const uint4 *>(tabt)[it]; else { const uint2 q = reinterpret_cast<const uint2 *>(tabt)[it]; d.dt[k].x = q.x; d.dt[k].y = q.y; } }
    }
    pin_odd_descriptors(d, nc, nt, L);                           // every descriptor before the first store
    return true;
}
template <typename ObsT, int L>
__device__ __forceinline__ void pack_blocks_odd(const Ctx<ObsT, L> &c, const OddDescriptors &d) {
    if constexpr (sizeof(ObsT) == 4) {
        const Params &p = c.p;
        int wc, wt, nc, nt;
        (void)odd_rows_apply(c, wc, wt, nc, nt);
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        float *cam = reinterpret_cast<float *>(c.g.cam_obs) + c.out * p.cam_elems, *tgt = reinterpret_cast<float *>(c.g.tgt_obs) + c.out * p.tgt_elems;
#pragma unroll
        for (int k = 0; k < kOddCap; ++k) {
            const int i = c.lane + L * k;
            if (L * k < nc && i < nc) {
                if (wc == 4) stream_store(f32x4{gather_one(c, d.dc[k].x), gather_one(c, d.dc[k].y), gather_one(c, d.dc[k].z), gather_one(c, d.dc[k].w)}, reinterpret_cast<f32x4 *>(cam) + i);
                else stream_store(f32x2{gather_one(c, d.dc[k].x), gather_one(c, d.dc[k].y)}, reinterpret_cast<f32x2 *>(cam) + i);
            }
        }
#pragma unroll
        for (int k = 0; k < kOddCap; ++k) {
            const int i = c.lane + L * k;
            if (L * k < nt && i < nt) {
                if (wt == 4) stream_store(f32x4{gather_one(c, d.dt[k].x), gather_one(c, d.dt[k].y), gather_one(c, d.dt[k].z), gather_one(c, d.dt[k].w)}, reinterpret_cast<f32x4 *>(tgt) + i);
                else stream_store(f32x2{gather_one(c, d.dt[k].x), gather_one(c, d.dt[k].y)}, reinterpret_cast<f32x2 *>(tgt) + i);
            }
        }
    }
}
template <typename ObsT, int L>
__device__ __forceinline__ bool pack_blocks_prefetched(const Ctx<ObsT, L> &c) {
    OddDescriptors d;
    if (!load_odd_descriptors(c, d)) return false;
    pack_blocks_odd(c, d);
    return true;
}

// HELD: the caller loaded this lane's descriptors before (the rollout kernel, once per launch); otherwise they are
// loaded here.  A template switch, not a pointer: a nullable pointer to the register array would force it into memory.
// PREFETCH (the fused rollouts of the compiled shapes): rows that are not whole 16-byte chunks go through pack_blocks_prefetched.
template <bool HELD, bool PREFETCH = false, typename ObsT, int L, typename D>
__device__ __forceinline__ void pack_observations(Ctx<ObsT, L> &c, D &held) {
    const Params &p = c.p;
    if (c.xdesc()) {
        const ObsT *xab = reinterpret_cast<const ObsT *>(c.g.xab);
        if (c.has_cam_obs() && p.cam_elems > 0)
            pack_block_xf<ObsT>(c, reinterpret_cast<ObsT *>(c.g.cam_obs) + c.out * p.cam_elems, c.g.xdesc, xab, p.cam_elems);
        if (c.has_tgt_obs())
            pack_block_xf<ObsT>(c, reinterpret_cast<ObsT *>(c.g.tgt_obs) + c.out * p.tgt_elems, c.g.xdesc + p.tgt_table_off,
                                xab + 2 * p.tgt_table_off, p.tgt_elems);
    } else if (packs_rows_f32(c)) {
        if constexpr (!HELD) load_pack_descriptors(c, held);
        pack_rows_f32(c, held);
    } else {
    bool packed = false;
    if constexpr (PREFETCH) packed = pack_blocks_prefetched(c);
    if (!packed) {
    if (c.has_cam_obs() && p.cam_elems > 0)
        pack_block<ObsT>(c, reinterpret_cast<ObsT *>(c.g.cam_obs) + c.out * p.cam_elems, c.table, p.cam_elems);
    if (c.has_tgt_obs())
        pack_block<ObsT>(c, reinterpret_cast<ObsT *>(c.g.tgt_obs) + c.out * p.tgt_elems, c.table + p.tgt_table_off, p.tgt_elems);
    }
    }
    store_masks(c);
}

// =============================================================================================
// Row-image mode of the fused rollouts (FixedShape<..., IMAGE = true>).
//
// joint_observation (environment.py:908-964) builds every row as [preserved | own private state | one block per other
// entity: its public state and a 1.0, or zeros when the viewer does not see it].  The descriptor packer above gathers those
// 1552 floats (MATE-4v8-9) element by element -- source slot, flag word, AND -- at every step: 13 vector instructions and 8 LDS
// reads per 16-byte chunk, 130 of a step's ~780 vector instructions with the scratch and flag writes that feed it, although
// a third of the elements never change inside an episode and every block shares ONE flag.  Here the environment's rows
// live in LDS for the whole launch (6.2 KB next to 3 KB of state: sixteen environments per CU still fit), laid out
// exactly as in the output buffers:
//   * image_statics, once per launch: the preserved blocks, the static parts of the private states, the camera rows'
//     obstacle blocks (camera_obstacle_view_mask is static), the static fields of the public-state table;
//   * per step, the lanes that own the kinematics write the three / three-to-eleven floats of a camera's / target's private
//     and public state that changed (simulate_cameras, image_targets);
//   * per step, THE LANE THAT MADE A VISIBILITY TEST WRITES ITS BLOCK (image_blocks): the (viewer, other) pair's public state
//     AND-ed with the verdict it holds in a register -- two 16-byte LDS reads, 4-7 ANDs, 2-4 LDS writes per pair, the block's
//     address held in a register since the launch began;
//   * image_store streams the rows out: one 16-byte LDS read and one non-temporal 16-byte store per chunk, no arithmetic.
// Same values, bit for bit, as the descriptor packer (tested against it and against the oracle).
template <typename ObsT>
__device__ __forceinline__ void image_statics(Ctx<ObsT> &c) {
    if constexpr (sizeof(ObsT) == 4) {
    const Params &p = c.p;
    const int lane = c.lane;
    for (int i = lane; i < (p.Nc + p.Nt) * 13; i += 64) {                        // preserved block of every row (environment.py:499-501, 941)
        const int row = (int)(((float)i + 0.5f) * (1.0f / 13.0f)), col = i - row * 13;
        const int agent = row < p.Nc ? row : row - p.Nc;
        // [Nc, Nt, No, agent index, the four warehouses' centres (+,+) (-,+) (-,-) (+,-) (constants.py:70-72), their half width]: made here --
        // read from the constants table in global memory, three dependent round trips to L2 stood at the head of every launch
        float v;
        if (col < 4) v = (float)(col == 0 ? p.Nc : col == 1 ? p.Nt : col == 2 ? p.No : agent);
        else if (col < 12) {
            const int k = col - 4, w = k >> 1;
            const bool plus = (k & 1) ? w < 2 : (w == 0 || w == 3);
            v = plus ? (float)kWarehouseCenter : -(float)kWarehouseCenter;
        } else v = (float)kWarehouseRadius;
        (row < p.Nc ? c.img_cam_row(row) : c.img_tgt_row(agent))[col] = v;
    }
    if (lane < p.Nc) {                                                           // Camera.state(private), entities.py:313-324
        float *pc = c.pub_cam(lane), *row = c.img_cam_row(lane) + 13;
        const float x = (float)c.cam_x(lane), y = (float)c.cam_y(lane), r = (float)p.cam_radius;
        pc[0] = x; pc[1] = y; pc[2] = r; pc[6] = 1.0f; pc[7] = 0.0f;
        row[0] = x; row[1] = y; row[2] = r; row[6] = (float)p.rmax; row[7] = (float)p.rot; row[8] = (float)p.zoom;
    }
    if (lane < p.Nt) {                                                           // Target.state(private), entities.py:631-637
        float *pt = c.pub_tgt(lane), *row = c.img_tgt_row(lane) + 13;
        const int cap = 1 + (int)((c.capword() >> lane) & 1ull);
        pt[2] = (float)p.tgt_sight; pt[4] = 1.0f; pt[5] = 0.0f; pt[6] = 0.0f; pt[7] = 0.0f;
        row[2] = (float)p.tgt_sight; row[4] = (float)(cap == 2 ? p.tgt_step * 0.5 : p.tgt_step); row[5] = (float)cap;
    }
    for (int o = lane; o < p.No; o += 64) {                                      // Obstacle.state, entities.py:147-148
        float *po = c.pub_obs(o);
        po[0] = (float)c.obs_x(o); po[1] = (float)c.obs_y(o); po[2] = (float)c.obs_r(o); po[3] = 1.0f;
    }
    if (lane < 4) c.pub_obs(p.No)[lane] = 0.0f;                                  // (what the second 16-byte read of the last obstacle's block sees)
    wave_sync();
    for (int q = lane; q < p.Nc * p.No; q += 64) {                               // camera rows: obstacle blocks, gated by the static mask
        const int cam = (int)(((float)q + 0.5f) * p.inv_No), o = q - cam * p.No;
        const uint32_t m = ((c.camobs(cam) >> o) & 1ull) ? ~0u : 0u;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(c.pub_obs(o));
        uint32_t *dst = reinterpret_cast<uint32_t *>(c.img_cam_row(cam) + 22 + 5 * p.Nt + 4 * o);
        dst[0] = src[0] & m; dst[1] = src[1] & m; dst[2] = src[2] & m; dst[3] = src[3] & m;
    }
    wave_sync();
    }
}

// The targets' private and public state (the row-image counterpart of fill_scratch); `last_gw` as there.
template <typename ObsT>
__device__ __forceinline__ void image_targets(Ctx<ObsT> &c, int &last_gw) {
    if constexpr (sizeof(ObsT) == 4) {
    const Params &p = c.p;
    const int lane = c.lane;
    if (lane < p.Nt) {
        float *pt = c.pub_tgt(lane), *row = c.img_tgt_row(lane) + 13;
        const int gw = c.ti(lane, TI_GW) & 0xffffff;            // bit 24 (colliding) is not part of the observation
        const float x = (float)c.tx(lane), y = (float)c.ty(lane);
        pt[0] = x; pt[1] = y; row[0] = x; row[1] = y;
        if (gw != last_gw) {
            last_gw = gw;
            const int goal = (gw & 0xff) - 1, weight = (gw >> 8) & 0xff, empty = (gw >> 16) & 0xf;
            const float loaded = goal >= 0 && weight > 0 ? 1.0f : 0.0f;
            pt[3] = loaded; row[3] = loaded;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                row[6 + w] = (float)(goal == w ? weight : 0);
                row[10 + w] = (float)((empty >> w) & 1);
            }
        }
    }
    wave_sync();
    }
}

// Every lane writes the blocks of the visibility pairs it tested (`seen`: update_view's seen_out).
template <typename ObsT>
__device__ __forceinline__ void image_blocks(Ctx<ObsT> &c, const RangeRoles &roles, uint32_t seen) {
    const Params &p = c.p;
    // all public states first, by every lane (a lane without a block in a slot reads offset 0 of the slice: harmless), so that
    // the slots' LDS round trips overlap instead of following one another under four different execution masks
    uint4 a[1 + kRoleRounds], b[1 + kRoleRounds];
#pragma unroll
    for (int slot = 0; slot < 1 + kRoleRounds; ++slot) {
        if (slot > p.range_rounds) break;
        const uint4 *src = reinterpret_cast<const uint4 *>(c.base + (roles.block[slot] & 0xffffu));
        a[slot] = src[0]; b[slot] = src[1];
    }
#pragma unroll
    for (int slot = 0; slot < 1 + kRoleRounds; ++slot) {
        if (slot > p.range_rounds) break;
        const uint32_t code = (roles.block_bits >> (2 * slot)) & 3u;
        if (code != 0u) {
            const uint32_t m = ((seen >> slot) & 1u) ? ~0u : 0u;
            uint32_t *dst = reinterpret_cast<uint32_t *>(c.base + (roles.block[slot] >> 16));
            dst[0] = a[slot].x & m; dst[1] = a[slot].y & m; dst[2] = a[slot].z & m; dst[3] = a[slot].w & m;
            if (code >= 2u) dst[4] = b[slot].x & m;
            if (code == 3u) { dst[5] = b[slot].y & m; dst[6] = b[slot].z & m; }
        }
    }
    wave_sync();
}

// The rows as they lie in LDS, 16 bytes per lane and chunk, to the output buffers (write-once stream: non-temporal).
// `cam_low` / `tgt_low`: 16-byte chunk index of the two output BLOCKS' bases inside their cache lines ((address >> 4) & 7), read once
// per launch -- the shift of a row follows from it and the row's index without waiting for the pointer itself.
template <bool SHIFTED, typename ObsT>
__device__ __forceinline__ void image_store_form(const Ctx<ObsT> &c, uint32_t cam_low, uint32_t tgt_low) {
    if constexpr (sizeof(ObsT) == 4) {
    const Params &p = c.p;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int nvc = p.cam_elems / 4, nvt = p.tgt_elems / 4;
    const f32x4 *src_c = reinterpret_cast<const f32x4 *>(c.img), *src_t = reinterpret_cast<const f32x4 *>(c.img + p.cam_elems);
    // SHIFTED: every store instruction covers a 128-byte-ALIGNED kilobyte of the output -- a row begins at a multiple of 32 bytes, not
    // of a cache line, so the lanes' chunks are shifted by the row's offset inside its first line (wave-uniform: 0..7 chunks).
    // Unshifted, every instruction straddles nine lines and the two partial ones are written again by its neighbour: the store
    // pattern BY ITSELF is 5 % slower (tools/store_roof.hip).  In the kernel the shifted form -- one store instruction and two
    // execution masks more -- is 1.3-2 % slower where the blocks take the rows fast (the arithmetic bounds the launch) and 3 %
    // faster where they do not (the stores do): Ptrs::store_shifted picks the form per launch (mate_engine_set_store_form).
    // All LDS reads of a block before its first store (a store issued between them would be waited for with them).
    constexpr int GC = 3, GT = 7;       // (image_fits shapes: at most 128 camera chunks and 384 target chunks + 7 of shift; asserted by the host)
    const int lane = c.lane & 63;       // (the range, for the compiler)
    constexpr int slack = SHIFTED ? 7 : 0;
    const int sc = SHIFTED ? (int)((cam_low + (uint32_t)c.out * (uint32_t)nvc) & 7u) : 0, st = SHIFTED ? (int)((tgt_low + (uint32_t)c.out * (uint32_t)nvt) & 7u) : 0;
    // (the shift goes into the wave-uniform bases; the LDS reads are unconditional -- a lane outside its row reads a neighbouring
    // part of the slice, or zeros past the workgroup's LDS, and stores nothing; only the first round needs the lower bound and
    // only the rounds that can reach the row's end the upper one)
    const f32x4 *from_c = src_c - sc, *from_t = src_t - st;
    // a target block of whole 8-byte chunks only (image_fits: at most 512 floats, four rounds; never shifted: its rows begin on
    // 8-byte boundaries that alternate with the environment's parity)
    const bool tgt8 = (p.tgt_elems % 4) != 0;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int GT2 = 4;
    const int nv2 = p.tgt_elems / 2;
    f32x2 vt2[GT2];
    f32x4 vc[GC], vt[GT];
#pragma unroll
    for (int k = 0; k < GC; ++k) if (64 * k - slack < nvc) vc[k] = from_c[lane + 64 * k];
    if (tgt8) {
        const f32x2 *from_t2 = reinterpret_cast<const f32x2 *>(c.img + p.cam_elems);
#pragma unroll
        for (int k = 0; k < GT2; ++k) if (64 * k < nv2) vt2[k] = from_t2[lane + 64 * k];
    } else {
#pragma unroll
    for (int k = 0; k < GT; ++k) if (64 * k - slack < nvt) vt[k] = from_t[lane + 64 * k];
    }
    // The blocks' base pointers come from the kernel-argument segment (a scalar load): first USED here, behind the LDS reads, so
    // that one wait covers both -- used earlier, the reads would queue up behind the pointers' round trip (+350 cycles per step).
    // (as integers through the barrier, and back as GLOBAL pointers: a generic pointer out of an asm stores through flat_store)
    typedef __attribute__((address_space(1))) f32x4 global_f32x4;
    uint64_t cam_base = reinterpret_cast<uint64_t>(c.g.cam_obs), tgt_base = reinterpret_cast<uint64_t>(c.g.tgt_obs);
    if (tgt8) asm volatile("" : "+s"(cam_base), "+s"(tgt_base), "+v"(vt2[0]));
    else asm volatile("" : "+s"(cam_base), "+s"(tgt_base), "+v"(vt[0]));
    // (what comes out of an asm counts as divergent: say again that it is not, or the row arithmetic runs on the vector unit)
    cam_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(cam_base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)cam_base);
    tgt_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(tgt_base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)tgt_base);
    global_f32x4 *to_c = (global_f32x4 *)(cam_base + (uint64_t)c.out * (uint64_t)(p.cam_elems * 4)) - sc;
    global_f32x4 *to_t = (global_f32x4 *)(tgt_base + (uint64_t)c.out * (uint64_t)(p.tgt_elems * 4)) - st;
#pragma unroll
    for (int k = 0; k < GC; ++k) if (64 * k - slack < nvc) {
        const bool inside = (k > 0 || lane >= sc) && (64 * (k + 1) <= nvc || lane + 64 * k - sc < nvc);
        if (inside) stream_store(vc[k], &to_c[lane + 64 * k]);
    }
    if (tgt8) {
        typedef __attribute__((address_space(1))) f32x2 global_f32x2;
        global_f32x2 *to_t2 = (global_f32x2 *)(tgt_base + (uint64_t)c.out * (uint64_t)(p.tgt_elems * 4));
#pragma unroll
        for (int k = 0; k < GT2; ++k) if (64 * k < nv2) { if (lane + 64 * k < nv2) stream_store(vt2[k], &to_t2[lane + 64 * k]); }
    } else {
#pragma unroll
    for (int k = 0; k < GT; ++k) if (64 * k - slack < nvt) {
        const bool inside = (k > 0 || lane >= st) && (64 * (k + 1) <= nvt || lane + 64 * k - st < nvt);
        if (inside) stream_store(vt[k], &to_t[lane + 64 * k]);
    }
    }
    }
}


template <typename ObsT>
__device__ __forceinline__ void image_store(const Ctx<ObsT> &c, uint32_t cam_low, uint32_t tgt_low) {
    if (c.g.store_shifted) image_store_form<true>(c, cam_low, tgt_low);      // (wave-uniform: a launch argument)
    else image_store_form<false>(c, cam_low, tgt_low);
}
template <typename ObsT>
__device__ __forceinline__ void image_store(const Ctx<ObsT> &c) {
    image_store(c, (uint32_t)(reinterpret_cast<uintptr_t>(c.g.cam_obs) >> 4) & 7u, (uint32_t)(reinterpret_cast<uintptr_t>(c.g.tgt_obs) >> 4) & 7u);
}

// =============================================================================================
// Register-resident step state of the fused random-policy rollout (row-image shapes).
//
// A step of one environment is a chain of ~1200 dependent instructions that a lone wave walks in ~12 k cycles, and four
// waves per SIMD add only a quarter to that: what a step costs is instructions AND LDS hand-offs (write, s_waitcnt, read: ~100
// cycles each, ~40 of them per step).  Most of those hand-offs pass a value from a lane to ITSELF one phase later -- a
// target's position from the kinematics to the warehouse test to the observation row, its bounty / goal word / step
// counters from one step's bookkeeping to the next, the tracked bit from the ballot to the reward.  Here those values stay in
// the registers of the lane that owns them for the whole launch (HeldState; target t on lane Nc + t, where its kinematics
// run; the environment's reward sums and counters on lane 0), LDS keeps what OTHER lanes read (the entity table, the
// cameras' angles, the masks), and the order-dependent goal logistics -- rare: a target stands in a warehouse -- runs on the
// LDS record as before, between a spill and a reload.  held_store puts everything back before the record is stored.
struct HeldState {
    double x, y;                                      // target lanes: position
    int32_t bounty, freight, gw, tsteps, trsteps;     // target lanes: the TI_* words of the record
    double ph, th;                                    // camera lanes: orientation, viewing angle
    double ep_reward, ep_delayed;                     // lane 0: the episode's reward sums ...
    int32_t epstep, delivered, awaiting;              // ... its step counter, delivered cargoes, "cargo still awaited"
    int32_t tick;                                     // lane 0: EI_TICK (the tick after the last executed step)
};

template <typename ObsT>
__device__ __forceinline__ void held_load(Ctx<ObsT> &c, HeldState &h) {
    const Params &p = c.p;
    const int t = c.lane - p.Nc;
    h = HeldState{};
    if (t >= 0 && t < p.Nt) {
        h.x = c.tx(t); h.y = c.ty(t);
        h.bounty = c.ti(t, TI_BOUNTY); h.freight = c.ti(t, TI_FREIGHT); h.gw = c.ti(t, TI_GW);
        h.tsteps = c.ti(t, TI_TSTEPS); h.trsteps = c.ti(t, TI_TRSTEPS);
    }
    if (c.lane < p.Nc) { h.ph = c.phi(c.lane); h.th = c.theta(c.lane); }
    if (c.lane == 0) {
        h.ep_reward = c.ep_reward(); h.ep_delayed = c.ep_delayed();
        h.epstep = c.ei(EI_EPSTEP); h.delivered = c.ei(EI_DELIVERED); h.tick = c.ei(EI_TICK);
        h.awaiting = (c.ei(EI_AWAITING) | c.ei(EI_AWAITING + 1) | c.ei(EI_AWAITING + 2) | c.ei(EI_AWAITING + 3)) != 0;
    }
}

template <typename ObsT>
__device__ __forceinline__ void held_store(Ctx<ObsT> &c, const HeldState &h) {
    const Params &p = c.p;
    const int t = c.lane - p.Nc;
    if (t >= 0 && t < p.Nt) {
        c.tx(t) = h.x; c.ty(t) = h.y;
        c.ti(t, TI_BOUNTY) = h.bounty; c.ti(t, TI_FREIGHT) = h.freight; c.ti(t, TI_GW) = h.gw;
        c.ti(t, TI_TSTEPS) = h.tsteps; c.ti(t, TI_TRSTEPS) = h.trsteps;
    }
    if (c.lane == 0) {
        c.ep_reward() = h.ep_reward; c.ep_delayed() = h.ep_delayed;
        c.ei(EI_EPSTEP) = h.epstep; c.ei(EI_TICK) = h.tick;
    }
    wave_sync();
}

// Camera.simulate with the angles in registers (simulate_cameras is the LDS form: same arithmetic, same stores for the others)
template <typename ObsT>
__device__ __forceinline__ void simulate_cameras_held(Ctx<ObsT> &c, const StepDraws &draws, HeldState &h) {
    if constexpr (sizeof(ObsT) == 4) {
    const Params &p = c.p;
    const int lane = c.lane;
    if (lane < p.Nc) {
        const double da = clip_uniform(draws.a0, -p.rot, p.rot);
        const double dz = clip_uniform(draws.a1, -p.zoom, p.zoom);
        const double ph = normalize_angle(h.ph + da);
        const double th = clip_uniform(h.th + dz, p.theta_min, kMaxViewingAngle);
        h.ph = ph; h.th = th;
        c.phi(lane) = ph; c.theta(lane) = th;                  // (the sector tests of the pair lanes read these)
        const double sr2 = div_nz(p.area, th);
        c.sight2(lane) = sr2;
        float sn, cs;
        sincos_deg_f32(ph, sn, cs);
        const float srf = sqrt_f32_1ulp((float)sr2);
        float *pc = c.pub_cam(lane), *row = c.img_cam_row(lane) + 13;
        const float x = srf * cs, y = srf * sn, tf = (float)th;
        pc[3] = x; pc[4] = y; pc[5] = tf; row[3] = x; row[4] = y; row[5] = tf;
    }
    }
}

// Target.simulate with the position in registers and the screen carried from the previous step (simulate_targets)
template <typename ObsT>
__device__ __forceinline__ void simulate_targets_held(Ctx<ObsT> &c, const StepDraws &draws, const NearCarry &carried, HeldState &h) {
    const Params &p = c.p;
    const int t = c.lane - p.Nc;
    if (t >= 0 && t < p.Nt) {
        const double ax = draws.a0, ay = draws.a1;
        const double step_size = ((c.capword() >> t) & 1ull) ? p.tgt_step * 0.5 : p.tgt_step;   // entities.py:612-615
        const double ox = h.x, oy = h.y;
        double vx = ax, vy = ay;
        double n = norm2(ax, ay);
#ifdef MATE_POLAR_CLAMP
        if (n > step_size) { set_norm_polar(vx, vy, step_size); n = step_size; }
#else
        if (n > step_size) { const double k = div_nz(step_size, n); vx = ax * k; vy = ay * k; n = step_size; }      // entities.py:649-650
#endif
        const double desx = ox + vx, desy = oy + vy;
        uint64_t todo = near_field(p, carried, t);
        bool n_known = true;
        SUB_ACC(c, 5);                                   // targets: the step vector
        SUB_COUNT(c, 2, todo != 0ull);                   // steps with a collision candidate
        while (todo) {
            const int k = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            double cx, cy, cr;
            c.circle(k, cx, cy, cr);
            if (!n_known) { n = norm2(vx, vy); n_known = true; }
            const double dx = cx - ox, dy = cy - oy;
            const double reach = n + cr;
            if (n != 0.0 && fma(dy, dy, dx * dx) > reach * reach * (1.0 + 1e-12)) continue;
            obstruct_tangential(ox, oy, vx, vy, n, n_known, cx, cy, cr);
        }
        SUB_ACC(c, 6);                                   // targets: the candidates' circles
        const double nx = clip_uniform(ox + vx, -kTerrain, kTerrain);   // entities.py:664-666
        const double ny = clip_uniform(oy + vy, -kTerrain, kTerrain);
        const bool colliding = (fabs(nx - desx) > 1e-6) || (fabs(ny - desy) > 1e-6);  // entities.py:668
        SUB_COUNT(c, 3, colliding);                      // steps in which a target was deflected
        h.x = nx; h.y = ny;
        const int slot = c.tgt_slot(t);
        c.ex[slot] = nx; c.ey[slot] = ny; c.exf[slot] = (float)nx; c.eyf[slot] = (float)ny;
        h.gw = (h.gw & ~(1 << 24)) | ((int)colliding << 24);
    }
    wave_sync();
}

// tracked bit and warehouse of the lane's target from the sector ballot and the position in registers (the tail of update_view)
template <typename ObsT>
__device__ __forceinline__ void view_tail_held(const Ctx<ObsT> &c, unsigned long long sector_ballot, const HeldState &h, bool &tracked, int &inside) {
    const Params &p = c.p;
    const int t = c.lane - p.Nc;
    unsigned long long any = 0ull;                       // tracked_bits = camera_target_view_mask.any(axis=0), environment.py:1388
    for (int cam = 0; cam < p.Nc; ++cam) any |= sector_ballot >> (cam * p.Nt);
    tracked = t >= 0 && t < p.Nt && ((any >> (t & 63)) & 1ull);
    const bool px = h.x > 0.0, py = h.y > 0.0;           // (see update_view: only the warehouse of the target's quadrant can hold it)
    const double wx = px ? kWarehouseCenter : -kWarehouseCenter, wy = py ? kWarehouseCenter : -kWarehouseCenter;
    const double sup = fmax(fabs(h.x - wx), fabs(h.y - wy));
    inside = (t >= 0 && t < p.Nt && sup <= kWarehouseRadius) ? (px ? (py ? 0 : 3) : (py ? 1 : 2)) : -1;
}

// assign_and_score on the registers; returns 1 when the episode ended at this step (wave-uniform)
template <typename ObsT>
__device__ __forceinline__ int assign_and_score_held(Ctx<ObsT> &c, uint32_t tick, float *scalars_out, HeldState &h, bool tracked, int inside) {
    const Params &p = c.p;
    const int lane = c.lane, t = lane - p.Nc;
    const bool is_target = t >= 0 && t < p.Nt;
    const int tr = (int)tracked;
    bool penal = false;
    if (is_target) {
        penal = tr && h.bounty > 0;                            // environment.py:1275
        const int nb = h.bounty - tr;
        h.bounty = nb > 0 ? nb : 0;                            // environment.py:1276
    }
    const int n_penal = __popcll(__ballot(penal));
    double reward = -(double)n_penal, delayed = 0.0;
    const unsigned long long inside_lanes = __ballot(inside >= 0);      // (target t on lane Nc + t)
    if (inside_lanes != 0ull) {                                // rare: through the LDS record, with the shared serial code
        if (is_target) {
            c.ti(t, TI_BOUNTY) = h.bounty; c.ti(t, TI_FREIGHT) = h.freight; c.ti(t, TI_GW) = h.gw;
            c.ti(t, TI_TSTEPS) = h.tsteps; c.ti(t, TI_TRSTEPS) = h.trsteps;
            c.inside(t) = inside;
        }
        wave_sync();
        if (lane == 0) {
            goal_logistics(c, tick, reward, delayed, (uint32_t)(inside_lanes >> p.Nc));
            c.xch(0) = __double2hiint(reward); c.xch(1) = __double2loint(reward);
            c.xch(2) = __double2hiint(delayed); c.xch(3) = __double2loint(delayed);
        }
        wave_sync();
        reward = __hiloint2double(c.xch(0), c.xch(1));
        delayed = __hiloint2double(c.xch(2), c.xch(3));
        if (is_target) {
            h.bounty = c.ti(t, TI_BOUNTY); h.freight = c.ti(t, TI_FREIGHT); h.gw = c.ti(t, TI_GW);
            h.tsteps = c.ti(t, TI_TSTEPS); h.trsteps = c.ti(t, TI_TRSTEPS);
        }
        if (lane == 0) {
            h.delivered = c.ei(EI_DELIVERED);
            h.awaiting = (c.ei(EI_AWAITING) | c.ei(EI_AWAITING + 1) | c.ei(EI_AWAITING + 2) | c.ei(EI_AWAITING + 3)) != 0;
        }
    }
    // metrics (environment.py:966-979), counters (environment.py:626-632)
    const bool with_bounty = is_target && h.bounty > 0;
    if (is_target) { h.tsteps += 1; h.trsteps += tr; }
    const int n_tracked = __popcll(__ballot(tracked));
    const int n_bounty = __popcll(__ballot(with_bounty));
    const int n_both = __popcll(__ballot(tracked && with_bounty));
    int done = 0;
    if (lane == 0) {
        const double epr = h.ep_reward + reward;
        const double epd = h.ep_delayed + delayed;
        h.ep_reward = epr; h.ep_delayed = epd;
        const int delivered = h.delivered;
        const double coverage = div_by_count((double)n_tracked, p.Nt);
        const double real_cov = n_bounty > 0 ? div_nz((double)n_both, (double)n_bounty) : 0.0;
        const double transport = delivered > 0 ? div_nz(epd, p.reward_scale * (double)delivered) : 0.0;
        const double r = p.sparse_reward ? delayed : reward;
        const int ep_step = h.epstep + 1;
        h.epstep = ep_step; h.tick = (int)(tick + 1u);
        done = !(ep_step <= p.max_episode_steps && h.awaiting);
        if (scalars_out) {
            float *o = scalars_out + c.out * 8;
            o[0] = (float)(-r); o[1] = (float)r; o[2] = (float)done; o[3] = (float)coverage;
            o[4] = (float)real_cov; o[5] = (float)transport; o[6] = (float)delivered; o[7] = (float)div_nz(r, p.max_team_reward);
        }
        if (done) {                                            // rare: the record's flag, the restart list, the statistics
            c.ei(EI_DONE) = c.g.done_count ? 3 : 1;            // (3: on the list of the next reset launch, see assign_and_score)
            if (c.g.done_count) {
                const int parity = c.list_parity();
                const int slot = atomicAdd(c.g.done_count + parity, 1);
                c.g.done_list[(int64_t)parity * c.g.N + slot] = (int32_t)c.env;
            }
            if (c.g.ep_stats) {
                double *es = c.g.ep_stats;
                atomicAdd(es + 0, 1.0); atomicAdd(es + 1, epr); atomicAdd(es + 2, (double)ep_step);
                atomicAdd(es + 3, coverage); atomicAdd(es + 4, (double)delivered);
            }
        }
    }
    return __builtin_amdgcn_readfirstlane(done);
}

// image_targets from the registers (target t on lane Nc + t)
template <typename ObsT>
__device__ __forceinline__ void image_targets_held(Ctx<ObsT> &c, const HeldState &h, int &last_gw) {
    if constexpr (sizeof(ObsT) == 4) {
    const Params &p = c.p;
    const int t = c.lane - p.Nc;
    if (t >= 0 && t < p.Nt) {
        float *pt = c.pub_tgt(t), *row = c.img_tgt_row(t) + 13;
        const int gw = h.gw & 0xffffff;                       // bit 24 (colliding) is not part of the observation
        const float x = (float)h.x, y = (float)h.y;
        pt[0] = x; pt[1] = y; row[0] = x; row[1] = y;
        if (gw != last_gw) {
            last_gw = gw;
            const int goal = (gw & 0xff) - 1, weight = (gw >> 8) & 0xff, empty = (gw >> 16) & 0xf;
            const float loaded = goal >= 0 && weight > 0 ? 1.0f : 0.0f;
            pt[3] = loaded; row[3] = loaded;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                row[6 + w] = (float)(goal == w ? weight : 0);
                row[10 + w] = (float)((empty >> w) & 1);
            }
        }
    }
    wave_sync();
    }
}

// =============================================================================================
// The step kernel: one wave per environment, 4 environments per workgroup.
template <typename ObsT, typename Shape, int FLOW = FLOW_ANY>
__global__ __launch_bounds__(256, 4) __attribute__((amdgpu_num_sgpr(96)))   // above 96 SGPRs a SIMD holds 7 waves instead of 8
void step_kernel(const Params *__restrict__ pp, const Ptrs g) {
    const Shape shape(pp);
    const Params &p = shape.get();   // scenario constants live in device memory: scalar loads on demand instead of ~80 pinned SGPRs
    extern __shared__ __align__(16) unsigned char smem[];
#ifdef MATE_PHASE_CLOCKS
    const long long t_begin = (long long)__builtin_amdgcn_s_memtime();
    const long long r_begin = (long long)__builtin_amdgcn_s_memrealtime();   // constant 100 MHz: calibrates the s_memtime ticks
#endif
    // tick and list parity: launch arguments, or the device-resident counter (graph-replayable launches, see Params)
    // (no select: the host keeps dev_tick = dev_group = 0 in the device parameters while it counts itself, and passes
    // parity = 0 and tick = the offset inside the reset interval while the device counts)
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.done_count) {
        const int32_t parity = (int32_t)((p.dev_group + (uint32_t)g.parity) & 1u);
        g.done_count[parity ^ 1] = 0;  // next step's counter
        g.ctrl[0] = parity;            // read by the auto-reset launch behind a device-counted interval
    }
    const uint32_t tick = p.dev_tick + g.tick;
    // wave-uniform by construction; readfirstlane lets the compiler keep everything derived from it in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int64_t env = (int64_t)blockIdx.x * 4 + wave;
    if (env >= g.N) return;
    phase_prio(g.stagger, 0);
    // The phases read the launch arguments from the kernel-argument segment where they use them (`gk` aliases `g`
    // there) instead of from the copy the compiler preloads into ~45 SGPRs at entry and keeps for the whole
    // kernel: with the scenario constants that was more than the 96 SGPRs a wave may hold at full occupancy.
    const Ptrs &gk = kernarg_ptrs(g);
    // the four waves of a workgroup never synchronise: each owns one environment and its LDS slice
    Ctx<ObsT> c(p, gk, smem + wave * p.lds_wave_bytes, lane, env, FLOW);
#ifdef MATE_PHASE_CLOCKS
    if (lane == 0 && g.phase_clocks) {
        g.phase_clocks[env * kClockStride + 0] = t_begin;
    }
#endif
    PHASE_STAMP(1);
#ifdef MATE_PHASE_CLOCKS
#define SKIP(bit) (g.debug_skip & (bit))
#else
#define SKIP(bit) false
#endif
    const StepDraws draws = load_records_with_draws(c, tick, !SKIP(1));
    const int mode = c.mode();
    if (c.freeze_done() && mode != MODE_OBSERVE) {
        wave_sync();
        if (c.ei(EI_DONE) != 0) {     // waiting for the next batched reset: no step, no new observation
            // (the whole row, like the idle rows of the fused kernels: the one-launch and the two-launch form of a step write the same bytes)
            if (lane == 0 && g.scalars) { float *o = g.scalars + c.out * 8; o[0] = 0.f; o[1] = 0.f; o[2] = 2.f; o[3] = o[4] = o[5] = o[6] = o[7] = 0.f; }
            if (lane == 0 && g.idle_steps) g.idle_steps[env] += 1;
            if (lane == 0 && g.done_count && c.ei(EI_DONE) == 1) {      // finished under auto_reset = 0 earlier: not on the list yet
                const int parity = c.list_parity();
                const int slot = atomicAdd(g.done_count + parity, 1);
                g.done_list[(int64_t)parity * g.N + slot] = (int32_t)env;
                reinterpret_cast<int32_t *>(g.dyn + env * p.DW + p.DF)[p.Nt * TI_STRIDE + EI_DONE] = 3;
            }
            return;
        }
    }
    wave_sync();
    build_entities(c);
    wave_sync();
    PHASE_STAMP(2);
    phase_prio(g.stagger, 1);
    SUB_STAMP(c, 9);
    if (!SKIP(2)) simulate_cameras(c, draws, mode != MODE_OBSERVE);
    SUB_STAMP(c, 12);
    if (mode != MODE_OBSERVE && !SKIP(4)) simulate_targets(c, draws);
    else wave_sync();
    PHASE_STAMP(3);
    phase_prio(g.stagger, 2);
    // shapes with one round of sector pairs: the tracked bits and the warehouses from the round's ballot, in registers
    const bool reg_tail = p.sector_rounds <= 1 && mode != MODE_OBSERVE && !SKIP(8) && !SKIP(32);
    int tracked_reg = 0, inside_reg = -1;
    // (Measured and dropped in round 4, profiles/r04_step_early_rows.txt: the target rows -- two thirds of a step's bytes, which need
    // only the range tests and the kinematics -- packed and stored right behind the visibility phase, ahead of the goals, and packed
    // once more in the rare step in which _assign_goals changed a cargo flag or a goal bit: 12.7-12.8 us against 12.4-12.5 us.  The
    // launch does not end with the drain of its stores, nor with the mean wave's chain -- four trims of that chain, 3 % of its
    // instructions and five LDS hand-offs, changed it by 0.0 us -- but with the YOUNGEST wave of each SIMD: lifetimes p50 20.5 k,
    // p99 26 k, max 28 k cycles, the slow waves slow in every phase alike.)
    PackDescriptors pack_desc;
    constexpr int GCE = Shape::kHeldGC, GTE = Shape::kHeldGT;
    const bool early_desc = FLOW != FLOW_ANY && Shape::kGreedyHeld && GCE + GTE <= 8 && packs_rows_f32(c);      // (wave-uniform; elsewhere the held registers cost the eighth wave per SIMD)
    if (reg_tail) {
        RangeRoles none;
        uint32_t seen_unused;
        unsigned long long sector_ballot = 0ull;
        update_view<false, false>(c, tick, S_TRANSMIT, true, none, seen_unused, nullptr, &sector_ballot);
        view_tail_regs(c, sector_ballot, tracked_reg, inside_reg);
    } else
    if (!SKIP(8)) update_view(c, tick, S_TRANSMIT, true);
    PHASE_STAMP(4);
    phase_prio(g.stagger, 3);
    // the packer's descriptors on their way (L2 / L1: every wave reads the same table) while the goals, the rewards and the gather
    // scratch are made
    if (early_desc) load_pack_descriptors(c, pack_desc);
    if (mode == MODE_OBSERVE) score_only(c, g.scalars);
    else if (reg_tail) assign_and_score(c, tick, g.scalars, &tracked_reg, &inside_reg);
    else if (!SKIP(32)) assign_and_score(c, tick, g.scalars);
    PHASE_STAMP(5);
    // the state record leaves as soon as it is final (nothing behind the goals writes it): its stores' acknowledgements then pass under
    // the packer instead of standing, with the rows', between the wave's last instruction and the end of the launch
    // (step_kernel 12.37 -> 12.17 us, step_greedy_kernel 17.93 -> 17.55 us at 4096 x MATE-4v8-9, A/B in one process: tools/archive/step_ab.py)
    if (Shape::kEarlyStateStore && mode != MODE_OBSERVE) store_dynamic(c);
    if (!SKIP(64)) fill_scratch(c);
    PHASE_STAMP(6);
    phase_prio(g.stagger, 4);
    if (!SKIP(128)) { if (early_desc) pack_observations<true>(c, pack_desc); else pack_observations<false>(c, pack_desc); }
    PHASE_STAMP(7);
    if (!Shape::kEarlyStateStore && mode != MODE_OBSERVE) store_dynamic(c);
    PHASE_STAMP(8);
#ifdef MATE_PHASE_CLOCKS
    if (lane == 0 && g.phase_clocks) g.phase_clocks[env * kClockStride + 15] = (long long)__builtin_amdgcn_s_memrealtime() - r_begin;
#endif
}

// =============================================================================================
// The step kernel with an environment split over TWO waves (one 128-thread workgroup per environment).
//
// At the headline batch step_kernel holds one generation of waves, four per SIMD, each walking ONE chain of ~1500 dependent
// instructions: its vector unit is busy 60 % of the time (0.15 of a wave's life, times four) and the launch lasts as long as
// that chain.  The chain has two halves that meet only twice:
//   wave A (cameras)  records -> Camera.simulate            | sector tests + occlusion lookups -> tracked bits ->   | camera rows,
//                                                            | _assign_goals, rewards, counters, targets' slots      | masks, state
//   wave B (targets)  records -> Target.simulate (screen,    | range tests, obstacle slots                           | target rows
//                     walk), entity table                    |                                                       |
//                                              barrier 1 ----^                                        barrier 2 -----^
// Everything in the environment's LDS slice has ONE writer per interval: A owns the camera angles, the integer words, the sector
// / camera->obstacle mask words and flags, the constant, camera and target slots of the gather scratch; B owns the targets'
// positions, the entity table, its collision screen, the range mask words and flags and the obstacle slots.  (Both write the
// static record: identical values.)  B reports the colliding bits as one word, which A folds into the goal words it owns.
// Same phase functions, same arithmetic, same bytes as step_kernel (tested); compiled for the two folded flows (f32 observations).
template <typename ObsT>
__device__ __forceinline__ void split_commit_records(Ctx<ObsT> &c, int role, double s0, double s1, double d0, double d1, const ObsT (&q)[4]) {
    const Params &p = c.p;
    const int lane = c.lane;
    const double *s = c.g.stat + c.env * p.SW;
    const double *d = c.g.dyn + c.env * p.DW;
    const int t_lo = 2 * p.Nc, t_hi = 2 * p.Nc + 2 * p.Nt;          // words of the dynamic record that hold the targets' positions: B's
    auto mine = [&](int i) { const bool pos = i >= t_lo && i < t_hi; return role == 0 ? !pos : pos; };
    if (lane < p.SW) c.st[lane] = s0;
    if (lane + 64 < p.SW) c.st[lane + 64] = s1;
    for (int i = lane + 128; i < p.SW; i += 64) c.st[i] = s[i];
    if (lane < p.DW && mine(lane)) c.dy[lane] = d0;
    if (lane + 64 < p.DW && mine(lane + 64)) c.dy[lane + 64] = d1;
    for (int i = lane + 128; i < p.DW; i += 64) if (mine(i)) c.dy[i] = d[i];
    // gather scratch: [0, sc_obs) constants, cameras, targets: A; [sc_obs, nscratch) obstacles: B
    const ObsT *si = reinterpret_cast<const ObsT *>(c.g.scratch_init);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k;
        if (i < p.nscratch && (role == 0 ? i < p.sc_obs : i >= p.sc_obs)) c.scratch[i] = q[k];
    }
    for (int i = lane + 256; i < p.nscratch; i += 64) if (role == 0 ? i < p.sc_obs : i >= p.sc_obs) c.scratch[i] = si[i];
    // mask words: the range rounds' are B's, all others A's
    const int r_lo = p.bit_range >> 5, r_hi = r_lo + 2 * p.range_rounds;
    for (int i = lane; i < p.MW; i += 64) { const bool rng = i >= r_lo && i < r_hi; if (role == 0 ? !rng : rng) c.mask[i] = 0u; }
}

// Camera.perceive for all (camera, target) and (camera, camera) pairs: update_view's sector rounds, on wave A
template <typename ObsT>
__device__ __forceinline__ void split_view_sector(Ctx<ObsT> &c, uint32_t tick) {
    const Params &p = c.p;
    const int lane = c.lane;
    double2 w[kDegWords];
    for (int round = 0; round + 1 < p.sector_rounds; ++round) {
        const SectorEval e = sector_eval(c, round * 64 + lane, tick, S_TRANSMIT, true);
        sector_fetch(c, e, w);
        const bool seen = sector_resolve(c, e, w);
        if (round * 64 + lane < p.n_sector) set_flag(c, round * 64 + lane, seen);
        const unsigned long long b = __ballot(seen);
        if (lane == 0) { c.mask[2 * round] = (uint32_t)b; c.mask[2 * round + 1] = (uint32_t)(b >> 32); }
    }
    const int last = p.sector_rounds - 1;
    SectorEval pending;
    pending.seen = false; pending.need = false;
    if (last >= 0) {
        pending = sector_eval(c, last * 64 + lane, tick, S_TRANSMIT, true);
        sector_fetch(c, pending, w);
    }
    // under the occlusion records' round trip: the static camera->obstacle bits (environment.py:752-755), the always-true bit ...
    if (lane < p.Nc) {
        const uint64_t m = c.camobs(lane);
        c.mask[(p.bit_camobs >> 5) + 2 * lane] = (uint32_t)m;
        c.mask[(p.bit_camobs >> 5) + 2 * lane + 1] = (uint32_t)(m >> 32);
    }
    if (lane == 0) { c.mask[p.bit_always >> 5] = 1u; set_flag(c, p.fs_always, true); }
    for (int q = lane; q < p.Nc * p.No; q += 64) {
        const int cam = (int)(((float)q + 0.5f) * p.inv_No);
        const int o = q - cam * p.No;
        set_flag(c, p.fs_camobs + cam * p.No + o, (c.camobs(cam) >> o) & 1ull);
    }
    // ... and which warehouse holds a target (the second half of update_view's tail)
    if (lane < p.Nt) {
        const double x = c.tx(lane), y = c.ty(lane);
        const bool px = x > 0.0, py = y > 0.0;
        const double wx = px ? kWarehouseCenter : -kWarehouseCenter, wy = py ? kWarehouseCenter : -kWarehouseCenter;
        const double sup = fmax(fabs(x - wx), fabs(y - wy));
        c.inside(lane) = sup <= kWarehouseRadius ? (px ? (py ? 0 : 3) : (py ? 1 : 2)) : -1;
    }
    unsigned long long b = 0ull;
    if (last >= 0) {
        const bool seen = sector_resolve(c, pending, w);
        if (last * 64 + lane < p.n_sector) set_flag(c, last * 64 + lane, seen);
        b = __ballot(seen);
        if (lane == 0) { c.mask[2 * last] = (uint32_t)b; c.mask[2 * last + 1] = (uint32_t)(b >> 32); }
    }
    wave_sync();
    // tracked_bits = camera_target_view_mask.any(axis=0) (environment.py:1388)
    if (lane < p.Nt) {
        int any = 0;
        for (int cam = 0; cam < p.Nc; ++cam) any |= (int)c.mask_bit(cam * p.Nt + lane);
        c.tracked(lane) = any;
    }
    wave_sync();
}

// Sensor.perceive for all (target, camera | obstacle | target) pairs: update_view's range rounds, on wave B
template <typename ObsT>
__device__ __forceinline__ void split_view_range(Ctx<ObsT> &c) {
    const Params &p = c.p;
    const int lane = c.lane;
    const int rbase = p.bit_range >> 5;
    uint32_t seen_bits = 0;
#pragma unroll 4
    for (int round = 0; round < p.range_rounds; ++round) {
        const int q = round * 64 + lane;
        const int qq = q < p.n_range ? q : 0;
        const int t = (int)(((float)qq + 0.5f) * p.inv_NJ);
        const int j = qq - t * p.NJ;
        const int tj = c.tgt_slot(t);
        const bool diag = (j == tj);
        const float dx = c.exf[tj] - c.exf[j], dy = c.eyf[tj] - c.eyf[j];
        const float d2 = fmaf(dy, dy, dx * dx);
        const float lim = (float)p.tgt_sight + c.erf[j], lim2 = lim * lim, rim = range_rim(lim);
        bool seen = d2 < lim2 - rim;
        if (!seen && !(d2 > lim2 + rim)) seen = range_exact(c, tj, j);
        seen_bits |= (uint32_t)((seen || diag) && q < p.n_range) << round;
    }
    for (int round = 0; round < p.range_rounds; ++round) {
        const int q = round * 64 + lane;
        const bool seen = (seen_bits >> round) & 1u;
        if (q < p.n_range) set_flag(c, p.fs_range + q, seen);
        const unsigned long long b = __ballot(seen);
        if (lane == 0) { c.mask[rbase + 2 * round] = (uint32_t)b; c.mask[rbase + 2 * round + 1] = (uint32_t)(b >> 32); }
    }
}

// one team's rows through the descriptor table (pack_rows_f32, one block): descriptors before the first store
template <int G, typename ObsT>
__device__ __forceinline__ void split_pack_rows(const Ctx<ObsT> &c, const uint32_t *table, float *dst, int elems) {
    if constexpr (sizeof(ObsT) == 4) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int nv = elems / 4;
    const uint4 *tab = reinterpret_cast<const uint4 *>(table);
    f32x4 *out = reinterpret_cast<f32x4 *>(dst);
    uint4 d[G > 0 ? G : 1];
#pragma unroll
    for (int k = 0; k < G; ++k) { const int i = c.lane + 64 * k; d[k] = tab[i < nv ? i : 0]; }
#pragma unroll
    for (int k = 0; k < G; ++k) asm volatile("" : "+v"(d[k].x), "+v"(d[k].y), "+v"(d[k].z), "+v"(d[k].w));
    auto chunk = [&](const uint4 &x) { return f32x4{gather_one(c, x.x), gather_one(c, x.y), gather_one(c, x.z), gather_one(c, x.w)}; };
#pragma unroll
    for (int k = 0; k < G; ++k) { const int i = c.lane + 64 * k; if (i < nv) stream_store(chunk(d[k]), &out[i]); }
    for (int i = c.lane + 64 * G; i < nv; i += 64) stream_store(chunk(tab[i]), &out[i]);
    }
}

template <typename ObsT, typename Shape, int FLOW, int EPW = 1>      // EPW: environments per workgroup (wave pairs)
__global__ __launch_bounds__(128 * EPW, 4) __attribute__((amdgpu_num_sgpr(96)))
void step_split_kernel(const Params *__restrict__ pp, const Ptrs g) {
    static_assert(sizeof(ObsT) == 4 && (FLOW == FLOW_RANDOM || FLOW == FLOW_ACT_F32), "the two-wave step: f32 observations, a folded flow");
    const Shape shape(pp);
    const Params &p = shape.get();
    extern __shared__ __align__(16) unsigned char smem[];
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.done_count) {
        const int32_t parity = (int32_t)((p.dev_group + (uint32_t)g.parity) & 1u);
        g.done_count[parity ^ 1] = 0;  // next step's counter
        g.ctrl[0] = parity;            // read by the auto-reset launch behind a device-counted interval
    }
    const uint32_t tick = p.dev_tick + g.tick;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int role = wave & 1, pair = wave >> 1;      // role 0: wave A (cameras), 1: wave B (targets)
    const int64_t env = (int64_t)blockIdx.x * EPW + pair;
    if (env >= g.N) return;
    phase_prio(g.stagger, 0);
    const Ptrs &gk = kernarg_ptrs(g);
    Ctx<ObsT> c(p, gk, smem + pair * p.lds_wave_bytes, lane, env, FLOW);
#ifdef MATE_PHASE_CLOCKS      // per-wave stamps: slots 0-7 wave A, 8-15 wave B (tools/archive/split_phases.py)
#define SPLIT_STAMP(i) do { if (lane == 0 && g.phase_clocks) g.phase_clocks[env * kClockStride + role * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define SPLIT_STAMP(i) do { } while (0)
#endif
    SPLIT_STAMP(0);
    // ---- records into registers (both waves: all of both records), the step's draws under their latency
    const double *s = c.g.stat + env * p.SW;
    const double *d = c.g.dyn + env * p.DW;
    const ObsT *si = reinterpret_cast<const ObsT *>(c.g.scratch_init);
    double s0 = s[lane < p.SW ? lane : 0], d0 = d[lane < p.DW ? lane : 0], s1 = 0.0, d1 = 0.0;
    if (p.SW > 64) s1 = s[lane + 64 < p.SW ? lane + 64 : 0];
    if (p.DW > 64) d1 = d[lane + 64 < p.DW ? lane + 64 : 0];
    ObsT q[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (p.nscratch > 64 * k) q[k] = si[lane + 64 * k < p.nscratch ? lane + 64 * k : 0];
    asm volatile("" : "+v"(s0), "+v"(s1), "+v"(d0), "+v"(d1));
    asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
    StepDraws draws{0.0, 0.0};
    {
        // A: its cameras' actions (random policy) and the transmittance draws of the camera->target pairs; B: its targets' actions
        DrawRole r = draw_role(c);
        const bool target_lane = lane >= p.Nc && lane < p.Nc + p.Nt;
        if (role == 0 ? (r.kind == 1 && target_lane) : (r.kind == 2 || (r.kind == 1 && !target_lane))) r.kind = 0;
        if (c.act_prefetched()) {
            StepDraws act = prefetch_action(c);
            asm volatile("" : "+v"(act.a0), "+v"(act.a1));
            if (role == 0) (void)step_draws(c, tick, nullptr, &r);
            draws = act;
        } else
        if (role == 0 || c.mode() == MODE_STEP_RANDOM) draws = step_draws(c, tick, nullptr, &r);
    }
    // the batched auto-reset's idle environments: decided from the record in registers, by both waves alike
    int done_word;
    {
        const int idx = p.Nt * TI_STRIDE + EI_DONE, word = p.DF + (idx >> 1);      // (wave-uniform)
        const double dw = word < 64 ? d0 : d1;
        const int half = (idx & 1) ? __double2hiint(dw) : __double2loint(dw);
        done_word = __builtin_amdgcn_readlane(half, word & 63);
        if (p.DW > 128 && word >= 128) done_word = reinterpret_cast<const int32_t *>(d + p.DF)[idx];
    }
    if (c.freeze_done() && done_word != 0) {
        if (role == 0 && lane == 0) {
            if (g.scalars) { float *o = g.scalars + c.out * 8; o[0] = 0.f; o[1] = 0.f; o[2] = 2.f; o[3] = o[4] = o[5] = o[6] = o[7] = 0.f; }
            if (g.idle_steps) g.idle_steps[env] += 1;
            if (g.done_count && done_word == 1) {      // finished under auto_reset = 0 earlier: not on the list yet
                const int parity = c.list_parity();
                const int slot = atomicAdd(g.done_count + parity, 1);
                g.done_list[(int64_t)parity * g.N + slot] = (int32_t)env;
                reinterpret_cast<int32_t *>(g.dyn + env * p.DW + p.DF)[p.Nt * TI_STRIDE + EI_DONE] = 3;
            }
        }
        return;
    }
    split_commit_records(c, role, s0, s1, d0, d1, q);
    wave_sync();
    SPLIT_STAMP(1);
    phase_prio(g.stagger, 1);
    if (role == 0) {
        simulate_cameras(c, draws, true);
        wave_sync();
    } else {
        build_entities(c);
        wave_sync();
        simulate_targets(c, draws, nullptr, &c.xch(4));
        // Obstacle.state slots of the gather scratch (fill_scratch's static part)
        for (int o = lane; o < p.No; o += 64) {
            ObsT *sc = c.scratch + p.sc_obs + o * 3;
            sc[0] = (ObsT)c.obs_x(o); sc[1] = (ObsT)c.obs_y(o); sc[2] = (ObsT)c.obs_r(o);
        }
    }
    SPLIT_STAMP(2);
    __syncthreads();          // barrier 1: camera angles and sight ranges, target positions, the entity table
    SPLIT_STAMP(3);
    phase_prio(g.stagger, 2);
    constexpr int GC = Shape::kHeldGC, GT = Shape::kHeldGT;
    if (role == 0) {
        split_view_sector(c, tick);
        SPLIT_STAMP(4);
        phase_prio(g.stagger, 3);
        if (lane < p.Nt) {          // the colliding bits B reported, into the goal words (Target.simulate, entities.py:668)
            const int gw = c.ti(lane, TI_GW) & ~(1 << 24);
            c.ti(lane, TI_GW) = gw | (((c.xch(4) >> lane) & 1) << 24);
        }
        wave_sync();
        assign_and_score(c, tick, g.scalars);
        // Target.state slots (fill_scratch's per-target part; obs_mode 0 in the folded flows)
        if (lane < p.Nt) {
            ObsT *sc = c.scratch + p.sc_tgt + lane * 14;
            const int gw = c.ti(lane, TI_GW) & 0xffffff;
            const int cap = 1 + (int)((c.capword() >> lane) & 1ull);
            sc[0] = (ObsT)c.tx(lane); sc[1] = (ObsT)c.ty(lane);
            sc[4] = (ObsT)(cap == 2 ? p.tgt_step * 0.5 : p.tgt_step); sc[5] = (ObsT)cap;
            const int goal = (gw & 0xff) - 1, weight = (gw >> 8) & 0xff, empty = (gw >> 16) & 0xf;
            sc[3] = (ObsT)(goal >= 0 && weight > 0 ? 1.0 : 0.0);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                sc[6 + w] = (ObsT)(goal == w ? weight : 0);
                sc[10 + w] = (ObsT)((empty >> w) & 1);
            }
        }
    } else {
        split_view_range(c);
    }
    SPLIT_STAMP(5);
    __syncthreads();          // barrier 2: flags, gather scratch, integer words
    SPLIT_STAMP(6);
    phase_prio(g.stagger, 4);
    if (role == 0) {
        if (p.cam_elems > 0) {
            float *dst = reinterpret_cast<float *>(c.g.cam_obs) + c.out * p.cam_elems;
            if ((p.cam_elems % 4) == 0) split_pack_rows<GC>(c, c.table, dst, p.cam_elems);
            else pack_block<ObsT>(c, dst, c.table, p.cam_elems);
        }
        store_masks(c);
        store_dynamic(c);
    } else {
        float *dst = reinterpret_cast<float *>(c.g.tgt_obs) + c.out * p.tgt_elems;
        if ((p.tgt_elems % 4) == 0) split_pack_rows<GT>(c, c.table + p.tgt_table_off, dst, p.tgt_elems);
        else pack_block<ObsT>(c, dst, c.table + p.tgt_table_off, p.tgt_elems);
    }
    SPLIT_STAMP(7);
}

// An environment whose episode had ended BEFORE a fused rollout began (a step or rollout with auto_reset = 0, or an
// imported done flag) is skipped by every step of the launch, so the step that would have listed it for the reset
// launch never runs: list it here, or it would idle forever.
template <typename ObsT, int L>
__device__ __forceinline__ void list_finished_at_entry(Ctx<ObsT, L> &c) {
    if (c.lane == 0 && c.g.done_count && c.ei(EI_DONE) == 1) {     // (3: a batched-reset step listed it already)
        const int parity = c.list_parity();
        const int slot = atomicAdd(c.g.done_count + parity, 1);
        c.g.done_list[(int64_t)parity * c.g.N + slot] = (int32_t)c.env;
        c.ei(EI_DONE) = 3;      // listed (stored with the record at the end of the launch): the one-step launches of a batched-reset interval meet it again
    }
}

// =============================================================================================
// K-step fused rollout under the on-device random policy: the same phases as step_kernel in a loop, with
// the environment's records resident in LDS for the whole launch.  Outputs of step r go to row r*N + env
// of the (rollout-shaped) output buffers.  Waves drift apart freely, so the launch takes about the MEAN
// wave time per step instead of the slowest wave's, and there is no kernel boundary between steps.
// An environment whose episode ends stops stepping (rows of the remaining steps carry done = 2 in the
// scalar record) and is reset by the host-launched reset kernel after the rollout.
// `E` environments per wave (Ctx: L = 64 / E lanes each; policy_kernels.hpp, rollout_greedy_kernel says what that means): E > 1 runs
// the phase functions written for any L -- the descriptor packer, no held roles, no row image, no register-resident state.
template <typename ObsT, typename Shape, int FLOW = FLOW_ANY, int E = 1>
__global__ __launch_bounds__(256, 4) void rollout_kernel(const Params *__restrict__ pp, const Ptrs g) {
    constexpr int L = 64 / E;
    static_assert(E == 1 || E == 2 || E == 4 || E == 8, "environments per wave");
    const Shape shape(pp, true);
    const Params &p = shape.get();
    extern __shared__ __align__(16) unsigned char smem[];
    if constexpr (E > 1) {          // (the per-step flows run this kernel with one step: tick and list parity may be the device's, see step_kernel)
        if (blockIdx.x == 0 && threadIdx.x == 0 && g.done_count) {
            const int32_t parity = (int32_t)((p.dev_group + (uint32_t)g.parity) & 1u);
            g.done_count[parity ^ 1] = 0;
            g.ctrl[0] = parity;
        }
    } else
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.done_count) g.done_count[g.parity ^ 1] = 0;
    // wave-uniform by construction; readfirstlane lets the compiler keep everything derived from it in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int hw_lane = threadIdx.x & 63, lane = hw_lane & (L - 1), shift = hw_lane & ~(L - 1);      // lane inside the environment's group; the group's first lane
    const int slot = E == 1 ? wave : wave * E + (hw_lane >> (E == 8 ? 3 : E == 4 ? 4 : 5));                       // the environment's slice of the workgroup's LDS
    const int64_t env = (int64_t)blockIdx.x * (4 * E) + slot;
    if (env >= g.N) return;               // (E > 1: the groups past the end of the batch leave; the others go on under their EXEC mask)
    const Ptrs &gk = kernarg_ptrs(g);     // launch arguments read where they are used (see step_kernel)
#ifdef MATE_PHASE_CLOCKS      // the launch's prologue, in s_memtime ticks since the wave began: slots 8..11 (tools/archive/rollout_prologue.py), 12 the epilogue
    const long long t_wave = (long long)__builtin_amdgcn_s_memtime();
#define PROLOGUE_STAMP(i) do { if (lane == 0 && g.phase_clocks) g.phase_clocks[env * kClockStride + (i)] = (long long)__builtin_amdgcn_s_memtime() - t_wave; } while (0)
#else
#define PROLOGUE_STAMP(i) do { } while (0)
#endif
    {
        Ctx<ObsT, L> c(p, gk, smem + slot * p.lds_wave_bytes, lane, env);
        c.shift = shift;
        load_records(c);
        wave_sync();
        build_entities(c);
        // (as one step of a per-step flow without a pending batched restart, a finished environment STEPS, as in step_kernel, and the step lists it)
        if (!(E > 1 && g.per_step && !g.freeze_done)) list_finished_at_entry(c);
        wave_sync();
    }
    PROLOGUE_STAMP(8);
    // The observation descriptors of this lane are the same for every step: loaded once and held in 32 VGPRs (the
    // headline batch runs 4 waves per SIMD, the register file has room), which takes the table's two global-load
    // round trips out of every step's pack phase.
    constexpr bool IMAGE = E == 1 && Shape::kImage;   // row-image mode: the observation rows live in LDS (see image_statics)
    constexpr bool HOLD_ROLES = E == 1 && Shape::kHoldRoles;
    static_assert(E == 1 || !Shape::kImage, "the row image is a whole-wave mode");
    static_assert(kNearWords == kRoleRounds, "one ballot per range round");
    static_assert(!IMAGE || (Shape::kHoldRoles && sizeof(ObsT) == 4 && FLOW != FLOW_ANY), "row-image mode: f32 rows, held lane roles, a folded flow");
    // the row chunks a lane holds descriptors of (E > 1: a group of L lanes takes L chunks per round -- all of them where that is at most twelve per lane)
    constexpr int kSubC = (Shape::kChunksC + L - 1) / L, kSubT = (Shape::kChunksT + L - 1) / L;
    constexpr int kGC = E == 1 ? Shape::kHeldGC : (kSubC + kSubT <= 12 ? kSubC : 2), kGT = E == 1 ? Shape::kHeldGT : (kSubC + kSubT <= 12 ? kSubT : 6);
    PackDescriptorsT<kGC, kGT> held;
    if constexpr (!IMAGE) {
        Ctx<ObsT, L> c(p, gk, smem + slot * p.lds_wave_bytes, lane, env, FLOW);
        load_pack_descriptors(c, held);      // (indices clamped: harmless when another pack path runs)
    }
    RangeRoles roles;                        // the lane's range-test pairs and limits, static inside an episode
    if constexpr (HOLD_ROLES) {
        Ctx<ObsT> c(p, gk, smem + wave * p.lds_wave_bytes, lane, env, FLOW);
        range_roles(c, roles);
        pin_roles(roles);
        PROLOGUE_STAMP(9);
        if constexpr (IMAGE) image_statics(c);
    }
    PROLOGUE_STAMP(10);
    NearCarry near{};                       // the collision screen of the step to come, made by the range tests of the step before
    if constexpr (HOLD_ROLES) {
        Ctx<ObsT> c(p, gk, smem + wave * p.lds_wave_bytes, lane, env, FLOW);
        near_seed(c, roles, near);
    }
    // Fair shares of the SIMD.  Its arbiter serves the oldest resident wave first, and in a launch that lasts for
    // tens of steps the age order never changes: of the four environment-waves of a SIMD the oldest ran a step in
    // 13 k cycles and the youngest in 21 k (measured), and the launch lasts as long as its slowest wave.
    // Rotating the issue priority by step and wave slot gives every wave each priority level equally often
    // (+13 % at 4096 environments; rotating at every phase boundary is no better, and a laggard-first controller that read the SIMD-mates' progress through a table keyed
    // by the hardware wave id equalised them perfectly and gained nothing more: waves kept in lockstep contend for the
    // same units at the same time).
    uint32_t hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    const int wave_slot = (int)(hw_id & 15u);
#ifdef MATE_PHASE_CLOCKS
    long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // cycles per phase summed over the steps of this launch
#ifdef MATE_SUB_CLOCKS
    long long sub[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // (SUB_ACC: 0..7 cycles of sub-phases, 15 the previous stamp)
#endif
    long long t_prev = (long long)__builtin_amdgcn_s_memtime();
    const long long t_first = t_prev, r_first = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    bool stepped = false;            // a full step has written the static mask words, flags and scratch slots
    int last_gw = -1;                // fill_scratch: the goal word behind the target's goal / cargo slots
    DrawCarry carry{0u, 0u, 0xffffffffu};
    DrawRole draws_of_lane;
    {
        Ctx<ObsT, L> c(p, gk, smem + slot * p.lds_wave_bytes, lane, env, FLOW);
        draws_of_lane = draw_role(c);
    }
    // The register-resident step (HeldState): the row-image shapes under the random-policy flow
    constexpr bool HELDSTATE = IMAGE && FLOW == FLOW_RANDOM;
    HeldState h{};
    int finished = 0;                // HELDSTATE: the episode is over (wave-uniform; the record's EI_DONE otherwise)
    const uint32_t tick0 = (E > 1 ? p.dev_tick : 0u) + g.tick;   // (launch arguments read once: inside the loop each read is a scalar load and a wait)
    const uint32_t cam_low = (uint32_t)(reinterpret_cast<uintptr_t>(g.cam_obs) >> 4) & 7u, tgt_low = (uint32_t)(reinterpret_cast<uintptr_t>(g.tgt_obs) >> 4) & 7u;
    const int n_steps = g.rollout_steps;
    if constexpr (HELDSTATE) {
        Ctx<ObsT> c(p, gk, smem + wave * p.lds_wave_bytes, lane, env, FLOW);
        held_load(c, h);
        finished = __builtin_amdgcn_readfirstlane((int)(c.ei(EI_DONE) != 0));
    }
    PROLOGUE_STAMP(11);
    // (Measured and dropped: the four waves of a workgroup -- one per SIMD of a CU, their environments' records all in this
    // workgroup's LDS -- handing their environments on to each other after every quarter of the launch, so that every SIMD of
    // the CU works on all sixteen environments.  It removed every dependence of a wave's pace on its environment (occlusion
    // lookups, collisions: +-4 % before) and changed nothing: what separates the fastest wave of a launch from the slowest,
    // ~15 %, goes with the SIMD / CU it runs on, not with the environment it steps.)
#pragma clang loop unroll(disable)
    for (int r = 0; r < n_steps; ++r) {
        // an opaque copy of the lane id per iteration keeps the compiler from hoisting every lane-role
        // computation of the body out of the loop (which costs >100 VGPRs of spills)
        int lane_r = lane, wave_r = wave, slot_r = slot;
        asm volatile("" : "+v"(lane_r));
        asm volatile("" : "+s"(wave_r));
        if constexpr (E == 1) slot_r = wave_r; else asm volatile("" : "+v"(slot_r));
        const Params *pr = pp;
        asm volatile("" : "+s"(pr));
        const Shape shape_r(pr, true);
        const Params &p = shape_r.get();
        const int64_t env_r = (int64_t)blockIdx.x * (4 * E) + slot_r;
        // (... and the held roles: predicates derived from them would otherwise be hoisted out of the loop as SGPR masks, which
        // the kernel has no scalar registers left for -- each came back as two v_readlane per step)
        if constexpr (HOLD_ROLES) pin_roles(roles, p.range_rounds, IMAGE, p.sector_rounds == 1);
        Ctx<ObsT, L> c(p, gk, smem + slot_r * p.lds_wave_bytes, lane_r, env_r, FLOW);
        c.shift = shift;
        c.out = (int64_t)r * g.N + env_r;
        c.statics_done = stepped;
        c.pivots = Shape::kGreedyHeld;      // (the compiled shapes; the generic kernel has no registers to spare)
        // (E > 1 as one step of the per-step flows: step()'s rule -- a finished environment idles only while a batched restart is pending)
        if (HELDSTATE ? finished != 0 : (c.ei(EI_DONE) != 0 && (E == 1 || !g.per_step || g.freeze_done))) {
            if (lane_r == 0 && g.scalars) { float *o = g.scalars + c.out * 8; o[0] = 0.f; o[1] = 0.f; o[2] = 2.f; o[3] = o[4] = o[5] = o[6] = o[7] = 0.f; }
            if (lane_r == 0 && g.idle_steps) g.idle_steps[env_r] += 1;      // a slot of the rollout, not an executed step
            continue;
        }
        const uint32_t tick = tick0 + (uint32_t)r;
        if (g.rotate_prio) {
            // (s_setprio takes an immediate: a two-level tree of scalar branches, not a chain of four)
            // rotate_prio >= 8 (experiment): the turn follows the shader clock >> rotate_prio instead of the wave's own step count --
            // the four waves of a SIMD read the same clock, so their turns never coincide however far their steps drift apart
            const int turn = g.rotate_prio >= 8 ? ((int)(__builtin_amdgcn_s_memtime() >> g.rotate_prio) + wave_slot) & 3 : (r + wave_slot) & 3;
            if (turn & 2) { if (turn & 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
            else { if (turn & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        }
#ifdef MATE_PHASE_CLOCKS
#define ROLL_STAMP(i) do { const long long t_now = (long long)__builtin_amdgcn_s_memtime(); acc[i] += t_now - t_prev; t_prev = t_now; } while (0)
#elif defined(MATE_ISA_MARKS)      // tools/isa_phases.py: phase boundaries as comments in the -S output (no instruction is emitted)
#define ROLL_STAMP(i) asm volatile("; ==== MATE_PHASE_END " #i)
#else
#define ROLL_STAMP(i) do { } while (0)
#endif
        ROLL_STAMP(7);         // loop overhead: from the end of the previous step to here
        StepDraws draws{0.0, 0.0};
        pin_draw_role(draws_of_lane);
        MATE_PHASE(1, draws = step_draws(c, tick, &carry, &draws_of_lane));
        if constexpr (FLOW == FLOW_ACT_F32) draws = prefetch_action(c);      // (the per-step flows' step(actions) on this kernel: the agents' lanes carry the caller's joint action)
        MATE_PHASE_AGAIN(1, DrawCarry again = carry; again.block = 0xffffffffu; const StepDraws d2 = step_draws(c, tick, &again); draws.a0 += 0.0 * d2.a0);
        ROLL_STAMP(0);
        if constexpr (HELDSTATE) {
            MATE_PHASE(2, simulate_cameras_held(c, draws, h));
            ROLL_STAMP(1);
#ifdef MATE_SUB_CLOCKS
            c.sub = sub;
#endif
            SUB_START(c);
            MATE_PHASE(4, simulate_targets_held(c, draws, near, h));
            SUB_ACC(c, 7);                               // targets: clip, entity table
            ROLL_STAMP(2);
            uint32_t seen = 0u;
            unsigned long long sector_ballot = 0ull;
            SUB_START(c);
            MATE_PHASE(8, update_view<true, true>(c, tick, S_TRANSMIT, true, roles, seen, &near, &sector_ballot));
            SUB_ACC(c, 3);                               // mask words, static bits
            MATE_PHASE_AGAIN(8, uint32_t again; update_view<true, true>(c, tick, S_TRANSMIT, true, roles, again, &near, &sector_ballot); seen |= again);
            bool tracked; int inside;
            view_tail_held(c, sector_ballot, h, tracked, inside);
            SUB_ACC(c, 4);                               // tracked bits, warehouses
            ROLL_STAMP(3);
            MATE_PHASE(16, finished = assign_and_score_held(c, tick, g.scalars, h, tracked, inside));
            ROLL_STAMP(4);
            MATE_PHASE(32, image_targets_held(c, h, last_gw); image_blocks(c, roles, seen));
            ROLL_STAMP(5);
            // (measured and dropped: the rows of step r leaving in the MIDDLE of step r + 1, behind its occlusion wait, so that their
            // acknowledgements have a whole step before the next wait instead of 40 % of one -- no faster)
            MATE_PHASE(64, image_store(c, cam_low, tgt_low); store_masks(c));
            wave_sync();
            stepped = true;
            ROLL_STAMP(6);
            continue;
        }
        MATE_PHASE(2, simulate_cameras(c, draws, true));
        MATE_PHASE_AGAIN(2, wave_sync(); simulate_cameras(c, StepDraws{0.0, 0.0}, true));      // (a zero action: the same instructions, the same state)
        ROLL_STAMP(1);
        MATE_PHASE(4, simulate_targets(c, draws, HOLD_ROLES ? &near : nullptr));
        ROLL_STAMP(2);
        uint32_t seen = 0u;
        MATE_PHASE(8, update_view<HOLD_ROLES, true>(c, tick, S_TRANSMIT, true, roles, seen, HOLD_ROLES ? &near : nullptr));
        MATE_PHASE_AGAIN(8, uint32_t again; update_view<HOLD_ROLES, true>(c, tick, S_TRANSMIT, true, roles, again, HOLD_ROLES ? &near : nullptr); seen |= again);
        ROLL_STAMP(3);
        MATE_PHASE(16, assign_and_score(c, tick, g.scalars));
        ROLL_STAMP(4);
        if constexpr (IMAGE) {
            MATE_PHASE(32, image_targets(c, last_gw); image_blocks(c, roles, seen));
            ROLL_STAMP(5);
            MATE_PHASE(64, image_store(c, cam_low, tgt_low); store_masks(c));
            MATE_PHASE_AGAIN(32, int gw2 = last_gw; image_targets(c, gw2); image_blocks(c, roles, seen));
            MATE_PHASE_AGAIN(64, wave_sync(); image_store(c, cam_low, tgt_low));
        } else {
        MATE_PHASE(32, fill_scratch(c, last_gw));
        ROLL_STAMP(5);
        MATE_PHASE(64, pack_observations<true, Shape::kGreedyHeld>(c, held));
        }
        wave_sync();
        stepped = true;
        ROLL_STAMP(6);
    }
#ifdef MATE_PHASE_CLOCKS
    if (lane == 0 && g.phase_clocks)
        for (int i = 0; i < 8; ++i) g.phase_clocks[env * kClockStride + i] = acc[i];
    if (lane == 0 && g.phase_clocks) {      // clock calibration: s_memtime ticks against the constant 100 MHz counter
        uint32_t hwid, xccid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xccid));
        g.phase_clocks[env * kClockStride + 13] = ((long long)xccid << 32) | (long long)hwid;
        g.phase_clocks[env * kClockStride + 14] = (long long)__builtin_amdgcn_s_memtime() - t_first;
        g.phase_clocks[env * kClockStride + 15] = (long long)__builtin_amdgcn_s_memrealtime() - r_first;
    }
#endif
#ifdef MATE_PHASE_CLOCKS
    const long long t_epilogue = (long long)__builtin_amdgcn_s_memtime();
#ifdef MATE_SUB_CLOCKS
    if (lane == 0 && g.phase_clocks)
        for (int i = 0; i < 8; ++i) g.phase_clocks[env * kClockStride + 16 + i] = sub[i];
#endif
#endif
    {
        Ctx<ObsT, L> c(p, gk, smem + slot * p.lds_wave_bytes, lane, env);
        if constexpr (HELDSTATE) held_store(c, h);
        store_dynamic(c);
    }
#ifdef MATE_PHASE_CLOCKS
    if (lane == 0 && g.phase_clocks) g.phase_clocks[env * kClockStride + 12] = (long long)__builtin_amdgcn_s_memtime() - t_epilogue;
#endif
}

}  // namespace mate
