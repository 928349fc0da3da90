// experiments.hpp -- the measurement hooks of the fused kernels.  NOT part of the product: engine_kernels.hpp includes this file
// only when a build defines one of the switches below (python -m mate_amd.build --variant NAME -DSWITCH[=bits]; tools/archive/ablate_rollout.sh,
// tools/archive/ablate_greedy.sh, tools/archive/double_phase.sh); the shipped library is compiled without any of them, from the one-line defaults
// at the top of engine_kernels.hpp, and its kernels are instruction for instruction what they were with the hooks spelled out in
// the step loops (lib/kernel_resources.json unchanged).
//
//   -DMATE_ABLATE=bits   a phase of the fused step loop COMPILED OUT, to weigh it (profiles/HISTORY.md section 5, round 2; a build without a phase is
//                        not a simulation -- the deltas over-attribute what the missing phase feeds): 1 draws, 2 cameras, 4 targets,
//                        8 visibility, 16 goals / rewards, 32 scratch or row-image blocks, 64 pack + stores, 128 the greedy agents,
//                        256 the zoom solve iterates once instead of twenty times
//   -DMATE_DOUBLE=bits   an idempotent phase executed TWICE per step: the difference of the dynamic instruction counters against the
//                        plain build is that phase's exact share, on real data (round 3): 1 draws, 2 cameras, 8 visibility,
//                        32 row-image blocks, 64 row-image store
//   -DMATE_LUT_FAKE      every occlusion lookup reads the SAME (cache-resident) record: what the real fetch's latency and its overflow
//                        paths cost (round 3: 1.9 %)
//   -DMATE_STORE_PLAIN   the observation rows leave through plain write-back stores instead of non-temporal ones (round 3: the pattern
//                        alone 15 % faster, the kernel 22 % slower: dirty rows wash the occlusion records out of the L2)
#pragma once

#ifndef MATE_ABLATE
#define MATE_ABLATE 0
#endif
#ifndef MATE_DOUBLE
#define MATE_DOUBLE 0
#endif

// a phase of the step loop: skipped when its bit is set in MATE_ABLATE
#define MATE_PHASE(bit, ...) do { if (!(MATE_ABLATE & (bit))) { __VA_ARGS__; } } while (0)
// the same phase once more (an idempotent re-execution, written out at the call site's names) when its bit is set in MATE_DOUBLE
#define MATE_PHASE_AGAIN(bit, ...) do { if (MATE_DOUBLE & (bit)) { __VA_ARGS__; } } while (0)
// GreedyCameraAgent's zoom solve (policy_kernels.hpp: zoom_fixed_point)
#define MATE_ZOOM_ITERATIONS ((MATE_ABLATE & 256) ? 1 : 20)

#ifdef MATE_LUT_FAKE
#define MATE_LUT_TABLE_OF(lc) (0 * (lc))
#else
#define MATE_LUT_TABLE_OF(lc) (lc)
#endif

#ifdef MATE_STORE_PLAIN
#define MATE_ROW_STORE(v, dst) (*(dst) = (v))
#else
#define MATE_ROW_STORE(v, dst) __builtin_nontemporal_store((v), (dst))
#endif
