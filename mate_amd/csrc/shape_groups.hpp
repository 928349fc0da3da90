// shape_groups.hpp -- the scenario shapes (cameras, targets, obstacles) with compiled specialisations of the kernels: EVERY scenario
// the reference ships (mate/assets/MATE-{1v1,1v2,2v2,2v4,4v2,4v4,4v8,8v8}-{0,9}.yaml, MATE-Navigation.yaml; MATE.yaml = 4v8-9).  A
// generic (AnyShape) fused rollout runs at less than half the rate of a specialised one (MATE-4v8-9 x 4096: 2.8e8 against 6.4e8
// env-steps/s), so until round 4 twelve of the seventeen shipped scenarios ran at half speed.  The shapes are compiled in six
// groups -- six translation units built in parallel by mate_amd/build.py (one unit with all of them takes four minutes).
// X: all kernels, f32 and f64 observations (the five shapes of BASELINE.json's configurations and the parity traces);
// Y: f32 observations only -- an f64-observation engine of such a shape runs the generic kernels.
#pragma once
#include "policy_kernels.hpp"

namespace mate {

using StepFn = void (*)(const Params *, const Ptrs);
using PolicyFn = void (*)(const Params *, const Ptrs, const PolicyPtrs);

#define MATE_SHAPES_G0(X, Y) X(4, 8, 9) X(4, 8, 0)
#define MATE_SHAPES_G1(X, Y) X(8, 8, 9) Y(8, 8, 0)
#define MATE_SHAPES_G2(X, Y) X(4, 2, 9) X(0, 8, 32) Y(4, 2, 0)
#define MATE_SHAPES_G3(X, Y) Y(4, 4, 9) Y(4, 4, 0) Y(2, 4, 9)
#define MATE_SHAPES_G4(X, Y) Y(2, 4, 0) Y(2, 2, 9) Y(2, 2, 0)
#define MATE_SHAPES_G5(X, Y) Y(1, 2, 9) Y(1, 2, 0) Y(1, 1, 9) Y(1, 1, 0)

struct KernelSet {
    StepFn step[3];            // [flow]
    StepFn split[3];           // step_split_kernel per flow, or null
    StepFn rollout[2];         // [0] generic flow, [1] FLOW_RANDOM (the row-image compilation where the shape has one)
    PolicyFn policy, rollout_greedy;
    PolicyFn step_greedy;      // step_greedy_kernel (f32 observations), or null
    int image;
    // E environments per wave (FixedShape::kSubWave, f32 observations): the fused rollouts of the small scenarios, or null / 1
    StepFn rollout_sub[3];     // [flow]: FLOW_ANY, FLOW_RANDOM, FLOW_ACT_F32 (the one-step form behind mate_engine_step)
    PolicyFn rollout_greedy_sub;
    int sub_wave;
};

// true = the group holds the shape and `out` is filled (false for an f64-observation engine of a Y shape)
bool pick_kernels_group0(int Nc, int Nt, int No, bool f64, bool no_image, KernelSet *out);
bool pick_kernels_group1(int Nc, int Nt, int No, bool f64, bool no_image, KernelSet *out);
bool pick_kernels_group2(int Nc, int Nt, int No, bool f64, bool no_image, KernelSet *out);
bool pick_kernels_group3(int Nc, int Nt, int No, bool f64, bool no_image, KernelSet *out);
bool pick_kernels_group4(int Nc, int Nt, int No, bool f64, bool no_image, KernelSet *out);
bool pick_kernels_group5(int Nc, int Nt, int No, bool f64, bool no_image, KernelSet *out);

}  // namespace mate
