// Compile-time sizes and the LDS layout of the fit constants (HIP-free: also included by the host
// emulation of the fit kernel under tests/emul).
#pragma once
#include "../../include/spsp.h"

namespace spsp {

constexpr int kV = SPSP_NUM_VERTS;          // 6890
constexpr int kV3 = kV * 3;                 // 20670
constexpr int kNJ = SPSP_NUM_JOINTS;        // 24
constexpr int kNB = SPSP_NUM_BETAS;         // 10
constexpr int kNP = SPSP_NUM_POSE_BASIS;    // 207
constexpr int kNCoef = kNB + kNP;           // 217 rows of the blend basis: [betas | R[1:]-I]
constexpr int kNFace = 5;                   // picked vertices that feed the 17 fit joints (nose, eyes, ears)

// Fit constants, one contiguous fp32 block copied into LDS by every fit workgroup.  Offsets in floats.
// Row pitches are ODD (73 / 17) so that both access patterns of the fit kernel -- lanes along a row and lanes
// along a column -- spread over distinct LDS banks.
constexpr int kJSPitch = 73, kF5Pitch = 17;
constexpr int kFcJbar = 0;                                // [24][3]
constexpr int kFcJS = kFcJbar + 72;                       // [10][73]   JS[i][k*3+c]
constexpr int kFcVt5 = kFcJS + 10 * kJSPitch;             // [16]       template coords of the 5 face verts (15 used)
constexpr int kFcS5 = kFcVt5 + 16;                        // [10][17]   shapedirs rows of the 5 face verts
constexpr int kFcP5 = kFcS5 + 10 * kF5Pitch;              // [208][17]  posedirs columns of the 5 face verts (+1 zero row)
constexpr int kFcW5 = kFcP5 + 208 * kF5Pitch;             // [5][24]    lbs weights of the 5 face verts
constexpr int kCPitch = 97;
constexpr int kFcC = kFcW5 + 120;                         // [10][97]   d(beta) coefficients over the 96-long adjoint
                                                          //            [dvp(15) | dJ(72) | 0(9)]: S5 | JS | 0
constexpr int kFcJrel = kFcC + 10 * kCPitch;              // [72 + 10*73]  same as Jbar/JS for J_k - J_parent(k) (root: J_0)
constexpr int kFcTotal = kFcJrel + 72 + 10 * kJSPitch;    // 6416 floats = 25.7 KB

// The kinematic tree the walks are compiled for: SMPL's (every SMPL / SMPL-family body model file carries this parents array).
constexpr int kSmplParents[24] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21};
constexpr int kWalkNB = 4;                                   // joints with a branching subtree, deepest first, and their children
constexpr int kWalkB[kWalkNB] = {9, 6, 3, 0};
constexpr int kWalkBChildren[kWalkNB][3] = {{12, 13, 14}, {9, 24, 24}, {6, 24, 24}, {1, 2, 3}};

}  // namespace spsp
