// tn_layout.h -- where a path position's records sit in HBM (the dense path state of the wavefront pipelines, SplitState in tn_kernels.h)
#pragma once

#include "tn_math.h"

namespace tn {

// WHERE a position's records sit.  Default (rounds 2-6): every field is an array of its own over all positions -- a wave of k_shade reads 8
// far-apart streams by position (five state fields, hit, hit primitive, NEE position) and writes 10 (five fields at each end of its region).
// -DTN_STATE_BLOCKS=1 (round 6; built, bit-identical under the GPU suite, measured, NOT the default): the fields of 64
// consecutive positions as ONE block, [field][lane] -- 5 x 1 KB of state per buffer, 1.5 KB of hand-over records (hit 1 KB | hit primitive
// 256 B | NEE position 256 B) -- so that a wave's loads and stores of a round are runs inside one contiguous piece: one stream per buffer
// instead of five to ten.  The premise was that the streaming kernels' run-to-run spread (+-10 % by which physical pages back the arrays,
// profiles/r05_y_realloc_speeds.txt) came from the NUMBER of concurrent streams.  It does not: over five fresh processes per build the
// blocks lose -- glass k_shade 9.24 ms (8.96-9.48) against 8.52 (8.49-8.79), k_generate 0.38 against 0.33, everything else equal, glass
// 1573 against 1598 Msamples/s, the 524k-triangle config 2257 / 2260, the fused scenes -0.3 % (profiles/r06_c_ab_state_blocks.md).  Five
// loads of a wave that land in ONE 5 KB piece queue up behind each other in one place; five arrays spread them.  The pointers in SplitState
// point at their field's first entry either way; positions go through sidx / hidx / hidx1 (the identity by default).
#ifndef TN_STATE_BLOCKS
#define TN_STATE_BLOCKS 0
#endif
constexpr uint32_t kStateFields = 5;
constexpr uint32_t kStateBlockF4 = kStateFields*64u;    // float4 per state block (5 KB)
constexpr uint32_t kHitBlockF4 = 96u;                   // float4 per hand-over block: 64 hits + 16 (64 hit primitives) + 16 (64 NEE positions)
TN_D uint32_t sidx(uint32_t pos) { return TN_STATE_BLOCKS ? (pos >> 6)*kStateBlockF4 + (pos & 63u) : pos; }          // SplitState::rayO .. rngId (float4 units)
TN_D uint32_t hidx(uint32_t pos) { return TN_STATE_BLOCKS ? (pos >> 6)*kHitBlockF4 + (pos & 63u) : pos; }            // SplitState::hit (float4 units)
TN_D uint32_t hidx1(uint32_t pos) { return TN_STATE_BLOCKS ? (pos >> 6)*(kHitBlockF4*4u) + (pos & 63u) : pos; }      // SplitState::hitPrim / pathNee (4-byte units)

} // namespace tn
