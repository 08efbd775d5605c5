// Argument block of the skinny step kernels (csrc/skinny.hip) and the internal two-problem launch used by the beam step.
#pragma once
#include "common.h"

namespace asr {

struct SkinnyArgs {
    const float* x1; int ld1; int K1; const int* gather1;   // row b of X1 = x1 + (gather1? gather1[b] : b)*ld1
    const float* x2; int ld2; int K2;
    const int* gather2;                   // row b of X2 (and of c_prev) = row gather2[b] (the beam step: parent rows); NULL = b
    const float* W; int ldw; const float* bias;
    int M, N;
    // linear mode
    float* out; int ldo; int accumulate;
    const int* zero_from; int zero_t;     // rows with zero_t >= zero_from[b] are written as zeros (raw_rnn emit)
    // LSTM mode (H > 0)
    int H; const float* c_prev; float* c_out; float* h_out; float* hdrop_out; float* gates_out;
    float keep; uint32_t seed; uint32_t step;
    int wperm;                            // LSTM mode: W's columns are already in tile order (column 16*tile + 4*unit + gate)
};

// Two independent problems of the same mode (both LSTM cells, or both plain linears) in ONE launch: blockIdx.x first
// covers problem 0's column tiles, then problem 1's.  The beam step's two LM cells (decoder's and external) and the two
// projections that follow them have no dependency on each other; as separate launches each costs 5-8 us of latency.
int skinny_launch_pair(hipStream_t s, bool lstm, const SkinnyArgs& a0, const SkinnyArgs& a1);
int skinny_launch(hipStream_t s, bool lstm, const SkinnyArgs& a);      // one problem, any SkinnyArgs (e.g. with gather2)

}  // namespace asr
